cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pd in 16 32 48; do
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r03_pad$pd -- python3 tools/resblock_only.py build/pad/pad$pd.so > gpurun_out/r03_pad$pd.log 2>&1
python3 tools/prof_summary.py gpurun_out/r03_pad$pd resblock > gpurun_out/r03_pad$pd.md 2>&1
grep "us per block" gpurun_out/r03_pad$pd.log; cat gpurun_out/r03_pad$pd.md | grep -v "^|---\|kernel"
done
find gpurun_out/r03_pad* -name "*.csv" -size +1M -delete
