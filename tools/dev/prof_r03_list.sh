# Round 3 (late): instruction counters of the movegen launches after the id lists became a gather
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MG="--steps 200 --warmup 20 --no-cpu-baseline --sat-boards 0 --selfplay-plies 0 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay --encode-boards 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r03_list_valu -- python3 bench.py $MG > gpurun_out/r03_list_valu.log 2>&1
echo rc=$?
python3 tools/prof_summary.py gpurun_out/r03_list_valu hive_piece > gpurun_out/r03_list_valu.md 2>&1
find gpurun_out/r03_list_valu -name "*.csv" -size +3M -delete
cat gpurun_out/r03_list_valu.md
