# Round 3: L2 requests per board-block of the tower's launch forms (does sharing weight fragments inside a workgroup reach the L2?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 0 1 2 3; do
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/r03_l2_m$m -- python3 tools/resblock_only.py - $m > gpurun_out/r03_l2_m$m.log 2>&1
python3 tools/prof_summary.py gpurun_out/r03_l2_m$m kernel > gpurun_out/r03_l2_m$m.md 2>&1
echo "mode $m"; grep "us per block" gpurun_out/r03_l2_m$m.log; grep -v "^|---\|counter |" gpurun_out/r03_l2_m$m.md
done
find gpurun_out/r03_l2_m* -name "*.csv" -size +1M -delete
