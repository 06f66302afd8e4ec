# Round 3: L1 (TCP) view of resblock_kernel's loads -- how many of the vector-memory reads reach the L2, and how long they take
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/r03_tcp -- python3 tools/resblock_only.py > gpurun_out/r03_tcp.log 2>&1
echo rc=$?
python3 tools/prof_summary.py gpurun_out/r03_tcp resblock > gpurun_out/r03_tcp.md 2>&1; cat gpurun_out/r03_tcp.md; tail -3 gpurun_out/r03_tcp.log
find gpurun_out/r03_tcp -name "*.csv" -size +1M -delete
