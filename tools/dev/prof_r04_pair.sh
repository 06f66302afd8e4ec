# Round 4: the movegen kernel in both lane layouts at 1 M boards (tools/dev/pair_bench.py launches both): kernel trace and PMC passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_pair_kt -- python3 tools/dev/pair_bench.py 1048576 > gpurun_out/r04_pair_kt.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04_pair_valu -- python3 tools/dev/pair_bench.py 1048576 > gpurun_out/r04_pair_valu.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/r04_pair_wait -- python3 tools/dev/pair_bench.py 1048576 > gpurun_out/r04_pair_wait.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04_pair_f -- python3 tools/dev/pair_bench.py 1048576 > gpurun_out/r04_pair_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04_pair_w -- python3 tools/dev/pair_bench.py 1048576 > gpurun_out/r04_pair_w.log 2>&1
echo rc=$?
for d in r04_pair_kt r04_pair_valu r04_pair_wait r04_pair_f r04_pair_w; do python3 tools/prof_summary.py gpurun_out/$d hive_piece > gpurun_out/$d.md 2>&1; done
cat gpurun_out/r04_pair_kt.log | tail -2
for d in r04_pair_kt r04_pair_valu r04_pair_wait r04_pair_f r04_pair_w; do echo "== $d"; cut -c1-220 gpurun_out/$d.md | head -12; done
find gpurun_out/r04_pair_* -name "*.csv" -size +3M -delete
