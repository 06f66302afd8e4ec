# Round 4: kernel trace of the default bench command (the driver's flags), for profiles/r04_movegen_kernel_trace.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_mg_kt -- python3 bench.py --steps 200 --warmup 20 --windows 5 --no-cpu-baseline --no-overlap --selfplay-plies 0 --no-whole-games --train-steps 0 --encode-boards 0 > gpurun_out/r04_mg_kt.log 2>&1
echo rc=$?
python3 tools/prof_summary.py gpurun_out/r04_mg_kt hive > gpurun_out/r04_mg_kt.md; cat gpurun_out/r04_mg_kt.md | cut -c1-160 | head
find gpurun_out/r04_mg_kt -name "*.csv" -size +1M -delete
