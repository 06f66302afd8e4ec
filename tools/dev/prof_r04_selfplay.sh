# Round 4: kernel trace of the self-play leg (8 plies x 50 simulations x 1024 games, default fp16 engine)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_sp_kt -- python3 bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --selfplay-plies 8 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay --encode-boards 0 --no-worker --sat-boards 0 --no-both-dtypes --no-reuse > gpurun_out/r04_sp_kt.log 2>&1
echo rc=$?
python3 tools/top_kernels.py gpurun_out/r04_sp_kt 30 > gpurun_out/r04_sp_top.md 2>&1
find gpurun_out/r04_sp_kt -name "*.csv" -size +3M -delete
tail -c 400 gpurun_out/r04_sp_kt.log
cut -c1-160 gpurun_out/r04_sp_top.md
