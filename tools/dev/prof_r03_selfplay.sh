# Round 3 (late): kernel trace of a self-play window with the leaf-batch economies on (row skipping, equal leaves shared)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_sp2_kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --selfplay-plies 8 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay --encode-boards 0 --no-worker --sat-boards 0 > gpurun_out/r03_sp2_kt.log 2>&1
echo rc=$?
python3 tools/top_kernels.py gpurun_out/r03_sp2_kt 200 > gpurun_out/r03_sp2_top.md 2>&1
find gpurun_out/r03_sp2_kt -name "*.csv" -size +3M -delete
grep -c . gpurun_out/r03_sp2_top.md
