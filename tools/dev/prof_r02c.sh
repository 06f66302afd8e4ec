cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_movegen_pmc_f -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sat-boards 0 --selfplay-plies 0 --no-whole-games --train-steps 0 --no-overlap > gpurun_out/r02_movegen_pmc_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_movegen_pmc_w -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sat-boards 0 --selfplay-plies 0 --no-whole-games --train-steps 0 --no-overlap > gpurun_out/r02_movegen_pmc_w.log 2>&1
echo rc=$?
python tools/prof_summary.py gpurun_out/r02_movegen_pmc_f hive > gpurun_out/r02_movegen_pmc_f.md 2>&1
python tools/prof_summary.py gpurun_out/r02_movegen_pmc_w hive > gpurun_out/r02_movegen_pmc_w.md 2>&1
