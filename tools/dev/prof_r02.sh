cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_movegen_kt -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --selfplay-plies 0 --no-whole-games --train-steps 0 --no-overlap > gpurun_out/r02_movegen_kt.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02_movegen_pmc_a -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --selfplay-plies 0 --no-whole-games --train-steps 0 --no-overlap > gpurun_out/r02_movegen_pmc_a.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_selfplay_kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --selfplay-plies 3 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay > gpurun_out/r02_selfplay_kt.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02_net_pmc_a -- python tools/net_latency.py 1024 > gpurun_out/r02_net_pmc_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/r02_net_pmc_b -- python tools/net_latency.py 1024 > gpurun_out/r02_net_pmc_b.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r02_net_pmc_c -- python tools/net_latency.py 1024 > gpurun_out/r02_net_pmc_c.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_train_kt -- python tools/train_prof.py > gpurun_out/r02_train_kt.log 2>&1
echo profiles rc=$?
for d in r02_movegen_kt r02_movegen_pmc_a r02_net_pmc_a r02_net_pmc_b r02_net_pmc_c; do python tools/prof_summary.py gpurun_out/$d hive > gpurun_out/$d.md 2>&1; done
python tools/prof_summary.py gpurun_out/r02_net_pmc_a conv > gpurun_out/r02_net_pmc_a_conv.md 2>&1
python tools/prof_summary.py gpurun_out/r02_net_pmc_b conv > gpurun_out/r02_net_pmc_b_conv.md 2>&1
python tools/prof_summary.py gpurun_out/r02_net_pmc_c conv > gpurun_out/r02_net_pmc_c_conv.md 2>&1
python tools/prof_summary.py gpurun_out/r02_net_pmc_a resblock > gpurun_out/r02_net_pmc_a_res.md 2>&1
python tools/prof_summary.py gpurun_out/r02_net_pmc_b resblock > gpurun_out/r02_net_pmc_b_res.md 2>&1
python tools/prof_summary.py gpurun_out/r02_net_pmc_c resblock > gpurun_out/r02_net_pmc_c_res.md 2>&1
python tools/top_kernels.py gpurun_out/r02_selfplay_kt 20 > gpurun_out/r02_selfplay_top.md 2>&1
python tools/top_kernels.py gpurun_out/r02_train_kt 25 > gpurun_out/r02_train_top.md 2>&1
find gpurun_out/r02_* -name "*.csv" -size +3M -delete
ls gpurun_out | tail -30
