# Round 3 profile passes (run on the GPU box through gpurun; every rocprofv3 pass has the program itself after `--`).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MG="--steps 200 --warmup 20 --no-cpu-baseline --sat-boards 0 --selfplay-plies 0 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay --encode-boards 65536"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_mg_kt -- python3 bench.py $MG > gpurun_out/r03_mg_kt.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r03_mg_valu -- python3 bench.py $MG > gpurun_out/r03_mg_valu.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03_mg_f -- python3 bench.py $MG > gpurun_out/r03_mg_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03_mg_w -- python3 bench.py $MG > gpurun_out/r03_mg_w.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r03_net_a -- python3 tools/net_latency.py 1024 > gpurun_out/r03_net_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/r03_net_b -- python3 tools/net_latency.py 1024 > gpurun_out/r03_net_b.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r03_net_c -- python3 tools/net_latency.py 1024 > gpurun_out/r03_net_c.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_sp_kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --selfplay-plies 8 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay --encode-boards 0 --no-worker > gpurun_out/r03_sp_kt.log 2>&1
echo rc=$?
for d in r03_mg_kt r03_mg_valu r03_mg_f r03_mg_w; do python3 tools/prof_summary.py gpurun_out/$d hive > gpurun_out/$d.md 2>&1; done
for d in r03_net_a r03_net_b r03_net_c; do python3 tools/prof_summary.py gpurun_out/$d resblock > gpurun_out/$d.md 2>&1; done
python3 tools/top_kernels.py gpurun_out/r03_sp_kt 24 > gpurun_out/r03_sp_top.md 2>&1
grep -h "leaf batch\|^| 1024" gpurun_out/r03_net_a.log | head -4
find gpurun_out/r03_* -name "*.csv" -size +3M -delete
ls gpurun_out | grep r03 | head -30
