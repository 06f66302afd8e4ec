cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_wgrad_pmc_f -- python tools/wgrad_bench.py 512 > gpurun_out/r02_wgrad_pmc_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_wgrad_pmc_w -- python tools/wgrad_bench.py 512 > gpurun_out/r02_wgrad_pmc_w.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r02_wgrad_pmc_c -- python tools/wgrad_bench.py 512 > gpurun_out/r02_wgrad_pmc_c.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r02_wgrad_pmc_m -- python tools/wgrad_bench.py 512 > gpurun_out/r02_wgrad_pmc_m.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_selfplay_kt2 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --selfplay-plies 12 --no-whole-games --train-steps 0 --no-overlap --no-cpu-baseline-selfplay > gpurun_out/r02_selfplay_kt2.log 2>&1
echo rc=$?
for d in r02_wgrad_pmc_f r02_wgrad_pmc_w r02_wgrad_pmc_c r02_wgrad_pmc_m; do python tools/prof_summary.py gpurun_out/$d wgrad > gpurun_out/$d.md 2>&1; done
python tools/top_kernels.py gpurun_out/r02_selfplay_kt2 40 > gpurun_out/r02_selfplay_top2.md 2>&1
find gpurun_out/r02_* -name "*.csv" -size +3M -delete
