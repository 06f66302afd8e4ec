# Round 4: counters of the heads kernels inside a 1024-leaf forward (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/r04_heads_a -- python3 tools/forward_kernels.py 1024 bf16 > gpurun_out/r04_heads_a.log 2>&1; echo a rc=$?
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04_heads_f -- python3 tools/forward_kernels.py 1024 bf16 > gpurun_out/r04_heads_f.log 2>&1; echo f rc=$?
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/r04_heads_b -- python3 tools/forward_kernels.py 1024 bf16 > gpurun_out/r04_heads_b.log 2>&1; echo b rc=$?
for p in a f b; do python3 tools/prof_summary.py gpurun_out/r04_heads_$p "heads" > gpurun_out/r04_heads_$p.md 2>&1; cat gpurun_out/r04_heads_$p.md | cut -c1-150; done
find gpurun_out -path "*r04_heads_*" -name "*.csv" -size +1M -delete
