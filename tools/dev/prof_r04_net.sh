# Round 4: counters of hive_tower72_bf16 (the 72-tile assembly tower) at 1024 boards x 19 blocks per launch, next to the
# launch-per-block chain (resblock_kernel).  Separate --pmc passes (kernel trace / SQ / LDS+waits / L2+L1), no other trace domain.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for form in ${FORMS:-72 0}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_net_kt_$form -- python3 tools/resblock_only.py - $form > gpurun_out/r04_net_kt_$form.log 2>&1; echo kt $form rc=$?
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04_net_a_$form -- python3 tools/resblock_only.py - $form > gpurun_out/r04_net_a_$form.log 2>&1; echo a $form rc=$?
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/r04_net_b_$form -- python3 tools/resblock_only.py - $form > gpurun_out/r04_net_b_$form.log 2>&1; echo b $form rc=$?
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/r04_net_c_$form -- python3 tools/resblock_only.py - $form > gpurun_out/r04_net_c_$form.log 2>&1; echo c $form rc=$?
  for p in kt a b c; do python3 tools/prof_summary.py gpurun_out/r04_net_${p}_$form "hive" > gpurun_out/r04_net_${p}_$form.md 2>&1; done
  cat gpurun_out/r04_net_kt_$form.md gpurun_out/r04_net_a_$form.md gpurun_out/r04_net_b_$form.md gpurun_out/r04_net_c_$form.md
  tail -1 gpurun_out/r04_net_kt_$form.log
done
find gpurun_out -path "*r04_net_*" -name "*.csv" -size +1M -delete
