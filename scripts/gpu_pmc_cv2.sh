# PMC counters of the rolling cost-volume microbenchmark (own runs, kernel trace only)
set -x
rm -rf gpurun_out/pmc_cv2; mkdir -p gpurun_out/pmc_cv2
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/scripts/exp_cv2.bin
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_cv2
timeout 120 $B 0 q > $O/plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -d $O -o p1 --output-format csv -- $B 0 q > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $O -o p2 --output-format csv -- $B 0 q > /dev/null 2>&1
ls -R $O | head; cat $O/plain.log
