# PMC counters on the Winograd microbenchmark variants (own runs, kernel trace only)
set -x
rm -rf gpurun_out/pmc_wino; mkdir -p gpurun_out/pmc_wino
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc_wino/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 -d $GRAFT_REPO_ROOT/gpurun_out/pmc_wino -o p1 --output-format csv -- $GRAFT_REPO_ROOT/scripts/exp_wino.bin pmc > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_wino -o p2 --output-format csv -- $GRAFT_REPO_ROOT/scripts/exp_wino.bin pmc > $GRAFT_REPO_ROOT/gpurun_out/pmc_wino/p2.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_wino
