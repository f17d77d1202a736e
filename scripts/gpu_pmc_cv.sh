set -x
mkdir -p gpurun_out/pmc_cv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cv -o p1 --output-format csv -- $GRAFT_REPO_ROOT/scripts/exp_cv.bin > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cv -o p2 --output-format csv -- $GRAFT_REPO_ROOT/scripts/exp_cv.bin > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_cv
