# PMC counters for the conv kernel (own run, kernel-trace only -- no sys/hip trace domains)
set -x
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc1 --output-format csv -- python $GRAFT_REPO_ROOT/scripts/exp_clock.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/stdout1.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc
