# MFMA-pipe / VALU / LDS counters of the default bench workload (own runs: --pmc with --kernel-trace only)
set -x
rm -rf gpurun_out/pmc_mfma; mkdir -p gpurun_out/pmc_mfma
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
C="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-op-timing --no-op-leg"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 -d $O -o p1 --output-format csv -- $C > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O -o p2 --output-format csv -- $C > /dev/null 2>&1
ls $O
