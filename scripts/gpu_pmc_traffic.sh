# HBM traffic counters of the default bench command (separate --pmc passes, kernel trace only)
set -x
mkdir -p gpurun_out/pmc_traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic -o $c --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-op-timing --no-op-leg > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic -o L2HIT --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-op-timing --no-op-leg > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
