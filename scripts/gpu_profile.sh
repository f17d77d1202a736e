# rocprofv3 kernel trace + stats of the default bench command (run on the GPU box via gpurun)
set -x
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-leg > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_stdout.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -30
