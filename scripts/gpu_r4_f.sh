mkdir -p gpurun_out/calib2 gpurun_out/pmc_req
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $GRAFT_REPO_ROOT/gpurun_out/calib2 -o RD --output-format csv -- $GRAFT_REPO_ROOT/scripts/exp_fetch_calib.bin > $GRAFT_REPO_ROOT/gpurun_out/calib2/run_rd.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $GRAFT_REPO_ROOT/gpurun_out/calib2 -o WR --output-format csv -- $GRAFT_REPO_ROOT/scripts/exp_fetch_calib.bin > $GRAFT_REPO_ROOT/gpurun_out/calib2/run_wr.log 2>&1
# the bench's kernels, single stream (per batch-8 launch)
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_req -o RD --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_req -o WR --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_req -o FS --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_req -o WS --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/calib2 gpurun_out/pmc_req | head -40
echo done
