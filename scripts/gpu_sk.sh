# Round 5: the small-launch conv kernel: its tests, the model tests, A/B of the forward with / without it, traces.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-sk}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "small_launch" 2>&1 | tail -15 > $O/new_tests.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -15 > $O/model_tests.txt
for b in 8 1 2; do timeout 300 python scripts/exp_ab_model.py small_conv $b 2>&1 | grep -v amdgpu.ids > $O/ab_small_conv_b$b.txt; done
cd /tmp && export TMPDIR=/tmp
for b in 8 1; do
  rm -rf /tmp/kt$b
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$b -o kt -- python $R/bench.py --batch $b --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
  python $R/scripts/kernel_trace_forward.py /tmp/kt$b > $O/forward_trace_b$b.txt 2>&1
done
cd $R
tail -4 $O/new_tests.txt; tail -3 $O/model_tests.txt; cat $O/ab_small_conv_b*.txt; head -1 $O/forward_trace_b8.txt; head -1 $O/forward_trace_b1.txt
