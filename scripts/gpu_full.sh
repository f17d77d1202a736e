# Round 5: the whole -m gpu suite, default bench line, A/B of the small-launch conv kernel, forward traces.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-full}
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_tests.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
for b in 8 2 1; do timeout 300 python scripts/exp_ab_model.py small_conv $b 2>&1 | grep -v amdgpu.ids > $O/ab_small_conv_b$b.txt; done
cd /tmp && export TMPDIR=/tmp
for b in 8 1; do
  rm -rf /tmp/kt$b
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$b -o kt -- python $R/bench.py --batch $b --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
  python $R/scripts/kernel_trace_forward.py /tmp/kt$b > $O/forward_trace_b$b.txt 2>&1
done
cd $R
tail -3 $O/gpu_tests.txt; cut -c1-200 $O/bench_default.json; cat $O/ab_small_conv_b*.txt; head -1 $O/forward_trace_b8.txt; head -1 $O/forward_trace_b1.txt
