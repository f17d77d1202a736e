# Round 6, baseline of the session: whole GPU suite, default bench line, kernel stats of the forward + the op-level leg, forward trace
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6d
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $O/gpu_tests.txt
cat $O/gpu_tests.txt
timeout 600 python bench.py 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
cut -c1-400 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
C="python $R/bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-op-leg --no-fp32-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- $C > $O/prof_stdout.log 2>&1
python $R/scripts/kernel_stats_table.py /tmp/prof 44 > $O/kernel_stats.txt 2>&1
rm -rf /tmp/kt8
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt8 -o kt -- python $R/bench.py --batch 8 --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
python $R/scripts/kernel_trace_forward.py /tmp/kt8 > $O/forward_trace_b8.txt 2>&1
rm -rf /tmp/opleg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/opleg -o op -- python $R/bench.py --op-leg-only > /dev/null 2>&1
python $R/scripts/kernel_stats_table.py /tmp/opleg 12 > $O/kernel_stats_op_leg.txt 2>&1
head -30 $O/kernel_stats.txt
