# Round 5 evidence, second call: the default bench line with roofline_hbm.traffic from the committed PMC passes, the whole GPU suite,
# smoke(), kernel stats of the forward alone, the batch-8 forward dispatch by dispatch.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final5b
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
C="python $R/bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-op-leg --no-fp32-leg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- $C > $O/prof_stdout.log 2>&1
python $R/scripts/kernel_stats_table.py $O/prof 44 > $O/kernel_stats.txt 2>&1
rm -rf /tmp/kt8
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt8 -o kt -- python $R/bench.py --batch 8 --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
python $R/scripts/kernel_trace_forward.py /tmp/kt8 > $O/forward_trace_b8.txt 2>&1
rm -rf $O/prof
cd $R
cat $O/gpu_tests.txt; tail -2 $O/smoke.txt; cut -c1-160 $O/bench_default.json; head -3 $O/forward_trace_b8.txt
