# Round 5, intermediate: the new / changed GPU tests first, then the whole -m gpu suite, then a bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r5t}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "concat_cost_volume or two_operand or out_of_range or f16x2_direct_physical" 2>&1 | tail -25 > $O/new_tests.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/gpu_tests.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
timeout 300 python scripts/exp_timeline.py 8 > $O/timeline_batch8.txt 2>/dev/null
tail -3 $O/new_tests.txt; tail -3 $O/gpu_tests.txt; cut -c1-300 $O/bench_default.json
