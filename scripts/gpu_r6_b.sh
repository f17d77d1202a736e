# Round 6, second GPU call: the whole GPU suite, default bench line
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6b
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests_all.txt 2>&1
tail -15 $O/tests_all.txt
grep -E "real motion|near the fp16" $O/tests_all.txt
timeout 600 python bench.py 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
cut -c1-400 $O/bench_default.json; tail -3 $O/bench_err.txt
