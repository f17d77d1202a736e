# Round 6, first GPU call: the new / changed tests, a default bench line, the correlation-in-context experiment (item 1c)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6a
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "real_motion or near_the_fp16 or lazy_range or fallback or out_of_range or channel_split or bench_json" > $O/tests_model.txt 2>&1
tail -5 $O/tests_model.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "small_launch or stride2 or thin_input or f16x2_direct or resize or final_flow" > $O/tests_ops.txt 2>&1
tail -3 $O/tests_ops.txt
timeout 600 python bench.py 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
bash scripts/gpu_clock_log.sh $O/clock_cv_in_context.log timeout 600 python scripts/exp_cv_in_context.py 8 > $O/exp_cv_in_context.txt 2>&1
cat $O/exp_cv_in_context.txt
grep -E "real motion|near the fp16" $O/tests_model.txt
PYTHONPATH=. timeout 300 python scripts/exp_resize_ab.py 8 2>&1 | grep -v amdgpu.ids | tee $O/exp_resize_ab.txt
