# Round 6: the three-operand estimator input (dense [cv | flow] records): new tests, the op-level leg, A/B in the forward
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6e
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "three_operands or flow_in_record or two_operand or concat_cost_volume" > $O/tests_ops.txt 2>&1
tail -5 $O/tests_ops.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "three_operand or two_operand or golden or real_motion" > $O/tests_model.txt 2>&1
tail -5 $O/tests_model.txt
timeout 300 python bench.py --op-leg-only 2>$O/op_leg_err.txt | tail -1 > $O/op_leg.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6e/op_leg.json"))["roofline_hbm"]
print("op leg: frac", round(d["frac"],4), "us", round(d["us_per_forward"],1), "marginal", round(d["us_per_forward_marginal_behind_conv"],1))
for k,v in d["per_kernel"].items(): print("  ", k, round(v["avg_us"],2), round(v["marginal_us_behind_conv"],2))
PY
for b in 8 1; do timeout 300 python scripts/exp_ab_model.py three_operand $b 2>&1 | grep -v amdgpu.ids >> $O/exp_ab_three_operand.txt; done
cat $O/exp_ab_three_operand.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
