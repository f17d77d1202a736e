# Round 6: ForwardPipeline -- tests, default bench (pipelined), the plain loop, batch 1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6g
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "pipeline or bench_json or out_of_range" > $O/tests_pipeline.txt 2>&1
tail -5 $O/tests_pipeline.txt
timeout 600 python bench.py 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6g/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "one-stream", d.get("value_one_stream"), d.get("ms_per_step_one_stream"), d["config"]["pipeline"])
print("roofline frac", d["roofline"]["frac"], d["roofline"]["frac_executed"], d["roofline"]["measured"][:80])
print("hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"].get("frac_in_step"))
PY
timeout 600 python bench.py --pipeline 0 --no-cpu-baseline --no-op-leg 2>>$O/bench_err.txt | tail -1 > $O/bench_pipeline0.json
timeout 600 python bench.py --pipeline 3 --no-cpu-baseline --no-op-leg 2>>$O/bench_err.txt | tail -1 > $O/bench_pipeline3.json
timeout 600 python bench.py --batch 1 --no-cpu-baseline --no-op-leg 2>>$O/bench_err.txt | tail -1 > $O/bench_b1.json
for f in bench_pipeline0 bench_pipeline3 bench_b1; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', round(d['value'],1), round(d['ms_per_step'],4), d.get('value_one_stream'), d['config'].get('pipeline',{}) and d['config']['pipeline']['depth'])"; done
tail -3 $O/bench_err.txt
