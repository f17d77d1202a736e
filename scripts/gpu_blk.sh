# Round 5: the block-per-workgroup correlation kernel of the small levels: its tests, the model tests, op leg + bench, timeline.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-blk}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "block_cost_volume" 2>&1 | tail -25 > $O/new_tests.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -15 > $O/model_tests.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
timeout 300 python scripts/exp_timeline.py 8 > $O/timeline_batch8.txt 2>/dev/null
tail -5 $O/new_tests.txt; tail -3 $O/model_tests.txt; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print(d["value"], d["ms_per_step"], json.dumps(d.get("roofline_hbm")))
PY
grep -i "cost_volume" $O/timeline_batch8.txt | head -12
