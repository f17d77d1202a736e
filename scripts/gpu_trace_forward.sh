# Round 5: kernel trace of single-stream forwards at batch 8 / 1 (per-dispatch durations and gaps), after the quick checks.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "block_cost_volume or bench" 2>&1 | tail -5 > $O/new_tests.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
for b in 8 1; do
  rm -rf /tmp/kt$b
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$b -o kt -- python $R/bench.py --batch $b --steps 4 --warmup 3 --streams 1 --no-cpu-baseline --no-op-timing --no-op-leg --no-fp32-leg > /dev/null 2>&1
  python $R/scripts/kernel_trace_forward.py /tmp/kt$b > $O/forward_trace_b$b.txt 2>&1
done
cd $R
tail -3 $O/new_tests.txt; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print(d["value"], d["ms_per_step"])
h=d["roofline_hbm"]; print(h["frac"], h["us_per_forward"], h["us_per_forward_event_pairs"])
for k,v in h["per_kernel"].items(): print(k, round(v["avg_us"],2), round(v["avg_us_event_pair"],2))
PY
head -3 $O/forward_trace_b8.txt; head -3 $O/forward_trace_b1.txt
