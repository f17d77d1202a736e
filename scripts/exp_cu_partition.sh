# Round 6 experiment: TWO / FOUR processes on one MI355X, each restricted to a share of the CUs (HSA_CU_MASK), each running the plain
# one-stream loop -- a spatial partition of the chip against the ForwardPipeline's time-sharing.  Aggregate pairs/s = sum over processes.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6t
rm -rf $O; mkdir -p $O
cd $R
C="python bench.py --pipeline 0 --steps 150 --warmup 10 --no-cpu-baseline --no-op-leg --no-fp32-leg --no-op-timing"
echo "== one process, all CUs, one-stream loop" | tee -a $O/exp_cu_partition.txt
timeout 300 $C 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')" | tee -a $O/exp_cu_partition.txt
echo "== one process, all CUs, pipeline depth 3" | tee -a $O/exp_cu_partition.txt
timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-op-leg --no-fp32-leg --no-op-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms')" | tee -a $O/exp_cu_partition.txt
for parts in 2 4; do
  echo "== $parts processes, 256 / $parts CUs each (HSA_CU_MASK), one-stream loops side by side" | tee -a $O/exp_cu_partition.txt
  per=$((256 / parts))
  for i in $(seq 0 $((parts - 1))); do
    lo=$((i * per)); hi=$((lo + per - 1))
    (HSA_CU_MASK="0:$lo-$hi" timeout 400 $C 2>$O/err_${parts}_$i.txt | tail -1 > $O/part_${parts}_$i.json) &
  done
  wait
  python - <<PY | tee -a $O/exp_cu_partition.txt
import json
tot = 0
for i in range($parts):
    try:
        d = json.load(open("$O/part_${parts}_%d.json" % i)); tot += d["value"]; print("  process", i, round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 3), "ms per step")
    except Exception as e:
        print("  process", i, "failed:", e, open("$O/err_${parts}_%d.txt" % i).read()[-300:])
print("  sum", round(tot, 1), "pairs/s")
PY
done
