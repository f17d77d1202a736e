R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cover; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-leg --no-op-timing > $O/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: the last 40 % of the launches
n = len(rows); w = rows[int(n * 0.55):int(n * 0.95)]
t0, t1 = w[0][0], max(r[1] for r in w)
ev = []
for s, e, _, _ in w: ev += [(s, 1), (e, -1)]
ev.sort()
cover = 0; conc = 0; last = t0; depth = 0; hist = {}
for t, d in ev:
    if depth > 0: cover += t - last
    hist[depth] = hist.get(depth, 0) + (t - last)
    conc += depth * (t - last); last = t; depth += d
span = t1 - t0
print(f"window {span/1e6:.2f} ms, {len(w)} launches; some kernel running {100*cover/span:.1f} % of the time; mean kernels in flight {conc/span:.2f}")
print("time share by number of kernels in flight:", {k: round(100*v/span, 1) for k, v in sorted(hist.items())})
qs = {}
for s, e, nme, q in w: qs[q] = qs.get(q, 0) + (e - s)
print("busy time per queue id (ms):", {k: round(v/1e6, 2) for k, v in qs.items()})
PY
rm -rf $O/t
