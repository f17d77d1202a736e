"""Round 6: a rocprofv3 --kernel-trace CSV of the PIPELINED bench (pwcnet_amd.ForwardPipeline: whole forwards on streams of their
own): how much of the steady state has one, two, three kernels in flight, each kernel family's durations beside its one-stream
durations (a second trace), and a window of the dispatches in start order with their queues.
usage: kernel_trace_pipeline.py <dir of the pipelined trace> [<dir of a one-stream trace>]"""
import collections, csv, glob, os, re, sys


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = re.sub(r"^void ", "", r["Kernel_Name"])
                k = re.sub(r"\(.*$", "", k)
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def forwards(rows, first="c16pair"):
    return [i for i, r in enumerate(rows) if first in r[2]]


rows = load(sys.argv[1])
st = forwards(rows)
n = len(st)
# steady state: from the start of forward n/2 to the start of the last forward but one
a, b = st[n // 2], st[-2]
t0, t1 = rows[a][0], rows[b][0]
nfw = (n - 2) - n // 2
print(f"{n} forwards in the trace; steady-state window: {nfw} forwards, {(t1 - t0) / 1e3 / nfw:.1f} us per forward")
ev = []
for s, e, k, q in rows:
    if e <= t0 or s >= t1:
        continue
    ev.append((max(s, t0), 1)); ev.append((min(e, t1), -1))
ev.sort()
depth_t = collections.Counter()
cur, prev = 0, t0
for t, d in ev:
    depth_t[cur] += t - prev
    cur += d; prev = t
depth_t[cur] += t1 - prev
tot = t1 - t0
print("kernels in flight: " + "  ".join(f"{k}: {100.0 * v / tot:.1f} %" for k, v in sorted(depth_t.items())))
fam = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in rows:
    if s >= t0 and s < t1:
        fam[k][0] += 1; fam[k][1] += e - s
one = {}
if len(sys.argv) > 2:
    r1 = load(sys.argv[2])
    s1 = forwards(r1)
    f1 = collections.defaultdict(lambda: [0, 0])
    for s, e, k, q in r1[s1[len(s1) // 2]:s1[-1]]:
        f1[k][0] += 1; f1[k][1] += e - s
    one = {k: v[1] / v[0] for k, v in f1.items()}
    nf1 = len(s1) - 1 - len(s1) // 2
    print(f"one-stream trace: {sum(v[1] for v in f1.values()) / 1e3 / nf1:.1f} us of kernels per forward")
print(f"pipelined: {sum(v[1] for v in fam.values()) / 1e3 / nfw:.1f} us of kernel durations per forward (overlapping kernels stretch each other)")
print(f"{'kernel':72s} {'per fwd':>8s} {'avg us':>9s} {'one-stream avg':>15s}")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:72]:72s} {v[0] / nfw:8.1f} {v[1] / v[0] / 1e3:9.1f} {one.get(k, 0) / 1e3:15.1f}")
print("\na window of the steady state (start offset us, duration us, queue, kernel):")
w0 = rows[st[n // 2 + 1]][0]
for s, e, k, q in rows:
    if s >= w0 and s < w0 + 1200e3 // 1:
        pass
cnt = 0
for s, e, k, q in rows:
    if s >= w0 and cnt < 150:
        print(f"{(s - w0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3s}  {k[:90]}")
        cnt += 1
