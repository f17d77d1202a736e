"""Per-dispatch view of ONE forward from a rocprofv3 --kernel-trace CSV (round 5): kernel duration and the gap since the
previous dispatch ended, in launch order -- what a forward's time is made of without event pairs in the stream.
usage: kernel_trace_forward.py <dir with *_kernel_trace.csv> [first-kernel substring]"""
import csv, glob, os, re, sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "c16pair"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < 2:
    sys.exit("no two forwards in the trace")
a, b = starts[-2], starts[-1]        # the last complete forward
seg = rows[a:b]
t_k = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
print(f"{len(seg)} dispatches, span {span / 1e3:.1f} us, kernels {t_k / 1e3:.1f} us, gaps {(span - t_k) / 1e3:.1f} us")
prev = None
for i, (s, e, k) in enumerate(seg):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{i:3d} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {k[:110]}")
    prev = e
