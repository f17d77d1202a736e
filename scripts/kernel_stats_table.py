"""Print the top rows of a rocprofv3 *kernel_stats.csv: python scripts/kernel_stats_table.py <dir> [rows]"""
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
if not fs:
    raise SystemExit(f"no kernel_stats.csv under {sys.argv[1]}")
rows = list(csv.DictReader(open(fs[0])))
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:nrows]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:9.1f} us  "
          f"{100 * float(r['TotalDurationNs']) / tot:5.1f}%")
