"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as the classic
--stats table: per kernel calls / total / average / percentage.  Usage:
    python scripts/rocpd_stats.py gpurun_out/prof/bench_results.db [skip_first_n_dispatches]
"""
import sqlite3
import sys


def main(path, skip=0):
    c = sqlite3.connect(path)
    rows = c.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count "
                     "from kernels order by start").fetchall()[skip:]
    agg = {}
    for name, dur, gx, wx, lds, vg, ag in rows:
        a = agg.setdefault(name, [0, 0, 0, 1 << 62, lds, vg, ag])
        a[0] += 1; a[1] += dur; a[2] = max(a[2], dur); a[3] = min(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, {tot/1e6:.3f} ms total kernel time")
    print(f"{'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'lds':>6} {'vgpr':>5} {'agpr':>5}  kernel")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{a[0]:6d} {a[1]/1e3:11.1f} {a[1]/a[0]/1e3:9.2f} {a[3]/1e3:9.2f} {a[2]/1e3:9.2f} {100*a[1]/tot:6.2f} "
              f"{a[4]:6d} {a[5]:5d} {a[6]:5d}  {name[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
