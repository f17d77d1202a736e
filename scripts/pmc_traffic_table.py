"""Per-kernel HBM traffic table from the three rocprofv3 --pmc CSV passes of scripts/gpu_pmc_traffic.sh.
usage: python scripts/pmc_traffic_table.py gpurun_out/pmc_traffic > profiles/rNN_pmc_traffic.txt
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so fetch_MB_x2 applies that correction."""
import collections, csv, glob, os, sys

root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if "at::" in name or "elementwise" in name:
            continue
        k = (name[-52:], r["Grid_Size"])
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} (3 separate passes) -- "
      "python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-op-timing")
print("# per-launch means.  FETCH_SIZE/WRITE_SIZE are in KB; fetch_MB_x2 applies the gfx950 2x correction for wide coalesced reads.")
print(f"{'kernel':62s} {'grid':>9s} {'n':>4s} {'fetch_KB':>10s} {'fetch_MB_x2':>11s} {'write_MB':>9s} {'L2_hit':>6s}")
rows = []
for (name, grid), d in vals.items():
    f = d.get("FETCH_SIZE", [0.0]); w = d.get("WRITE_SIZE", [0.0])
    h = sum(d.get("TCC_HIT_sum", [0.0])); m = sum(d.get("TCC_MISS_sum", [0.0]))
    fk = sum(f) / len(f); wk = sum(w) / len(w)
    rows.append((fk * len(f), name, grid, len(f), fk, wk, h / (h + m) if h + m else float("nan")))
for _, name, grid, n, fk, wk, hit in sorted(rows, reverse=True):
    print(f"{name:62s} {grid:>9s} {n:4d} {fk:10.0f} {fk * 2 / 1024:11.1f} {wk / 1024:9.1f} {hit:6.2f}")

# machine-readable aggregate per kernel family (all grids), read by bench.py for roofline.traffic
import json, re
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for _, name, grid, n, fk, wk, hit in rows:
    base = re.sub(r"<.*", "", name.replace("void ", "").strip()).split()[-1]
    fam[base][0] += n; fam[base][1] += n * fk * 2 * 1024; fam[base][2] += n * wk * 1024
out = {k: {"launches": v[0], "hbm_read_bytes_per_launch": v[1] / v[0], "hbm_write_bytes_per_launch": v[2] / v[0]} for k, v in fam.items()}
if len(sys.argv) > 2:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction; "
                         "python bench.py --steps 3 --warmup 2 (batch 8, 448x1024, use_dc=False)", "kernels": out},
              open(sys.argv[2], "w"), indent=1)
