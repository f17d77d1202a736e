"""FETCH_SIZE / WRITE_SIZE calibration factors from the two rocprofv3 --pmc passes over scripts/exp_fetch_calib.bin.
usage: python scripts/fetch_calib_table.py gpurun_out/calib > profiles/rNN_pmc_calibration.txt
factor = true bytes / (counter x 1024): what a counter reading has to be multiplied by for that access pattern."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
TRUE = {"read_lines_dma": 1 << 30, "read_half_dma": 1 << 30, "read_lines_vgpr": 1 << 30, "read_gather_vgpr": 1 << 30,
        "write_lines<false>": 1 << 30, "write_lines<true>": 1 << 30, "write_records336": ((1 << 30) // 336) * 336}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- scripts/exp_fetch_calib.bin; 1 GiB per launch (4x the Infinity Cache)")
print(f"{'kernel':24s} {'true MB':>9s} {'FETCH_SIZE KB':>14s} {'fetch factor':>12s} {'WRITE_SIZE KB':>14s} {'write factor':>12s}")
out = {}
for name, true in TRUE.items():
    d = vals.get(name, {})
    f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", [])
    fk = sum(f) / len(f) if f else float("nan"); wk = sum(w) / len(w) if w else float("nan")
    ff = true / (fk * 1024) if fk and name.startswith("read") else float("nan")
    wf = true / (wk * 1024) if wk and name.startswith("write") else float("nan")
    print(f"{name:24s} {true / 1e6:9.1f} {fk:14.0f} {ff:12.3f} {wk:14.0f} {wf:12.3f}")
    out[name] = {"true_bytes": true, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "fetch_factor": ff, "write_factor": wf}
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
