"""Per-kernel MFMA-pipe / VALU / LDS counter table from the passes of scripts/gpu_pmc_mfma.sh.
usage: python scripts/pmc_mfma_table.py gpurun_out/pmc_mfma [pmc_traffic.json to merge mfma_busy into]
SQ_VALU_MFMA_BUSY_CYCLES counts cycles the matrix pipe is busy (summed over the SIMDs it is sampled on),
SQ_BUSY_CU_CYCLES the cycles a CU has waves; their ratio / 4 SIMDs is the MFMA-pipe utilisation while the
kernel is resident (gfx94x formula MfmaUtil = 100 * MFMA_BUSY / (BUSY_CU * 4), reused for gfx950)."""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if "at::" in name or "elementwise" in name or "rocclr" in name:
            continue
        vals[name[-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_F32", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU",
        "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
print("# rocprofv3 --kernel-trace --pmc ... (2 passes, scripts/gpu_pmc_mfma.sh) -- python bench.py --steps 3 --warmup 2 (batch 8, 448x1024)")
print("# sums over all launches of the kernel family; mfma_util = MFMA_BUSY / (BUSY_CU * 4)")
print(f"{'kernel':62s} {'n':>5s} {'mfma_util':>9s} " + " ".join(f"{c[3:][:16]:>16s}" for c in cols))
out = {}
for name, d in sorted(vals.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CU_CYCLES", [0]))):
    n = max(len(v) for v in d.values())
    tot = {c: sum(d.get(c, [0.0])) for c in cols}
    util = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * tot["SQ_BUSY_CU_CYCLES"]) if tot["SQ_BUSY_CU_CYCLES"] else float("nan")
    print(f"{name:62s} {n:5d} {util:9.3f} " + " ".join(f"{tot[c]:16.4g}" for c in cols))
    base = re.sub(r"<.*", "", name.replace("void ", "").strip()).split()[-1]
    o = out.setdefault(base, {"mfma_busy_cycles": 0.0, "busy_cu_cycles": 0.0, "launches": 0})
    o["mfma_busy_cycles"] += tot["SQ_VALU_MFMA_BUSY_CYCLES"]; o["busy_cu_cycles"] += tot["SQ_BUSY_CU_CYCLES"]; o["launches"] += n
for o in out.values():
    o["mfma_util"] = o["mfma_busy_cycles"] / (4 * o["busy_cu_cycles"]) if o["busy_cu_cycles"] else None
if len(sys.argv) > 2:
    j = json.load(open(sys.argv[2]))
    j["mfma_busy"] = out
    j["mfma_busy_source"] = "scripts/gpu_pmc_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), same bench workload"
    json.dump(j, open(sys.argv[2], "w"), indent=1)
