"""HBM-side traffic of the correlation + warp launches of bench.py's op-level leg (roofline_hbm.traffic), from the L2's request-size
counters -- the same counters and arithmetic as scripts/pmc_request_table.py (read bytes = 128 n128 + 64 n64 + 32 n32; write bytes =
64 n64 + 32 (n - n64); calibration profiles/r04_pmc_calibration.txt), collected over `python bench.py --op-leg-only` (every
pyramid level's production launch on cold operand sets, flows ~ N(0, 3^2) px).
usage: python scripts/pmc_op_leg_table.py gpurun_out/<dir>/pmc_op [profiles/pmc_traffic.json] > profiles/rNN_pmc_traffic_op_leg.txt"""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if not name.startswith(("cost_volume", "warp_kernel")):
            continue
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(d, k):
    v = d.get(k, [])
    return sum(v) / len(v) if v else 0.0
print("# rocprofv3 --kernel-trace --pmc <L2 request-size counters> (two passes: reads, writes) -- python bench.py --op-leg-only")
print("# (batch 8 at 448x1024: one launch per pyramid level per 'forward'; per-launch means)")
print(f"{'kernel':60s} {'n':>4s} {'read MB':>9s} {'write MB':>9s}")
tot_r = tot_w = 0.0
per = {}
for name, d in sorted(vals.items()):
    n = len(d.get("TCC_EA0_RDREQ_sum", [])) or len(d.get("TCC_EA0_WRREQ_sum", [])) or 1
    rd = 128 * mean(d, "TCC_EA0_RDREQ_128B_sum") + 64 * mean(d, "TCC_EA0_RDREQ_64B_sum") + 32 * mean(d, "TCC_EA0_RDREQ_32B_sum")
    w64 = mean(d, "TCC_EA0_WRREQ_64B_sum")
    wr = 64 * w64 + 32 * (mean(d, "TCC_EA0_WRREQ_sum") - w64)
    print(f"{name[-60:]:60s} {n:4d} {rd / 1e6:9.1f} {wr / 1e6:9.1f}")
    tot_r += rd; tot_w += wr
    per[name] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "launches_profiled": n}
print(f"{'sum = one forward':60s} {'':>4s} {tot_r / 1e6:9.1f} {tot_w / 1e6:9.1f}")
if len(sys.argv) > 2:
    t = json.load(open(sys.argv[2])) if os.path.exists(sys.argv[2]) else {}
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from pwcnet_amd.profiler import source_stamp
    t["op_leg"] = {"hbm_read_bytes_per_forward": tot_r, "hbm_write_bytes_per_forward": tot_w, "per_kernel": per,
                   "stamp": {"source_sha": source_stamp(), "kernel_symbols": sorted(per)},
                   "source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ{,_64B}_sum (separate passes) over "
                             "`python bench.py --op-leg-only` (scripts/gpu_round5_final.sh); bytes = sum(size x requests)"}
    json.dump(t, open(sys.argv[2], "w"), indent=1)
