"""Per-kernel HBM-side traffic of the bench's kernels from the L2's request-SIZE counters (round 4; replaces the FETCH_SIZE x2
rule of thumb with counted request sizes).
usage: python scripts/pmc_request_table.py gpurun_out/pmc_req [profiles/pmc_traffic.json] > profiles/rNN_pmc_traffic.txt
Passes (scripts/gpu_r4_f.sh, `bench.py --streams 1`: every launch is a whole batch-8 launch):
  RD: TCC_EA0_RDREQ_sum, _32B_sum, _64B_sum, _128B_sum     WR: TCC_EA0_WRREQ_sum, _64B_sum     FS: FETCH_SIZE     WS: WRITE_SIZE
read bytes = 128 n128 + 64 n64 + 32 n32;  write bytes = 64 n64 + 32 (n - n64).
Calibration (scripts/exp_fetch_calib.hip, profiles/r04_pmc_calibration.txt): on gfx950 EVERY read request of every pattern
tried (whole lines, 64-byte half-line pieces, scattered 128-byte records; LDS-DMA and VGPR loads) is a 128-byte request, which
FETCH_SIZE tallies at 64 bytes: FETCH_SIZE x 2 = bytes for every pattern; a half-line access pattern really moves its lines
twice when they are evicted in between (1.36 GiB for 1 GiB of 64-byte pieces).  WRITE_SIZE needs no factor."""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if "at::" in name or "elementwise" in name or "rocclr" in name:
            continue
        vals[(name.replace("void ", "").strip()[-60:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(d, k):
    v = d.get(k, [])
    return sum(v) / len(v) if v else 0.0
rows = []
for (name, grid), d in vals.items():
    n = len(d.get("TCC_EA0_RDREQ_sum", [])) or len(d.get("FETCH_SIZE", [])) or 1
    rd = 128 * mean(d, "TCC_EA0_RDREQ_128B_sum") + 64 * mean(d, "TCC_EA0_RDREQ_64B_sum") + 32 * mean(d, "TCC_EA0_RDREQ_32B_sum")
    w64 = mean(d, "TCC_EA0_WRREQ_64B_sum")
    wr = 64 * w64 + 32 * (mean(d, "TCC_EA0_WRREQ_sum") - w64)
    rows.append((rd * n, name, grid, n, rd, wr, mean(d, "FETCH_SIZE") * 1024, mean(d, "WRITE_SIZE") * 1024,
                 mean(d, "TCC_EA0_RDREQ_128B_sum") / max(1.0, mean(d, "TCC_EA0_RDREQ_sum"))))
print("# rocprofv3 --kernel-trace --pmc <L2 request-size counters> (separate passes) -- python bench.py --steps 3 --warmup 2 --streams 1 "
      "(batch 8, 448x1024; per-launch means, every launch a whole batch)")
print(f"{'kernel':62s} {'grid':>9s} {'n':>4s} {'read MB':>9s} {'write MB':>9s} {'128B share':>10s} {'FETCH_SIZEx2 MB':>15s} {'WRITE_SIZE MB':>13s}")
for _, name, grid, n, rd, wr, fs, ws, s128 in sorted(rows, reverse=True):
    print(f"{name:62s} {grid:>9s} {n:4d} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {s128:10.3f} {2 * fs / 1e6:15.1f} {ws / 1e6:13.1f}")
if len(sys.argv) > 2:
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for _, name, grid, n, rd, wr, fs, ws, s128 in rows:
        base = re.sub(r"<.*", "", name).split()[-1]
        fam[base][0] += n; fam[base][1] += n * rd; fam[base][2] += n * wr
    old = json.load(open(sys.argv[2])) if os.path.exists(sys.argv[2]) else {}
    # the stamp: which sources these counters belong to, and how many forwards the passes saw -- bench.py
    # compares both (and the dominant kernel's launches per forward) with the run it is printing and reports the traffic as
    # stale instead of printing it when they differ.  git_sha is filled in by scripts/stamp_pmc.py where a .git exists.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from pwcnet_amd.profiler import source_stamp
    forwards = int(sys.argv[3]) if len(sys.argv) > 3 else (fam.get("conv3x3_c16pair_kernel", [0])[0] or fam.get("resize_kernel", [0])[0] or 0)
    out = {"source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ{,_64B}_sum (separate passes), bytes = sum(size x "
                     "requests); python bench.py --steps 3 --warmup 2 --streams 1 (batch 8, 448x1024, use_dc=False: every launch "
                     "a whole batch-8 launch); calibration profiles/r04_pmc_calibration.txt",
           "stamp": {"source_sha": source_stamp(), "git_sha": None, "forwards_profiled": forwards,
                     "kernel_symbols": sorted(fam)},
           "kernels": {k: {"launches": v[0], "launches_per_forward": (v[0] / forwards if forwards else None),
                           "hbm_read_bytes_per_launch": v[1] / v[0], "hbm_write_bytes_per_launch": v[2] / v[0]}
                       for k, v in fam.items()}}
    for k, v in old.items():            # (the other tables of the file: mfma_busy, op_leg, ...)
        if k not in out:
            out[k] = v
    json.dump(out, open(sys.argv[2], "w"), indent=1)
