#!/usr/bin/env python
"""bench.py -- image-pairs/sec of the PWC-Net forward at 448x1024 on N MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py`,
  one rank per GPU.  A step = one PWCDCNet forward over this rank's batch of synthetic
  pairs (default 8 x 448x1024, BASELINE.json configs[1]); inputs are resident in HBM
  before the timed region.  Pairs shard across ranks with no data-path collective
  (weak scaling); one RCCL all-gather of per-rank stats per run.  Rank 0 prints ONE
  JSON line.

Extra objects in the line:
  roofline      dominant kernel (by summed duration): algorithmic flops / HIP-event
                duration measured over the timed region, vs the fp32-MFMA peak;
  roofline_hbm  the cost-volume (+fused warp) kernel against the HBM peak;
  kernels       per-kernel launches / ms per step;
  cpu_baseline  the CPU oracle (oracle/, a port -- the TF reference cannot run) timed on
                the host cores on whole 448x1024 pairs, N=1 / rank 0 only;
  parity        max-abs / EPE of flows_final between the HIP path and the oracle on the
                cpu_baseline pair.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 chip peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU per step")
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--use-dc", action="store_true", help="dense-connection estimator (configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget")
    ap.add_argument("--no-op-timing", action="store_true", help="skip the per-launch HIP events")
    return ap.parse_args()


# HIP events bracket the instrumented kernels in every SAMPLE_EVERY-th step of the timed region only:
# an event pair serialises the launches around it (45 instrumented launches cost 9 % of a step)
SAMPLE_EVERY = 8


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import pwcnet_amd
    from pwcnet_amd import weights as W
    from pwcnet_amd.profiler import OpTimer
    from pwcnet_amd.sharding import gather_stats

    # identical seeded glorot-uniform weights on every rank
    specs = W.conv_specs(use_dc=args.use_dc)
    wts = W.init_weights(specs, seed=0)
    net = pwcnet_amd.PWCDCNet(use_dc=args.use_dc)
    net.load_weights(wts)

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    B, H, Wd = args.batch, args.height, args.width
    im0 = torch.rand((B, H, Wd, 3), generator=g, device=dev, dtype=torch.float32)
    im1 = torch.rand((B, H, Wd, 3), generator=g, device=dev, dtype=torch.float32)

    for _ in range(args.warmup):
        net(im0, im1)
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed profile pass: every launch bracketed by HIP events -> per-kernel table and the
    # dominant kernel.  (An event pair per launch costs ~19 % of the step on MI355X, so the
    # timed region below instruments only the dominant kernel and the correlation/warp
    # kernels -- a handful of launches per step.)
    full = None
    dominant = None
    if not args.no_op_timing:
        full = OpTimer()
        with full:
            for _ in range(min(args.steps, 5)):
                net(im0, im1)
        full_summary = full.summary()
        full_steps = min(args.steps, 5)
        dominant = max(full_summary.items(), key=lambda kv: kv[1]["ms"])[0]

    timer = None if args.no_op_timing else OpTimer(only=(dominant, "cost_volume", "warp_kernel"))
    sync_all()
    t0 = time.perf_counter()
    if timer is not None:
        with timer:
            for i in range(args.steps):
                timer.enabled = (i % SAMPLE_EVERY == 0)
                out = net(im0, im1)
    else:
        for _ in range(args.steps):
            out = net(im0, im1)
    sync_all()
    elapsed = time.perf_counter() - t0

    stats = gather_stats(dict(pairs=float(B * args.steps), seconds=elapsed), dist, dev)
    max_elapsed = max(s["seconds"] for s in stats)
    total_pairs = sum(s["pairs"] for s in stats)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    line = {
        "metric": "image_pairs_per_sec_448x1024" if (H, Wd) == (448, 1024) else f"image_pairs_per_sec_{H}x{Wd}",
        "value": total_pairs / max_elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * max_elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (uniform[0,1) images, seeded glorot-uniform weights; trained weights absent)",
        "config": {
            "workload": (f"batch={B} {H}x{Wd} random-init PWC-Net (PWCDCNet use_dc={args.use_dc}) forward "
                         f"per GPU on {world}xMI355X"
                         + (" = BASELINE.json configs[1]" if (B, H, Wd, args.use_dc) == (8, 448, 1024, False) else "")),
            "global_batch": B * world,
            "height": H,
            "width": Wd,
            "parallelism": f"dp{world}: pairs sharded across ranks, no data-path collective",
        },
    }

    if timer is not None:
        summ = timer.summary()          # events recorded INSIDE the timed region (sampled steps)
        n_sampled = len(range(0, args.steps, SAMPLE_EVERY))
        dd = summ[dominant]
        ach = dd["flops"] / (dd["ms"] * 1e-3) / 1e12
        line["roofline"] = {"kernel": dominant, "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                            "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                            "avg_launch_us": 1e3 * dd["ms"] / dd["launches"],
                            "launches_per_step": dd["launches"] / n_sampled,
                            "flops_per_launch": dd["flops"] / dd["launches"],
                            "measured": f"HIP events around each launch of this kernel in every {SAMPLE_EVERY}th step of the "
                                        f"timed region ({n_sampled} of {args.steps} steps)"}
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and (B, H, Wd, args.use_dc) == (8, 448, 1024, False):
            # HBM bytes per launch from the committed PMC passes of this same workload
            # (scripts/gpu_pmc_traffic.sh; counters cannot be collected from inside the process)
            t = json.load(open(pmc))
            fam = t["kernels"].get(dominant.split("<")[0])
            if fam:
                line["roofline"]["traffic"] = fam["hbm_read_bytes_per_launch"] + fam["hbm_write_bytes_per_launch"]
                line["roofline"]["traffic_unit"] = "bytes per launch (mean over the kernel's launches)"
                line["roofline"]["traffic_source"] = "profiles/pmc_traffic.json: " + t["source"]
        if dominant.startswith("conv3x3_wino"):
            # Winograd F(2x2,3x3) executes 16 multiplies per 2x2 outputs instead of 36: `achieved`
            # counts the ALGORITHMIC (direct-convolution) flops, so it can exceed the MFMA peak
            line["roofline"]["executed_mfma_tflops"] = ach / 2.25
            line["roofline"]["mfma_pipe_frac"] = ach / 2.25 / PEAK_F32_MFMA_TFLOPS
            line["roofline"]["note"] = ("Winograd F(2x2,3x3): algorithmic flops = 2*M*9*Cin*Cout per launch; the MFMA "
                                        "units execute 2.25x fewer (executed_mfma_tflops, mfma_pipe_frac)")
        hb = [(k, d) for k, d in summ.items() if k.startswith(("cost_volume", "warp_kernel"))]
        if hb:
            ms = sum(d["ms"] for _, d in hb)
            by = sum(d["bytes"] for _, d in hb)
            a3 = by / (ms * 1e-3) / 1e9
            line["roofline_hbm"] = {"kernel": "+".join(k for k, _ in hb), "bound": "hbm", "achieved": a3,
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a3 / PEAK_HBM_GBS,
                                    "traffic": None, "ms_per_step": ms / n_sampled,
                                    "per_kernel": {k: {"avg_us": 1e3 * d["ms"] / d["launches"],
                                                       "gbs": d["bytes"] / (d["ms"] * 1e-3) / 1e9}
                                                   for k, d in hb},
                                    "measured": f"HIP events in every {SAMPLE_EVERY}th step of the timed region; bytes = N*h*w*(2C+81)*4 "
                                                "(cost volume), N*h*w*(2C+2)*4 (warp), N*h*w*(3C+2+81)*4 (coarse-level fused warp + cost "
                                                "volume + f0 copy), all 5 pyramid levels"}
        # per-kernel table from the untimed, fully instrumented profile pass
        kernels = {}
        for k, d in full_summary.items():
            kernels[k] = {"launches_per_step": d["launches"] / full_steps,
                          "ms_per_step": d["ms"] / full_steps,
                          "avg_us": 1e3 * d["ms"] / d["launches"],
                          "tflops": (d["flops"] / (d["ms"] * 1e-3) / 1e12) if d["ms"] > 0 else 0.0,
                          "gbs": (d["bytes"] / (d["ms"] * 1e-3) / 1e9) if d["ms"] > 0 else 0.0}
        conv_ms = sum(d["ms"] for k, d in full_summary.items() if k.startswith("conv3x3_mfma"))
        conv_fl = sum(d["flops"] for k, d in full_summary.items() if k.startswith("conv3x3_mfma"))
        if conv_ms > 0:
            a2 = conv_fl / (conv_ms * 1e-3) / 1e12
            line["roofline_all_mfma_convs"] = {"bound": "mfma", "achieved": a2, "peak": PEAK_F32_MFMA_TFLOPS,
                                               "unit": "TFLOP/s", "frac": a2 / PEAK_F32_MFMA_TFLOPS,
                                               "ms_per_step": conv_ms / full_steps,
                                               "measured": "untimed profile pass (every launch instrumented)"}
        line["kernels"] = kernels
        line["gpu_busy_ms_per_step_profile_pass"] = sum(d["ms"] for d in full_summary.values()) / full_steps

    if world == 1 and not args.no_cpu_baseline:
        line.update(cpu_baseline_and_parity(net, wts, args, dev))

    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline_and_parity(net, wts, args, dev):
    """Times the CPU oracle (oracle/: C restatement, OpenMP over the host cores) on whole
    pairs of the bench shape, and compares the HIP forward with it on the first pair."""
    from oracle import oracle as orc
    H, Wd = args.height, args.width
    onet = orc.OraclePWCDCNet(wts, use_dc=args.use_dc)
    rng = np.random.RandomState(4321)
    n_done, t_cpu, first = 0, 0.0, None
    while t_cpu < args.cpu_seconds and n_done < 8:
        a = rng.uniform(0, 1, size=(1, H, Wd, 3)).astype(np.float32)
        b = rng.uniform(0, 1, size=(1, H, Wd, 3)).astype(np.float32)
        t0 = time.perf_counter()
        e_final, _ = onet(a, b)
        t_cpu += time.perf_counter() - t0
        n_done += 1
        if first is None:
            first = (a, b, e_final)
    a, b, e_final = first
    final, _ = net(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    got = final.cpu().numpy()
    return {
        "cpu_baseline": {"value": n_done / t_cpu, "unit": "pairs/s", "cores": orc.num_threads(), "kind": "port",
                         "host_cpu_count": os.cpu_count(),
                         "sample": f"{n_done} whole {H}x{Wd} pair(s), {t_cpu:.1f} s of CPU time; oracle/ C "
                                   "restatement (OpenMP), same weights; the TF-1.8 reference cannot run here"},
        "parity": {"max_abs_flows_final": float(np.abs(got - e_final).max()),
                   "epe": orc.epe(e_final, got), "tolerance": 1e-3,
                   "max_abs_flow_value": float(np.abs(e_final).max())},
    }


if __name__ == "__main__":
    main()
