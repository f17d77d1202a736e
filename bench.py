#!/usr/bin/env python
"""bench.py -- image-pairs/sec of the PWC-Net forward at 448x1024 on N MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  * `--gpus N` with no WORLD_SIZE in the environment: this process re-launches itself as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    bench.py ...` (one rank per GPU, RCCL) and relays rank 0's JSON line -- for N = 1 the
    forward runs in this process (`--spawn` forces the launcher, and with it the nccl
    process group, at N = 1 too);
  * launched BY torch.distributed.run (WORLD_SIZE set): runs as that rank; `--gpus` must
    equal WORLD_SIZE or the run aborts (a silently ignored N measured the wrong thing).
  A step = one PWCDCNet forward over this rank's batch of synthetic pairs (default
  8 x 448x1024 per GPU, BASELINE.json configs[1]); inputs are resident in HBM before the
  timed region.  Round 6: the K steps of the timed region are dealt to the replicas of a
  pwcnet_amd.ForwardPipeline (--pipeline D, default 3): step i runs on replica i % D, each
  replica on a HIP stream with a hardware queue of its own, so whole forwards overlap; every
  step is complete inside the bracket.  `value_one_stream` is the same K steps one after the
  other on one stream (the round 1-5 form); the per-kernel legs are measured in THAT loop.  Pairs shard across ranks with no data-path collective (weak scaling);
  one RCCL all-gather of per-rank stats per run.  Rank 0 prints ONE JSON line.
  `--config configs3` = PWCDCNet use_dc=True batch 8; `--config configs4` = 960x1920
  batch 8 per GPU on 2 GPUs (BASELINE.json configs[4]; an explicit `--gpus 1` runs one
  rank's share of it); `--config configs2` = configs[1] per GPU on 8 GPUs.

Extra objects in the line:
  roofline       dominant kernel (by summed duration).  For the Winograd kernel `achieved`
                 is the rate of the multiplies the MFMA units EXECUTE (a fraction of a
                 hardware peak, always <= 1); the direct-convolution ("algorithmic") rate it
                 replaces and the reduction factor are separate keys;
  roofline_hbm   correlation + warp kernels, op-level leg: every pyramid level of the
                 workload, flows ~ N(0, 3^2) px, buffers rotated through
                 > 256 MB so the Infinity Cache does not serve them, HIP events per launch,
                 bytes = N*h*w*(2C+81)*4 (cost volume) + N*h*w*(2C+2)*4 (warp);
  roofline_hbm_in_step  the same kernels as they run inside the timed forward;
  kernels        per-kernel launches / ms per step (untimed, fully instrumented pass);
  cpu_baseline   the CPU oracle (oracle/, a port -- the TF reference cannot run) timed on
                 the host cores on whole pairs of the bench shape, N=1 / rank 0 only;
  parity         max-abs / EPE of flows_final between the HIP path and the oracle on the
                 cpu_baseline pair; parity_moving: the same with every conv kernel scaled up so
                 that the flows, and with them the warps, are several pixels.
"""
import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 chip peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense F16 matrix peak (v_mfma_f32_32x32x16_f16: 32 cycles per instruction and SIMD at 2.4 GHz)
# the matrix pipe each MFMA kernel runs on
KERNEL_PEAK = {"conv3x3_h2_kernel": PEAK_F16_MFMA_TFLOPS, "conv3x3_c16pair_kernel": PEAK_F16_MFMA_TFLOPS}


def peak_of(kernel):
    return KERNEL_PEAK.get(kernel, PEAK_F32_MFMA_TFLOPS)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
# the arithmetic the path computes in (a bare "f32" hides the split)
DTYPE = "f32 via fp16x2-split MFMA (fp32 tensors and accumulation; big stride-1 convs, level-1 extractor and correlation on the F16 pipe)"

CONFIGS = {
    # name -> overrides (BASELINE.json `configs` indices)
    "configs1": {},
    "configs2": {"gpus": 8},
    "configs3": {"use_dc": True},
    "configs4": {"height": 960, "width": 1920, "gpus": 2},
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU per step")
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--use-dc", action="store_true", help="dense-connection estimator (configs[3])")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="preset of BASELINE.json `configs` (overrides height/width/use-dc; configs2 / configs4 also imply "
                         "--gpus 8 / 2 unless --gpus is given)")
    ap.add_argument("--spawn", action="store_true", help="go through torch.distributed.run even for --gpus 1")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST form of --gpus N on a box with fewer GPUs: rank r runs on device r %% (visible GPUs) and the ranks' "
                         "statistics are gathered over gloo (RCCL refuses two ranks on one device).  Exercises the N > 1 branch "
                         "(barriers, per-rank timing, aggregation, rank-0-only legs) on real hardware; its pairs/s is NOT a "
                         "scaling figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget")
    ap.add_argument("--no-op-timing", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--no-op-leg", action="store_true", help="skip the op-level correlation/warp leg")
    ap.add_argument("--op-leg-only", action="store_true",
                    help="run ONLY the op-level correlation/warp leg (what scripts/gpu_pmc_op_leg.sh profiles with rocprofv3 --pmc "
                         "to fill roofline_hbm.traffic) and print its object")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the f16x2=False comparison forward (value_fp32_only)")
    ap.add_argument("--mode", choices=("infer", "train"), default="infer",
                    help="train: one optimisation step per bench step (pwcnet_amd.train.Trainer: forward, backward, "
                         "one RCCL all-reduce of the gradients, Adam) -- SURVEY.md 8 f4, not the headline metric")
    ap.add_argument("--loss", choices=("multiscale", "robust"), default="multiscale", help="--mode train")
    ap.add_argument("--streams", type=int, default=0,
                    help="PWCDCNet(streams=K): the batch runs as K sub-batches on side HIP streams whose kernels overlap; "
                         "0 = the model's default (2 for even batches >= 4 and for 2 large pairs, else 1), 1 = single stream.  The per-kernel "
                         "roofline legs always come from single-stream passes.")
    ap.add_argument("--pipeline", type=int, default=-1,
                    help="pwcnet_amd.ForwardPipeline(depth=D): consecutive steps (whole batches) are dealt to D replicas of the "
                         "model, each on a HIP stream with a hardware queue of its own, so that the launch-bound coarse levels of "
                         "one step run under the matrix-bound launches of another; -1 = 3 (as deep as hardware queues are found); "
                         "0 / 1 = the plain one-stream loop.  The per-kernel legs always come from the one-stream loop.")
    ap.add_argument("--persistent-outputs", action="store_true",
                    help="PWCDCNet(persistent_outputs=True): replays write into the plan's own output tensors")
    args = ap.parse_args(argv)
    raw = sys.argv[1:] if argv is None else list(argv)
    args.gpus_given = any(a == "--gpus" or a.startswith("--gpus=") for a in raw)
    if args.config:
        for k, v in CONFIGS[args.config].items():
            if k == "gpus":
                if not args.gpus_given:      # an explicit --gpus always wins over the preset
                    args.gpus = v
            else:
                setattr(args, k, v)
    return args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn(args):
    """Re-launch this script with one rank per GPU under torch.distributed.run; rank 0's
    stdout (the JSON line) is this process's stdout."""
    n_dev = torch.cuda.device_count()
    if args.gpus > n_dev and not args.share_gpu:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
    cmd += [a for a in sys.argv[1:] if a != "--spawn"]
    if not args.gpus_given:                  # --gpus came from a --config preset: hand it to the ranks
        cmd += ["--gpus", str(args.gpus)]
    return subprocess.call(cmd, env=env)


# HIP events bracket the instrumented kernels in ONE step of every SAMPLE_EVERY of the timed region only:
# an event pair serialises the launches around it (44 instrumented launches make such a step 0.5 ms = 14 % longer;
# measured: 2160 pairs/s with 2 sampled steps of 20, 2189 with none)
SAMPLE_EVERY = 20


def train_mode(args, dist, dev, rank, world):
    """--mode train: K optimisation steps of the training path on this rank's batch (synthetic pairs and flows,
    seeded glorot-uniform weights).  Data-parallel: every rank steps on its own pairs, gradients are summed with one
    all-reduce per step.  Same timing bracket as the inference mode."""
    from pwcnet_amd.sharding import aggregate_throughput, gather_stats
    from pwcnet_amd.train import Trainer
    B, H, Wd = args.batch, args.height, args.width
    tn = Trainer(use_dc=args.use_dc, loss=args.loss, device=str(dev), dist=dist)
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + rank)
    im0 = torch.rand((B, H, Wd, 3), generator=g, device=dev, dtype=torch.float32)
    im1 = torch.rand((B, H, Wd, 3), generator=g, device=dev, dtype=torch.float32)
    gt = torch.randn((B, H, Wd, 2), generator=g, device=dev, dtype=torch.float32) * 3.0
    loss0 = None
    for _ in range(max(args.warmup, 1)):
        v = float(tn.step(im0, im1, gt))
        loss0 = v if loss0 is None else loss0

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tn.step(im0, im1, gt)
    sync_all()
    elapsed = time.perf_counter() - t0
    stats = gather_stats(dict(pairs=float(B * args.steps), seconds=elapsed), dist, dev)
    if rank == 0:
        value, ms_per_step, _, _ = aggregate_throughput(stats, args.steps)
        print(json.dumps({
            "metric": f"training_image_pairs_per_sec_{H}x{Wd}",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (uniform[0,1) images, N(0, 3^2) px flows, seeded glorot-uniform weights)",
            "config": {"workload": f"training step (forward + backward + Adam), batch={B} pairs per GPU, {H}x{Wd}, "
                                   f"PWCDCNet use_dc={args.use_dc}, {args.loss} loss, on {world}xMI355X",
                       "global_batch": B * world, "per_gpu_batch": B, "height": H, "width": Wd,
                       "parallelism": f"dp{world}: pairs sharded across ranks"
                                      + ("" if dist is None else "; one RCCL all-reduce of the 20 MB gradient buffer per step")},
            "loss_first_step": loss0, "loss_last_step": float(loss),
            "peak_memory_gib": torch.cuda.max_memory_allocated() / 2 ** 30,
        }))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus > 1 or args.spawn):
        sys.exit(spawn(args))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python bench.py --gpus {args.gpus}` (it spawns the ranks itself) or pass the "
                         "matching --nproc-per-node")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if env_world is not None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev_index = local_rank % max(1, torch.cuda.device_count()) if args.share_gpu else local_rank
        torch.cuda.set_device(dev_index)
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the ONE JSON
        # line (file descriptor 1 points at stderr while the process group comes up)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.share_gpu:
                dist.init_process_group("gloo")           # (test form: ranks may share a device, which RCCL refuses)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl = RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            try:                                    # the banner sits in C stdio's buffer when stdout is a pipe
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    if args.mode == "train":
        if args.share_gpu:
            raise SystemExit("bench.py: --share-gpu is a test form of the inference bench only (the training step's all-reduce "
                             "runs on device tensors: RCCL)")
        return train_mode(args, dist, dev, rank, world)

    import pwcnet_amd
    from pwcnet_amd import weights as W
    from pwcnet_amd.profiler import OpTimer
    from pwcnet_amd.sharding import aggregate_throughput, gather_stats

    # identical seeded glorot-uniform weights on every rank
    specs = W.conv_specs(use_dc=args.use_dc)
    wts = W.init_weights(specs, seed=0)
    depth = args.pipeline if args.pipeline >= 0 else 3
    if args.persistent_outputs or args.streams > 1:
        depth = 1                       # (plan-owned outputs / sub-batch streams: the round 2-5 forms, one forward at a time)
    pipe = None
    if depth >= 2:
        from pwcnet_amd.pipeline import ForwardPipeline
        pipe = ForwardPipeline(depth=depth, device=dev, use_dc=args.use_dc)
        pipe.load_weights(wts)
        net = pipe.nets[0]              # the one-stream legs (per-kernel events, op leg, fp32 leg) run on replica 0
    else:
        net = pwcnet_amd.PWCDCNet(use_dc=args.use_dc, persistent_outputs=args.persistent_outputs,
                                  streams=args.streams if args.streams > 0 else None)
        net.load_weights(wts)
    eff_streams = 1 if args.persistent_outputs else net.effective_streams((args.batch, args.height, args.width, 3))
    if args.op_leg_only:
        if rank == 0:
            print(json.dumps({"roofline_hbm": corr_warp_op_leg(net, args.batch, args.height, args.width, dev)}))
        return

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    B, H, Wd = args.batch, args.height, args.width
    im0 = torch.rand((B, H, Wd, 3), generator=g, device=dev, dtype=torch.float32)
    im1 = torch.rand((B, H, Wd, 3), generator=g, device=dev, dtype=torch.float32)

    for _ in range(args.warmup):
        net(im0, im1)
    torch.cuda.synchronize()
    if pipe is not None:
        # every replica records its launch plan on its first forward and replays it from the second on: two untimed forwards
        # per lane (on top of the W warm-up steps above, which ran on replica 0 on this stream)
        for _ in range(2 * depth):
            pipe.submit(im0, im1)
        torch.cuda.synchronize()
        if getattr(pipe, "effective_depth", 1) < 2:
            pipe = None                 # no second hardware queue was found: the plain loop
    # the model's own verdict on its side streams (device-timed queue probe); no vetted stream -> it ran single-stream
    ss_report = net.side_stream_report
    if eff_streams > 1 and (ss_report is None or ss_report.get("verdict") != "vetted"):
        eff_streams = 1
    overlapped = eff_streams > 1      # HIP events on the caller's stream do not bracket side-stream kernels

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed profile pass: every launch bracketed by HIP events -> per-kernel table and the
    # dominant kernel.  (An event pair per launch costs ~19 % of the step on MI355X, so the
    # timed region below instruments only the dominant kernel and the correlation/warp
    # kernels -- a handful of launches per step.)
    full_summary = None
    dominant = None
    full_steps = min(args.steps, 5)
    if not args.no_op_timing and full_steps > 0:
        keep_streams, net.streams = net.streams, 1      # per-kernel durations: single-stream form of the same forward
        for _ in range(2):
            net(im0, im1)
        torch.cuda.synchronize()
        full = OpTimer()
        with full:
            for _ in range(full_steps):
                net(im0, im1)
        full_summary = full.summary()
        dominant = max(full_summary.items(), key=lambda kv: kv[1]["ms"])[0]
        net.streams = keep_streams
        for _ in range(2):
            net(im0, im1)
        torch.cuda.synchronize()

    # events inside the timed region only when its kernels run one after the other on the caller's stream
    timer = None if (dominant is None or overlapped) else OpTimer(only=(dominant, "cost_volume", "warp_kernel"))
    # sampled steps: the middle one of every SAMPLE_EVERY (at least one)
    sampled = set(i for i in range(args.steps) if i % SAMPLE_EVERY == SAMPLE_EVERY // 2) or {args.steps - 1}

    def one_stream_loop():
        """K steps one after the other on this stream, HIP events around the dominant kernel's and the correlation's launches in
        the sampled steps -> (seconds, host seconds to issue)"""
        sync_all()
        t0 = time.perf_counter()
        if timer is not None:
            with timer:
                for i in range(args.steps):
                    timer.enabled = i in sampled
                    out = net(im0, im1)
        else:
            for _ in range(args.steps):
                out = net(im0, im1)
        issue = time.perf_counter() - t0              # host time to ISSUE the K forwards (the GPU runs behind)
        sync_all()
        return time.perf_counter() - t0, issue

    one_stream = None
    if pipe is not None:
        # THE TIMED REGION: K steps, step i on replica i % depth (each a whole batch through the whole network); the bracket is the
        # contract's -- synchronize + barrier on both sides, every step complete inside it
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ticket = pipe.submit(im0, im1)
        issue_elapsed = time.perf_counter() - t0
        sync_all()
        elapsed = time.perf_counter() - t0
        del ticket
        st_rep = pipe.status()
    else:
        elapsed, issue_elapsed = one_stream_loop()
        st_rep = net.status()                         # the kernels' status words (fp16 range / stream-K): nothing may have fired

    stats = gather_stats(dict(pairs=float(B * args.steps), seconds=elapsed, issue_seconds=issue_elapsed), dist,
                         "cpu" if args.share_gpu else dev)
    value, ms_per_step, total_pairs, n_ranks = aggregate_throughput(stats, args.steps)
    assert n_ranks == world
    # every rank's own step time next to the job's (= the slowest rank's): a straggler shows as a rank, not as a mystery
    per_rank_ms = [1e3 * st["seconds"] / args.steps for st in stats]
    per_rank_issue_ms = [1e3 * st["issue_seconds"] / args.steps for st in stats]
    used_rccl = dist is not None
    if dist is not None:
        # every rank leaves the process group together, before rank 0's single-GPU legs (op-level leg, CPU baseline)
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    if pipe is not None:
        # the plain loop beside it: K more steps one after the other on one stream (rank 0), with the HIP events the per-kernel
        # legs need -- in the pipelined region the kernels of two steps overlap and a launch has no duration of its own
        e1, i1 = one_stream_loop()
        one_stream = {"value": B * args.steps / e1, "ms_per_step": 1e3 * e1 / args.steps,
                      "host_issue_ms_per_step": 1e3 * i1 / args.steps}
        net.status()
    cfg_idx = None
    if (B, H, Wd) == (8, 448, 1024):
        cfg_idx = 3 if args.use_dc else (1 if world == 1 else (2 if world == 8 else None))
    elif (B, H, Wd, args.use_dc, world) == (8, 960, 1920, False, 2):
        cfg_idx = 4
    line = {
        "metric": "image_pairs_per_sec_448x1024" if (H, Wd) == (448, 1024) else f"image_pairs_per_sec_{H}x{Wd}",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE if net.f16x2 else "f32",
        "timed_region_ms": 1e3 * elapsed,
        "per_rank_ms_per_step": per_rank_ms,
        "per_rank_host_issue_ms_per_step": per_rank_issue_ms,
        "data": "synthetic (uniform[0,1) images, seeded glorot-uniform weights; trained weights absent)",
        "range_status": {"flags": st_rep["flags"], "f16x2_still_on": st_rep["f16x2"], "range_check": net.range_check,
                         "note": "PWC_STATUS_* bits the F16-pipe kernels raised during the run (0: every operand stayed below "
                                 "65504, no stream-K timeout); a raised bit would have moved the model to the fp32 kernels"},
        "config": {
            "workload": (f"batch={B} pairs per GPU, {H}x{Wd}, random-init PWC-Net (PWCDCNet use_dc={args.use_dc}) "
                         f"forward on {world}xMI355X"
                         + (f" = BASELINE.json configs[{cfg_idx}]" if cfg_idx is not None else "")),
            "global_batch": B * world,
            "per_gpu_batch": B,
            "height": H,
            "width": Wd,
            "parallelism": f"dp{world}: pairs sharded across ranks, no data-path collective"
                           + (("; gloo all-gather of per-rank stats, ranks SHARE GPUs (--share-gpu: a test of the N > 1 branch, not a "
                               "scaling figure)") if (used_rccl and args.share_gpu) else
                              ("; RCCL all-gather of per-rank stats" if used_rccl else "")),
            "outputs": "persistent (plan-owned)" if args.persistent_outputs else "fresh tensors per call",
            "gpus_from": ("--gpus" if getattr(args, "gpus_given", True) or not args.config
                          else f"implied by --config {args.config}"),
            "streams": eff_streams,
            "pipeline": ({"depth": pipe.effective_depth, "asked": depth,
                          "what": "pwcnet_amd.ForwardPipeline: step i runs on replica i % depth of the model (own activations, "
                                  "stream-K workspace and status words), each replica on a HIP stream with a hardware queue of "
                                  "its own; every step is a whole batch through the whole network and all K steps complete "
                                  "inside the timed bracket; results are bit-identical to the one-stream loop "
                                  "(tests/test_gpu_model.py::test_pipeline_matches_single_stream); value_one_stream = the same "
                                  "K steps one after the other on one stream",
                          "streams": pipe.stream_report.get("verdict") if pipe.stream_report else None}
                         if pipe is not None else None),
            "conv3x3_arithmetic": ("fp32 tensors throughout; the big stride-1 layers (pwc_conv3x3_h2_supported) form each fp32 "
                                   "product from exact-to-22-bit fp16 operand pairs on the F16 matrix pipe with fp32 accumulation "
                                   "(more accurate than an fp32 MFMA chain, tolerances unchanged; PWCDCNet(f16x2=False) = fp32 "
                                   "MFMA everywhere), the rest run on v_mfma_f32_16x16x4_f32"),
            "side_streams": (None if ss_report is None else
                             {"verdict": ss_report.get("verdict"), "picked": len(ss_report.get("picked", [])),
                              "rejected": ss_report.get("rejected"), "probes": len(ss_report.get("probes", []))}),
        },
    }

    if one_stream is not None:
        line["value_one_stream"] = one_stream["value"] * (world if world > 1 else 1)
        line["ms_per_step_one_stream"] = one_stream["ms_per_step"]
        line["one_stream_note"] = ("the same K steps one after the other on one stream (PWCDCNet.__call__ in a loop, rank 0"
                                   + (f"; x {world} ranks" if world > 1 else "") + "): the round 1-5 headline form, and the "
                                   "loop the per-kernel roofline legs are measured in")
    if dominant is not None:
        if timer is not None:
            summ = timer.summary()          # events recorded INSIDE the timed region (sampled steps)
            n_sampled = len(sampled)
            where = (f"HIP events around each launch of this kernel in one step of every {SAMPLE_EVERY} of the "
                     + (f"timed region ({n_sampled} of {args.steps} steps)" if pipe is None else
                        f"ONE-STREAM loop of the same {args.steps} steps ({n_sampled} sampled; ms_per_step_one_stream): in the "
                        f"pipelined timed region the kernels of {pipe.effective_depth} steps overlap and a launch has no "
                        "duration of its own"))
        else:
            summ = full_summary             # overlapped timed region: the untimed single-stream profile pass
            n_sampled = full_steps
            where = (f"HIP events around each launch in the UNTIMED single-stream profile pass ({full_steps} forwards of "
                     f"the whole batch, every launch bracketed); the timed region runs the batch as {eff_streams} "
                     "sub-batches on side HIP streams whose kernels overlap, so per-kernel durations exist only here")
        dd = summ[dominant]
        alg = dd["flops"] / (dd["ms"] * 1e-3) / 1e12
        exe = dd["exec_flops"] / (dd["ms"] * 1e-3) / 1e12
        # achieved / frac: ALGORITHMIC flops of the launches (2*M*9*Cin*Cout of the direct convolution) over their
        # durations, against the peak of the pipe the kernel runs on.  What the pipe EXECUTES for them (three fp16 products per
        # multiply-add, padded Cin; Winograd's fewer multiplies) is frac_executed -- the pipe's utilisation, not the work done.
        roof = {"kernel": dominant, "bound": "mfma", "achieved": alg, "peak": peak_of(dominant),
                "unit": "TFLOP/s", "frac": alg / peak_of(dominant), "frac_algorithmic": alg / peak_of(dominant),
                "executed_tflops": exe, "frac_executed": exe / peak_of(dominant), "traffic": None,
                "avg_launch_us": 1e3 * dd["ms"] / dd["launches"],
                "launches_per_step": dd["launches"] / n_sampled,
                "flops_per_launch": dd["exec_flops"] / dd["launches"],
                "algorithmic_tflops": alg,
                "algorithmic_flops_per_launch": dd["flops"] / dd["launches"],
                "algorithmic_over_executed": dd["flops"] / dd["exec_flops"],
                "measured": where,
                "note": "achieved/frac = algorithmic: 2*M*9*Cin*Cout of the direct convolution the launch replaces (SURVEY.md 8d); "
                        "executed_tflops/frac_executed = multiply-adds the MFMA units execute (Winograd: 16 per 2x2 outputs "
                        "and 36 per 4x4 outputs instead of 9 per output, physical Cin; conv3x3_h2_kernel: three fp16 "
                        "products per fp32 multiply-add, on the F16 pipe, against the dense F16 peak)"}
        if dominant == "conv3x3_h2_kernel":
            roof["arithmetic"] = ("fp32 in / fp32 out / fp32 accumulation; every operand as h + 2^-11 m' (h = fp16(x), m' = "
                                  "fp16((x - h) 2^11)), products uh vh + 2^-11 (uh vm' + um' vh) on v_mfma_f32_32x32x16_f16: "
                                  "error 0.1x that of the fp32 F(4x4) kernel it replaces (profiles/r04_exp_h2.txt)")
            roof["sustained_matrix_rate"] = {
                "frac_of_peak": 0.69,
                "why": "a loop of nothing but F16 matrix instructions on all 256 CUs runs at 2398 MHz / 848 W on all-zero operands "
                       "(2.46 PFLOP/s) and at ~1700 MHz under the ~1300 W cap on N(0,1) or split operands = 1.74 PFLOP/s "
                       "(sclk / power log: profiles/r05_exp_h2_micro_clock_power.txt; round 4's 493 ns per 24 instructions, "
                       "profiles/r04_exp_h2_micro.txt); frac_executed against THAT rate = frac_executed / 0.69"}
        caps = {"conv3x3_wino_kernel": 0.83, "conv3x3_wino4_kernel": 0.62}
        if dominant in caps:
            roof["instruction_mix_cap"] = {
                "frac": caps[dominant],
                "why": "on gfx950 no VALU / LDS / memory instruction overlaps an fp32 MFMA of the same SIMD (measured: "
                       "profiles/r03_exp_fp32_mfma_excludes_valu.txt); cap = 32 N_mfma / (32 N_mfma + 6.3 N_valu + 3 N_lds + "
                       "7 N_dma) of the kernel's main loop (DESIGN.md 3.4)"}
        total_ms = sum(v["ms"] for v in summ.values())
        roof["other_mfma_kernels"] = {
            k: {"frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak_of(k), "peak": peak_of(k),
                "frac_executed": v["exec_flops"] / (v["ms"] * 1e-3) / 1e12 / peak_of(k),
                "algorithmic_tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                "avg_launch_us": 1e3 * v["ms"] / v["launches"], "launches_per_step": v["launches"] / n_sampled,
                "share_of_kernel_time": v["ms"] / total_ms}
            for k, v in summ.items() if k != dominant and v.get("exec_flops", 0) > 0 and v["ms"] > 0.05 * total_ms}
        prof = os.path.join(ROOT, "profiles")
        pmc = os.path.join(prof, "pmc_traffic.json")
        if os.path.exists(pmc) and (B, H, Wd, args.use_dc) == (8, 448, 1024, False):
            # HBM bytes per launch from the committed PMC passes of this same workload
            # (scripts/gpu_pmc_traffic.sh; counters cannot be collected from inside the process)
            t = json.load(open(pmc))
            fam = t["kernels"].get(dominant.split("<")[0])
            # the counters are printed only if they were taken from THIS build and this launch pattern: the
            # file carries the hash of the kernel sources it was collected from and the launches per forward of every kernel
            stale = traffic_stale(t.get("stamp"), fam, roof["launches_per_step"])
            if stale:
                roof["traffic_stale"] = True
                roof["traffic_stale_why"] = stale
                fam = None
            elif fam:
                roof["traffic_stale"] = False
                roof["traffic_stamp"] = t.get("stamp")
            if fam:
                roof["traffic"] = fam["hbm_read_bytes_per_launch"] + fam["hbm_write_bytes_per_launch"]
                roof["traffic_unit"] = "bytes per launch (mean over the kernel's launches, each a whole batch)"
                roof["traffic_read"] = fam["hbm_read_bytes_per_launch"]
                roof["traffic_write"] = fam["hbm_write_bytes_per_launch"]
                roof["algorithmic_bytes_per_launch"] = dd["bytes"] / dd["launches"]     # input + weights + output, once each
                roof["traffic_over_algorithmic"] = roof["traffic"] / max(1.0, roof["algorithmic_bytes_per_launch"])
                roof["traffic_source"] = "profiles/pmc_traffic.json: " + t["source"]
            busy = t.get("mfma_busy", {}).get(dominant.split("<")[0])
            if busy:
                roof["mfma_busy_pmc"] = busy
        line["roofline"] = roof
        hb = [(k, d) for k, d in summ.items() if k.startswith(("cost_volume", "warp_kernel"))]
        if hb:
            ms = sum(d["ms"] for _, d in hb)
            by = sum(d["bytes"] for _, d in hb)
            a3 = by / (ms * 1e-3) / 1e9
            line["roofline_hbm_in_step"] = {
                "kernel": "+".join(k for k, _ in hb), "bound": "hbm", "achieved": a3,
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a3 / PEAK_HBM_GBS,
                "ms_per_step": ms / n_sampled,
                "per_kernel": {k: {"avg_us": 1e3 * d["ms"] / d["launches"],
                                   "gbs": d["bytes"] / (d["ms"] * 1e-3) / 1e9} for k, d in hb},
                "measured": where + " (random-init net: flows ~ 0); bytes = N*h*w*(2C+81)*4 (cost volume), N*h*w*(2C+2)*4 (warp), "
                            "N*h*w*(2C+2+81)*4 (fused warp + cost volume), all 5 pyramid levels; the f0 concat "
                            "copy that rides in some launches is NOT counted"}
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

    if not args.no_op_leg:
        line["roofline_hbm"] = corr_warp_op_leg(net, B, H, Wd, dev)
        if "roofline_hbm_in_step" in line:
            # the same launches inside the timed forward (HIP events around them in sampled steps; flows ~ 0 there): VERDICT r5 1c
            line["roofline_hbm"]["frac_in_step"] = line["roofline_hbm_in_step"]["frac"]
            line["roofline_hbm"]["us_per_forward_in_step"] = 1e3 * line["roofline_hbm_in_step"]["ms_per_step"]
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and (B, H, Wd, args.use_dc) == (8, 448, 1024, False):
            t = json.load(open(pmc)).get("op_leg")
            stale = traffic_stale((t or {}).get("stamp"), t, None)
            if t and stale:
                line["roofline_hbm"]["traffic_stale"] = True
                line["roofline_hbm"]["traffic_stale_why"] = stale
            elif t:     # HBM-side bytes of the SAME leg from the committed PMC passes (scripts/gpu_pmc_op_leg.sh)
                hb = line["roofline_hbm"]
                hb["traffic_stale"] = False
                hb["traffic_stamp"] = t.get("stamp")
                hb["traffic"] = t["hbm_read_bytes_per_forward"] + t["hbm_write_bytes_per_forward"]
                hb["traffic_unit"] = "bytes per forward's worth of launches (all five levels)"
                hb["traffic_read"], hb["traffic_write"] = t["hbm_read_bytes_per_forward"], t["hbm_write_bytes_per_forward"]
                hb["traffic_over_algorithmic"] = hb["traffic"] / hb["algorithmic_bytes_per_forward"]
                hb["traffic_per_kernel"] = t.get("per_kernel")
                hb["traffic_source"] = "profiles/pmc_traffic.json: " + t["source"]

    if not args.no_fp32_leg and world == 1 and net.f16x2:
        # the same forward with fp32 MFMA everywhere (PWCDCNet(f16x2=False)): what the split arithmetic buys, in the same line
        net32 = pwcnet_amd.PWCDCNet(use_dc=args.use_dc, persistent_outputs=args.persistent_outputs,
                                    streams=args.streams if args.streams > 0 else None, f16x2=False)
        net32.load_weights(wts)
        for _ in range(max(2, min(args.warmup, 3))):
            net32(im0, im1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o32 = net32(im0, im1)
        torch.cuda.synchronize()
        e32 = time.perf_counter() - t0
        line["value_fp32_only"] = B * args.steps / e32
        line["ms_per_step_fp32_only"] = 1e3 * e32 / args.steps
        o16 = net(im0, im1)
        line["max_abs_flow_diff_vs_fp32_only"] = float((o16[0] - o32[0]).abs().max())
        del net32, o32, o16

    if world == 1 and not args.no_cpu_baseline:
        line.update(cpu_baseline_and_parity(net, wts, args, dev))

    print(json.dumps(line))
    sys.stdout.flush()


def traffic_stale(stamp, fam, launches_per_step):
    """None if the committed counter passes describe this build and launch pattern, else the reason (a string): the file's
    source hash must equal pwcnet_amd.profiler.source_stamp() of the tree bench.py runs from, and the kernel's launches per
    forward in the passes must equal this run's."""
    from pwcnet_amd.profiler import source_stamp
    if not stamp or not stamp.get("source_sha"):
        return "profiles/pmc_traffic.json carries no stamp (collected before round 6)"
    now = source_stamp()
    if stamp["source_sha"] != now:
        return f"collected from kernel sources {stamp['source_sha']}, this run is {now}"
    if launches_per_step is not None and fam is not None and fam.get("launches_per_forward") is not None \
            and abs(fam["launches_per_forward"] - launches_per_step) > 1e-6:
        return f"the passes saw {fam['launches_per_forward']} launches per forward, this run {launches_per_step}"
    return None


def corr_warp_op_leg(net, B, H, Wd, dev, reps=12):
    """Op-level roofline of the correlation + warp kernels: for every pyramid
    level of the workload the production launch sequence of PWCDCNet._forward (level 0: cost
    volume; levels >= 1: warp by flows_up*scales[l], then cost volume) on random features and
    flows ~ N(0, 3^2) level-pixels.  Each launch is bracketed by HIP events on the launch
    stream; the operand sets rotate through > 256 MB so neither L2 nor the Infinity Cache
    holds them between repetitions."""
    from pwcnet_amd import modules as M
    from pwcnet_amd.profiler import OpTimer
    from pwcnet_amd.weights import pyramid_channels
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    chans = pyramid_channels(net.num_levels)
    levels = []
    for l in range(net.output_level + 1):
        h, w = H >> (net.num_levels - l), Wd >> (net.num_levels - l)
        levels.append((l, h, w, chans[l]))
    totals = collections.OrderedDict()
    per_level = {}
    # Round 6: inside the forward a correlation launch follows matrix-bound launches that hold the chip at
    # its power cap (sclk ~1.7 GHz, ~1300 W); among its own kind (the chains below) it runs at ~2.35 GHz / 400 W
    # (profiles/r06_exp_cv_in_context.txt + the sclk / power trace beside it).  Beside each level's own chain the leg therefore
    # times a chain of [conv3x3_h2 128 -> 128 at B x 112 x 256 ; the level's launches] and subtracts the chain of the convolutions
    # alone: what the level ADDS behind a matrix-bound launch (a diagnostic, see the return value).
    import ctypes
    from pwcnet_amd import _lib
    Lc = _lib.lib()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    cx = torch.randn((B, 112, 256, 128), generator=g, device=dev)
    cy = torch.empty((B, 112, 256, 128), device=dev)
    cw = torch.randn((3, 3, 128, 128), generator=g, device=dev) * 0.03
    cb = torch.zeros(128, device=dev)
    cpk = torch.empty(Lc.pwc_conv3x3_h2_packed_floats(128, 128), device=dev)
    _lib.check(Lc.pwc_conv3x3_h2_pack_f32(vp(cw), None, 128, 128, 128, vp(cpk), None))
    cwsf = int(Lc.pwc_conv3x3_h2_workspace_floats(B, 112, 256, 128, 128, 1))
    cws = torch.full((max(cwsf, 1),), -1, dtype=torch.int32, device=dev).view(torch.float32)

    def neighbour():
        _lib.check(Lc.pwc_conv3x3_h2_f32(vp(cx), 128, vp(cpk), vp(cb), vp(cy), 128, B, 112, 256, 128, 128, 1, 1, 0.1,
                                         vp(cws) if cwsf else None, cws.numel() if cwsf else 0, _lib.current_stream()))

    def replay_us(fn, n):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for r in range(n):
                fn(r)
        graph.replay()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        del graph
        return sorted(ts)[2]

    neighbour()
    torch.cuda.synchronize()
    conv_us = None
    try:
        conv_us = replay_us(lambda r: neighbour(), 24)
    except RuntimeError as e:
        print(f"op leg: no graph capture of the neighbour chain ({e})", file=sys.stderr)
    for l, h, w, C in levels:
        # the estimator input buffer of this level, as PWCDCNet lays it out (non-DC geometry)
        lay = net._est_layout(l, B, h, w, C, l > 0, list(range(32)) if l > 0 else None)
        est_cs = lay.n_phys
        # round 6: levels on the three-tensor estimator input write dense [cv 81 | flow 2 | 0] records (PWCDCNet._level_input)
        three = l > 0 and net._three_operand_level(l, B, h, w, C, list(range(32)))
        if three:
            est_cs = net.of_estimators[l].CVX_CS
        set_bytes = 4 * B * h * w * (3 * C + 2 + est_cs)
        nsets = max(2, int(300e6 // set_bytes) + 1)
        nsets = min(nsets, 64)
        sets = []
        for _ in range(nsets):
            f0 = torch.randn((B, h, w, C), generator=g, device=dev)
            f1 = torch.randn((B, h, w, C), generator=g, device=dev)
            # flows_up are px/20 of the full-resolution motion; the warp multiplies by scales[l]
            sc = net.scales[l] if l > 0 else 1.0
            fl = torch.randn((B, h, w, 2), generator=g, device=dev) * (3.0 / sc)
            E = torch.zeros((B, h, w, est_cs), device=dev)
            f1w = torch.empty_like(f1)
            sets.append((f0, f1, fl, E, f1w))

        def run(s):
            f0, f1, fl, E, f1w = s
            v0 = M.View(f0.data_ptr(), C, B, h, w, C)
            v1 = M.View(f1.data_ptr(), C, B, h, w, C)
            Ev = M.View(E.data_ptr(), est_cs, B, h, w, est_cs)
            if three:
                net.cv_layer._run(v0, v1, M.View(E.data_ptr(), est_cs, B, h, w, 81), flow=M.View(fl.data_ptr(), 2, B, h, w, 2),
                                  flow_scale=net.scales[l], concat=True, out_pad_writable=2)
                return
            cv_out = M.sub_view(Ev, lay.offset("cv"), 81)
            f0_dst = M.sub_view(Ev, lay.offset("f0"), C) if "f0" in lay.segments else None
            flv = M.View(fl.data_ptr(), 2, B, h, w, 2)
            net._corr_level(l, v0, v1, flv if l > 0 else None, cv_out, f0_dst, Ev, dev)

        for s in sets[:2]:
            run(s)
        torch.cuda.synchronize()
        lt = OpTimer()
        with lt:
            for r in range(reps):
                run(sets[r % nsets])
        torch.cuda.synchronize()
        lsum = lt.summary()
        ev_ms = sum(d["ms"] for d in lsum.values())
        # the same launches as ONE captured chain over every operand set (a graph replay: no host in the loop, no event
        # between launches -- an event pair around a single small launch adds the pipeline drain and refill around it,
        # 5-10 us on a 6 us kernel); one event pair around the replay, median of 5
        chain_us = None
        try:
            n_chain = max(nsets, 24)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for r in range(n_chain):
                    run(sets[r % nsets])
            graph.replay()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / n_chain)
            chain_us = sorted(ts)[2]
            del graph
        except RuntimeError as e:      # capture refused (a launch path that allocates): the event pairs stay the measure
            print(f"op leg level {l}: no graph capture ({e}); event pairs per launch only", file=sys.stderr)
        ctx_us = None
        if conv_us is not None and chain_us is not None:
            try:
                n_chain = max(nsets, 24)
                ctx_us = replay_us(lambda r: (neighbour(), run(sets[r % nsets])), n_chain) - conv_us
            except RuntimeError as e:
                print(f"op leg level {l}: no graph capture of the in-context chain ({e})", file=sys.stderr)
        level_us = chain_us if chain_us is not None else 1e3 * ev_ms / reps
        for k, d in lsum.items():
            share = d["ms"] / ev_ms if ev_ms > 0 else 1.0 / len(lsum)
            t = totals.setdefault(k, dict(us=0.0, bytes=0.0, launches=0.0, ev_us=0.0, ctx_us=0.0))
            t["us"] += level_us * share                 # a level of several launches: the chain time split by event-pair shares
            t["ctx_us"] += (ctx_us if ctx_us is not None else level_us) * share
            t["bytes"] += d["bytes"] / reps
            t["launches"] += d["launches"] / reps
            t["ev_us"] += 1e3 * d["ms"] / reps
        per_level[l] = dict(h=h, w=w, C=C, sets=nsets, est_buffer_channels=est_cs, dense_cv_records=bool(three),
                            f0_in_buffer=(not three) and "f0" in lay.segments,
                            us_chain=chain_us, us_event_pairs=1e3 * ev_ms / reps, us_in_context=ctx_us,
                            algorithmic_bytes=sum(d["bytes"] for d in lsum.values()) / reps)
        del sets
    # HEADLINE = the chains of the launches themselves (a duration per launch).  The "marginal" figure -- chain of [conv3x3_h2 ; level]
    # minus the chain of the convolutions -- is what a level ADDS to a matrix-bound neighbour: it is a difference of two chains, not a
    # duration (a small launch starts under the neighbour's tail: the difference can be ~0 or negative), so it is reported as a
    # diagnostic of the forward's clock / power state only and no bandwidth is derived from it.
    us = sum(t["us"] for t in totals.values())
    us_marginal = sum(t["ctx_us"] for t in totals.values())
    by = sum(t["bytes"] for t in totals.values())
    ach = by / (us * 1e-6) / 1e9
    for lv in per_level.values():
        lv["us_marginal_behind_conv"] = lv.pop("us_in_context")
        lv["frac"] = (lv["algorithmic_bytes"] / (lv["us_chain"] * 1e-6) / 1e9 / PEAK_HBM_GBS) if lv["us_chain"] else None
    return {"kernel": "+".join(totals), "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBS, "traffic": None,
            "us_per_forward": us,
            "headline": "chains of each level's production launches over rotating operand sets (graph replay, one event pair "
                        "around the chain): a duration per launch",
            "neighbour_conv_us": conv_us,
            "us_per_forward_marginal_behind_conv": us_marginal,
            "marginal_note": "chain of [conv3x3_h2 128 -> 128 ; level] minus the chain of the convolutions alone: what a level adds "
                             "behind a matrix-bound launch in the forward's clock / power state; a difference, not a duration (small "
                             "launches start under the neighbour's tail) -- no bandwidth is derived from it",
            "us_per_forward_event_pairs": sum(t["ev_us"] for t in totals.values()),
            "algorithmic_bytes_per_forward": by,
            "per_kernel": {k: {"avg_us": t["us"] / t["launches"], "avg_us_event_pair": t["ev_us"] / t["launches"],
                               "marginal_us_behind_conv": t["ctx_us"] / t["launches"],
                               "gbs": t["bytes"] / (t["us"] * 1e-6) / 1e9,
                               "launches_per_forward": t["launches"]} for k, t in totals.items()},
            "per_level": per_level,
            "measured": f"op-level leg: production launch sequence of every pyramid level (batch {B}), flows ~ "
                        "N(0,3^2) px, operand sets rotating through > 256 MB; avg_us = a captured chain of the level's "
                        "launches over all its operand sets (>= 24 launches), one HIP event pair around the graph replay, "
                        f"median of 5; avg_us_event_pair = an event pair around every launch ({reps} repetitions; what "
                        "rounds 2-4 reported: includes the pipeline drain and refill around a lone launch); "
                        "bytes = N*h*w*(2C+81)*4 per cost volume + N*h*w*(2C+2)*4 per warp "
                        "(N*h*w*(2C+2+81)*4 for a fused launch); concat-copy bytes not counted (levels whose first conv "
                        "reads features_0 from the pyramid tensor have no such copy)"}


def _physical_cores(cpus):
    """Number of distinct (package, core) pairs among the logical CPUs `cpus` (sysfs topology);
    falls back to len(cpus)."""
    seen = set()
    try:
        for c in cpus:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
    except OSError:
        return max(1, len(cpus))
    return max(1, len(seen))


def cpu_baseline_and_parity(net, wts, args, dev):
    """Times the CPU oracle (oracle/: C restatement, OpenMP over the host cores) on whole
    pairs of the bench shape, and compares the HIP forward with it on the first pair."""
    from oracle import oracle as orc
    H, Wd = args.height, args.width
    # threads = physical cores this process may run on.  TF's intra-op pool would take every
    # logical CPU, but for this OpenMP port the SMT siblings are a loss: measured on the GPU box
    # (2 x 64 cores, 256 logical CPUs) 1.29 pairs/s on 128 threads vs 0.089 pairs/s on 256.
    try:
        usable = sorted(os.sched_getaffinity(0))
    except AttributeError:
        usable = list(range(os.cpu_count() or 1))
    threads = _physical_cores(usable)
    orc.set_num_threads(threads)
    onet = orc.OraclePWCDCNet(wts, use_dc=args.use_dc)
    rng = np.random.RandomState(4321)
    n_done, t_cpu, first, per_pair = 0, 0.0, None, []
    while (t_cpu < args.cpu_seconds and n_done < 8) or n_done < 3:      # at least three repeats: the host is shared
        a = rng.uniform(0, 1, size=(1, H, Wd, 3)).astype(np.float32)
        b = rng.uniform(0, 1, size=(1, H, Wd, 3)).astype(np.float32)
        t0 = time.perf_counter()
        e_final, _ = onet(a, b)
        per_pair.append(time.perf_counter() - t0)
        t_cpu += per_pair[-1]
        n_done += 1
        if first is None:
            first = (a, b, e_final)
    a, b, e_final = first
    final, _ = net(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    got = final.cpu().numpy()
    # Random-init weights give flows of a fraction of a pixel: the warps barely move anything.  Second check with every
    # kernel scaled up (flows of several pixels at every level, as in tests/test_gpu_model.py::test_e2e_large_flows):
    # same pair, same bound.
    import pwcnet_amd
    gain = 1.25 if args.use_dc else 1.35
    w2 = {k: (v * gain).astype(np.float32) if k.endswith("/kernel") else v for k, v in wts.items()}
    net2 = pwcnet_amd.PWCDCNet(use_dc=args.use_dc)
    net2.load_weights(w2)
    e2, _ = orc.OraclePWCDCNet(w2, use_dc=args.use_dc)(a, b)
    g2 = net2(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))[0].cpu().numpy()
    del net2
    # Third check at REAL motion (tests/test_gpu_model.py::test_e2e_real_motion_vs_oracle): the coarsest flow
    # head's bias set to (5.2, -3.1) px/20 -- every level adds its residual to the upsampled flow below, so flows_final is a
    # translation of ~(104, -62) px plus what the (gain 1.3) convs add: > 100 px flows, warps of up to 26 px at level 4.
    w3 = {k: (v * 1.3).astype(np.float32) if k.endswith("/kernel") else v for k, v in wts.items()}
    w3["pwcdcnet/optflow_0/conv2d_5/bias"] = np.asarray((5.2, -3.1), np.float32)
    net3 = pwcnet_amd.PWCDCNet(use_dc=args.use_dc)
    net3.load_weights(w3)
    e3, _ = orc.OraclePWCDCNet(w3, use_dc=args.use_dc)(a, b)
    g3 = net3(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))[0].cpu().numpy()
    st3 = net3.status()
    del net3
    srt = sorted(per_pair)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    return {
        "cpu_baseline": {"value": n_done / t_cpu, "unit": "pairs/s", "cores": orc.num_threads(), "kind": "port",
                         "repeats": n_done, "value_median": 1.0 / med, "value_best": 1.0 / srt[0], "value_worst": 1.0 / srt[-1],
                         "seconds_per_pair": [round(t, 3) for t in per_pair],
                         "host_cpu_count": os.cpu_count(), "cpus_in_affinity_mask": len(usable),
                         "sample": f"{n_done} whole {H}x{Wd} pair(s), {t_cpu:.1f} s of wall time on "
                                   f"{orc.num_threads()} OpenMP threads = one per physical core in this process's "
                                   "affinity mask (SMT siblings measured 14x slower); oracle/ C restatement, same "
                                   "weights; the TF-1.8 reference cannot run here"},
        "parity": {"max_abs_flows_final": float(np.abs(got - e_final).max()),
                   "epe": orc.epe(e_final, got), "tolerance": 1e-3,
                   "max_abs_flow_value": float(np.abs(e_final).max())},
        "parity_moving": {"max_abs_flows_final": float(np.abs(g2 - e2).max()), "epe": orc.epe(e2, g2), "tolerance": 1e-3,
                          "max_abs_flow_value": float(np.abs(e2).max()), "kernel_gain": gain,
                          "note": "same pair, every conv kernel scaled by kernel_gain so that the flows (and the warps) "
                                  "are several pixels"},
        "parity_real_motion": {"max_abs_flows_final": float(np.abs(g3 - e3).max()), "epe": orc.epe(e3, g3), "tolerance": 1e-3,
                               "max_abs_flow_value": float(np.abs(e3).max()), "kernel_gain": 1.3, "coarsest_head_bias": [5.2, -3.1],
                               "f16x2_kept": bool(st3["f16x2"]),
                               "note": "same pair, flows_final of > 100 px (a translation injected through the coarsest flow "
                                       "head's bias, carried up by every level's residual), default kernel routing"},
    }


if __name__ == "__main__":
    main()
