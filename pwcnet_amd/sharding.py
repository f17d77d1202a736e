"""Multi-GPU sharding of image pairs.

Pairs are independent (N is a pure batch dimension in every op; the 20 MB of weights are
replicated), so one process per GPU handles a contiguous slice of the pairs and nothing
crosses GPUs on the data path.  The only collective is one all-gather of a few floats of
per-rank statistics (RCCL over xGMI under the `nccl` backend, gloo on CPU in the tests).
"""
import torch


def shard_range(n_pairs, world, rank):
    """Contiguous [lo, hi) slice of `n_pairs` pairs for `rank`; sizes differ by at most 1,
    low ranks take the remainder."""
    if not (0 <= rank < world) or n_pairs < 0:
        raise ValueError(f"bad shard request: n_pairs={n_pairs} world={world} rank={rank}")
    base, rem = divmod(n_pairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_stats(stats, dist=None, device="cpu"):
    """All-gather a flat {name: float} dict; returns the list of every rank's dict (rank
    order).  `dist` is torch.distributed (initialised) or None for a single process."""
    keys = sorted(stats)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [dict(stats)]
    mine = torch.tensor([float(stats[k]) for k in keys], dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [dict(zip(keys, t.cpu().tolist())) for t in out]


def aggregate_throughput(stats, steps):
    """The whole-job numbers of a bench run from the gathered per-rank {pairs, seconds} dicts: every rank timed
    the same K steps, the job took as long as its slowest rank (MAX over ranks), and processed the SUM of the
    ranks' pairs.  Returns (value in pairs/s, ms per step, total pairs, world size)."""
    if not stats:
        raise ValueError("no rank statistics")
    slowest = max(s["seconds"] for s in stats)
    total = sum(s["pairs"] for s in stats)
    if slowest <= 0 or steps <= 0:
        raise ValueError(f"bad timing: slowest rank {slowest} s over {steps} steps")
    return total / slowest, 1e3 * slowest / steps, total, len(stats)


def allreduce_sum_(flat, dist=None):
    """In-place sum of a flat (gradient) buffer over the ranks -- one collective per training step (RCCL over xGMI
    under `nccl`; gloo in the CPU tests).  Returns the world size (1: nothing done) so that the caller can average."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 1
    dist.all_reduce(flat)
    return dist.get_world_size()


def epe(flows_gt, flows):
    """End-point error, reference losses.py:11-13: mean over (batch, y, x) of the L2 norm of
    the flow difference; both flows unscaled (pixels)."""
    return torch.linalg.vector_norm(flows_gt - flows, ord=2, dim=3).mean()


def gather_flows(flows, n_total, dist=None):
    """All-gather the per-rank (n_r, h, w, 2) flow batches of a contiguous shard_range split
    into the full (n_total, h, w, 2) tensor on every rank (3.67 MB per
    448x1024 pair).  Ranks hold different numbers of pairs: shards are padded to the largest."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert flows.shape[0] == n_total
        return flows
    world = dist.get_world_size()
    sizes = [hi - lo for lo, hi in (shard_range(n_total, world, r) for r in range(world))]
    assert flows.shape[0] == sizes[dist.get_rank()], "shard does not match shard_range"
    # a rank without pairs (n_total < world) only holds a placeholder: every rank pads to the
    # trailing shape of the ranks that do hold pairs, agreed on by one small all-gather
    mine = torch.tensor(list(flows.shape[1:]) if flows.shape[0] > 0 else [0] * (flows.dim() - 1),
                        dtype=torch.int64, device=flows.device)
    shapes = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(shapes, mine)
    held = [tuple(int(v) for v in t.cpu().tolist()) for t, n in zip(shapes, sizes) if n > 0]
    if not held:
        return flows[:0]
    trail = held[0]
    assert all(h == trail for h in held), f"gather_flows: ranks hold different shapes {held}"
    pad = torch.zeros((max(sizes),) + trail, dtype=flows.dtype, device=flows.device)
    if flows.shape[0] > 0:
        pad[:flows.shape[0]] = flows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def evaluate_pairs(forward, load_pair, n_pairs, batch=8, dist=None, device="cpu", gather=False):
    """Sharded evaluation loop (counterpart of the validation block of reference
    train.py:124-131 without TensorFlow): every rank runs `forward(images_0, images_1) ->
    flows_final` on its shard_range slice of the pairs, `load_pair(i) -> (image_0, image_1,
    flow_gt)` as (h,w,3),(h,w,3),(h,w,2) float tensors.  Returns a dict with the global
    pixel-weighted EPE, the per-pair EPEs in pair order and, if `gather`, all predicted flows
    (all pairs must then have one size)."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_range(n_pairs, world, rank)
    per_pair, flows_out = [], []
    err_sum, px_sum = 0.0, 0.0
    i = lo
    while i < hi:
        items = [load_pair(j) for j in range(i, min(i + batch, hi))]
        # pairs of one batch must share a size; split the batch where the size changes
        k = 1
        while k < len(items) and items[k][0].shape == items[0][0].shape:
            k += 1
        items = items[:k]
        im0 = torch.stack([it[0] for it in items]).to(device)
        im1 = torch.stack([it[1] for it in items]).to(device)
        gt = torch.stack([it[2] for it in items]).to(device)
        flows = forward(im0, im1)
        norms = torch.linalg.vector_norm(gt - flows, ord=2, dim=3)          # (k, h, w)
        per_pair.extend(norms.mean(dim=(1, 2)).double().cpu().tolist())
        err_sum += float(norms.double().sum())
        px_sum += float(norms.numel())
        if gather:
            flows_out.append(flows.clone())
        i += k
    stats = gather_stats({"err": err_sum, "px": px_sum, "n": float(hi - lo)}, dist if world > 1 else None, device)
    res = {"epe": sum(s["err"] for s in stats) / max(sum(s["px"] for s in stats), 1.0),
           "pairs": int(sum(s["n"] for s in stats))}
    mine = torch.tensor(per_pair, dtype=torch.float64, device=device).reshape(-1, 1, 1, 1)
    res["per_pair_epe"] = gather_flows(mine, n_pairs, dist if world > 1 else None).reshape(-1).cpu().tolist()
    if gather:
        local = torch.cat(flows_out, dim=0) if flows_out else torch.zeros((0, 1, 1, 2), device=device)
        res["flows"] = gather_flows(local, n_pairs, dist if world > 1 else None)
    return res
