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
