"""Multi-GPU layer: problems are independent, so the batch shards contiguously over ranks with no
data-path communication during the solve and ONE gather of the results at the end
(torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference (mpt0816/Cilqr) is single-process; this replaces nothing there -- it is the
sharding rule SURVEY 8(e) specifies: rank r owns problems [r*B/G, (r+1)*B/G).
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of problems owned by `rank` (first `total % world` ranks get one extra)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_scene(scene: dict, rank: int, world: int) -> dict:
    """Per-problem arrays of a problem-major scene dict restricted to this rank's block."""
    B = scene["coarse"].shape[0]
    lo, hi = shard_range(B, rank, world)
    out = {}
    for k, v in scene.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k not in ("left", "right"):
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def _gather_equal(t, dst, dist):
    """dist.gather of equally shaped tensors; returns the concatenation on dst, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        parts = [t.new_empty(t.shape) for _ in range(world)]
        dist.gather(t, parts, dst=dst)
        return parts
    dist.gather(t, None, dst=dst)
    return None


def gather_results(traj, cost_hist, n_cost, status, dst: int = 0):
    """One collective per output tensor: rank `dst` receives every rank's block, in rank order.

    Every rank must hold the same per-rank batch size (weak scaling; pad the last shard
    otherwise).  cost_hist is trimmed to the longest live history before it travels
    (rows >= n_cost are unspecified by the ABI).  Returns a dict of concatenated tensors on
    `dst`, None on the other ranks.
    """
    import torch
    import torch.distributed as dist

    h = n_cost.max().to(torch.int32).reshape(1)
    dist.all_reduce(h, op=dist.ReduceOp.MAX)
    H = int(h.item())
    hist = cost_hist[:, :H].contiguous()
    parts = {
        "traj": _gather_equal(traj.contiguous(), dst, dist),
        "cost_hist": _gather_equal(hist, dst, dist),
        "n_cost": _gather_equal(n_cost.contiguous(), dst, dist),
        "status": _gather_equal(status.contiguous(), dst, dist),
    }
    if dist.get_rank() != dst:
        return None
    return {k: torch.cat(v, dim=0) for k, v in parts.items()}
