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


def gather_results(traj, cost_hist, n_cost, status, dst: int = 0):
    """ONE gather: every rank packs its results into a single [B_rank, W] fp64 tensor
    (trajectory | live cost-history rows | n_cost | status) and rank `dst` receives the blocks in
    rank order.  The only other collective is a 4-byte all-reduce(MAX) that agrees on the number
    of history rows to ship (rows >= n_cost are unspecified by the ABI, and shipping all
    max_iter+1 rows would quadruple the payload).

    Every rank must hold the same per-rank batch size (weak scaling; pad the last shard
    otherwise).  Returns a dict of concatenated tensors on `dst`, None on the other ranks.
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    B, K = traj.shape[0], traj.shape[1]
    h = n_cost.max().to(torch.int32).reshape(1)
    dist.all_reduce(h, op=dist.ReduceOp.MAX)
    H = int(h.item())
    packed = torch.cat([traj.reshape(B, -1), cost_hist[:, :H].reshape(B, -1),
                        n_cost.to(torch.float64).reshape(B, 1), status.to(torch.float64).reshape(B, 1)], dim=1)
    if rank == dst:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, parts, dst=dst)
    else:
        dist.gather(packed, None, dst=dst)
        return None
    full = torch.cat(parts, dim=0)
    w_traj, w_hist = K * traj.shape[2], H * cost_hist.shape[2]
    return {
        "traj": full[:, :w_traj].reshape(world * B, K, traj.shape[2]),
        "cost_hist": full[:, w_traj:w_traj + w_hist].reshape(world * B, H, cost_hist.shape[2]),
        "n_cost": full[:, w_traj + w_hist].to(n_cost.dtype),
        "status": full[:, w_traj + w_hist + 1].to(status.dtype),
    }
