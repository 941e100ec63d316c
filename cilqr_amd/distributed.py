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


def gather_results(traj, cost_hist, n_cost, status, dst: int = 0, densify: bool = True, derive=None):
    """ONE gather: every rank packs its results into a single flat fp64 tensor
    (trajectories | live cost-history rows, ragged | n_cost | status) and rank `dst` receives the
    blocks in rank order.  Only the n_cost[b] live rows of each problem's history travel (9 of
    the 201 rows on average; rows >= n_cost are unspecified by the ABI).  The only other
    collective is a 4-byte all-reduce(MAX) that agrees on the padded length of the ragged part.

    Every rank must hold the same per-rank batch size (weak scaling; pad the last shard
    otherwise).  Returns a dict of concatenated tensors on `dst`, None on the other ranks:
    traj, n_cost, status, and the history either dense again (`cost_hist`, as many rows as the
    longest history; densify=True) or ragged as it travelled (`hist_rows` [sum n_cost, 5] in problem
    order, problem b's rows start at cumsum(n_cost)[b-1]; densify=False, no extra passes on rank 0).

    derive=(dt, wheel_base): only the 8 independent columns of a trajectory point travel (x, y, theta, v,
    a, delta, jerk, delta_rate -- SURVEY 8(e)); rank `dst` rebuilds column 0 (time = i dt) and column 7
    (kappa = tan(delta) / wheel_base, TransformToTrajectory cc:771-791).  kappa is then recomputed with
    torch's tan, which may differ from the device's in the last bit.
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    full_F = traj.shape[2]
    if derive is not None:
        traj = traj[:, :, [1, 2, 3, 4, 5, 6, 8, 9]].contiguous()
    B, K, F = traj.shape
    C = cost_hist.shape[2]
    nc = n_cost.to(torch.int64)
    h_loc = int(nc.max().item()) if B > 0 else 0
    live = torch.arange(h_loc, device=nc.device)[None, :] < nc[:, None]          # [B, h_loc]
    rows = cost_hist[:, :h_loc][live]                                             # [R_loc, C], problem-major
    r_max = torch.tensor([rows.shape[0]], dtype=torch.int64, device=nc.device)
    dist.all_reduce(r_max, op=dist.ReduceOp.MAX)
    R = int(r_max.item())
    pad = torch.zeros((R - rows.shape[0], C), dtype=cost_hist.dtype, device=cost_hist.device)
    packed = torch.cat([traj.reshape(-1), rows.reshape(-1), pad.reshape(-1),
                        n_cost.to(torch.float64), status.to(torch.float64)])
    if rank == dst:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, parts, dst=dst)
    else:
        dist.gather(packed, None, dst=dst)
        return None
    n_traj, n_rows = B * K * F, R * C
    trajs, ncs, sts, hists = [], [], [], []
    for p in parts:
        trajs.append(p[:n_traj].reshape(B, K, F))
        ncs.append(p[n_traj + n_rows:n_traj + n_rows + B].to(n_cost.dtype))
        sts.append(p[n_traj + n_rows + B:].to(status.dtype))
    all_traj = torch.cat(trajs)
    if derive is not None:
        dt_, wb_ = derive
        full = torch.empty((all_traj.shape[0], K, full_F), dtype=all_traj.dtype, device=all_traj.device)
        full[:, :, 1:7] = all_traj[:, :, 0:6]
        full[:, :, 8:10] = all_traj[:, :, 6:8]
        full[:, :, 0] = torch.arange(K, dtype=all_traj.dtype, device=all_traj.device)[None, :] * dt_
        full[:, :, 7] = torch.tan(all_traj[:, :, 5]) / wb_
        all_traj = full
    out = {"traj": all_traj, "n_cost": torch.cat(ncs), "status": torch.cat(sts)}
    if not densify:
        out["hist_rows"] = torch.cat([p[n_traj:n_traj + int(c.to(torch.int64).sum().item()) * C].reshape(-1, C)
                                      for p, c in zip(parts, ncs)])
        return out
    H = int(max(int(c.max().item()) for c in ncs)) if B > 0 else 0
    for p, c in zip(parts, ncs):
        c64 = c.to(torch.int64)
        dense = torch.zeros((B, H, C), dtype=cost_hist.dtype, device=cost_hist.device)
        m = torch.arange(H, device=c64.device)[None, :] < c64[:, None]
        dense[m] = p[n_traj:n_traj + int(c64.sum().item()) * C].reshape(-1, C)
        hists.append(dense)
    out["cost_hist"] = torch.cat(hists)
    return out
