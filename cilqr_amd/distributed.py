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
    n_traj, n_rows = B * K * F, R * C
    if rank == dst:
        # the ranks' blocks land side by side in ONE buffer (the gather list are its rows): no concatenation pass on the root
        buf = torch.empty((world, packed.numel()), dtype=packed.dtype, device=packed.device)
        parts = list(buf.unbind(0))
        dist.gather(packed, parts, dst=dst)
    else:
        dist.gather(packed, None, dst=dst)
        return None
    ncs = [p[n_traj + n_rows:n_traj + n_rows + B].to(n_cost.dtype) for p in parts]
    sts = [p[n_traj + n_rows + B:].to(status.dtype) for p in parts]
    trav = buf[:, :n_traj].reshape(world, B, K, F)                 # a view: only the rank dimension is strided
    if derive is not None:
        # 8 travelling columns -> the 10 of a trajectory point, written once into the final tensor
        dt_, wb_ = derive
        full = torch.empty((world, B, K, full_F), dtype=trav.dtype, device=trav.device)
        full[..., 1:7] = trav[..., 0:6]
        full[..., 8:10] = trav[..., 6:8]
        full[..., 0] = torch.arange(K, dtype=trav.dtype, device=trav.device)[None, None, :] * dt_
        full[..., 7] = torch.tan(trav[..., 5]) / wb_
        all_traj = full.reshape(world * B, K, full_F)
    else:
        all_traj = trav.reshape(world * B, K, F)                    # one copy (the rank dimension is strided in the buffer)
    out = {"traj": all_traj, "n_cost": torch.cat(ncs), "status": torch.cat(sts)}
    if not densify:
        out["hist_rows"] = torch.cat([p[n_traj:n_traj + int(c.to(torch.int64).sum().item()) * C].reshape(-1, C)
                                      for p, c in zip(parts, ncs)])
        return out
    H = int(max(int(c.max().item()) for c in ncs)) if B > 0 else 0
    hists = []
    for p, c in zip(parts, ncs):
        c64 = c.to(torch.int64)
        dense = torch.zeros((B, H, C), dtype=cost_hist.dtype, device=cost_hist.device)
        m = torch.arange(H, device=c64.device)[None, :] < c64[:, None]
        dense[m] = p[n_traj:n_traj + int(c64.sum().item()) * C].reshape(-1, C)
        hists.append(dense)
    out["cost_hist"] = torch.cat(hists)
    return out


class GatherThread:
    """The per-job results gather of a rank, off the thread that keeps the solver fed.

    A job's results are complete when its wait returns; packing them, agreeing on the ragged length (a host
    synchronisation), the collective and -- on the root -- unpacking world x the payload then run here, on a thread and (on a
    GPU) a stream of their own, while the caller submits the next jobs.  Every rank must `put` its finished jobs in the same
    order: the ranks' collectives pair up by position in that order.  `on_done(tag)` is called after the gather of `tag`
    (hand the result buffers back to whoever reuses them).  `drain()` blocks until everything put so far is gathered and
    re-raises the first error of the thread; call it before any collective of the caller's own.

    After an error the thread gathers nothing more (its tags keep coming back through `on_done`, so the caller's loop does
    not stall on its own slots) -- but the OTHER ranks are then waiting for this one: inside `dist.gather` on the collective
    path, where only the end of this process (what `drain()` raising leads to in bench.py, and what every launcher turns into
    the end of the job) releases them; in the barrier of `PeerGather` on the one-process path, which the failing rank breaks
    (`threading.BrokenBarrierError` in every waiter).  `on_error(exc)`, if given, is called once, on the gather thread, at the
    first error: the place to tear a process group down early."""

    def __init__(self, device=None, dst=0, derive=None, on_done=None, gather_fn=None, on_error=None):
        """gather_fn(traj, hist, nc, st) -> result replaces the torch.distributed gather (bench.py --multi: the ranks are
        threads of one process and results travel by peer copies, see PeerGather)."""
        import queue
        import threading
        self.device, self.dst, self.derive, self.on_done = device, dst, derive, on_done
        self.on_error = on_error
        self.gather_fn = gather_fn or (lambda t, h, n, s: gather_results(t, h, n, s, dst=self.dst, densify=False, derive=self.derive))
        self.q = queue.Queue()
        self.error = None
        self.busy_s = 0.0
        self.count = 0
        self.last = None        # result of the last gather (root) / None (other ranks)
        self.last_tag = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        import time
        stream = None
        if self.device is not None:
            import torch
            torch.cuda.set_device(self.device)
            stream = torch.cuda.Stream(device=self.device)
        while True:
            item = self.q.get()
            if item is None:
                self.q.task_done()
                return
            tag, traj, hist, nc, st = item
            t0 = time.perf_counter()
            try:
                if self.error is None:
                    if stream is not None:
                        import torch
                        with torch.cuda.stream(stream):
                            res = self.gather_fn(traj, hist, nc, st)
                        stream.synchronize()
                    else:
                        res = self.gather_fn(traj, hist, nc, st)
                    self.last, self.last_tag = res, tag
            except Exception as e:   # noqa: BLE001  (kept for drain(); the tags keep coming back so the local loop goes on)
                first = self.error is None
                self.error = e
                if first and self.on_error is not None:
                    try:
                        self.on_error(e)
                    except Exception:   # noqa: BLE001
                        pass
            self.busy_s += time.perf_counter() - t0
            self.count += 1
            if self.on_done is not None:
                self.on_done(tag)
            self.q.task_done()

    def put(self, tag, traj, hist, nc, st):
        self.q.put((tag, traj, hist, nc, st))

    def drain(self):
        self.q.join()
        if self.error is not None:
            raise self.error

    def reset_stats(self):
        self.busy_s, self.count = 0.0, 0

    def close(self):
        self.q.put(None)
        self.thread.join(30.0)


class PeerGather:
    """Results of N GPUs driven by ONE process (bench.py --multi; the reference's caller is one process, planning_node.cc:9-31)
    brought to the root device without a process group: rank r -- a thread -- copies its block into rows [r B, (r + 1) B) of
    tensors that live on the root device (peer-to-peer copies over xGMI: torch's cross-device copy_), the 8 independent
    trajectory columns, the first max(n_cost) Cost rows, n_cost, status; the root rebuilds time and kappa as gather_results
    does.  Same order rule as the collective: every rank gathers its jobs in the order they finished."""

    def __init__(self, world, B, K, M, root_device=None, derive=(0.1, 1.0), on_root=None, timeout_s=600.0):
        """on_root(result): called on the root between the arrival of every rank's block and the release of the rows for the
        next job -- the only window in which the root tensors hold exactly one job (the returned dict aliases them)."""
        import threading
        import torch
        self.world, self.B, self.K, self.derive = world, B, K, derive
        kw = dict(device=root_device) if root_device is not None else {}
        self.traj = torch.zeros((world * B, K, 10), dtype=torch.float64, **kw)
        self.hist = torch.zeros((world * B, M + 1, 5), dtype=torch.float64, **kw)
        self.nc = torch.zeros(world * B, dtype=torch.int32, **kw)
        self.st = torch.zeros(world * B, dtype=torch.int32, **kw)
        self.barrier = threading.Barrier(world)
        self.timeout_s = timeout_s     # a rank that never arrives breaks the barrier instead of hanging the others
        self.on_root = on_root

    def gather_fn(self, rank):
        import torch

        def fn(traj, hist, nc, st):
            try:
                return body(traj, hist, nc, st)
            except BaseException:
                # a rank that fails must not leave the others waiting in the barrier for ever: breaking it raises
                # BrokenBarrierError in every waiter, now and on every later job (GatherThread keeps the first error)
                self.barrier.abort()
                raise

        def body(traj, hist, nc, st):
            lo, hi = rank * self.B, (rank + 1) * self.B
            cols = [1, 2, 3, 4, 5, 6, 8, 9]
            root = self.traj.device
            self.traj[lo:hi, :, cols] = traj[:, :, cols].to(root, non_blocking=True)
            h = int(nc.max().item()) if self.B > 0 else 0
            self.hist[lo:hi, :h] = hist[:, :h].to(root, non_blocking=True)
            self.hist[lo:hi, h:] = 0.0     # rows past this job's longest history: zero, as every other host path pads them
            self.nc[lo:hi] = nc.to(root, non_blocking=True)
            self.st[lo:hi] = st.to(root, non_blocking=True)
            if traj.is_cuda:
                # the peer copies are ordered on this rank's stream, the writes into the root tensors on the ROOT device's
                # current stream of this thread: both must have drained before the barrier says "every block has landed"
                torch.cuda.current_stream().synchronize()
            if root.type == "cuda":
                torch.cuda.current_stream(root).synchronize()
            self.barrier.wait(self.timeout_s)          # every rank's block of this job has landed
            res = None
            if rank == 0:
                dt_, wb_ = self.derive
                self.traj[:, :, 0] = torch.arange(self.K, dtype=torch.float64, device=root)[None, :] * dt_
                self.traj[:, :, 7] = torch.tan(self.traj[:, :, 6]) / wb_
                if self.traj.is_cuda:
                    torch.cuda.current_stream(root).synchronize()
                res = {"traj": self.traj, "cost_hist": self.hist, "n_cost": self.nc, "status": self.st}
                if self.on_root is not None:
                    self.on_root(res)
            self.barrier.wait(self.timeout_s)          # the root has read / rebuilt before the next job overwrites the rows
            return res
        return fn
