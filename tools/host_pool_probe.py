#!/usr/bin/env python
"""Host arrays through the pool: B problems per batch in HOST memory in and out (CILQR_MEM_HOST), `steps` batches through a pool
of `handles` handles (cilqr_pool_submit / cilqr_pool_wait).  Compares with the same stream of batches on device-resident arrays
and checks that both give the same bits.  Prints one JSON object.  usage: host_pool_probe.py [B] [steps] [handles]"""
import json
import os
import resource
import sys
import time

import numpy as np
import torch

torch.cuda.init()      # before the library's first HIP call (the other order leaves torch without a device on these boxes)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilqr_amd import api, scenario  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ORDER = sys.argv[4] if len(sys.argv) > 4 else "host-first"      # which stream of batches is timed first
PINNED = len(sys.argv) > 5 and sys.argv[5] == "pinned"          # the caller's arrays in page-locked memory (hipHostMalloc through torch)


def host_array(shape, dtype, fill=0):
    if not PINNED:
        return np.full(shape, fill, dtype)
    t = torch.empty(tuple(shape) if not isinstance(shape, int) else (shape,), dtype={np.float64: torch.float64, np.int32: torch.int32}[dtype],
                    pin_memory=True)
    v = t.numpy()
    v[...] = fill
    _keep.append(t)
    return v


_keep = []
spec = scenario.SPECS["mix11"]
K, cmax = spec.n_steps + 1, spec.cmax
sc = scenario.generate(spec, B, seed=2, workers=min(32, os.cpu_count() or 8))
cfg = api.default_config(spec.n_steps)
M = cfg.max_iter
left, right = np.ascontiguousarray(sc["left"]), np.ascontiguousarray(sc["right"])
pool = api.HandlePool(cfg, device=0, handles=P, batch_capacity=B, cmax=cmax, max_lane_segments=max(left.shape[0], right.shape[0]))
depth = pool.depth()
a = {}
for k_, dt_np in (("start", np.float64), ("coarse", np.float64), ("corridor", np.float64), ("ccount", np.int32)):
    a[k_] = host_array(sc[k_].shape, dt_np)
    a[k_][...] = sc[k_]
in_bytes = sum(v.nbytes for v in a.values())


class HostSlot:
    def __init__(self):
        self.traj = host_array((B, K, 10), np.float64)
        self.hist = host_array((B, M + 1, 5), np.float64)
        self.nc = host_array(B, np.int32)
        self.st = host_array(B, np.int32)
        self.ni = host_array(B, np.int32)
        self.sol = api.SolutionBatch(api.MEM_HOST, 0, self.traj.ctypes.data, self.hist.ctypes.data, self.nc.ctypes.data,
                                     self.st.ctypes.data, self.ni.ctypes.data, None, None, None)


slots = [HostSlot() for _ in range(depth)]
out_bytes = slots[0].traj.nbytes + slots[0].hist.nbytes + 3 * slots[0].nc.nbytes
prob = api.ProblemBatch(B, K, cmax, api.MEM_HOST, a["start"].ctypes.data, a["coarse"].ctypes.data, a["corridor"].ctypes.data,
                        a["ccount"].ctypes.data, left.shape[0], right.shape[0], left.ctypes.data, right.ctypes.data)


def stream_of(prob_, sols, n):
    sub = col = 0
    for _ in range(n):
        if sub - col == depth:
            assert pool.wait() == api.OK
            col += 1
        assert pool.submit_raw(prob_, sols[sub % depth]) == api.OK
        sub += 1
    while col < sub:
        assert pool.wait() == api.OK
        col += 1


def timed(prob_, sols):
    stream_of(prob_, sols, depth + 1)
    torch.cuda.synchronize()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    stream_of(prob_, sols, STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
    return {"value": round(B * STEPS / dt, 1), "ms_per_step": round(1e3 * dt / STEPS, 3), "cores_busy": round(cpu / dt, 2)}


# the same on device-resident arrays (torch only as the allocator)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(v).to(dev) for k, v in a.items()}
prob_d = api.ProblemBatch(B, K, cmax, api.MEM_DEVICE, d["start"].data_ptr(), d["coarse"].data_ptr(), d["corridor"].data_ptr(),
                          d["ccount"].data_ptr(), left.shape[0], right.shape[0], left.ctypes.data, right.ctypes.data)
dt_ = [dict(traj=torch.zeros((B, K, 10), dtype=torch.float64, device=dev), hist=torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
            nc=torch.zeros(B, dtype=torch.int32, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev),
            ni=torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(depth)]
sol_d = [api.SolutionBatch(api.MEM_DEVICE, 0, s["traj"].data_ptr(), s["hist"].data_ptr(), s["nc"].data_ptr(), s["st"].data_ptr(),
                           s["ni"].data_ptr(), None, None, None) for s in dt_]
torch.cuda.synchronize()
res = {"batch": B, "steps": STEPS, "handles": P, "order": ORDER, "caller_memory": "pinned" if PINNED else "pageable", "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
       "input_bytes": in_bytes, "output_bytes": out_bytes}
for which in (("host_memory", "device_memory") if ORDER == "host-first" else ("device_memory", "host_memory", "device_memory_again")):
    res[which] = timed(prob, [s.sol for s in slots]) if which == "host_memory" else timed(prob_d, sol_d)
g = dt_[0]
nc = g["nc"].cpu().numpy()
live = np.arange(M + 1)[None, :] < nc[:, None]
res["identical"] = bool(np.array_equal(slots[0].traj, g["traj"].cpu().numpy()) and np.array_equal(slots[0].nc, nc)
                        and np.array_equal(slots[0].st, g["st"].cpu().numpy()) and np.array_equal(slots[0].ni, g["ni"].cpu().numpy())
                        and np.array_equal(slots[0].hist[live], g["hist"].cpu().numpy()[live]) and not slots[0].hist[~live].any())
res["live_cost_row_bytes"] = int(nc.sum()) * 40
res["device_bytes"] = pool.device_bytes()
pool.close()
print(json.dumps(res))
