"""One synchronous call over 65536 problems, cut into k logical shards on ONE GPU (cilqr_multi_* with a device listed
k times): do the shards' latency-bound kernels fill each other's gaps?  gpurun: python tools/sharded_single_call.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cilqr_amd import api, scenario

B = int(os.environ.get("B", 65536))
sc = scenario.generate("mix11", B, seed=2)
cfg = api.default_config(sc["n_steps"])
K, M, cmax = cfg.n_steps + 1, cfg.max_iter, sc["cmax"]
dev = torch.device("cuda:0")
d = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("start", "coarse", "corridor", "ccount")}
left, right = np.ascontiguousarray(sc["left"]), np.ascontiguousarray(sc["right"])
out = dict(traj=torch.zeros((B, K, 10), dtype=torch.float64, device=dev), hist=torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
           nc=torch.zeros(B, dtype=torch.int32, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev), ni=torch.zeros(B, dtype=torch.int32, device=dev))
sol = api.SolutionBatch(api.MEM_DEVICE, 0, out["traj"].data_ptr(), out["hist"].data_ptr(), out["nc"].data_ptr(), out["st"].data_ptr(), out["ni"].data_ptr(), None, None)
ref = None
for k in (1, 2, 3, 4, 6, 8):
    m = api.MultiDeviceOptimizer(cfg, devices=[0] * k, batch_capacity=B, cmax=cmax, max_lane_segments=max(left.shape[0], right.shape[0]))
    one = api.BatchIlqrOptimizer(cfg, batch_capacity=1, cmax=cmax)
    prob = one.make_problem(B, d["start"].data_ptr(), d["coarse"].data_ptr(), d["corridor"].data_ptr(), d["ccount"].data_ptr(), cmax,
                            left.ctypes.data, right.ctypes.data, left.shape[0], right.shape[0], api.MEM_DEVICE)
    torch.cuda.synchronize()
    ts = []
    for it in range(8):
        t0 = time.perf_counter()
        rc = m.solve_raw(prob, sol)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        assert rc == api.OK, rc
    got = out["traj"].clone()
    if ref is None:
        ref = got
    print(f"shards {k}: {1e3 * min(ts[2:]):.2f} ms best, {1e3 * np.mean(ts[2:]):.2f} ms mean -> {B / np.mean(ts[2:]):.0f} solves/s; "
          f"{m.device_bytes() / 1e9:.1f} GB; identical to 1 shard: {bool(torch.equal(got, ref))}", flush=True)
    m.close(); one.close()
