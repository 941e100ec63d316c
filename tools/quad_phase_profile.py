#!/usr/bin/env python
"""Phase times inside k_quadratize (tuning build -DCILQR_QUAD_PROFILE: wall-clock stamps per wave).
    python tools/quad_phase_profile.py [batch]        (needs cilqr_amd/lib/variants/libcilqr_hip_quadprof.so)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from cilqr_amd import api, scenario  # noqa: E402

api.LIB_PATH = os.path.join(os.path.dirname(api.LIB_PATH), "variants", "libcilqr_hip_quadprof.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
base = scenario.generate("mix11", 256, seed=3)
rep = (B + 255) // 256
sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B] if isinstance(v, np.ndarray) and v.shape[:1] == (256,) else v)
      for k, v in base.items()}
opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
opt.stage_load(sc)
opt.stage_init_guess()
L = opt.L
L.cilqr_debug_quad_profile.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    opt.stage_quadratize()
L.cilqr_debug_quad_profile(None, 1)
t0 = time.perf_counter()
opt.stage_quadratize()
dt = time.perf_counter() - t0
W = 1 << 16
buf = np.zeros(W * 8, np.uint64)
L.cilqr_debug_quad_profile(buf.ctypes.data, 0)
p = buf.reshape(W, 8).astype(np.float64) * 0.01      # us
p = p[p[:, 5] > 0]
names = ["state, goals, first planes; bounds, Jacobian, early stores, sincos", "corridor (planes x discs)", "lanes (10 searches + 10 planes)"]
print(f"B={B}: stage_quadratize {dt * 1e6:.1f} us for this call; {len(p)} waves recorded")
for k, nm in enumerate(names):
    print(f"  {nm:68s} mean {p[:, k].mean():7.2f}  median {np.median(p[:, k]):7.2f}  p95 {np.quantile(p[:, k], 0.95):7.2f} us")
print(f"  {'whole function':68s} mean {p[:, 5].mean():7.2f}  median {np.median(p[:, 5]):7.2f}  p95 {np.quantile(p[:, 5], 0.95):7.2f} us")
opt.close()
