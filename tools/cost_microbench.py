#!/usr/bin/env python
"""Times cilqr_stage_total_cost (k_cost_knots + reduction + 2.6 MB read-back) at full batch size.
    [CILQR_LIB=path/to/variant.so] python tools/cost_microbench.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process: torch first)
from cilqr_amd import api, scenario  # noqa: E402

if os.environ.get("CILQR_LIB"):
    api.LIB_PATH = os.path.abspath(os.environ["CILQR_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
base = scenario.generate("mix11", 256, seed=3)
rep = (B + 255) // 256
sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B] if isinstance(v, np.ndarray) and v.shape[:1] == (256,) else v)
      for k, v in base.items()}
opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
opt.stage_load(sc)
opt.stage_init_guess()
for _ in range(3):
    c = opt.stage_total_cost()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    c = opt.stage_total_cost()
dt = (time.perf_counter() - t0) / n
print(f"{os.path.basename(api.LIB_PATH):24s} B={B}  stage_total_cost {dt * 1e6:8.1f} us per call, mean cost {c[:, 0].mean():.6g}")
opt.close()
