#!/usr/bin/env python
"""Times cilqr_stage_backward alone at several batch sizes (wall time over many calls).
    python tools/bwd_microbench.py [sizes...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from cilqr_amd import api, scenario  # noqa: E402

if os.environ.get("CILQR_LIB"):
    api.LIB_PATH = os.path.abspath(os.environ["CILQR_LIB"])
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 2048, 65536]
base = scenario.generate("mix11", 256, seed=3)
for B in sizes:
    rep = (B + 255) // 256
    sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B] if isinstance(v, np.ndarray) and v.shape[:1] == (256,) else v)
          for k, v in base.items()}
    opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
    opt.stage_load(sc)
    opt.stage_init_guess()
    opt.stage_quadratize()
    lam = np.ones(B)
    for _ in range(3):
        opt.stage_backward(lam)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        opt.stage_backward(lam)
    dt = (time.perf_counter() - t0) / n
    print(f"{os.path.basename(api.LIB_PATH):28s} B={B:6d}  stage_backward {dt * 1e6:8.1f} us per call (incl. lambda upload + sync)")
    opt.close()
