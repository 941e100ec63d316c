#!/usr/bin/env python
"""Which mapping of the backward pass is fastest at which launch size: one lane / eight lanes / one wavefront per problem
(cilqr_amd/csrc/backward_core.hpp), forced through CILQR_OPT_TEAM_THRESHOLD / CILQR_OPT_WAVE_THRESHOLD, timed on
cilqr_stage_backward (wall time per call over many calls: ~25 us of it are the lambda upload and the sync, the same for all).
    python tools/bwd_forms_sweep.py [sizes...]   ->  one JSON object"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from cilqr_amd import api, scenario  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768]
base = scenario.generate("mix11", 256, seed=3)
out = {"unit": "us per cilqr_stage_backward call (wall, mean of 40)", "sizes": {}}
BIG = 1 << 30
for B in sizes:
    rep = (B + 255) // 256
    sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B] if isinstance(v, np.ndarray) and v.shape[:1] == (256,) else v)
          for k, v in base.items()}
    opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
    opt.stage_load(sc)
    opt.stage_init_guess()
    opt.stage_quadratize()
    lam = np.ones(B)
    row = {}
    for name, team, wave in (("lane", 0, 0), ("team", BIG, 0), ("wave", BIG, BIG)):
        opt.set_option(api.OPT_TEAM_THRESHOLD, team)
        opt.set_option(api.OPT_WAVE_THRESHOLD, wave)
        for _ in range(5):
            opt.stage_backward(lam)
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            opt.stage_backward(lam)
        row[name] = round((time.perf_counter() - t0) / n * 1e6, 1)
    row["best"] = min(("lane", "team", "wave"), key=lambda k: row[k])
    out["sizes"][str(B)] = row
    opt.close()
    print(B, row, file=sys.stderr)
print(json.dumps(out))
