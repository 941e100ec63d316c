#!/usr/bin/env python
"""VERDICT r03 item 6, the upper bound before building anything: would the cost kernels run faster if the corridor planes came
from the Infinity Cache (MALL, 256 MB) instead of HBM -- what tiling the line search into ~8 k-problem tiles could buy?

k_cost_knots over a batch small enough for its planes to stay MALL-resident (8192 problems: 160 MB of planes + 20 MB of
states) is timed twice at the SAME launch size: back to back (warm: the previous call left the planes in the cache) and with a
2 GB device-to-device copy between the calls (cold: the planes come from HBM, as they do for every round of a 65536-problem
iteration, which streams 1.28 GB of planes).  warm / cold is the most a MALL-resident tile could gain on this kernel.
    python tools/mall_probe.py            (prints one JSON line)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cilqr_amd import api, scenario  # noqa: E402

base = scenario.generate("mix11", 256, seed=3)
dev = torch.device("cuda", 0)
flush_src = torch.empty(2 * 1024 ** 3, dtype=torch.uint8, device=dev)
flush_dst = torch.empty_like(flush_src)
out = {}
for B in (4096, 8192, 16384, 65536):
    rep = (B + 255) // 256
    sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B] if isinstance(v, np.ndarray) and v.shape[:1] == (256,) else v)
          for k, v in base.items()}
    opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
    opt.stage_load(sc)
    opt.stage_init_guess()
    cost_d = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    L = opt.L

    def call():
        rc = L.cilqr_stage_total_cost(opt.h, cost_d.data_ptr(), api.MEM_DEVICE)
        assert rc == api.OK
        torch.cuda.synchronize()

    for _ in range(5):
        call()
    warm, cold = [], []
    for _ in range(30):
        t0 = time.perf_counter(); call(); warm.append(time.perf_counter() - t0)
    for _ in range(30):
        flush_dst.copy_(flush_src); torch.cuda.synchronize()
        t0 = time.perf_counter(); call(); cold.append(time.perf_counter() - t0)
    w, c = float(np.median(warm)), float(np.median(cold))
    out[str(B)] = {"planes_mb": round(B * 51 * 16 * 24 / 1e6, 1), "warm_us": round(w * 1e6, 1), "cold_us": round(c * 1e6, 1),
                   "warm_ns_per_problem": round(w * 1e9 / B, 2), "cold_ns_per_problem": round(c * 1e9 / B, 2),
                   "cold_over_warm": round(c / w, 3)}
    opt.close()
print(json.dumps({"kernel": "k_cost_knots + reduction (cilqr_stage_total_cost, device output)", "what": __doc__.split(chr(10))[0], "by_batch": out}))
