#!/usr/bin/env python
"""PCIe-inclusive rate: the same workload as bench.py but with inputs and outputs in HOST memory
(the C-ABI stages them through the device).  Reported in DESIGN.md; never bench.py's `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilqr_amd import api, scenario  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sc = scenario.generate("mix11", B, seed=2, workers=os.cpu_count() or 8)
opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
opt.plan(sc)                      # warm-up (allocates the staging buffers)
t = []
for _ in range(3):
    t0 = time.perf_counter()
    opt.plan(sc)
    t.append(time.perf_counter() - t0)
best = min(t)
print(json.dumps({"batch": B, "seconds": t, "solves_per_s_host_memory": B / best,
                  "note": "pageable numpy arrays in, pageable numpy arrays out (traj 268 MB, cost_hist 527 MB)"}))
