#!/usr/bin/env python
"""Per-phase cycle counts inside k_tail (tuning build -DCILQR_TAIL_PROFILE prints them per block):
    python tools/tail_phase_profile.py [batch]     needs cilqr_amd/lib/variants/libcilqr_hip_tailprof.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cilqr_amd import api, scenario
api.LIB_PATH = os.path.join(os.path.dirname(api.LIB_PATH), "variants", os.environ.get("CILQR_TAILPROF_LIB", "libcilqr_hip_tailprof.so"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenario.generate("mix11", B, seed=12)
opt = api.BatchIlqrOptimizer(n_steps=50, batch_capacity=B, cmax=16)
g = opt.plan(sc)
print("iterations mean", g["n_iter"].mean(), "max", g["n_iter"].max())
opt.close()
