#!/usr/bin/env python
"""(Under tests/: it runs the CPU oracle.)
Parity of the HIP path against the CPU oracle on a larger sample than the test suite uses: for every
scene family, how many problems match within 1e-4 relative (status, iteration count, every Cost row,
final trajectory), how many are ill-conditioned in the oracle itself (a 4e-16 relative perturbation
of the inputs changes the ORACLE's result by more than 1e-5), and the largest deviation among the
well-conditioned ones.  One JSON document on stdout.
    python tests/parity_report.py [problems-per-family]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from cilqr_amd import api, scenario  # noqa: E402
from parity_util import compare_solutions, oracle_cfg_from, oracle_reference, rel_err  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
report = {"tolerance": 1e-4, "problems_per_family": n, "families": {}}
for family, seed in (("ped6", 201), ("mix11", 202), ("demo80", 203), ("dyn20", 204)):
    nb = n if family != "dyn20" else max(64, n // 4)
    sc = scenario.generate(family, nb, seed=seed, workers=8)
    cfg = api.default_config(sc["n_steps"])
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=nb, cmax=sc["cmax"])
    t0 = time.time()
    gpu = opt.plan(sc)
    t_gpu = time.time() - t0
    t0 = time.time()
    ref = oracle_reference(sc, oracle_cfg_from(opt.cfg))
    t_cpu = time.time() - t0
    n_pass, n_exc, fails = compare_solutions(gpu, ref, tol=1e-4, margin_tol=0.0)
    stable = ref["stable"]
    failed = {b for b, _ in fails}
    worst_cost, worst_traj = 0.0, 0.0
    for b in range(nb):
        if not stable[b] or b in failed:
            continue
        nc = int(ref["n_cost"][b])
        worst_cost = max(worst_cost, rel_err(gpu["cost_hist"][b, :nc], ref["cost_hist"][b, :nc]))
        worst_traj = max(worst_traj, rel_err(gpu["traj"][b], ref["traj"][b]))
    report["families"][family] = {
        "problems": nb, "n_steps": int(sc["n_steps"]),
        "oracle_stable": int(stable.sum()), "oracle_unstable": int((~stable).sum()),
        "match_within_tolerance": int(n_pass),
        "stable_and_matching": int(sum(1 for b in range(nb) if stable[b] and b not in failed)),
        "stable_but_different": int(sum(1 for b in failed if stable[b])),
        "unstable_but_matching": int(sum(1 for b in range(nb) if not stable[b] and b not in failed)),
        "max_rel_err_cost_rows_stable": worst_cost, "max_rel_err_trajectory_stable": worst_traj,
        "status_histogram_gpu": np.bincount(gpu["status"], minlength=6).tolist(),
        "status_histogram_oracle": np.bincount(ref["status"], minlength=6).tolist(),
        "mean_cost_rows": float(gpu["n_cost"].mean()),
        "gpu_seconds_incl_transfers": round(t_gpu, 3), "oracle_seconds_3_runs": round(t_cpu, 1)}
    opt.close()
print(json.dumps(report, indent=1))
