#!/usr/bin/env python
"""(Under tests/: it runs the CPU oracle.)
Parity of the HIP path against the CPU oracle on a larger sample than the unit tests use.  Per scene
family: whole solves (status, iteration count, accepted step size of every iteration, every Cost row,
final trajectory within 1e-4) on the oracle-stable problems, the share of problems the oracle cannot
reproduce itself under a 4e-16 input perturbation (8 re-runs), and the step-by-step replay of EVERY
problem, stable or not (tests/parity_util.py).  `python tests/parity_report.py [problems-per-family]`
prints one JSON document; tests/test_gpu_parity.py::test_parity_report runs the same function in the
`-m gpu` suite and asserts on it."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

FAMILIES = (("ped6", 201), ("mix11", 202), ("demo80", 203), ("dyn20", 204))


def family_report(family, seed, nb, lib_path=None):
    from cilqr_amd import api, scenario
    from parity_util import (PERTURB_EPS, N_PERTURB, check_steps, compare_solutions, oracle_cfg_from,
                             oracle_reference, second_look, solution_errors)
    sc = scenario.generate(family, nb, seed=seed, workers=8)
    cfg = api.default_config(sc["n_steps"])
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=nb, cmax=sc["cmax"])
    exact = "--fast-lane-ties" not in sys.argv   # the library's default: the reference's nearest-segment tie rule
    if not exact:   # the opt-in fast rule (squared distances only): the step replay then may use the lane_tie excuse
        opt.set_option(api.OPT_EXACT_LANE_TIES, 0)
    t0 = time.time()
    gpu = opt.plan(sc, max_iter_trajs=48, alpha_trace=True)
    t_gpu = time.time() - t0
    ocfg = oracle_cfg_from(opt.cfg)
    t0 = time.time()
    ref = oracle_reference(sc, ocfg)
    t_cpu = time.time() - t0
    n_pass, fails = compare_solutions(gpu, ref, tol=1e-4)
    stable = ref["stable"]
    failed = {b for b, _ in fails}
    worst_cost = worst_traj = 0.0
    for b in range(nb):
        if not stable[b] or b in failed:
            continue
        _, ec, et = solution_errors(gpu, ref, b)
        worst_cost, worst_traj = max(worst_cost, ec), max(worst_traj, et)
    looks = [second_look(sc, ocfg, b, gpu=gpu) for b in sorted(b for b in failed if stable[b])]
    t0 = time.time()
    steps = check_steps(gpu, sc, ocfg, allow_lane_tie=not exact)
    t_steps = time.time() - t0
    errs = np.asarray(steps.pop("errors"))
    opt.close()
    return {
        "problems": nb, "n_steps": int(sc["n_steps"]),
        "perturbation": {"eps": PERTURB_EPS, "oracle_reruns": N_PERTURB},
        "oracle_stable": int(stable.sum()), "oracle_unstable": int((~stable).sum()),
        "match_within_tolerance": int(n_pass),
        "stable_and_matching": int(sum(1 for b in range(nb) if stable[b] and b not in failed)),
        "stable_but_different": sorted(int(b) for b in failed if stable[b]),
        # ... each of them looked at again with 64+ fresh oracle re-runs (parity_util.second_look): excused only if the oracle itself
        # ends elsewhere in at least 4 of 64 AND the library's result equals one of those perturbed endings within 1e-4;
        # "confirmed" = everything else (a mismatch)
        "stable_but_different_second_look": looks,
        "stable_but_different_confirmed": sorted(l["problem"] for l in looks if not l["excused"]),
        "stable_but_different_excused": sorted(l["problem"] for l in looks if l["excused"]),
        "unstable_but_matching": int(sum(1 for b in range(nb) if not stable[b] and b not in failed)),
        "max_err_cost_rows_stable": worst_cost, "max_err_trajectory_stable": worst_traj,
        "steps": {"replayed": steps["steps"], "within_1e-8": steps["tight"], "excused_discontinuous_in_oracle": steps["excused"],
                  "failed": steps["failed"][:20], "n_failed": len(steps["failed"]), "iterates_not_kept": steps["truncated"],
                  "worst_error_among_matching": steps["worst"],
                  "error_quantiles_0.5_0.9_0.99_0.999": [float(q) for q in np.quantile(errs, [0.5, 0.9, 0.99, 0.999])] if errs.size else None},
        "status_histogram_gpu": np.bincount(gpu["status"], minlength=7).tolist(),
        "status_histogram_oracle": np.bincount(ref["status"], minlength=7).tolist(),
        "mean_cost_rows": float(gpu["n_cost"].mean()),
        "seconds": {"gpu_incl_transfers": round(t_gpu, 2), "oracle_9_runs": round(t_cpu, 1), "step_replay": round(t_steps, 1)},
    }


def build_report(n=1024, families=FAMILIES, with_reference_order=True):
    """Product library against the oracle, and -- in a child process, because a process holds one libcilqr_hip --
    the TEST-ONLY build of the same kernels with the cost / quadratisation arithmetic in the reference's operation
    order and the library's log / sin / cos / tan (cilqr_amd/csrc/ref_order.hpp, lib/libcilqr_hip_reforder.so).  The
    share of problems that differ from the oracle in THAT build is what the reference's ill-conditioning costs any
    implementation with another libm; the product's share on top of it is what its re-associations add."""
    import torch  # noqa: F401  (one HIP runtime per process: torch's first)
    report = {"tolerance_whole_solves": 1e-4, "tolerance_steps": 1e-8, "problems_per_family": n,
              "exact_lane_ties": "--fast-lane-ties" not in sys.argv, "families": {}}
    for family, seed in families:
        report["families"][family] = family_report(family, seed, n)
    if with_reference_order:
        import subprocess
        lib = os.path.join(ROOT, "cilqr_amd", "lib", "libcilqr_hip_reforder.so")
        if not os.path.exists(lib):
            report["reference_order_build"] = {"error": f"{lib} is missing (make -C cilqr_amd/csrc reforder)"}
        else:
            env = dict(os.environ, CILQR_LIB=lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), "--plain"], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                report["reference_order_build"] = {"error": r.stderr[-2000:]}
            else:
                sub = json.loads(r.stdout)
                report["reference_order_build"] = {"library": os.path.relpath(lib, ROOT), "families": sub["families"]}
        rob = report["reference_order_build"].get("families")
        if rob:
            # the comparison the two builds are there for: problems (of all, stable or not) that do not match the oracle
            report["mismatch_share_all_problems"] = {
                fam: {"product_build": round(1.0 - report["families"][fam]["match_within_tolerance"] / report["families"][fam]["problems"], 4),
                      "reference_order_build": round(1.0 - rob[fam]["match_within_tolerance"] / rob[fam]["problems"], 4),
                      "oracle_unstable_share": round(report["families"][fam]["oracle_unstable"] / report["families"][fam]["problems"], 4)}
                for fam in report["families"] if fam in rob}
    return report


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_ = int(args[0]) if args else 1024
    print(json.dumps(build_report(n_, with_reference_order="--plain" not in sys.argv), indent=1))
