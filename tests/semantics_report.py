#!/usr/bin/env python
"""(Under tests/: it runs the CPU oracle.  No GPU needed.)
What the two UNPINNED readings of Eigen's semantics are worth (SURVEY 7 "hard parts").

The oracle (oracle/cilqr_oracle.cc) carries two switches for what nothing in this image can check against Eigen 3.4:
  CILQR_DV_EVAL    lazy (default: cc:383-384 re-evaluate the `auto` expressions Qu, Quu on the UPDATED Vx, Vxx) / eager
  CILQR_DOT_ORDER  eigen_sse2 (default since round 5: X^T * Y products: even/odd packet redux; plain-lhs products:
                   sequential pmadd) / sequential (rounds 1-4) / eigen_redux (halving tree, an Eigen build without SIMD)
This script solves the same scenes under every variant and reports, per scene family, how many problems move away from
the default variant by more than the parity tolerance (1e-4: status, iteration count, every accepted step size, every
Cost row, final trajectory) -- over all problems and over the ORACLE-STABLE ones (those the default oracle reproduces
itself under a 4e-16 input perturbation, the set the GPU parity gate is held on), next to the share of problems that are
unstable under that perturbation anyway.  A variant that moves no stable problem is indistinguishable from the default
at the gate's resolution: whichever reading is right, the verdict of the parity tests is the same.

    python tests/semantics_report.py [problems-per-family]  >  profiles/r05_semantics_sensitivity.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

FAMILIES = (("ped6", 201), ("mix11", 202), ("demo80", 203), ("dyn20", 204))
DOT = {"sequential": 0, "eigen_redux": 1, "eigen_sse2": 2}
VARIANTS = (("dv_lazy+dot_sequential", 0, 0), ("dv_lazy+dot_eigen_redux", 0, 1), ("dv_eager+dot_eigen_sse2", 1, 2),
            ("dv_eager+dot_sequential", 1, 0))     # against the default: dv_lazy + dot_eigen_sse2


def family_report(family, seed, nb, variants=VARIANTS):
    from cilqr_amd import scenario
    from oracle import oracle as orc
    from parity_util import PERTURB_EPS, N_PERTURB, REL_TOL, oracle_reference, solution_errors
    sc = scenario.generate(family, nb, seed=seed, workers=8)
    cfg = orc.default_config(sc["n_steps"])
    orc.reset_semantics()
    t0 = time.time()
    ref = oracle_reference(sc, cfg)               # default variant + stability mask (8 perturbed re-runs)
    stable = ref["stable"]
    out = {"problems": nb, "n_steps": int(sc["n_steps"]), "oracle_unstable_under_4e-16": int((~stable).sum()),
           "perturbation": {"eps": PERTURB_EPS, "oracle_reruns": N_PERTURB}, "tolerance": REL_TOL, "variants": {}}
    for name, dv, dot in variants:
        orc.set_semantics(dv, dot)
        try:
            r = orc.solve_batch(sc, cfg, want_trace=True)
        finally:
            orc.reset_semantics()
        moved, moved_stable, flow, flow_stable = 0, 0, 0, 0
        worst_stable = 0.0
        errs = []
        for b in range(nb):
            same_flow, ec, et = solution_errors(r, ref, b)
            e = max(ec, et)
            if not same_flow:
                flow += 1
                flow_stable += int(stable[b])
            if not same_flow or e > REL_TOL:
                moved += 1
                moved_stable += int(stable[b])
            elif stable[b]:
                worst_stable = max(worst_stable, e)
            if same_flow:
                errs.append(e)
        out["variants"][name] = {
            "moved_beyond_tolerance": moved, "moved_beyond_tolerance_among_stable": moved_stable,
            "control_flow_changed": flow, "control_flow_changed_among_stable": flow_stable,
            "largest_deviation_among_stable_within_tolerance": worst_stable,
            "deviation_quantiles_0.5_0.9_0.99_same_flow": [float(q) for q in np.quantile(errs, [0.5, 0.9, 0.99])] if errs else None,
            "bit_identical_solves": int(sum(1 for b in range(nb) if np.array_equal(r["traj"][b], ref["traj"][b])
                                            and np.array_equal(r["cost_hist"][b], ref["cost_hist"][b]))),
        }
    out["seconds"] = round(time.time() - t0, 1)
    return out


def build_report(n=1024, families=FAMILIES):
    rep = {"what": "oracle variants against the default oracle (dv lazy, dot eigen_sse2); see tests/semantics_report.py",
           "problems_per_family": n, "families": {}}
    for fam, seed in families:
        rep["families"][fam] = family_report(fam, seed, n)
    tot = {}
    for fam in rep["families"].values():
        for name, v in fam["variants"].items():
            t = tot.setdefault(name, {"moved_beyond_tolerance": 0, "moved_beyond_tolerance_among_stable": 0})
            t["moved_beyond_tolerance"] += v["moved_beyond_tolerance"]
            t["moved_beyond_tolerance_among_stable"] += v["moved_beyond_tolerance_among_stable"]
    rep["totals"] = {"problems": n * len(families),
                     "oracle_unstable_under_4e-16": sum(f["oracle_unstable_under_4e-16"] for f in rep["families"].values()),
                     "variants": tot}
    return rep


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(json.dumps(build_report(int(args[0]) if args else 1024), indent=1))
