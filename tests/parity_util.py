"""Shared helpers for the parity tests: scene -> oracle / HIP outputs, comparison rules."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

REL_TOL = 1e-4  # north_star: per-iteration costs and final trajectories within 1e-4 relative


def oracle_cfg_from(cfg) -> "orc.OracleConfig":
    """Build the oracle's config from the product's (same field names, separate structs)."""
    o = orc.OracleConfig()
    for name, _ in orc.OracleConfig._fields_:
        setattr(o, name, getattr(cfg, name))
    return o


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor) -- relative with an absolute floor for near-zero entries."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def compare_solutions(gpu: dict, ref: dict, tol=REL_TOL, margin_tol=1e-7):
    """Per-problem comparison of a HIP solve against the oracle.

    A problem PASSES when status, n_cost, every cost row and the final trajectory agree within
    `tol` relative.  A problem whose control flow differs is EXCUSED only if the oracle reports
    that one of its accept/converge decisions sat within `margin_tol` (relative) of its threshold,
    i.e. the decision is not determined at fp64 rounding level.  Returns (n_pass, n_excused,
    failures[list of (index, reason)]).
    """
    B = ref["traj"].shape[0]
    n_pass = n_exc = 0
    fails = []
    for b in range(B):
        why = None
        nc = int(ref["n_cost"][b])
        if int(gpu["n_cost"][b]) != nc or int(gpu["status"][b]) != int(ref["status"][b]):
            why = (f"control flow: n_cost {int(gpu['n_cost'][b])} vs {nc}, status "
                   f"{int(gpu['status'][b])} vs {int(ref['status'][b])}")
        else:
            e_cost = rel_err(gpu["cost_hist"][b, :nc], ref["cost_hist"][b, :nc])
            e_traj = rel_err(gpu["traj"][b], ref["traj"][b])
            if e_cost > tol:
                why = f"cost history rel err {e_cost:.3e}"
            elif e_traj > tol:
                why = f"trajectory rel err {e_traj:.3e}"
        if why is None:
            n_pass += 1
        elif ref.get("min_margin") is not None and ref["min_margin"][b] < margin_tol:
            n_exc += 1
        else:
            fails.append((b, why))
    return n_pass, n_exc, fails


def oracle_reference(scene: dict, cfg=None, n_perturb: int = 2, eps: float = 4e-16, stable_tol: float = 1e-5):
    """Oracle solve + a per-problem conditioning mask.

    The reference algorithm is chaotic on a few percent of scenes: re-running the ORACLE ITSELF on
    inputs perturbed by ~2 ulp (coarse trajectory scaled by 1 + eps*N(0,1)) changes cost histories
    by far more than 1e-4 or even the iteration count (long line-search chains amplify rounding
    noise).  No implementation with a different libm / summation order can match the oracle on
    those scenes, so parity is asserted on the problems whose oracle result is stable under such
    perturbations (`stable`), and the unstable fraction is reported and bounded separately.
    """
    r0 = orc.solve_batch(scene, cfg)
    B = scene["coarse"].shape[0]
    stable = np.ones(B, bool)
    rng = np.random.default_rng(12345)
    for _ in range(n_perturb):
        sc2 = dict(scene)
        sc2["coarse"] = scene["coarse"] * (1.0 + eps * rng.standard_normal(scene["coarse"].shape))
        r1 = orc.solve_batch(sc2, cfg, want_margin=False)
        for b in range(B):
            if not stable[b]:
                continue
            nc = int(r0["n_cost"][b])
            if int(r1["n_cost"][b]) != nc or int(r1["status"][b]) != int(r0["status"][b]):
                stable[b] = False
            elif rel_err(r1["cost_hist"][b, :nc], r0["cost_hist"][b, :nc]) > stable_tol or \
                    rel_err(r1["traj"][b], r0["traj"][b]) > stable_tol:
                stable[b] = False
    r0["stable"] = stable
    return r0


def assert_parity(gpu: dict, ref: dict, tol=REL_TOL, max_unstable_frac=0.15, what="", margin_tol=0.0):
    """Every oracle-stable problem must match within tol; unstable ones are counted, not compared.

    margin_tol > 0 additionally treats a problem as unstable when one of the oracle's own
    accept/converge decisions sat within that relative distance of its threshold (used for the
    zero-tolerance configuration, where the iteration runs into the rounding-noise plateau)."""
    stable = ref.get("stable")
    B = ref["traj"].shape[0]
    if stable is None:
        stable = np.ones(B, bool)
    if margin_tol > 0.0 and ref.get("min_margin") is not None:
        stable = stable & (ref["min_margin"] >= margin_tol)
    n_pass, n_exc, fails = compare_solutions(gpu, ref, tol=tol, margin_tol=0.0)
    bad = [(b, why) for b, why in fails if stable[b]]
    n_unstable = int((~stable).sum())
    assert n_unstable <= max(1, int(max_unstable_frac * B)), \
        f"{what}: {n_unstable}/{B} scenes are ill-conditioned in the oracle itself"
    assert not bad, f"{what}: {len(bad)} oracle-stable problems differ: {bad[:5]}"
    return dict(n=B, n_stable=int(stable.sum()), n_match=n_pass, n_unstable=n_unstable,
                n_unstable_matching=int(sum(1 for b in range(B) if not stable[b]) - sum(1 for b, _ in fails if not stable[b])))
