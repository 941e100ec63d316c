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
