#!/usr/bin/env python
"""Child process of tests/test_gpu_parity.py::test_four_row_candidate_arena_is_bit_identical.

Runs with CILQR_SPEC_ROWS=4 (the candidate arena of the big batches: four step sizes per slot, re-strided for the rest --
cilqr_amd/csrc/kernels_search.hip: spec_view) and CILQR_SPEC_PASS_ENTRIES=8 (so that the problems which rejected every
pre-rolled round need SEVERAL passes over the re-used cells even in a batch of a few hundred), lockstep loop to the end.
Solves the scenes of the .npz given on the command line under three schedules and checks every result bit for bit against
the arrays in that file (the parent's solve with the plain eleven-row arena):
  rounds   CILQR_OPT_SPEC_THRESHOLD = 0: pre-rolled rounds + multi-pass remainder in every iteration
  default  the default threshold: rounds while the active set exceeds what four rows hold of all eleven step sizes, then all
           eleven at once at the shorter stride
  rounds6  CILQR_OPT_SEQ_ROUNDS = 6 > 4 rows: falls back to the round-by-round rollouts, remainder over the whole arena"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    from cilqr_amd import api
    assert os.environ.get("CILQR_SPEC_ROWS") == "4" and os.environ.get("CILQR_SPEC_PASS_ENTRIES")
    z = np.load(sys.argv[1])
    sc = {k: z[k] for k in ("start", "coarse", "corridor", "ccount", "left", "right")}
    sc.update(n_steps=int(z["n_steps"]), cmax=int(z["cmax"]))
    B = sc["start"].shape[0]
    cfg = api.default_config(sc["n_steps"], **json.loads(str(z["cfg_over"])))
    out = {}
    for name, opts in (("rounds", {api.OPT_SPEC_THRESHOLD: 0}), ("default", {}), ("rounds6", {api.OPT_SPEC_THRESHOLD: 0, api.OPT_SEQ_ROUNDS: 6})):
        opt = api.BatchIlqrOptimizer(cfg, batch_capacity=B, cmax=sc["cmax"])
        opt.set_option(api.OPT_TAIL_THRESHOLD, 0)
        for k, v in opts.items():
            opt.set_option(k, v)
        g = opt.plan(sc, max_iter_trajs=3, alpha_trace=True)
        for k in ("traj", "cost_hist", "status", "n_cost", "n_iter", "alpha_trace"):
            assert np.array_equal(g[k], z["ref_" + k]), f"{name}: {k} differs from the eleven-row arena's"
        out[name] = int((g["alpha_trace"] >= 4).sum())     # accepted step sizes beyond the pre-rolled four: the remainder passes' work
        opt.close()
    print(json.dumps({"ok": True, "problems": int(B), "accepted_beyond_four_rounds": out}))


if __name__ == "__main__":
    main()
