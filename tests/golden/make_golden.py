#!/usr/bin/env python
"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle.

The reference (mpt0816/Cilqr) has no tests or recorded vectors and cannot run in this
environment, so these fixtures pin the build's own oracle (oracle/cilqr_oracle.cc): inputs of
IlqrOptimizer::Plan plus the oracle's outputs and per-stage dumps.  They are data (npz), not code.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from cilqr_amd import scenario  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from parity_util import oracle_reference  # noqa: E402

CASES = [  # (file stem, scene family, seed, pool size, scenes kept)
    ("mix11_n50", "mix11", 101, 24, 6),
    ("ped6_n50", "ped6", 102, 16, 4),
    ("dyn20_n100", "dyn20", 103, 8, 2),
]


def pick(ref, n):
    """Well-conditioned scenes with varied behaviour: most iterations first, then a quick one."""
    idx = [b for b in np.argsort(-ref["n_iter"]) if ref["stable"][b]]
    chosen = idx[: n - 1] + idx[-1:]
    return sorted(chosen)


def main():
    for stem, fam, seed, pool, keep in CASES:
        sc = scenario.generate(fam, pool, seed=seed)
        ref = oracle_reference(sc, n_perturb=3, eps=1e-13)
        sel = pick(ref, keep)
        out = dict(family=fam, seed=seed, picked=np.array(sel), n_steps=sc["n_steps"], dt=sc["dt"],
                   cmax=sc["cmax"], left=sc["left"], right=sc["right"])
        for k in ("start", "coarse", "corridor", "ccount"):
            out[k] = sc[k][sel]
        for k in ("traj", "cost_hist", "n_cost", "status", "n_iter"):
            out["ref_" + k] = ref[k][sel]
        nmax = int(out["ref_n_cost"].max())
        out["ref_cost_hist"] = out["ref_cost_hist"][:, :nmax]
        # stage dumps of the first kept scene
        o = orc.Oracle(n_steps=sc["n_steps"])
        b = sel[0]
        o.set_problem(sc["start"][b], sc["coarse"][b], sc["corridor"][b], sc["ccount"][b], sc["left"], sc["right"])
        goals, cor, la, ra, rad = o.constraints()
        X, U = o.init_guess()
        q = o.quadratize(X, U)
        Kfb, kff, dV = o.backward(1.0, q)
        Xn, Un = o.forward(0.5012, X, U, Kfb, kff)
        r = o.plan(max_iter_trajs=8, want_trace=True)
        out.update(st_goals=goals, st_corridor=cor, st_left=la, st_right=ra, st_disc_radius=rad,
                   st_X=X, st_U=U, st_cost=o.total_cost(X, U), st_gnorm=o.grad_norm(kff, U),
                   st_K=Kfb, st_k=kff, st_dV=dV, st_Xn=Xn, st_Un=Un, st_cost_n=o.total_cost(Xn, Un),
                   st_trace=r["trace"], st_iter_trajs=r["iter_trajs"], st_n_iter_trajs=r["n_iter_trajs"],
                   **{"st_q_" + k: v for k, v in q.items()})
        path = os.path.join(HERE, stem + ".npz")
        np.savez_compressed(path, **out)
        print(stem, "scenes", sel, "n_cost", out["ref_n_cost"], "status", out["ref_status"],
              f"{os.path.getsize(path) / 1024:.0f} KiB")


def make_scene_file():
    """tests/golden/scenes_mix11_4.cqs (+ .json): the wire format of cilqr_amd/scene_io.py, pinned."""
    import hashlib
    import json
    from cilqr_amd import scene_io
    sc = scenario.generate("mix11", 4, seed=71, obstacle_points=True, scenarios=True)
    f = scene_io.from_generator(sc)
    path = os.path.join(HERE, "scenes_mix11_4.cqs")
    scene_io.save(path, f)
    raw = open(path, "rb").read()
    K = sc["coarse"].shape[1]
    meta = {"sha256": hashlib.sha256(raw).hexdigest(), "bytes": len(raw), "scenes": 4, "knots": K, "dt": sc["dt"],
            "n_static": [len(s.static) for s in f.scenes], "n_dynamic": [len(s.dynamic) for s in f.scenes],
            "points_per_knot": [scene_io.environment_points(s, np.arange(K) * sc["dt"])[1].tolist() for s in f.scenes],
            "made_by": "tests/golden/make_golden.py (scene file section): scenario.generate('mix11', 4, seed=71, scenarios=True)"}
    json.dump(meta, open(os.path.join(HERE, "scenes_mix11_4.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
    make_scene_file()
