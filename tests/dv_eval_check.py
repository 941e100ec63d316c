#!/usr/bin/env python
"""(Under tests/: it runs the CPU oracle.)  Child process of tests/test_gpu_parity.py::test_delta_v_evaluation_switch.

Runs with CILQR_LIB = cilqr_amd/lib/libcilqr_hip_dveager.so, the TEST-ONLY build of the product sources with
-DCILQR_DV_EVAL_EAGER (cilqr_amd/csrc/backward_core.hpp): delta_V_ of Backward (ilqr_optimizer.cc:383-384) from the Qu / Quu
the gains were computed from, instead of the lazy re-evaluation on the updated Vx / Vxx.  Checks, and prints one JSON line:
  * the backward stage in all three mappings (wavefront / eight lanes / one lane per problem): delta_V_ equals the oracle's
    EAGER variant (oracle_set_semantics(1, .)) to 1e-9, the three mappings agree bit for bit, gains are those of the
    default reading -- and delta_V_ is NOT the lazy oracle's (the switch does something);
  * whole solves: every step of every problem replays in the eager oracle within 1e-8."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    from cilqr_amd import api, scenario
    from oracle import oracle as orc
    from parity_util import assert_steps, oracle_cfg_from, rel_err
    assert "dveager" in api.LIB_PATH, api.LIB_PATH
    B = 96
    sc = scenario.generate("mix11", B, seed=52)
    cfg = api.default_config(sc["n_steps"])
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=B, cmax=sc["cmax"])
    ocfg = oracle_cfg_from(opt.cfg)
    opt.stage_load(sc)
    opt.stage_init_guess()
    X, U = opt.read(api.T_X), opt.read(api.T_U)
    opt.stage_quadratize()
    q = {k: opt.read(t) for k, t in dict(A=api.T_A, B=api.T_B, lx=api.T_LX, lu=api.T_LU, lxx=api.T_LXX, luu=api.T_LUU).items()}
    lam = np.linspace(0.5, 3.0, B)
    forms = {}
    for name, team, wave in (("wave", 4096, 1024), ("team", 4096, 0), ("lane", 0, 0)):
        opt.set_option(api.OPT_TEAM_THRESHOLD, team)
        opt.set_option(api.OPT_WAVE_THRESHOLD, wave)
        opt.stage_backward(lam)
        forms[name] = (opt.read(api.T_KFB), opt.read(api.T_KFF), opt.read(api.T_DV))
    for name in ("team", "lane"):
        for a, b in zip(forms[name], forms["wave"]):
            assert np.array_equal(a, b), f"{name} and wave mappings differ"
    Kfb, kff, dV = forms["wave"]
    worst_eager, n_differs_from_lazy = 0.0, 0
    for b in range(B):
        o = orc.Oracle(ocfg)
        assert o.set_problem(sc["start"][b], sc["coarse"][b], sc["corridor"][b], sc["ccount"][b], sc["left"], sc["right"]) == 0
        qb = {k: q[k][b] for k in q}
        orc.set_semantics(0, -1)
        oK, ok_, dV_lazy = o.backward(float(lam[b]), qb)
        orc.set_semantics(1, -1)
        oK1, ok1, dV_eager = o.backward(float(lam[b]), qb)
        assert np.array_equal(oK, oK1) and np.array_equal(ok_, ok1)        # the switch touches delta_V_ only
        assert rel_err(Kfb[b], oK, 1e-6) < 1e-9 and rel_err(kff[b], ok_, 1e-6) < 1e-9
        worst_eager = max(worst_eager, rel_err(dV[b], dV_eager, 1e-6))
        n_differs_from_lazy += int(rel_err(dV[b], dV_lazy, 1e-6) > 1e-6)
    assert worst_eager < 1e-9, worst_eager
    assert n_differs_from_lazy >= B // 2, n_differs_from_lazy
    opt.close()
    # whole solves, replayed step by step in the eager oracle
    n = 160
    sc = scenario.generate("mix11", n, seed=77)
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=n, cmax=sc["cmax"])
    g = opt.plan(sc, max_iter_trajs=48, alpha_trace=True)
    orc.set_semantics(1, -1)
    rep = assert_steps(g, sc, ocfg, what="eager delta_V build against the eager oracle")
    orc.set_semantics(0, -1)
    opt.close()
    print(json.dumps({"ok": True, "worst_dV_error_vs_eager_oracle": worst_eager, "stage_problems_differing_from_lazy": n_differs_from_lazy,
                      "steps": {k: v for k, v in rep.items() if k != "failed"}}))


if __name__ == "__main__":
    main()
