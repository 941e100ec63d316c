"""Ad-hoc GPU probe (not a pytest): stage-by-stage and full-solve parity + timing printout."""
import sys
import time

import numpy as np

from parity_util import ROOT, compare_solutions, oracle_cfg_from, rel_err  # noqa: F401
from cilqr_amd import api, scenario
from oracle import oracle as orc


def stages(name="mix11", B=8, seed=5):
    sc = scenario.generate(name, B, seed=seed)
    N = sc["n_steps"]
    opt = api.BatchIlqrOptimizer(n_steps=N, batch_capacity=B, cmax=sc["cmax"])
    opt.stage_load(sc)
    ocfg = oracle_cfg_from(opt.cfg)
    worst = {}

    def upd(k, e):
        worst[k] = max(worst.get(k, 0.0), e)

    goals = opt.read(api.T_GOALS)
    cor = opt.read(api.T_CORRIDOR)
    lanes = opt.read(api.T_LANES)
    opt.stage_init_guess()
    X = opt.read(api.T_X)
    U = opt.read(api.T_U)
    cost = opt.stage_total_cost()
    opt.stage_quadratize()
    q = {k: opt.read(t) for k, t in dict(A=api.T_A, B=api.T_B, lx=api.T_LX, lu=api.T_LU,
                                         lxx=api.T_LXX, luu=api.T_LUU).items()}
    opt.stage_backward(1.0)
    Kfb, kff, dV, gn = opt.read(api.T_KFB), opt.read(api.T_KFF), opt.read(api.T_DV), opt.read(api.T_GNORM)
    opt.stage_forward(0.5012)
    Xc, Uc = opt.read(api.T_XCAND), opt.read(api.T_UCAND)
    for b in range(B):
        o = orc.Oracle(ocfg)
        o.set_problem(sc["start"][b], sc["coarse"][b], sc["corridor"][b], sc["ccount"][b], sc["left"], sc["right"])
        og, oc, ol, orr, _ = o.constraints()
        upd("goals", rel_err(goals[b], og))
        m = np.arange(sc["cmax"])[None, :] < sc["ccount"][b][:, None]
        upd("corridor", rel_err(cor[b][m], oc[m], 1e-3))
        upd("lanes", rel_err(lanes, np.concatenate([ol, orr]), 1e-3))
        oX, oU = o.init_guess()
        upd("init X", rel_err(X[b], oX))
        upd("init U", rel_err(U[b], oU, 1e-3))
        upd("cost", rel_err(cost[b], o.total_cost(X[b], U[b])))
        oq = o.quadratize(X[b], U[b])
        for k in q:
            upd("quad " + k, rel_err(q[k][b], oq[k], 1e-6))
        gq = {k: q[k][b] for k in q}
        oK, ok_, odV = o.backward(1.0, gq)
        upd("K", rel_err(Kfb[b], oK, 1e-6))
        upd("k", rel_err(kff[b], ok_, 1e-6))
        upd("dV", rel_err(dV[b], odV, 1e-6))
        upd("gnorm", rel_err(gn[b], o.grad_norm(kff[b], U[b]), 1e-9))
        oXn, oUn = o.forward(0.5012, X[b], U[b], Kfb[b], kff[b])
        upd("fwd X", rel_err(Xc[b], oXn))
        upd("fwd U", rel_err(Uc[b], oUn, 1e-3))
    for k, v in worst.items():
        print(f"  stage {k:12s} max rel err {v:.3e}")
    opt.close()


def full(name="mix11", B=256, seed=7, iter_trajs=0):
    sc = scenario.generate(name, B, seed=seed)
    N = sc["n_steps"]
    opt = api.BatchIlqrOptimizer(n_steps=N, batch_capacity=B, cmax=sc["cmax"])
    opt.set_profiling(True)
    t = time.time()
    g = opt.plan(sc, max_iter_trajs=iter_trajs)
    t_gpu = time.time() - t
    t = time.time()
    g = opt.plan(sc, max_iter_trajs=iter_trajs)
    t_gpu2 = time.time() - t
    r = orc.solve_batch(sc, oracle_cfg_from(opt.cfg))
    n_pass, n_exc, fails = compare_solutions(g, r)
    p = opt.profile()
    print(f"  full {name} B={B}: pass {n_pass} excused {n_exc} fail {len(fails)}; gpu {t_gpu:.3f}s/{t_gpu2:.3f}s "
          f"oracle {r['seconds']:.2f}s; iters {p.iterations} bwd {p.backward_ms:.3f}ms quad {p.quadratize_ms:.3f} "
          f"ls {p.linesearch_ms:.3f} other {p.other_ms:.3f} total {p.total_ms:.3f}")
    for b, why in fails[:10]:
        print("    FAIL", b, why, "margin", r["min_margin"][b])
    opt.close()
    return len(fails)


if __name__ == "__main__":
    print("abi", api.lib().cilqr_abi_version())
    stages()
    nf = full("ped6", 256)
    nf += full("mix11", 1024)
    nf += full("dyn20", 128)
    sys.exit(1 if nf else 0)
