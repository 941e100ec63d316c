"""GPU parity tests: every test calls the HIP path through the C-ABI (include/cilqr.h) and checks
it against the CPU oracle or the committed golden fixtures.

Tolerance: north_star asks for per-iteration costs and final trajectories within 1e-4 relative
(REL_TOL); stage outputs are checked much tighter (1e-9 relative, the GPU differs from the oracle
only through libm rounding and summation order)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from parity_util import (REL_TOL, entry_err, assert_parity, assert_steps, cost_err, oracle_cfg_from, oracle_reference, rel_err,
                         traj_err)
from cilqr_amd import api, scenario
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STAGE_TOL = 1e-9
ITER_CAP = 48   # iterates kept per problem for the step-by-step replay (longer solves: decisions and costs only)


def _plan(opt, sc, **kw):
    """A solve with everything the parity rules look at: iterates and the per-iteration decisions."""
    return opt.plan(sc, max_iter_trajs=ITER_CAP, alpha_trace=True, **kw)


# Which solve loop the optimisers of a test run.  None = the product default (CILQR_OPT_TAIL_THRESHOLD 256: batches
# of the sizes used here run entirely in the per-problem tail kernel, kernels_tail.hip); 0 = lockstep kernels to the
# end.  The `both_paths` fixture runs a test once with each.
_TAIL = [None]


def _opt(sc, B=None, **cfg_over):
    cfg = api.default_config(sc["n_steps"], **cfg_over)
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=B or sc["coarse"].shape[0], cmax=sc["cmax"])
    if _TAIL[0] is not None:
        opt.set_option(api.OPT_TAIL_THRESHOLD, _TAIL[0])
    return opt


@pytest.fixture(params=["tail", "lockstep"])
def both_paths(request):
    _TAIL[0] = None if request.param == "tail" else 0
    yield request.param
    _TAIL[0] = None


@pytest.fixture
def lockstep_only():
    """Tests of the lockstep kernels' schedules (team backward, speculative rounds, re-packing): the tail kernel
    would take these small batches over from the first iteration."""
    _TAIL[0] = 0
    yield
    _TAIL[0] = None


@pytest.fixture(scope="module", autouse=True)
def _build(built):
    return built


# ---------------------------------------------------------------------------------------------
# stages
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,B,seed", [("mix11", 24, 31), ("dyn20", 6, 32)])
def test_stage_parity(family, B, seed):
    sc = scenario.generate(family, B, seed=seed)
    opt = _opt(sc)
    ocfg = oracle_cfg_from(opt.cfg)
    opt.stage_load(sc)
    goals, cor, lanes = opt.read(api.T_GOALS), opt.read(api.T_CORRIDOR), opt.read(api.T_LANES)
    opt.stage_init_guess()
    X, U = opt.read(api.T_X), opt.read(api.T_U)
    cost = opt.stage_total_cost()
    opt.stage_quadratize()
    q = {k: opt.read(t) for k, t in dict(A=api.T_A, B=api.T_B, lx=api.T_LX, lu=api.T_LU, lxx=api.T_LXX,
                                         luu=api.T_LUU).items()}
    lam = np.linspace(0.5, 3.0, B)
    opt.stage_backward(lam)
    Kfb, kff, dV, gn = opt.read(api.T_KFB), opt.read(api.T_KFF), opt.read(api.T_DV), opt.read(api.T_GNORM)
    opt.stage_forward(0.2512)
    Xc, Uc = opt.read(api.T_XCAND), opt.read(api.T_UCAND)
    for b in range(B):
        o = orc.Oracle(ocfg)
        assert o.set_problem(sc["start"][b], sc["coarse"][b], sc["corridor"][b], sc["ccount"][b],
                             sc["left"], sc["right"]) == 0
        og, oc, ol, orr, _ = o.constraints()
        assert np.array_equal(goals[b], og)
        m = np.arange(sc["cmax"])[None, :] < sc["ccount"][b][:, None]
        assert rel_err(cor[b][m], oc[m], 1e-3) < 1e-12
        assert rel_err(lanes, np.concatenate([ol, orr]), 1e-3) < 1e-12
        oX, oU = o.init_guess()
        assert rel_err(X[b], oX) < STAGE_TOL and rel_err(U[b], oU, 1e-3) < STAGE_TOL
        # from here on feed the oracle the GPU's own iterate: each stage is checked in isolation
        assert rel_err(cost[b], o.total_cost(X[b], U[b])) < STAGE_TOL
        oq = o.quadratize(X[b], U[b])
        for k in q:
            worst, where = entry_err(q[k][b], oq[k])     # per entry, floor = 1e-3 of the knot's largest (parity_util)
            assert worst < STAGE_TOL, (k, worst, where)
        oK, ok_, odV = o.backward(float(lam[b]), {k: q[k][b] for k in q})
        assert rel_err(Kfb[b], oK, 1e-6) < STAGE_TOL and rel_err(kff[b], ok_, 1e-6) < STAGE_TOL
        assert rel_err(dV[b], odV, 1e-6) < STAGE_TOL
        assert gn[b] == pytest.approx(o.grad_norm(kff[b], U[b]), rel=1e-12)
        oXn, oUn = o.forward(0.2512, X[b], U[b], Kfb[b], kff[b])
        # a first-iteration step can fling the rollout through tan() poles, where 1-ulp libm
        # differences are amplified; the tight bound applies to rollouts that stay physical
        tame = np.all(np.abs(oXn[:, 5]) < 0.7) and np.all(np.abs(oXn[:, 3]) < 30.0)
        ftol = STAGE_TOL if tame else REL_TOL
        assert rel_err(Xc[b], oXn) < ftol and rel_err(Uc[b], oUn, 1e-3) < ftol
    opt.close()


@pytest.mark.parametrize("family", ["mix11", "dyn20"])
def test_wave_init_guess_is_bit_identical_to_the_lane_kernel(family):
    """Small batches build the init guess (iqr, ilqr_optimizer.cc:793-842) with a wavefront per problem (k_init_guess_wave:
    Jacobians of all steps side by side, one output element of the LQR sweep per lane), large ones with a lane per problem.
    The same scenes must come out with the same bits from both -- states, controls and the whole solve that starts there."""
    n_small, n_big = 96, 4200                      # launch_init_guess switches kernels above 4096 problems
    base = scenario.generate(family, n_small, seed=71)
    rep = (n_big + n_small - 1) // n_small
    big = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:n_big] if isinstance(v, np.ndarray) and v.shape[:1] == (n_small,) else v)
           for k, v in base.items()}
    small_opt = _opt(base)
    small_opt.stage_load(base)
    small_opt.stage_init_guess()
    Xs, Us = small_opt.read(api.T_X), small_opt.read(api.T_U)
    big_opt = _opt(big)
    big_opt.stage_load(big)
    big_opt.stage_init_guess()
    Xb, Ub = big_opt.read(api.T_X), big_opt.read(api.T_U)
    assert np.array_equal(Xs, Xb[:n_small]) and np.array_equal(Us, Ub[:n_small])
    assert np.array_equal(Xb[:n_small], Xb[n_small:2 * n_small])          # and the position in the batch does not matter
    o = orc.Oracle(oracle_cfg_from(small_opt.cfg))
    for b in range(0, n_small, 7):
        assert o.set_problem(base["start"][b], base["coarse"][b], base["corridor"][b], base["ccount"][b], base["left"], base["right"]) == 0
        oX, oU = o.init_guess()
        assert rel_err(Xs[b], oX) < STAGE_TOL and rel_err(Us[b], oU, 1e-3) < STAGE_TOL
    a, b_ = small_opt.plan(base), big_opt.plan(big)
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter"):
        assert np.array_equal(a[k], b_[k][:n_small]), k
    small_opt.close()
    big_opt.close()


def test_open_loop_rollout():
    rng = np.random.default_rng(5)
    B, N = 70, 50
    x0 = np.stack([rng.normal(size=B) * 30, rng.normal(size=B) * 30, rng.uniform(-3.1, 3.1, B),
                   rng.uniform(0, 15, B), rng.uniform(-3, 3, B), rng.uniform(-0.6, 0.6, B)], axis=1)
    U = np.stack([rng.uniform(-10, 10, (B, N)), rng.uniform(-0.23, 0.23, (B, N))], axis=2)
    opt = api.BatchIlqrOptimizer(n_steps=N, batch_capacity=B)
    X = opt.open_loop_rollout(x0, U)
    o = orc.Oracle(n_steps=N)
    for b in range(B):
        assert rel_err(X[b], o.open_loop_rollout(x0[b], U[b])) < 1e-11
    # a step's angle wraps take a straight-line form unless some lane of the wavefront leaves (-3 pi, 3 pi), in which case
    # the wavefront repeats the step with the complete function (dev_model.hpp: dynamics): start states whose heading and
    # steering angle are several turns off the principal range, on every third problem, exercise that path next to lanes
    # that stay on the short one
    x02 = x0.copy()
    x02[::3, 2] += 2.0 * np.pi * rng.integers(2, 9, len(x02[::3])) * rng.choice([-1.0, 1.0], len(x02[::3]))
    x02[::3, 5] += 2.0 * np.pi * rng.integers(2, 9, len(x02[::3])) * rng.choice([-1.0, 1.0], len(x02[::3]))
    X2 = opt.open_loop_rollout(x02, U)
    keep = np.ones(B, bool); keep[::3] = False
    assert np.array_equal(X2[keep], X[keep])
    for b in range(0, B, 3):
        assert rel_err(X2[b], o.open_loop_rollout(x02[b], U[b])) < 1e-11
    opt.close()


# ---------------------------------------------------------------------------------------------
# full solves
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,B,seed", [("ped6", 200, 41), ("mix11", 333, 42), ("dyn20", 96, 43), ("demo80", 64, 44)])
def test_full_solve_parity(family, B, seed, both_paths):
    """Whole solves vs the oracle (status, iteration count, accepted step size of every iteration, every
    Cost row, final trajectory: 1e-4 on every oracle-stable problem; stability = 8 oracle re-runs at a
    4e-16 input perturbation) AND every single step of every problem, stable or not, replayed in the
    oracle from the HIP path's own iterate (1e-8).  Unstable shares of these scene sets, which depend on
    the oracle alone: 5.0 / 3.9 / 2.1 / 9.4 %."""
    sc = scenario.generate(family, B, seed=seed)
    opt = _opt(sc)
    g = _plan(opt, sc)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc, ocfg)
    rep = assert_parity(g, ref, what=f"{family} B={B}")
    steps = assert_steps(g, sc, ocfg, what=f"{family} B={B}")
    print(f"\n[{family}] whole solves {rep}\n[{family}] steps {steps}")
    opt.close()


def test_parity_report():
    """The large-sample comparison (tests/parity_report.py: 1024 scenes of each family), run by the driver:
    no oracle-stable problem differs, the unstable share stays under 10 %, no step fails its replay and
    under 1 % of the steps are excused as discontinuous in the oracle.  The report goes to
    gpurun_out/parity_report.json when that directory exists (copied to profiles/ by hand)."""
    import json
    from parity_report import build_report
    rep = build_report(1024)
    out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
            json.dump(rep, f, indent=1)
    for fam, r in rep["families"].items():
        # a problem the 8-run stability mask called stable and the library solves differently is looked at again with 64+ fresh
        # oracle re-runs (parity_util.second_look): it is a mismatch (none allowed) unless the oracle itself ends elsewhere in at
        # least 4 of 64 AND the library's result is one of those endings within 1e-4 -- then the mask missed an unstable problem
        # (a flip rate of 20 % passes eight runs one time in six); at most one in a thousand may
        assert not r["stable_but_different_confirmed"], (fam, r["stable_but_different_second_look"])
        assert len(r["stable_but_different"]) <= max(1, r["problems"] // 1000), (fam, r["stable_but_different_second_look"])
        assert r["oracle_unstable"] <= 0.10 * r["problems"], (fam, r["oracle_unstable"])
        assert r["steps"]["n_failed"] == 0, (fam, r["steps"]["failed"][:5])
        assert r["steps"]["excused_discontinuous_in_oracle"] <= 0.01 * r["steps"]["replayed"], (fam, r["steps"])


def test_wide_corridors_and_stage_state_after_a_solve():
    """cmax above 42 (the corridor producer can emit up to 64 planes per knot; the load kernel tiles its
    LDS transpose so any cmax fits), and the handle's staged state is invalid after a solve: the stage
    entry points must refuse to run until cilqr_stage_load is called again."""
    sc = scenario.generate("mix11", 40, seed=77)
    B, K, c0 = 40, sc["n_steps"] + 1, sc["cmax"]
    wide = 50
    cor = np.zeros((B, K, wide, 3))
    cor[:, :, :c0] = sc["corridor"]
    cnt = sc["ccount"].copy()
    rng = np.random.default_rng(3)
    for b in range(0, B, 2):                      # far-away copies of the live planes: up to 50 per knot
        for i in range(0, K, 3):
            n0 = int(cnt[b, i])
            extra = int(rng.integers(20, wide - n0 + 1))
            src = rng.integers(0, n0, extra)
            pl = cor[b, i, src].copy()
            pl[:, 2] += (3.0 + 20.0 * rng.random(extra)) * np.hypot(pl[:, 0], pl[:, 1])
            cor[b, i, n0:n0 + extra] = pl
            cnt[b, i] = n0 + extra
    sc2 = dict(sc, corridor=cor, ccount=cnt, cmax=wide)
    assert cnt.max() > 42
    opt = _opt(sc2)
    g = _plan(opt, sc2)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc2, ocfg)
    assert_parity(g, ref, max_unstable_frac=0.15, what="wide corridors")
    assert_steps(g, sc2, ocfg, what="wide corridors")
    L, h = opt.L, opt.h
    assert L.cilqr_stage_init_guess(h) == api.ERR_STATE and L.cilqr_stage_quadratize(h) == api.ERR_STATE
    cost = np.zeros((B, 5))
    assert L.cilqr_stage_total_cost(h, cost.ctypes.data, api.MEM_HOST) == api.ERR_STATE
    x = np.zeros((B, K, 6))
    assert L.cilqr_stage_read(h, api.T_X, x.ctypes.data, api.MEM_HOST) == api.ERR_STATE
    opt.stage_load(sc2)
    opt.stage_init_guess()
    opt.stage_quadratize()                        # works again after a load
    opt.close()


@pytest.mark.parametrize("family,seed,min_share", [("dyn20", 43, 0.9), ("dyn20x", 46, 1.0)])
def test_config4_barriers_are_active_at_the_init_guess(family, seed, min_share):
    """BASELINE configs[4] is "constrained-ILQR barrier active" (at least one corridor / lane barrier
    in the relaxed region at the init guess).  From the device's own init guess and shrunk + normalised planes:
    the share of problems with a barrier argument g > -eps (RelaxBarrierFunction's relaxed branch,
    barrier_function.h:104-113) at some (knot, disc, plane); then the solve itself against the oracle.  "dyn20x"
    keeps every obstacle down to 0.6 m from the coarse path, so all of its problems start inside a barrier."""
    B = 96
    sc = scenario.generate(family, B, seed=seed)
    opt = _opt(sc)
    opt.stage_load(sc)
    opt.stage_init_guess()
    X, cor = opt.read(api.T_X), opt.read(api.T_CORRIDOR)
    cfg = opt.cfg
    Ld = (cfg.front_hang + cfg.wheel_base + cfg.rear_hang) / cfg.num_of_disc
    off = np.array([Ld * (j - 0.5) - cfg.rear_hang for j in range(cfg.num_of_disc)])       # cc:556-565
    px = X[:, :, 0, None] + off * np.cos(X[:, :, 2, None])                                   # [B,K,D]
    py = X[:, :, 1, None] + off * np.sin(X[:, :, 2, None])
    g = cor[..., 0, None] * px[:, :, None, :] + cor[..., 1, None] * py[:, :, None, :] - cor[..., 2, None]   # [B,K,C,D]
    live = np.arange(cor.shape[2])[None, None, :, None] < sc["ccount"][:, :, None, None]
    relaxed = ((g > -cfg.barrier_eps) & live).reshape(B, -1).sum(axis=1)
    share = float((relaxed > 0).mean())
    print(f"\n[{family}] problems with a relaxed corridor barrier at the init guess: {share:.2f}, "
          f"relaxed terms per problem: mean {relaxed.mean():.0f}")
    assert share >= min_share
    cost0 = opt.stage_total_cost()
    assert np.median(cost0[:, 3]) > 100.0            # the corridor component dominates the initial cost
    g_ = _plan(opt, sc)
    ocfg = oracle_cfg_from(cfg)
    ref = oracle_reference(sc, ocfg)
    assert_parity(g_, ref, what=family)
    assert_steps(g_, sc, ocfg, what=family)
    opt.close()


def test_exit_paths_at_batch_scale():
    """UNSOLVED / MAX_ITER / converged exits with thousands of problems in lockstep (compaction, the team backward
    kernel and the speculative line search all switch on and off while problems leave through different exits):
    256 distinct "dyn20x" scenes tiled 32x with the tolerances at 0 and 24 iterations -- every copy bit-identical,
    statuses and iteration counts of the distinct scenes equal to the oracle's on its stable problems."""
    base = scenario.generate("dyn20x", 256, seed=93)
    rep = 32
    sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if isinstance(v, np.ndarray) and v.shape[:1] == (256,) else v)
          for k, v in base.items()}
    opt = _opt(sc, max_iter=24, abs_cost_tol=0.5, rel_cost_tol=0.0)
    g = opt.plan(sc, alpha_trace=True)
    for k in ("traj", "n_cost", "status", "n_iter", "alpha_trace"):
        v = g[k].reshape(rep, 256, *g[k].shape[1:])
        assert np.array_equal(v, np.broadcast_to(v[:1], v.shape)), k
    hist = np.bincount(g["status"][:256], minlength=7)
    print(f"\nstatus histogram of the 256 distinct scenes: {hist.tolist()}")
    assert hist[api.ST_MAX_ITER] > 0 and hist[api.ST_CONVERGED_ABS] > 0
    first = {k: v[:256] for k, v in g.items() if isinstance(v, np.ndarray)}
    ref = oracle_reference(base, oracle_cfg_from(opt.cfg))
    assert_parity(first, ref, max_unstable_frac=0.15, what="exit paths at batch scale")
    opt.close()


def test_golden_fixtures_through_the_c_abi():
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))):
        g = np.load(path)
        sc = {k: g[k] for k in ("start", "coarse", "corridor", "ccount", "left", "right")}
        sc.update(n_steps=int(g["n_steps"]), cmax=int(g["cmax"]))
        opt = _opt(sc)
        r = opt.plan(sc, max_iter_trajs=8)
        assert np.array_equal(r["n_cost"], g["ref_n_cost"]) and np.array_equal(r["status"], g["ref_status"])
        assert np.array_equal(r["n_iter"], g["ref_n_iter"])
        for b in range(sc["start"].shape[0]):
            nc = int(g["ref_n_cost"][b])
            assert cost_err(r["cost_hist"][b, :nc], g["ref_cost_hist"][b, :nc]) < REL_TOL
            assert traj_err(r["traj"][b], g["ref_traj"][b]) < REL_TOL
        # iter_trajs of the first scene: init guess + accepted non-final iterates (cc:170,294)
        n_it = int(g["st_n_iter_trajs"])
        assert r["n_iter_trajs"][0] == n_it
        k = min(n_it, 8)
        assert traj_err(r["iter_trajs"][0, :k], g["st_iter_trajs"][:k]) < REL_TOL
        opt.close()


@pytest.mark.parametrize("over,expect", [
    (dict(max_iter=3), api.ST_MAX_ITER),
    (dict(rel_cost_tol=0.0, abs_cost_tol=5.0), api.ST_CONVERGED_ABS),
    (dict(rel_cost_tol=0.0, abs_cost_tol=0.0, max_iter=60), None),
])
def test_exit_paths(over, expect, both_paths):
    """Every exit of Optimize() (max-iter, abs tol, lambda > 1e11 / gnorm) agrees with the oracle."""
    noisy = over.get("rel_cost_tol", 1.0) == 0.0 and over.get("abs_cost_tol", 1.0) == 0.0
    if noisy:
        # With both tolerances at 0 the solver iterates into the rounding-noise plateau, where accept / reject decisions
        # hang on the last bits of a cost difference and most problems are not reproducible by the oracle itself.  The
        # scenes of this case are screened (tests/golden/make_zero_tolerance_set.py): 32 of the 48 are problems whose
        # zero-tolerance solve the oracle reproduces under 4e-16 input noise with every decision >= 1e-9 from its threshold.
        import json
        sel = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zero_tolerance_scenes.json")))
        assert sel["config"] == {k: over[k] for k in sel["config"]}
        pool = scenario.generate(sel["family"], sel["pool"], seed=sel["pool_seed"])
        idx = np.asarray(sel["indices"])
        sc = {k: (np.ascontiguousarray(v[idx]) if isinstance(v, np.ndarray) and v.shape[:1] == (sel["pool"],) else v) for k, v in pool.items()}
    else:
        sc = scenario.generate("ped6", 48, seed=51)
    opt = _opt(sc, **over)
    g = _plan(opt, sc)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc, ocfg)
    if noisy:
        # whole solves are only comparable where no decision of the oracle hung on the last bits of a cost
        # difference (relative distance to its threshold under 1e-9): the others count as unstable too
        ref["stable"] &= ref["min_margin"] >= 1e-9
        assert ref["stable"].sum() >= 24, "the screened set must be stable for at least half of its problems"
    assert_parity(g, ref, max_unstable_frac=0.5 if noisy else 0.125, what=str(over))
    # the lambda > 1e11 exit is reached through noisy rejections by nature: no stable problem ends there, every one of
    # its steps still has to replay in the oracle or be shown discontinuous there
    steps = assert_steps(g, sc, ocfg, what=str(over), max_excused_frac=0.10 if noisy else 0.02)
    print(f"\n{over}: steps {steps}")
    if expect is not None:
        assert (g["status"] == expect).sum() >= 1
    else:
        assert set(np.unique(g["status"])) <= {api.ST_GNORM, api.ST_UNSOLVED, api.ST_MAX_ITER}
        assert (g["status"] == api.ST_UNSOLVED).sum() >= 1 and (g["status"] == api.ST_GNORM).sum() >= 1
    opt.close()


# ---------------------------------------------------------------------------------------------
# hostile inputs (VERDICT r04 missing #4): what the reference's arithmetic does with NaN / Inf, inside a batch
# ---------------------------------------------------------------------------------------------
def _poison_cases(cmax):
    """(problem index, what, how).  Every one of them makes the reference's first TotalCost non-finite (cc:172), so every trial
    of every iteration is rejected -- z = dcost / expected is NaN, cc:255-258 -- and the solve leaves through lambda > 1e11
    after ten iterations (cc:298-307) with one Cost row and the init guess as its trajectory; a NaN behind a knot's live
    plane count is never read."""
    return [
        (11, "start x NaN", lambda s, b: s["start"].__setitem__((b, 0), np.nan)),
        (37, "start x +Inf", lambda s, b: s["start"].__setitem__((b, 0), np.inf)),
        (58, "start v NaN", lambda s, b: s["start"].__setitem__((b, 3), np.nan)),
        (64, "coarse[20].x NaN", lambda s, b: s["coarse"].__setitem__((b, 20, 0), np.nan)),
        (65, "coarse[20].x +Inf", lambda s, b: s["coarse"].__setitem__((b, 20, 0), np.inf)),
        (99, "coarse[N].y NaN", lambda s, b: s["coarse"].__setitem__((b, -1, 1), np.nan)),
        (127, "coarse[20].theta -Inf", lambda s, b: s["coarse"].__setitem__((b, 20, 2), -np.inf)),
        (128, "plane a NaN", lambda s, b: s["corridor"].__setitem__((b, 10, 0, 0), np.nan)),
        (191, "plane c +Inf", lambda s, b: s["corridor"].__setitem__((b, 10, 0, 2), np.inf)),
        (192, "plane c -Inf", lambda s, b: s["corridor"].__setitem__((b, 30, 1, 2), -np.inf)),
        (255, "plane b NaN, last knot", lambda s, b: s["corridor"].__setitem__((b, -1, 2, 1), np.nan)),
        (299, "NaN behind the live planes", lambda s, b: s["corridor"].__setitem__((b, 10, cmax - 1, 0), np.nan)),
    ]


def _same_or_both_nonfinite(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert np.array_equal(fa, fb), f"{what}: finite / non-finite pattern differs"
    if fa.any():
        scale = max(1.0, float(np.abs(b[fb]).max()))
        assert float(np.abs(a[fa] - b[fb]).max()) <= tol * scale, what


def test_hostile_inputs_inside_a_batch(both_paths):
    """One NaN / Inf problem of every kind inside a batch of 300 (start, coarse knot, plane coefficient): status, iteration
    count and Cost-row count as the oracle's on the same poisoned inputs (UNSOLVED after ten iterations), the trajectory =
    the init guess with the oracle's finite / non-finite pattern, a non-finite Cost total and every category the oracle has finite; every other
    problem of the batch bit-identical to the same batch without the poisoned problems; no hang."""
    B = 300
    clean = scenario.generate("mix11", B, seed=97)
    assert (clean["ccount"][:, [10, 30, -1]] >= 3).all() and (clean["ccount"][299, 10] < clean["cmax"])
    opt = _opt(clean)
    g0 = _plan(opt, clean)
    sc = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in clean.items()}
    cases = _poison_cases(clean["cmax"])
    for b, _, f in cases:
        f(sc, b)
    g1 = _plan(opt, sc)
    bad = np.zeros(B, bool)
    bad[[b for b, _, _ in cases[:-1]]] = True          # the last case poisons nothing that is read
    for k in ("traj", "cost_hist", "status", "n_cost", "n_iter"):
        assert np.array_equal(g1[k][~bad], g0[k][~bad]), f"{k}: a finite neighbour changed"
    ocfg = oracle_cfg_from(opt.cfg)
    sub = {k: (np.ascontiguousarray(v[bad]) if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    ref = orc.solve_batch(sub, ocfg)
    idx = np.nonzero(bad)[0]
    assert (ref["status"] == api.ST_UNSOLVED).all() and (ref["n_cost"] == 1).all() and (ref["n_iter"] == 10).all()
    for j, b in enumerate(idx):
        what = [w for i, w, _ in cases if i == b][0]
        assert g1["status"][b] == ref["status"][j] and g1["n_cost"][b] == ref["n_cost"][j] and g1["n_iter"][b] == ref["n_iter"][j], \
            (what, g1["status"][b], g1["n_cost"][b], g1["n_iter"][b])
        _same_or_both_nonfinite(g1["traj"][b], ref["traj"][j], 1e-9, f"{what}: trajectory")
        # the only Cost row: its total is non-finite on both sides (that is what rejects every trial).  Category by category
        # the library may be finite where the reference is NaN: the branch-free barrier takes max(-g, eps) (dev_model.hpp),
        # which drops the NaN of a constraint evaluated at a NaN *state* -- that state's J term carries the NaN into the total
        # instead.  Never the other way round, and finite on both sides means equal.
        row, rrow = g1["cost_hist"][b, 0], ref["cost_hist"][j, 0]
        assert not np.isfinite(row[0]) and not np.isfinite(rrow[0]), (what, row, rrow)
        if what.startswith("plane"):
            # the documented difference (non-finite inputs): a live plane with a non-finite coefficient is stored
            # as (0, 0, -inf) at load, so total and corridor category are +inf where the reference's arithmetic gives NaN
            assert np.isposinf(row[0]) and np.isposinf(row[3]) and np.isnan(rrow[0]) and np.isnan(rrow[3]), (what, row, rrow)
        for c in range(1, 5):
            if np.isfinite(rrow[c]):
                assert np.isfinite(row[c]) and abs(row[c] - rrow[c]) <= 1e-9 * max(1.0, abs(rrow[c])), (what, c, row, rrow)
    opt.close()


def test_hostile_lane_tables(both_paths):
    """A NaN in a lane point: the two segments that touch it have NaN distances, the strict `<` of cc:610-613 never picks
    them, and the solve goes on with the others -- oracle and library alike."""
    sc = scenario.generate("mix11", 64, seed=98)
    sc["left"] = sc["left"].copy()
    sc["left"][7, 0] = np.nan
    opt = _opt(sc)
    g = _plan(opt, sc)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc, ocfg)
    assert_parity(g, ref, max_unstable_frac=0.125, what="one NaN lane point")
    assert_steps(g, sc, ocfg, what="one NaN lane point")
    assert np.isfinite(g["traj"]).all()
    # no finite lane end point at all: the reference's loop indexes [-1] (undefined behaviour; the oracle takes segment 0 and
    # ends UNSOLVED).  The boundary refuses the call instead -- there is no box to build the lane grid over
    sc["left"] = np.full_like(sc["left"], np.nan)
    sc["right"] = np.full_like(sc["right"], np.nan)
    with pytest.raises(api.CilqrError) as e:
        _plan(opt, sc)
    assert e.value.code == api.ERR_ARG
    ref = orc.solve_batch(sc, ocfg)
    assert (ref["status"] == api.ST_UNSOLVED).all() and (ref["n_iter"] == 10).all()
    opt.close()


def test_degenerate_limits(both_paths):
    """max_iter = 1 (the smallest the boundary accepts; 0 is CILQR_ERR_ARG like every other non-positive size) and a batch whose
    every knot has a plane count of 0 (no corridor term at all, cc:560-581 loop bodies never run)."""
    sc = scenario.generate("ped6", 40, seed=99)
    with pytest.raises(api.CilqrError):
        _opt(sc, max_iter=0)
    opt = _opt(sc, max_iter=1)
    g = _plan(opt, sc)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc, ocfg)
    assert_parity(g, ref, max_unstable_frac=0.125, what="max_iter = 1")
    assert set(np.unique(g["status"])) <= {api.ST_MAX_ITER, api.ST_CONVERGED_ABS, api.ST_CONVERGED_REL}
    opt.close()
    sc["ccount"] = np.zeros_like(sc["ccount"])
    opt = _opt(sc)
    g = _plan(opt, sc)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc, ocfg)
    assert_parity(g, ref, max_unstable_frac=0.125, what="no corridor planes")
    assert_steps(g, sc, ocfg, what="no corridor planes")
    assert (g["cost_hist"][:, 0, 3] == 0.0).all()          # the corridor category of every Cost row is an empty sum
    opt.close()


def test_ragged_counts_single_problem_and_odd_batches(both_paths):
    sc = scenario.generate("mix11", 130, seed=61)
    # ragged corridor: drop to the 4 box planes on some knots, keep everything on others
    sc["ccount"][::3, ::2] = 4
    full = _opt(sc)
    g = _plan(full, sc)
    ref = oracle_reference(sc, oracle_cfg_from(full.cfg))
    assert_parity(g, ref, what="ragged")
    assert_steps(g, sc, oracle_cfg_from(full.cfg), what="ragged")
    # problems are independent: any sub-batch (1, 63, 65) gives bit-identical results
    for lo, hi in [(7, 8), (0, 63), (65, 130)]:
        sub = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (130,) else v) for k, v in sc.items()}
        r = full.plan(sub)
        assert np.array_equal(r["traj"], g["traj"][lo:hi]) and np.array_equal(r["cost_hist"], g["cost_hist"][lo:hi])
        assert np.array_equal(r["status"], g["status"][lo:hi])
    # idempotence: same handle, same input, same bits
    g2 = full.plan(sc)
    assert np.array_equal(g2["traj"], g["traj"]) and np.array_equal(g2["cost_hist"], g["cost_hist"])
    full.close()


def test_argument_errors_mirror_plan():
    sc = scenario.generate("ped6", 4, seed=71)
    opt = _opt(sc)
    L, h = opt.L, opt.h
    prob, keep = opt._host_problem(sc)
    K, M = opt.K, opt.cfg.max_iter
    traj, hist = np.zeros((4, K, 10)), np.zeros((4, M + 1, 5))
    nc, st = np.zeros(4, np.int32), np.zeros(4, np.int32)

    def sol(**kw):
        d = dict(memory=api.MEM_HOST, max_iter_trajs=0, traj=traj.ctypes.data, cost_hist=hist.ctypes.data,
                 n_cost=nc.ctypes.data, status=st.ctypes.data, n_iter=None, iter_trajs=None, n_iter_trajs=None)
        d.update(kw)
        return api.SolutionBatch(**d)

    assert L.cilqr_solve_batch(h, C.byref(prob), C.byref(sol())) == api.OK
    assert L.cilqr_solve_batch(h, C.byref(prob), C.byref(sol(traj=None))) == api.ERR_NULL      # cc:64
    assert L.cilqr_solve_batch(h, C.byref(prob), None) == api.ERR_NULL
    p2 = api.ProblemBatch.from_buffer_copy(prob); p2.n_left = 0
    assert L.cilqr_solve_batch(h, C.byref(p2), C.byref(sol())) == api.ERR_CONSTRAINTS           # cc:68-73
    p3 = api.ProblemBatch.from_buffer_copy(prob); p3.corridor = None
    assert L.cilqr_solve_batch(h, C.byref(p3), C.byref(sol())) == api.ERR_CONSTRAINTS
    p4 = api.ProblemBatch.from_buffer_copy(prob); p4.n_knots = K - 1
    assert L.cilqr_solve_batch(h, C.byref(p4), C.byref(sol())) == api.ERR_KNOTS                 # cc:75-78
    p5 = api.ProblemBatch.from_buffer_copy(prob); p5.batch = 5
    assert L.cilqr_solve_batch(h, C.byref(p5), C.byref(sol())) == api.ERR_CAPACITY
    fresh = _opt(sc)
    assert fresh.L.cilqr_stage_quadratize(fresh.h) == api.ERR_STATE
    assert fresh.L.cilqr_stage_init_guess(fresh.h) == api.ERR_STATE
    fresh.close()
    del keep
    opt.close()


def test_device_memory_interface_and_stream():
    torch = pytest.importorskip("torch")
    sc = scenario.generate("mix11", 100, seed=81)
    B, K = 100, sc["n_steps"] + 1
    opt = _opt(sc)
    host = opt.plan(sc)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    opt.set_stream(stream.cuda_stream)
    t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("start", "coarse", "corridor", "ccount")}
    M = opt.cfg.max_iter
    o_traj = torch.zeros((B, K, 10), dtype=torch.float64, device=dev)
    o_hist = torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev)
    o_nc = torch.zeros(B, dtype=torch.int32, device=dev)
    o_st = torch.zeros(B, dtype=torch.int32, device=dev)
    left, right = np.ascontiguousarray(sc["left"]), np.ascontiguousarray(sc["right"])
    torch.cuda.synchronize()
    prob = opt.make_problem(B, t["start"].data_ptr(), t["coarse"].data_ptr(), t["corridor"].data_ptr(),
                            t["ccount"].data_ptr(), sc["cmax"], left.ctypes.data, right.ctypes.data,
                            left.shape[0], right.shape[0], api.MEM_DEVICE)
    sol = api.SolutionBatch(api.MEM_DEVICE, 0, o_traj.data_ptr(), o_hist.data_ptr(), o_nc.data_ptr(),
                            o_st.data_ptr(), None, None, None)
    assert opt.solve_raw(prob, sol) == api.OK
    torch.cuda.synchronize()
    assert np.array_equal(o_traj.cpu().numpy(), host["traj"])
    assert np.array_equal(o_nc.cpu().numpy(), host["n_cost"]) and np.array_equal(o_st.cpu().numpy(), host["status"])
    hist = o_hist.cpu().numpy()
    for b in range(B):
        assert np.array_equal(hist[b, :host["n_cost"][b]], host["cost_hist"][b, :host["n_cost"][b]])
    opt.close()


def test_per_problem_lane_tables_as_groups():
    """Every Plan call of the reference carries its own lane constraints (ilqr_optimizer.h:41-48); a batch carries them
    as groups of problems that share a table (cilqr_problem_batch::n_lane_groups).  Two tables over the same road
    (boundaries sampled every 5 m and every 7 m -- different segments, different planes): the grouped solve equals
    the two separate solves bit for bit, each group matches the oracle run with ITS table, and the tables do differ."""
    sc = scenario.generate("mix11", 60, seed=170)
    road = scenario.build_road()
    left7, right7 = scenario.lane_constraints(road, seg_len=7.0)
    n0 = 25
    part = lambda lo, hi, l, r: dict(sc, **{k: sc[k][lo:hi] for k in ("start", "coarse", "corridor", "ccount")}, left=l, right=r)
    a, b = part(0, n0, sc["left"], sc["right"]), part(n0, 60, left7, right7)
    grouped = dict(sc, left=np.concatenate([sc["left"], left7]), right=np.concatenate([sc["right"], right7]),
                   lane_groups=[(0, len(sc["left"]), len(sc["right"])), (n0, len(left7), len(right7))])
    opt = _opt(sc, B=60)
    opt.close()
    opt = api.BatchIlqrOptimizer(api.default_config(sc["n_steps"]), batch_capacity=60, cmax=sc["cmax"], max_lane_segments=128)
    g = _plan(opt, grouped)
    ga, gb = _plan(opt, a), _plan(opt, b)
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "alpha_trace", "iter_trajs"):
        assert np.array_equal(g[k][:n0], ga[k]) and np.array_equal(g[k][n0:], gb[k]), k
    ocfg = oracle_cfg_from(opt.cfg)
    for sub_, res in ((a, ga), (b, gb)):
        ref = oracle_reference(sub_, ocfg)
        assert_parity(res, ref, max_unstable_frac=0.15, what="lane groups")
        assert_steps(res, sub_, ocfg, what="lane groups")
    shared = _plan(opt, part(n0, 60, sc["left"], sc["right"]))
    assert not np.array_equal(shared["traj"], gb["traj"])            # the second table is a different constraint set
    # the stage API works on one table
    prob, keep = opt._host_problem(grouped)
    assert opt.L.cilqr_stage_load(opt.h, C.byref(prob)) == api.ERR_ARG
    opt.close()


def test_tracker_init_guess():
    """CILQR_INIT_TRACKER: IlqrOptimizer::InitGuess through the closed-loop Tracker (ilqr_optimizer.cc:107-139,
    tracker.cc) -- the alternative init guess the reference keeps commented out (cc:168) and recommends
    (README.md:61-67).  The device's init guess against oracle/tracker_oracle.cc with the stations of the DP planner's
    coarse trajectories and with chord-length stations, then whole solves started from it, every step replayed in the
    oracle from the device's own iterates.  A problem is left out of the stage comparison when one of the oracle's DARE
    loops stopped within 1e-9 (relative) of its tolerance: one iteration more or less changes the gains by ~1e-2."""
    g = scenario.generate_dp("demo80", 24, seed=180, workers=8)
    ok = np.nonzero(g["found"])[0]
    assert len(ok) >= 12
    take = lambda a: np.ascontiguousarray(a[ok])
    B, K = len(ok), 81
    for with_station in (True, False):
        sc = dict(start=take(g["start"]), coarse=take(g["coarse"]), left=g["left"], right=g["right"], n_steps=80, cmax=16)
        if with_station:
            sc["coarse_station"] = take(g["dp"][:, :, 1])
        # corridors for the solve: a plain box around every knot (the init guess does not read them)
        th = sc["coarse"][:, :, 2]
        n = np.stack([np.stack([np.cos(th), np.sin(th)], -1), np.stack([-np.cos(th), -np.sin(th)], -1),
                      np.stack([-np.sin(th), np.cos(th)], -1), np.stack([np.sin(th), -np.cos(th)], -1)], 2)   # [B,K,4,2]
        c = (n * sc["coarse"][:, :, None, :2]).sum(-1) + 10.0
        sc["corridor"] = np.zeros((B, K, 16, 3))
        sc["corridor"][:, :, :4, :2] = n
        sc["corridor"][:, :, :4, 2] = c
        sc["ccount"] = np.full((B, K), 4, np.int32)
        cfg = api.default_config(80, init_guess=api.INIT_TRACKER)
        opt = api.BatchIlqrOptimizer(cfg, batch_capacity=B, cmax=16, max_lane_segments=64)
        opt.stage_load(sc)
        opt.stage_init_guess()
        X, U = opt.read(api.T_X), opt.read(api.T_U)
        compared = 0
        for b in range(B):
            oX, oU, margin = orc.tracker_init_guess(sc["start"][b], sc["coarse"][b], sc.get("coarse_station", [None] * B)[b])
            if margin < 1e-9:
                continue
            compared += 1
            assert traj_err(X[b], oX) < STAGE_TOL and traj_err(U[b], oU) < STAGE_TOL, (b, with_station)
        assert compared >= B - 2
        # the tracker follows the coarse path far better than iqr's open-loop-ish rollout: lower initial cost
        cost_tr = opt.stage_total_cost()[:, 0]
        iq = api.BatchIlqrOptimizer(api.default_config(80), batch_capacity=B, cmax=16, max_lane_segments=64)
        iq.stage_load(sc)
        iq.stage_init_guess()
        assert np.median(cost_tr) < np.median(iq.stage_total_cost()[:, 0])
        iq.close()
        res = _plan(opt, sc)
        assert ((res["status"] >= 1) & (res["status"] <= 5)).all()
        assert traj_err(res["iter_trajs"][:, 0, :, 1:7], X) == 0.0            # iter_trajs[0] is the init guess (cc:170)
        assert_steps(res, sc, oracle_cfg_from(opt.cfg), what=f"tracker init guess, stations {with_station}")
        if with_station:
            # the same batch as two lane groups (the table twice): every group must project onto ITS problems' stations
            n0 = B // 2
            grouped = dict(sc, left=np.concatenate([sc["left"], sc["left"]]), right=np.concatenate([sc["right"], sc["right"]]),
                           lane_groups=[(0, len(sc["left"]), len(sc["right"])), (n0, len(sc["left"]), len(sc["right"]))])
            assert not np.array_equal(sc["coarse_station"][:B - n0], sc["coarse_station"][n0:])
            gr = _plan(opt, grouped)
            for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "alpha_trace", "iter_trajs"):
                assert np.array_equal(gr[k], res[k]), k
        opt.close()
    with pytest.raises(api.CilqrError):
        api.BatchIlqrOptimizer(api.default_config(50, init_guess=7), batch_capacity=4)


def test_gather_results_through_the_c_abi_single_rank():
    """cilqr_comm_* / cilqr_gather_results (librccl loaded with dlopen, no PyTorch involved in the exchange) with a
    one-rank communicator -- all a 1-GPU box can hold: RCCL initialises, the results are packed (8 trajectory
    columns, live Cost rows only), unpacked on the root and equal the local results bit for bit (time and kappa
    rebuilt), rows past n_cost stay untouched.  State errors mirror the rest of the ABI."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    sc = scenario.generate("mix11", 150, seed=160)
    B, K = 150, sc["n_steps"] + 1
    opt = _opt(sc)
    M = opt.cfg.max_iter
    t = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(dev) for k in ("start", "coarse", "corridor", "ccount")}
    loc = dict(traj=torch.zeros((B, K, 10), dtype=torch.float64, device=dev), hist=torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
               nc=torch.zeros(B, dtype=torch.int32, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev),
               ni=torch.zeros(B, dtype=torch.int32, device=dev))
    gat = {k: torch.full_like(v, -7) for k, v in loc.items()}
    left, right = np.ascontiguousarray(sc["left"]), np.ascontiguousarray(sc["right"])
    torch.cuda.synchronize()
    prob = opt.make_problem(B, t["start"].data_ptr(), t["coarse"].data_ptr(), t["corridor"].data_ptr(), t["ccount"].data_ptr(),
                            sc["cmax"], left.ctypes.data, right.ctypes.data, left.shape[0], right.shape[0], api.MEM_DEVICE)

    def sol(d):
        return api.SolutionBatch(api.MEM_DEVICE, 0, d["traj"].data_ptr(), d["hist"].data_ptr(), d["nc"].data_ptr(),
                                 d["st"].data_ptr(), d["ni"].data_ptr(), None, None)

    assert opt.solve_raw(prob, sol(loc)) == api.OK
    assert opt.gather_results_raw(B, sol(loc), 0, sol(gat)) == api.ERR_STATE         # no communicator yet
    uid = api.comm_unique_id()
    assert len(uid) == api.UNIQUE_ID_BYTES
    opt.comm_create(uid, 0, 1)
    with pytest.raises(api.CilqrError):
        opt.comm_create(uid, 0, 1)                                                    # one communicator per handle
    assert opt.gather_results_raw(B, sol(loc), 1, sol(gat)) == api.ERR_ARG           # root outside the world
    assert opt.gather_results_raw(B, sol(loc), 0, None) == api.ERR_NULL              # the root needs a destination
    assert opt.gather_results_raw(B, sol(loc), 0, sol(gat)) == api.OK
    torch.cuda.synchronize()
    for k in ("traj", "nc", "st", "ni"):
        assert torch.equal(gat[k], loc[k]), k
    live = torch.arange(M + 1, device=dev)[None, :] < loc["nc"][:, None].long()
    assert torch.equal(gat["hist"][live], loc["hist"][live])
    assert bool((gat["hist"][~live] == -7).all())                                     # rows >= n_cost untouched
    opt.comm_destroy()
    assert opt.L.cilqr_comm_destroy(opt.h) == api.ERR_STATE
    opt.close()


def test_gather_results_more_live_rows_than_the_message_region_holds():
    """A rank's message has room for 32 live Cost rows per problem; zero tolerances make a batch keep 43 on average: the rows
    beyond the region travel in the second exchange (comm.hip, step 4).  One rank: packed in two windows, unpacked from two
    buffers -- the gathered history equals the local one row for row."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    sc = scenario.generate("mix11", 60, seed=161)
    B, K = 60, sc["n_steps"] + 1
    cfg = api.default_config(sc["n_steps"], rel_cost_tol=0.0, abs_cost_tol=0.0, max_iter=90)
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=B, cmax=sc["cmax"])
    M = cfg.max_iter
    g = opt.plan(sc)
    assert int(g["n_cost"].sum()) > 32 * B + 200
    loc = dict(traj=torch.from_numpy(g["traj"]).to(dev), hist=torch.from_numpy(g["cost_hist"]).to(dev),
               nc=torch.from_numpy(g["n_cost"]).to(dev), st=torch.from_numpy(g["status"]).to(dev), ni=torch.from_numpy(g["n_iter"]).to(dev))
    gat = {k: torch.full_like(v, -7) for k, v in loc.items()}

    def sol(d):
        return api.SolutionBatch(api.MEM_DEVICE, 0, d["traj"].data_ptr(), d["hist"].data_ptr(), d["nc"].data_ptr(),
                                 d["st"].data_ptr(), d["ni"].data_ptr(), None, None)

    torch.cuda.synchronize()
    opt.comm_create(api.comm_unique_id(), 0, 1)
    for _ in range(2):      # (the second call: the same staging blocks again)
        assert opt.gather_results_raw(B, sol(loc), 0, sol(gat)) == api.OK
        torch.cuda.synchronize()
        for k in ("traj", "nc", "st", "ni"):
            assert torch.equal(gat[k], loc[k]), k
        live = torch.arange(M + 1, device=dev)[None, :] < loc["nc"][:, None].long()
        assert torch.equal(gat["hist"][live], loc["hist"][live]) and bool((gat["hist"][~live] == -7).all())
        gat["hist"].fill_(-7)
    opt.comm_destroy()
    opt.close()


def test_submit_wait_keeps_batches_in_flight_on_separate_handles():
    """cilqr_submit / cilqr_wait: three handles on three streams solve three different batches
    concurrently; every result equals the synchronous solve of the same batch, bit for bit.
    State errors: waiting with nothing submitted, submitting twice."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    scenes = [scenario.generate("mix11", 700, seed=130 + i) for i in range(3)]
    sync = []
    for sc in scenes:
        o = _opt(sc)
        sync.append(o.plan(sc))
        o.close()
    K, M = scenes[0]["n_steps"] + 1, None
    jobs = []
    for sc in scenes:
        o = _opt(sc)
        M = o.cfg.max_iter
        st = torch.cuda.Stream()
        o.set_stream(st.cuda_stream)
        B = sc["coarse"].shape[0]
        keep = {k: np.ascontiguousarray(sc[k]) for k in ("start", "coarse", "corridor", "ccount", "left", "right")}
        bufs = (torch.zeros((B, K, 10), dtype=torch.float64, device=dev),
                torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
                torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
        prob = o.make_problem(B, keep["start"].ctypes.data, keep["coarse"].ctypes.data, keep["corridor"].ctypes.data,
                              keep["ccount"].ctypes.data, sc["cmax"], keep["left"].ctypes.data,
                              keep["right"].ctypes.data, keep["left"].shape[0], keep["right"].shape[0], api.MEM_HOST)
        sol = api.SolutionBatch(api.MEM_DEVICE, 0, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
                                bufs[3].data_ptr(), None, None, None)
        jobs.append((o, st, keep, bufs, prob, sol))
    torch.cuda.synchronize()
    assert jobs[0][0].wait() == api.ERR_STATE
    for rep in range(2):                               # a handle is reusable after wait()
        for j in jobs:
            assert j[0].submit_raw(j[4], j[5]) == api.OK
        for j in jobs:
            assert j[0].wait() == api.OK
        torch.cuda.synchronize()
        for j, ref in zip(jobs, sync):
            assert np.array_equal(j[3][0].cpu().numpy(), ref["traj"])
            assert np.array_equal(j[3][2].cpu().numpy(), ref["n_cost"])
            assert np.array_equal(j[3][3].cpu().numpy(), ref["status"])
    for j in jobs:
        j[0].close()


@pytest.mark.parametrize("finish_threshold", [-1, 3000, 0])
def test_two_solves_in_flight_on_one_handle(finish_threshold):
    """cilqr_submit keeps TWO solves in flight on one handle (and a third queued behind them): the survivors of solve i move into the handle's finishing
    arena (CILQR_OPT_FINISH_THRESHOLD) and finish on a second stream while solve i+1 is iterated in the main arena.
    Five different batches -- larger than the finishing arena, smaller than it (handed over before the first iteration),
    smaller than the tail threshold (never handed over) -- through one handle, results bit-identical to
    cilqr_solve_batch on the same handle, in submission order; a third submit and a synchronous call in between are
    refused; host and device result buffers."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    sizes = [9000, 700, 8500, 100, 9000]
    scenes = [scenario.generate("mix11", n, seed=140 + i) for i, n in enumerate(sizes)]
    Bmax = max(sizes)
    opt = api.BatchIlqrOptimizer(api.default_config(scenes[0]["n_steps"]), batch_capacity=Bmax, cmax=scenes[0]["cmax"],
                                 max_lane_segments=64)
    if finish_threshold >= 0:
        opt.set_option(api.OPT_FINISH_THRESHOLD, finish_threshold)
    sync = [opt.plan(sc, alpha_trace=True) for sc in scenes]
    K, M = scenes[0]["n_steps"] + 1, opt.cfg.max_iter
    jobs = []
    for i, sc in enumerate(scenes):
        B = sc["coarse"].shape[0]
        keep = {k: np.ascontiguousarray(sc[k]) for k in ("start", "coarse", "corridor", "ccount", "left", "right")}
        prob = opt.make_problem(B, keep["start"].ctypes.data, keep["coarse"].ctypes.data, keep["corridor"].ctypes.data,
                                keep["ccount"].ctypes.data, sc["cmax"], keep["left"].ctypes.data,
                                keep["right"].ctypes.data, keep["left"].shape[0], keep["right"].shape[0], api.MEM_HOST)
        if i % 2 == 0:   # device buffers
            bufs = (torch.zeros((B, K, 10), dtype=torch.float64, device=dev), torch.zeros((B, M + 1, 5), dtype=torch.float64, device=dev),
                    torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                    torch.zeros(B, dtype=torch.int32, device=dev))
            sol = api.SolutionBatch(api.MEM_DEVICE, 0, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
                                    bufs[3].data_ptr(), bufs[4].data_ptr(), None, None)
        else:            # host buffers
            bufs = (np.zeros((B, K, 10)), np.zeros((B, M + 1, 5)), np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32))
            sol = api.SolutionBatch(api.MEM_HOST, 0, bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data,
                                    bufs[3].ctypes.data, bufs[4].ctypes.data, None, None)
        jobs.append((keep, bufs, prob, sol))
    torch.cuda.synchronize()
    assert opt.wait() == api.ERR_STATE                       # nothing submitted
    host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else a
    for rep in range(2):
        for b in jobs:
            for a in b[1]:
                if hasattr(a, "zero_"):
                    a.zero_()
                else:
                    a[...] = -3      # host arrays arrive dirty
        torch.cuda.synchronize()
        assert opt.submit_raw(jobs[0][2], jobs[0][3]) == api.OK
        assert opt.submit_raw(jobs[1][2], jobs[1][3]) == api.OK
        assert opt.submit_raw(jobs[2][2], jobs[2][3]) == api.OK            # the third is queued (its host arrays travel meanwhile)
        assert opt.submit_raw(jobs[3][2], jobs[3][3]) == api.ERR_STATE     # two in flight and one queued: collect first
        assert opt.solve_raw(jobs[3][2], jobs[3][3]) == api.ERR_STATE
        nxt = 3
        for i in range(len(jobs)):
            assert opt.wait() == api.OK                       # the oldest one
            if nxt < len(jobs):
                assert opt.submit_raw(jobs[nxt][2], jobs[nxt][3]) == api.OK
                nxt += 1
            ref, got = sync[i], jobs[i][1]
            torch.cuda.synchronize()
            assert np.array_equal(host(got[0]), ref["traj"]), (rep, i)
            nc = ref["n_cost"]
            assert np.array_equal(host(got[2]), nc) and np.array_equal(host(got[3]), ref["status"])
            assert np.array_equal(host(got[4]), ref["n_iter"])
            hist = host(got[1])
            for b in range(0, sizes[i], 37):
                assert np.array_equal(hist[b, :nc[b]], ref["cost_hist"][b, :nc[b]])
            if not hasattr(got[1], "cpu"):       # host memory: the whole array, zero rows included
                assert np.array_equal(hist, ref["cost_hist"]), (rep, i)
        assert opt.wait() == api.ERR_STATE
    again = opt.plan(scenes[0], alpha_trace=True)            # the synchronous call works again
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "alpha_trace"):
        assert np.array_equal(again[k], sync[0][k]), k
    opt.close()


def test_one_process_several_devices_sharded_solve():
    """cilqr_multi_*: one host call cut into contiguous shards, one solver handle per listed device, results written
    into the caller's arrays at each shard's offset.  On a one-GPU box the device is listed several times (logical
    shards): the sharded solve equals cilqr_solve_batch over the whole batch bit for bit -- uneven splits, the tracker
    init guess with caller-supplied stations (a per-problem input that must be offset per shard), and the argument
    errors."""
    sc = scenario.generate("mix11", 203, seed=190)
    one = _opt(sc)
    ref = _plan(one, sc)
    one.close()
    for devices in ((0, 0), (0, 0, 0), (0,)):
        m = api.MultiDeviceOptimizer(api.default_config(sc["n_steps"]), devices=devices, batch_capacity=203, cmax=sc["cmax"])
        n, first, dev = m.shards(203)
        assert n == len(devices) and first[0] == 0 and list(dev) == list(devices)
        assert np.all(np.diff(np.append(first, 203)) >= 203 // len(devices))          # contiguous, balanced
        g = m.plan(sc, max_iter_trajs=ITER_CAP, alpha_trace=True)
        for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "alpha_trace", "iter_trajs", "n_iter_trajs"):
            assert np.array_equal(g[k], ref[k]), (devices, k)
        sub = {k: (v[:77] if isinstance(v, np.ndarray) and v.shape[:1] == (203,) else v) for k, v in sc.items()}
        g2 = m.plan(sub)                                                                # a smaller batch on the same handle
        assert np.array_equal(g2["traj"], ref["traj"][:77]) and np.array_equal(g2["n_cost"], ref["n_cost"][:77])
        if len(devices) == 2:
            assert m.device_bytes() > 0
            grouped = dict(sc, lane_groups=[(0, len(sc["left"]), len(sc["right"])), (100, len(sc["left"]), len(sc["right"]))],
                           left=np.concatenate([sc["left"], sc["left"]]), right=np.concatenate([sc["right"], sc["right"]]))
            assert m.plan(grouped, check=False)["rc"] == api.ERR_ARG                   # one lane table per sharded call
            big = scenario.generate("mix11", 204, seed=1)
            assert m.plan(big, check=False)["rc"] == api.ERR_CAPACITY
        m.close()
    with pytest.raises(api.CilqrError):
        api.MultiDeviceOptimizer(api.default_config(50), devices=(0, 97), batch_capacity=8)   # no such device
    # per-problem stations follow their shard
    gdp = scenario.generate_dp("demo80", 16, seed=181, workers=8)
    ok = np.nonzero(gdp["found"])[0]
    take = lambda a_: np.ascontiguousarray(a_[ok])
    B, K = len(ok), 81
    th = take(gdp["coarse"])[:, :, 2]
    nrm = np.stack([np.stack([np.cos(th), np.sin(th)], -1), np.stack([-np.cos(th), -np.sin(th)], -1),
                    np.stack([-np.sin(th), np.cos(th)], -1), np.stack([np.sin(th), -np.cos(th)], -1)], 2)
    cor = np.zeros((B, K, 16, 3))
    cor[:, :, :4, :2] = nrm
    cor[:, :, :4, 2] = (nrm * take(gdp["coarse"])[:, :, None, :2]).sum(-1) + 10.0
    sct = dict(start=take(gdp["start"]), coarse=take(gdp["coarse"]), left=gdp["left"], right=gdp["right"], n_steps=80, cmax=16,
               corridor=cor, ccount=np.full((B, K), 4, np.int32), coarse_station=take(gdp["dp"][:, :, 1]))
    cfg = api.default_config(80, init_guess=api.INIT_TRACKER)
    one = api.BatchIlqrOptimizer(cfg, batch_capacity=B, cmax=16, max_lane_segments=64)
    ref = one.plan(sct)
    one.close()
    m = api.MultiDeviceOptimizer(cfg, devices=(0, 0), batch_capacity=B, cmax=16)
    g = m.plan(sct)
    for k in ("traj", "cost_hist", "n_cost", "status"):
        assert np.array_equal(g[k], ref[k]), k
    m.close()


def test_lean_log_and_reciprocal_are_accurate_to_an_ulp_or_two():
    """The barrier kernels use their own log / reciprocal (dev_model.hpp: log_pos, fast_rcp) instead
    of the library routines.  Against numpy on 400k points spanning the whole normal range and the
    barrier's working range: relative error of the reciprocal <= 2 ulp, absolute error of the log
    <= 2 ulp of max(|log x|, 1) -- nine orders of magnitude inside the 1e-4 parity tolerance."""
    sc = scenario.generate("ped6", 4, seed=3)
    opt = _opt(sc)
    rng = np.random.default_rng(5)
    x = np.concatenate([
        np.exp(rng.uniform(-700, 700, 100000)),            # whole normal range
        np.exp(rng.uniform(np.log(1e-3), np.log(1e3), 200000)),   # distances in metres
        1.0 + rng.uniform(-1e-3, 1e-3, 50000),             # log(x) ~ 0: cancellation-prone
        np.array([1.0, 2.0, 0.5, np.sqrt(0.5), np.nextafter(np.sqrt(0.5), 0), 0.01, 2.2250738585072014e-308,
                  1.7976931348623157e308]),
        rng.uniform(0.5, 2.0, 50000)])
    eps = np.finfo(np.float64).eps
    for fn in (0, 2):
        got = opt.device_math(fn, x)
        ref = np.log(x)
        err = np.abs(got - ref) / (eps * np.maximum(np.abs(ref), 1.0))
        assert err.max() <= 2.0, (fn, err.max(), x[err.argmax()])
        # and where log(x) is tiny the RELATIVE error stays small too
        near1 = np.abs(x - 1.0) < 1e-2
        rel = np.abs(got[near1] - ref[near1]) / np.maximum(np.abs(ref[near1]), 1e-300)
        assert rel[np.abs(ref[near1]) > 0].max() <= 4 * eps
    assert opt.device_math(0, np.array([1.0]))[0] == 0.0
    y = np.concatenate([x[:300000], -x[:300000]])
    y = y[(np.abs(y) > 1e-300) & (np.abs(y) < 1e300)]
    got = opt.device_math(1, y)
    assert (np.abs(got * y - 1.0) <= 4 * eps).all()
    assert np.abs(got - 1.0 / y).max() <= 0 or (np.abs(got - 1.0 / y) / np.abs(1.0 / y)).max() <= 2 * eps
    opt.close()


def test_lean_sin_cos_tan_are_accurate_to_a_few_ulp():
    """dev_model.hpp lean_sincos / lean_tan (dynamics, Jacobian, disc positions) against numpy's
    libm: sin / cos within 2 ulp of max(|value|, tiny) over the angle range the states live in
    ([-pi, pi) after NormalizeAngle) and well beyond; tan within 4 ulp relative."""
    sc = scenario.generate("ped6", 4, seed=3)
    opt = _opt(sc)
    rng = np.random.default_rng(6)
    k = np.arange(-40, 41)
    near = np.concatenate([k * (np.pi / 2) + d for d in (0.0, 1e-9, -1e-9, 1e-5, -1e-5, 1e-13)])
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-0.8, 0.8, 100000),
                        rng.uniform(-100.0, 100.0, 100000), rng.uniform(-1e5, 1e5, 50000),
                        rng.uniform(-1e-6, 1e-6, 10000), near, np.array([0.0, np.pi, -np.pi, np.pi / 4, 1e-300])])
    eps = np.finfo(np.float64).eps
    for fn, ref in ((3, np.sin(x)), (4, np.cos(x))):
        got = opt.device_math(fn, x)
        small = np.abs(x) <= 100.0
        err = np.abs(got - ref) / (eps * np.maximum(np.abs(ref), 1e-300))
        # relative accuracy wherever the argument reduction has bits to spare (|x| <= 100) ...
        assert err[small].max() <= 3.0, (fn, err[small].max(), x[small][err[small].argmax()])
        # ... and absolute accuracy everywhere
        assert (np.abs(got - ref) <= 2 * eps).all(), fn
    got = opt.device_math(5, x)
    ref = np.tan(x)
    small = np.abs(x) <= 100.0
    rel = np.abs(got - ref)[small] / np.maximum(np.abs(ref[small]), 1e-300)
    assert rel.max() <= 5 * eps, (rel.max(), x[small][rel.argmax()])
    assert opt.device_math(3, np.array([0.0]))[0] == 0.0 and opt.device_math(4, np.array([0.0]))[0] == 1.0
    opt.close()


def test_normalize_angle_is_the_reference_expression_bit_for_bit():
    """NormalizeAngle (math_utils.cpp:53-59): a = fmod(angle + pi, 2 pi); if (a < 0) a += 2 pi; return a - pi.
    The device function takes exact short cuts for arguments within a turn of the principal range and the
    library fmod otherwise; fmod is exact, so every result must equal the reference expression evaluated in
    numpy -- including the ends of the range, multiples of pi, huge and tiny arguments."""
    sc = scenario.generate("ped6", 4, seed=3)
    opt = _opt(sc)
    rng = np.random.default_rng(8)
    k = np.arange(-9, 10)
    edges = np.concatenate([k * np.pi + d for d in (0.0, 1e-16, -1e-16, 4e-16, -4e-16, 1e-9, -1e-9)])
    edges = np.concatenate([edges, np.nextafter(k * np.pi, np.inf), np.nextafter(k * np.pi, -np.inf)])
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 100000), rng.uniform(-4 * np.pi, 4 * np.pi, 100000),
                        rng.uniform(-1e3, 1e3, 50000), rng.uniform(-1e9, 1e9, 20000), rng.uniform(-1e-6, 1e-6, 5000),
                        edges, np.array([0.0, -0.0, np.pi, -np.pi, 2 * np.pi, -2 * np.pi, 1e300, -1e300, 5e-324])])
    a = np.fmod(x + np.pi, 2.0 * np.pi)
    ref = np.where(a < 0.0, a + 2.0 * np.pi, a) - np.pi
    got = opt.device_math(6, x)
    assert np.array_equal(got, ref), (x[got != ref][:5], got[got != ref][:5], ref[got != ref][:5])
    assert ((got >= -np.pi) & (got <= np.pi)).all()
    # the form the rollouts use (short paths as a straight line, the complete function for a wavefront in which a lane
    # leaves them): the same bits, whether the rare arguments sit alone in their wavefront, mixed with common ones, or absent
    for xs in (x, rng.permutation(x), x[:200000], np.concatenate([[np.nan, np.inf, -np.inf], x[:61]])):
        a = np.fmod(xs + np.pi, 2.0 * np.pi)
        with np.errstate(invalid="ignore"):
            ref = np.where(a < 0.0, a + 2.0 * np.pi, a) - np.pi
        got = opt.device_math(10, xs)
        assert np.array_equal(got, ref, equal_nan=True), (xs[got != ref][:5], got[got != ref][:5], ref[got != ref][:5])
    opt.close()


def test_team_backward_is_bit_identical_to_one_lane_per_problem(lockstep_only):
    """CILQR_OPT_TEAM_THRESHOLD / CILQR_OPT_WAVE_THRESHOLD: small backward launches spread a problem over eight
    lanes (k_backward_team) or over a whole wavefront (k_backward_wave).  Stage outputs (gains, delta_V, gradient
    norm) and whole solves must not change by a bit, for batch sizes that leave teams idle, fill them exactly, or
    cross blocks."""
    for B in (1, 7, 8, 9, 70):
        sc = scenario.generate("mix11", B, seed=140 + B)
        opt = _opt(sc)
        outs = []
        for team, wave in ((0, 0), (4096, 0), (0, 4096)):
            opt.set_option(api.OPT_TEAM_THRESHOLD, team)
            opt.set_option(api.OPT_WAVE_THRESHOLD, wave)
            opt.stage_load(sc)
            opt.stage_init_guess()
            opt.stage_quadratize()
            opt.stage_backward(np.linspace(0.3, 2.0, B))
            outs.append([opt.read(t) for t in (api.T_KFB, api.T_KFF, api.T_DV, api.T_GNORM)])
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert np.array_equal(a, b)
        opt.close()
    for family, B, seed in (("mix11", 300, 150), ("dyn20x", 40, 151)):
        sc = scenario.generate(family, B, seed=seed)
        opt = _opt(sc)
        opt.set_option(api.OPT_WAVE_THRESHOLD, 0)
        opt.set_option(api.OPT_TEAM_THRESHOLD, 4096)
        a = opt.plan(sc, max_iter_trajs=2)
        outs = []
        for team, wave in ((0, 0), (64, 0), (4096, 16), (0, 4096)):   # incl. switching kernels in the middle of a solve
            opt.set_option(api.OPT_TEAM_THRESHOLD, team)
            opt.set_option(api.OPT_WAVE_THRESHOLD, wave)
            outs.append(opt.plan(sc, max_iter_trajs=2))
        for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "iter_trajs"):
            for o in outs:
                assert np.array_equal(a[k], o[k]), (family, k)
        opt.close()


@pytest.mark.parametrize("N", [1, 2, 3, 5, 7])
def test_short_horizons_empty_knots_and_single_segment_lanes(N, both_paths):
    """Horizons shorter than the rollout's prefetch depth (4 steps) and than a backward team's
    pipeline, knots without any corridor plane, lane tables of one segment, batches of 1 / 3 / 65."""
    import dataclasses
    for B in (1, 3, 65):
        spec = dataclasses.replace(scenario.SPECS["mix11"], n_steps=N)
        sc = scenario.generate(spec, B, seed=300 + N)
        sc["ccount"][:, ::2] = 0
        if N == 5:
            sc["left"] = np.ascontiguousarray(sc["left"][3:4])
            sc["right"] = np.ascontiguousarray(sc["right"][3:4])
        opt = _opt(sc)
        g = _plan(opt, sc)
        ref = oracle_reference(sc, oracle_cfg_from(opt.cfg))
        assert_parity(g, ref, max_unstable_frac=0.0, what=f"N={N} B={B}")   # every one of them is stable
        assert_steps(g, sc, oracle_cfg_from(opt.cfg), what=f"N={N} B={B}")
        opt.close()


def test_long_horizon_more_knots_than_a_tail_workgroup_has_threads(both_paths):
    """N = 280 (K = 281 knots): more knots than the 256 threads of a tail workgroup, 5.6 times the bench horizon for
    the wave / team backward kernels and the rollouts.  Whole solves and every step against the oracle, both loops."""
    import dataclasses
    spec = dataclasses.replace(scenario.SPECS["mix11"], n_steps=280)
    sc = scenario.generate(spec, 12, seed=333)
    opt = _opt(sc)
    g = _plan(opt, sc)
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sc, ocfg)
    rep = assert_parity(g, ref, max_unstable_frac=0.34, what="N=280")
    steps = assert_steps(g, sc, ocfg, what="N=280", max_excused_frac=0.05)
    print(f"\nN=280: whole solves {rep}; steps {steps}")
    opt.close()


def test_horizon_beyond_the_lds_budgets_of_the_wave_kernels():
    """N = 1100: the per-step rows of the wave backward pass no longer fit the 64 KiB a launch gets without asking
    (launch_backward_wave declines, the eight-lane kernel takes over) and the tail kernel's fixed LDS block no longer fits
    either (tail_supported: the solve stays in the lockstep loop).  Both guards must leave a working solve behind: every
    step replays in the oracle, and the options that would select the declined kernels change nothing."""
    import dataclasses
    spec = dataclasses.replace(scenario.SPECS["ped6"], n_steps=1100)
    sc = scenario.generate(spec, 3, seed=71)
    opt = _opt(sc, max_iter=25)
    g = _plan(opt, sc)
    assert set(np.unique(g["status"])) <= {api.ST_CONVERGED_ABS, api.ST_CONVERGED_REL, api.ST_GNORM, api.ST_MAX_ITER, api.ST_UNSOLVED}
    steps = assert_steps(g, sc, oracle_cfg_from(opt.cfg), what="N=1100", max_excused_frac=0.1)
    print(f"\nN=1100: steps {steps}")
    opt.set_option(api.OPT_TAIL_THRESHOLD, 0)
    opt.set_option(api.OPT_WAVE_THRESHOLD, 0)
    h = _plan(opt, sc)
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "iter_trajs", "n_iter_trajs", "alpha_trace"):
        assert np.array_equal(g[k], h[k], equal_nan=True), k
    opt.close()


@pytest.mark.parametrize("family,distinct", [("mix11", 256), ("dyn20x", 128)])
def test_full_size_batch_properties(family, distinct):
    """BASELINE configs[2] (B = 65536, N = 50, mix11) and configs[4] (B = 65536, N = 100, 20 dynamic obstacles, barriers
    active at the init guess: dyn20x) at full size: `distinct` scenes tiled to 65536.  Size-independent properties: every
    copy of a scene gives bit-identical output wherever it sits in the batch, all problems terminate, accepted costs
    decrease monotonically, and the distinct scenes match the oracle."""
    base = scenario.generate(family, distinct, seed=91)
    rep = 65536 // distinct
    sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if isinstance(v, np.ndarray) and v.shape[:1] == (distinct,) else v)
          for k, v in base.items()}
    B = distinct * rep
    opt = _opt(sc)
    g = opt.plan(sc, alpha_trace=True)
    assert ((g["status"] >= 1) & (g["status"] <= 5)).all()
    tr = g["traj"].reshape(rep, distinct, *g["traj"].shape[1:])
    assert np.array_equal(tr, np.broadcast_to(tr[:1], tr.shape))
    nc = g["n_cost"].reshape(rep, distinct)
    assert np.array_equal(nc, np.broadcast_to(nc[:1], nc.shape))
    tot = g["cost_hist"][:distinct, :, 0]
    for b in range(distinct):
        assert np.all(np.diff(tot[b, :g["n_cost"][b]]) < 0)
    first = {k: v[:distinct] for k, v in g.items() if isinstance(v, np.ndarray)}
    ref = oracle_reference(base, oracle_cfg_from(opt.cfg))
    assert_parity(first, ref, what=f"full-size batch {family}", max_unstable_frac=0.125 if family == "mix11" else 0.2)
    assert B == 65536


@pytest.mark.parametrize("family,sample,max_unstable", [("mix11", 640, 0.10), ("dyn20x", 512, 0.16)])
def test_full_size_batch_of_distinct_scenes_against_the_oracle(family, sample, max_unstable):
    """BASELINE configs[2] (mix11, N = 50) and configs[4] (dyn20x, N = 100) with 65536 DIFFERENT scenes (in
    the tiled test every copy of a scene leaves the active list in the same iteration, so survivor re-packing, the finishing
    arena and the four-row candidate layout never see the ragged thinning of 65536 different problems).
      * one cilqr_solve_batch and one solve submitted through a pool, host arrays in and out (the upload-ahead / ragged-download
        path): bit-identical to each other, every array;
      * a seeded sample of `sample` problems solved again as a batch of their own (another arena layout: eleven candidate rows,
        no hand-over at 8192): bit-identical to their rows of the full batch -- so what holds for the sample's solve holds for
        the full batch's;
      * that sample against the oracle: whole solves on its stable problems (status, iteration count, every accepted step
        size, cost rows, trajectory at 1e-4), every step of every sampled problem replayed at 1e-8."""
    B = 65536
    sc = scenario.generate(family, B, seed=606, workers=min(16, os.cpu_count() or 4))
    cfg = api.default_config(sc["n_steps"])
    pool = api.HandlePool(cfg, device=0, handles=1, batch_capacity=B, cmax=sc["cmax"], max_lane_segments=64)
    opt = pool.handle_at(0, batch_capacity=B, cmax=sc["cmax"])
    full = opt.plan(sc, alpha_trace=True)                      # cilqr_solve_batch
    assert ((full["status"] >= 1) & (full["status"] <= 5)).all()
    # how ragged the batch is: the iteration counts must spread (this is what the tiled test cannot offer)
    assert len(np.unique(full["n_iter"])) >= 12
    prob, keep = opt._host_problem(sc)
    K, M = sc["n_steps"] + 1, cfg.max_iter
    out = dict(traj=np.full((B, K, 10), np.nan), cost_hist=np.full((B, M + 1, 5), np.nan), n_cost=np.full(B, -1, np.int32),
               status=np.full(B, -1, np.int32), n_iter=np.full(B, -1, np.int32), alpha_trace=np.full((B, M), 9, np.int8))
    sol = api.SolutionBatch(api.MEM_HOST, 0, out["traj"].ctypes.data, out["cost_hist"].ctypes.data, out["n_cost"].ctypes.data,
                            out["status"].ctypes.data, out["n_iter"].ctypes.data, None, None, out["alpha_trace"].ctypes.data)
    assert pool.submit_raw(prob, sol) == api.OK and pool.wait() == api.OK        # cilqr_pool_submit
    del keep
    for k in out:
        assert np.array_equal(out[k], full[k]), k
    idx = np.sort(np.random.default_rng(17).choice(B, sample, replace=False))
    sub = {k: (np.ascontiguousarray(v[idx]) if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in sc.items()}
    g = opt.plan(sub, max_iter_trajs=48, alpha_trace=True)
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "alpha_trace"):
        assert np.array_equal(g[k], full[k][idx]), k
    ocfg = oracle_cfg_from(opt.cfg)
    ref = oracle_reference(sub, ocfg)
    rep = assert_parity(g, ref, max_unstable_frac=max_unstable, what=f"65536 distinct {family} scenes, sample of {sample}")
    steps = assert_steps(g, sub, ocfg, what=f"65536 distinct {family} scenes, sample of {sample}")
    stable = ref["stable"]
    assert np.array_equal(np.bincount(g["status"][stable], minlength=7), np.bincount(ref["status"][stable], minlength=7))
    assert np.array_equal(g["n_iter"][stable], ref["n_iter"][stable])
    print(f"\n{family}: whole solves {rep}; steps {steps}; iteration counts of the full batch: "
          f"{np.bincount(full['n_iter']).nonzero()[0][[0, -1]].tolist()}, mean {full['n_iter'].mean():.2f}")
    pool.close()
    opt.close()


def _dist_matrix(tab, pts):
    """LineSegment2d::DistanceTo (line_segment2d.cpp:61-75) of every point to every segment."""
    s, e = tab[:, 3:5], tab[:, 5:7]
    d = e - s
    ln = np.hypot(d[:, 0], d[:, 1])
    u = d / ln[:, None]
    x0 = pts[:, None, 0] - s[None, :, 0]
    y0 = pts[:, None, 1] - s[None, :, 1]
    proj = x0 * u[None, :, 0] + y0 * u[None, :, 1]
    d_start = np.hypot(x0, y0)
    d_end = np.hypot(pts[:, None, 0] - e[None, :, 0], pts[:, None, 1] - e[None, :, 1])
    d_perp = np.abs(x0 * u[None, :, 1] - y0 * u[None, :, 0])
    return np.where(proj <= 0.0, d_start, np.where(proj >= ln[None, :], d_end, d_perp))


def _nearest_numpy(tab, pts):
    dist = _dist_matrix(tab, pts)
    return np.argmin(dist, axis=1), np.sort(dist, axis=1)


def test_nearest_lane_grid_equals_linear_scan():
    """The accelerated lookup must return the reference's linear-scan answer everywhere: on the road,
    far outside the grid, and in the wedge regions where two segments tie exactly."""
    sc = scenario.generate("ped6", 4, seed=3)
    opt = _opt(sc)
    opt.stage_load(sc)
    rng = np.random.default_rng(7)
    road = scenario.build_road()
    s = rng.uniform(0, road.length, 200000)
    lat = rng.uniform(-14.0, 10.0, s.size)
    x, y = road.cartesian(s, lat)
    pts = [np.stack([x, y], axis=1),
           rng.uniform(-400, 400, (20000, 2)),                       # mostly outside the grid
           np.concatenate([sc["left"][:, 3:5], sc["right"][:, 5:7]])]  # exact segment end points (ties)
    # points pushed outwards from every joint: both neighbours are "past the end" -> exact tie
    for tab in (sc["left"], sc["right"]):
        d = tab[:, 5:7] - tab[:, 3:5]
        nrm = np.stack([tab[:, 0], tab[:, 1]], axis=1) / np.hypot(tab[:, 0], tab[:, 1])[:, None]
        for r in (0.3, 2.0, 6.0):
            pts.append(tab[:, 5:7] + r * nrm + 1e-3 * d)
            pts.append(tab[:, 3:5] + r * nrm - 1e-3 * d)
    n_random = sum(len(q) for q in pts[:2])
    pts = np.concatenate(pts)
    gl, gr = opt.nearest_lane(pts, use_grid=True)
    sl, sr = opt.nearest_lane(pts, use_grid=False)
    assert np.array_equal(gl, sl) and np.array_equal(gr, sr)
    # and the linear scan itself follows the reference's DistanceTo semantics (hypot-based) except
    # where the two best distances agree to rounding
    for tab, got in ((sc["left"], sl), (sc["right"], sr)):
        ref, srt = _nearest_numpy(tab, pts)
        clear = (srt[:, 1] - srt[:, 0]) > 1e-9 * (1.0 + srt[:, 0])
        assert np.array_equal(got[clear], ref[clear])
        assert clear.mean() > 0.8
        # wherever the answers differ the two candidates are equidistant to rounding
        d_got = np.take_along_axis(_dist_matrix(tab, pts), got[:, None].astype(np.int64), axis=1)[:, 0]
        assert np.all(d_got - srt[:, 0] <= 1e-9 * (1.0 + srt[:, 0]))
    opt.close()


def test_a_failing_solve_between_two_good_ones_on_one_handle():
    """Three submitted solves with host arrays on one handle (two in flight, one queued, the transfer thread uploading ahead):
    the middle one is refused by the argument checks (`n_knots` wrong: CILQR_ERR_KNOTS, cc:75-78) -- its wait returns that code,
    the solves either side of it return their results bit for bit, nothing deadlocks, an input buffer the bad solve's upload
    never got is not leaked (a second round of three goes through), and destroying the handle with solves in flight waits."""
    sc = scenario.generate("mix11", 3000, seed=171)
    opt = _opt(sc)
    ref = opt.plan(sc)
    B, K, M = 3000, sc["n_steps"] + 1, opt.cfg.max_iter
    prob, keep = opt._host_problem(sc)
    bad = api.ProblemBatch.from_buffer_copy(prob)
    bad.n_knots = K - 1
    null_corridor = api.ProblemBatch.from_buffer_copy(prob)
    null_corridor.corridor = None

    def outs():
        o = dict(traj=np.full((B, K, 10), np.nan), hist=np.full((B, M + 1, 5), np.nan), nc=np.full(B, -1, np.int32),
                 st=np.full(B, -1, np.int32), ni=np.full(B, -1, np.int32))
        o["sol"] = api.SolutionBatch(api.MEM_HOST, 0, o["traj"].ctypes.data, o["hist"].ctypes.data, o["nc"].ctypes.data,
                                     o["st"].ctypes.data, o["ni"].ctypes.data, None, None, None)
        return o

    for wrong, code in ((bad, api.ERR_KNOTS), (null_corridor, api.ERR_CONSTRAINTS)):
        a, b, c = outs(), outs(), outs()
        assert opt.submit_raw(prob, a["sol"]) == api.OK
        assert opt.submit_raw(wrong, b["sol"]) == api.OK          # accepted: the checks run when its turn comes
        assert opt.submit_raw(prob, c["sol"]) == api.OK
        assert opt.wait() == api.OK
        assert opt.wait() == code
        assert opt.wait() == api.OK
        for o in (a, c):
            assert np.array_equal(o["traj"], ref["traj"]) and np.array_equal(o["hist"], ref["cost_hist"])
            assert np.array_equal(o["nc"], ref["n_cost"]) and np.array_equal(o["st"], ref["status"])
    again = opt.plan(sc)                                           # the synchronous call still works on the handle
    assert np.array_equal(again["traj"], ref["traj"])
    a, b, c = outs(), outs(), outs()
    for o in (a, b, c):
        assert opt.submit_raw(prob, o["sol"]) == api.OK
    opt.close()                                                    # cilqr_destroy with three solves outstanding: waits for them
    del keep


def test_handle_pool_deals_batches_round_robin_bit_identically():
    """cilqr_pool_*: two handles on one GPU, seven different batches submitted as a stream (up to depth = 6 submitted:
    two in flight and one queued per handle, the oldest collected first).  Every batch comes back bit-identical to the synchronous call; a submit beyond
    the depth and a wait on an empty pool are refused; destroy collects what is still in flight."""
    torch = pytest.importorskip("torch")
    sizes = [9000, 700, 8500, 100, 9000, 3000, 5000]
    scenes = [scenario.generate("mix11", n, seed=300 + i) for i, n in enumerate(sizes)]
    cfg = api.default_config(scenes[0]["n_steps"])
    one = api.BatchIlqrOptimizer(cfg, batch_capacity=max(sizes), cmax=scenes[0]["cmax"], max_lane_segments=64)
    sync = [one.plan(sc) for sc in scenes]
    pool = api.HandlePool(cfg, device=0, handles=2, batch_capacity=max(sizes), cmax=scenes[0]["cmax"], max_lane_segments=64)
    assert pool.depth() == 6 and pool.device_bytes() > 1.5 * api.BatchIlqrOptimizer.device_bytes(pool.handle_at(0))
    K, M = cfg.n_steps + 1, cfg.max_iter
    jobs = []
    for sc in scenes:
        B = sc["coarse"].shape[0]
        keep = {k: np.ascontiguousarray(sc[k]) for k in ("start", "coarse", "corridor", "ccount", "left", "right")}
        prob = one.make_problem(B, keep["start"].ctypes.data, keep["coarse"].ctypes.data, keep["corridor"].ctypes.data,
                                keep["ccount"].ctypes.data, sc["cmax"], keep["left"].ctypes.data, keep["right"].ctypes.data,
                                keep["left"].shape[0], keep["right"].shape[0], api.MEM_HOST)
        # (the caller's arrays arrive dirty: rows >= n_cost of cost_hist come back as zeros on every host path, include/cilqr.h)
        bufs = (np.full((B, K, 10), np.nan), np.full((B, M + 1, 5), np.nan), np.full(B, -9, np.int32), np.full(B, -9, np.int32),
                np.full(B, -9, np.int32))
        sol = api.SolutionBatch(api.MEM_HOST, 0, bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data,
                                bufs[3].ctypes.data, bufs[4].ctypes.data, None, None)
        jobs.append((keep, bufs, prob, sol))
    assert pool.wait() == api.ERR_STATE                                   # nothing submitted
    for j in jobs[:6]:
        assert pool.submit_raw(j[2], j[3]) == api.OK
    assert pool.submit_raw(jobs[6][2], jobs[6][3]) == api.ERR_STATE       # depth reached: collect first
    assert pool.wait() == api.OK                                          # batch 0
    assert pool.submit_raw(jobs[6][2], jobs[6][3]) == api.OK
    assert pool.profile().iterations > 0
    for _ in range(6):
        assert pool.wait() == api.OK
    assert pool.wait() == api.ERR_STATE
    for i, (j, ref) in enumerate(zip(jobs, sync)):
        for got, key in zip(j[1], ("traj", "cost_hist", "n_cost", "status", "n_iter")):
            assert np.array_equal(got, ref[key]), (i, key)
    # a pool destroyed with solves in flight waits for them (their arrays are still alive here)
    for j in jobs[:4]:
        j[1][0][...] = 0
        assert pool.submit_raw(j[2], j[3]) == api.OK
    pool.close()
    for j, ref in zip(jobs[:4], sync):
        assert np.array_equal(j[1][0], ref["traj"])
    one.close()


def test_exact_lane_ties_follow_the_reference_rule():
    """CILQR_OPT_EXACT_LANE_TIES: FindNeastLaneSegment (ilqr_optimizer.cc:605-618) compares DistanceTo values (hypot /
    |cross|, line_segment2d.cpp:61-75) with a strict '<', first index wins.  With the option on, the device search --
    grid and full scan -- must return numpy's first minimum of those distances on EVERY point, the tie strips included:
    joints, points on the normal through a joint (perpendicular distance to one segment = end-point distance to its
    neighbour, the strip the iteration parks on), the same points moved by a few ulp."""
    sc = scenario.generate("ped6", 4, seed=3)
    opt = _opt(sc)
    opt.set_option(api.OPT_EXACT_LANE_TIES, 1)
    opt.stage_load(sc)
    rng = np.random.default_rng(11)
    road = scenario.build_road()
    s_ = rng.uniform(0, road.length, 100000)
    x, y = road.cartesian(s_, rng.uniform(-14.0, 10.0, s_.size))
    pts = [np.stack([x, y], axis=1), np.concatenate([sc["left"][:, 3:5], sc["right"][:, 5:7]])]
    for tab in (sc["left"], sc["right"]):
        d = tab[:, 5:7] - tab[:, 3:5]
        u = d / np.hypot(d[:, 0], d[:, 1])[:, None]
        nrm = np.stack([-u[:, 1], u[:, 0]], axis=1)
        for r in (-6.0, -2.5, -0.7, 0.3, 2.0, 6.0):
            for end in (tab[:, 3:5], tab[:, 5:7]):
                base = end + r * nrm                       # on the normal through the joint
                pts.append(base)
                for k in (-3, -1, 1, 3):                   # and a few ulp along the segment on either side
                    pts.append(base + k * np.spacing(np.abs(base)) * np.sign(u))
                pts.append(base + 1e-9 * u)
                pts.append(base - 1e-9 * u)
    pts = np.concatenate(pts)
    gl, gr = opt.nearest_lane(pts, use_grid=True)
    sl, sr = opt.nearest_lane(pts, use_grid=False)
    n_tie = 0
    for tab, grid, scan in ((sc["left"], gl, sl), (sc["right"], gr, sr)):
        ref, srt = _nearest_numpy(tab, pts)
        n_tie += int((srt[:, 1] == srt[:, 0]).sum())
        bad = np.nonzero((grid != ref) | (scan != ref))[0]
        assert bad.size == 0, (bad[:5], pts[bad[:5]], grid[bad[:5]], scan[bad[:5]], ref[bad[:5]])
    assert n_tie > 100          # the set does contain exact ties of the reference's distances
    # the default search differs on some of them (that is what the option is for)
    opt.set_option(api.OPT_EXACT_LANE_TIES, 0)
    dl, dr = opt.nearest_lane(pts, use_grid=True)
    print(f"\nexact ties in the point set: {n_tie}; default search differs from the reference on "
          f"{int((dl != _nearest_numpy(sc['left'], pts)[0]).sum() + (dr != _nearest_numpy(sc['right'], pts)[0]).sum())} points")
    opt.close()


def test_exact_lane_ties_need_no_excuse():
    """The library's DEFAULT is the reference's tie rule (CILQR_OPT_EXACT_LANE_TIES = 1, nothing set here): the step
    replay runs with the `lane_tie` excuse switched OFF -- the zero-tolerance configuration that iterates into the noise
    plateau (where iterates come to rest on tie strips), and a default run."""
    for over, n, frac in ((dict(rel_cost_tol=0.0, abs_cost_tol=0.0, max_iter=60), 48, 0.25), (dict(), 160, 0.02)):
        sc = scenario.generate("mix11", n, seed=77)
        cfg = api.default_config(sc["n_steps"], **over)
        opt = api.BatchIlqrOptimizer(cfg, batch_capacity=n, cmax=sc["cmax"])
        g = _plan(opt, sc)
        rep = assert_steps(g, sc, oracle_cfg_from(cfg), what=f"exact ties {over}", max_excused_frac=frac, allow_lane_tie=False)
        assert rep["lane_tie"] == 0
        # setting the option to 1 explicitly is the same solve, bit for bit
        opt.set_option(api.OPT_EXACT_LANE_TIES, 1)
        g1 = _plan(opt, sc)
        for key in ("traj", "cost_hist", "status", "n_iter", "alpha_trace"):
            assert np.array_equal(g[key], g1[key]), key
        print(f"\nexact ties (default), {over}: steps {rep}")
        opt.close()


def test_fast_lane_tie_rule_is_the_opt_in():
    """CILQR_OPT_EXACT_LANE_TIES = 0: nearest segments by squared distances alone.  Same results as the default wherever
    no iterate meets a tie strip; the step replay holds with the `lane_tie` excuse available (and only then may use it)."""
    sc = scenario.generate("mix11", 160, seed=77)
    cfg = api.default_config(sc["n_steps"])
    opt = api.BatchIlqrOptimizer(cfg, batch_capacity=160, cmax=sc["cmax"])
    g_ref = _plan(opt, sc)
    opt.set_option(api.OPT_EXACT_LANE_TIES, 0)
    g = _plan(opt, sc)
    rep = assert_steps(g, sc, oracle_cfg_from(cfg), what="fast lane ties", allow_lane_tie=True)
    same = int(np.sum([np.array_equal(g["traj"][b], g_ref["traj"][b]) for b in range(160)]))
    print(f"\nfast tie rule: steps {rep}; {same} of 160 solves bit-identical to the default rule")
    assert same >= 150
    opt.close()


def test_iter_trajs_beyond_the_count_are_unspecified_on_every_path():
    """include/cilqr.h: entries of iter_trajs at or beyond n_iter_trajs[b] are UNSPECIFIED -- small host batches copy out
    only the iterates that exist (whatever the caller's buffer held stays behind them), large host batches and device
    outputs deliver the whole block.  Pinned here for both host paths: the entries below the count are identical whichever
    path produced them, and a caller that pre-fills its buffer finds the fill value (small path) or anything (large path)
    behind them -- never something to rely on.  INTEGRATION.md section 3 says the same."""
    sc = scenario.generate("mix11", 300, seed=31)
    K, cap = sc["n_steps"] + 1, 12
    opt = _opt(sc)

    def solve(n, fill):
        sub = {k: (v[:n] if isinstance(v, np.ndarray) and v.shape[:1] == (300,) else v) for k, v in sc.items()}
        prob, keep = opt._host_problem(sub)
        M = opt.cfg.max_iter
        traj, hist = np.zeros((n, K, 10)), np.zeros((n, M + 1, 5))
        nc, st, ni, nit = (np.zeros(n, np.int32) for _ in range(4))
        it = np.full((n, cap, K, 10), fill)
        sol = api.SolutionBatch(api.MEM_HOST, cap, traj.ctypes.data, hist.ctypes.data, nc.ctypes.data, st.ctypes.data,
                                ni.ctypes.data, it.ctypes.data, nit.ctypes.data, None)
        assert opt.solve_raw(prob, sol) == api.OK
        del keep
        return it, nit, traj

    small_it, small_n, small_traj = solve(3, -7.5)          # a few problems from host arrays: the packed small-transfer path
    big_it, big_n, big_traj = solve(300, -7.5)              # the staged path
    assert np.array_equal(small_n, big_n[:3]) and np.array_equal(small_traj, big_traj[:3])
    for b in range(3):
        n = min(int(small_n[b]), cap)
        assert n >= 1 and np.array_equal(small_it[b, :n], big_it[b, :n])          # what exists is the same on both paths
        assert np.all(small_it[b, n:] == -7.5)                                     # small path: the caller's bytes stay
    opt.close()


def test_small_host_batches_fetch_their_iterates_in_two_steps():
    """A small host batch copies out the head of the staging block first (trajectory, cost rows, counts -- and, for a batch of
    one, its first 16 iterates: solver.hip, job_finish) and the iterates that exist beyond that once their counts are known.
    With tolerances of zero a solve keeps iterating, so a batch of one has more than 16 iterates and a batch of three fetches
    every iterate in the second step: what arrives must be what the staged path of a large batch delivers for the same
    problems -- trajectories, every live cost row (zeros behind them), every iterate below the count, the caller's bytes behind."""
    sc = scenario.generate("mix11", 300, seed=33)
    K, cap = sc["n_steps"] + 1, 64
    opt = _opt(sc, rel_cost_tol=0.0, abs_cost_tol=0.0, max_iter=60)
    M = opt.cfg.max_iter

    def solve(n, first=0):
        sub = {k: (v[first:first + n] if isinstance(v, np.ndarray) and v.shape[:1] == (300,) else v) for k, v in sc.items()}
        prob, keep = opt._host_problem(sub)
        traj, hist = np.zeros((n, K, 10)), np.full((n, M + 1, 5), 3.25)
        nc, st, ni, nit = (np.zeros(n, np.int32) for _ in range(4))
        it = np.full((n, cap, K, 10), -7.5)
        sol = api.SolutionBatch(api.MEM_HOST, cap, traj.ctypes.data, hist.ctypes.data, nc.ctypes.data, st.ctypes.data,
                                ni.ctypes.data, it.ctypes.data, nit.ctypes.data, None)
        assert opt.solve_raw(prob, sol) == api.OK
        del keep
        return dict(traj=traj, hist=hist, nc=nc, st=st, ni=ni, it=it, nit=nit)

    big = solve(300)
    assert (big["nit"] > 16).sum() >= 100            # the case the second copy exists for is the common one here
    b_long = int(np.argmax(big["nit"]))
    for first, n in ((b_long, 1), (0, 1), (0, 3), (5, 2)):
        small = solve(n, first)
        sl = slice(first, first + n)
        for k in ("traj", "nc", "st", "ni", "nit"):
            assert np.array_equal(small[k], big[k][sl]), (k, first, n)
        for b in range(n):
            rows = int(small["nc"][b])
            assert np.array_equal(small["hist"][b, :rows], big["hist"][first + b, :rows])
            assert np.all(small["hist"][b, rows:] == 0.0) and np.all(big["hist"][first + b, rows:] == 0.0)
            cnt = min(int(small["nit"][b]), cap)
            assert np.array_equal(small["it"][b, :cnt], big["it"][first + b, :cnt]), (first, n, b, cnt)
            assert np.all(small["it"][b, cnt:] == -7.5)
    opt.close()


def test_delta_v_evaluation_switch():
    """whether cc:383-384 see the updated Vx / Vxx (lazy `auto`, the default in product and oracle) or
    the ones the gains came from (eager) cannot be run against Eigen here, so BOTH readings stay built and checked: the
    test-only library libcilqr_hip_dveager.so (-DCILQR_DV_EVAL_EAGER, cilqr_amd/csrc/backward_core.hpp) against the oracle's
    eager variant, in a child process (a process holds one libcilqr_hip): tests/dv_eval_check.py."""
    import json
    import subprocess
    import sys
    lib = os.path.join(ROOT, "cilqr_amd", "lib", "libcilqr_hip_dveager.so")
    assert os.path.exists(lib), f"{lib} is missing (make -C cilqr_amd/csrc dveager; __graft_entry__.build() does it)"
    r = subprocess.run([sys.executable, os.path.join(HERE, "dv_eval_check.py")], env=dict(os.environ, CILQR_LIB=lib),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["ok"] and rep["steps"]["steps"] > 1000
    print(f"\neager delta_V build: {rep}")


def test_dot_order_switch():
    """the product sums every `X.transpose() * Y` coefficient of Backward / iqr as the reference's default
    x86-64 -O2 (SSE2) build of Eigen does, (t0 + (t2 + t4)) + (t1 + (t3 + t5)) (cilqr_amd/csrc/dev_model.hpp: sum6_xty) -- the
    oracle's default too, so every other parity test of this file holds that reading.  The other reading (index order, what
    rounds 1-4 shipped) stays built and checked like the eager delta_V: libcilqr_hip_dotseq.so against the oracle's
    sequential variant, in a child process: tests/dot_order_check.py."""
    import json
    import subprocess
    import sys
    lib = os.path.join(ROOT, "cilqr_amd", "lib", "libcilqr_hip_dotseq.so")
    assert os.path.exists(lib), f"{lib} is missing (make -C cilqr_amd/csrc dotseq; __graft_entry__.build() does it)"
    r = subprocess.run([sys.executable, os.path.join(HERE, "dot_order_check.py")], env=dict(os.environ, CILQR_LIB=lib),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["ok"] and rep["steps"]["steps"] > 1000
    print(f"\nsequential dot-order build: {rep}")


def test_backward_gains_follow_the_references_dot_order():
    """The same statement for the PRODUCT library: fed with identical stage inputs, its gains equal the default (eigen_sse2)
    oracle's more closely than the sequential oracle's on most problems, in all three mappings (bit-identical to each other)."""
    from oracle import oracle as orc
    B = 96
    sc = scenario.generate("mix11", B, seed=53)
    opt = _opt(sc)
    ocfg = oracle_cfg_from(opt.cfg)
    opt.stage_load(sc)
    opt.stage_init_guess()
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
    Kfb = forms["wave"][0]
    closer, exact, exact_seq = 0, 0, 0
    try:
        for b in range(B):
            o = orc.Oracle(ocfg)
            assert o.set_problem(sc["start"][b], sc["coarse"][b], sc["corridor"][b], sc["ccount"][b], sc["left"], sc["right"]) == 0
            qb = {k: q[k][b] for k in q}
            orc.set_semantics(-1, orc.DOT_ORDER_SEQUENTIAL)
            oK0, _, _ = o.backward(float(lam[b]), qb)
            orc.set_semantics(-1, orc.DOT_ORDER_EIGEN_SSE2)
            oK2, _, _ = o.backward(float(lam[b]), qb)
            closer += int(np.abs(Kfb[b] - oK2).max() < np.abs(Kfb[b] - oK0).max())
            exact += int(np.array_equal(Kfb[b], oK2))
            exact_seq += int(np.array_equal(Kfb[b], oK0))
    finally:
        orc.reset_semantics()
    print(f"\ngains of {B} problems: closer to the eigen_sse2 oracle {closer}, bit-equal to it {exact}, bit-equal to the sequential oracle {exact_seq}")
    assert closer >= B // 2 and exact >= exact_seq, (closer, exact, exact_seq)
    opt.close()


def test_four_row_candidate_arena_is_bit_identical(lockstep_only, tmp_path):
    """Arenas of >= 32768 slots hold FOUR candidates per slot instead of eleven (solver.hip: spec_rows): the
    pre-rolled rounds use them, and what is left dead afterwards is re-strided for the remaining step sizes of the problems
    that rejected every round -- in several passes when there are more of those than fit.  A child process forces that
    layout onto a small batch (CILQR_SPEC_ROWS=4) with passes of 8 entries and holds three schedules against this process's
    solve with the plain eleven-row arena, bit for bit (tests/spec_rows_check.py); dyn20x scenes with zero tolerances: many
    iterations, many rejected rounds."""
    import json
    import subprocess
    import sys
    sc = scenario.generate("dyn20x", 200, seed=96)
    over = dict(max_iter=30, abs_cost_tol=0.0, rel_cost_tol=0.0)
    opt = _opt(sc, **over)
    g = opt.plan(sc, max_iter_trajs=3, alpha_trace=True)
    opt.close()
    beyond = int((g["alpha_trace"] >= 4).sum())
    assert beyond >= 64, beyond          # enough problem-iterations accept a step size of the remainder passes to fill several
    path = str(tmp_path / "ref.npz")
    np.savez(path, **{k: sc[k] for k in ("start", "coarse", "corridor", "ccount", "left", "right")}, n_steps=sc["n_steps"],
             cmax=sc["cmax"], cfg_over=json.dumps(over), **{"ref_" + k: g[k] for k in ("traj", "cost_hist", "status", "n_cost", "n_iter", "alpha_trace")})
    r = subprocess.run([sys.executable, os.path.join(HERE, "spec_rows_check.py"), path],
                       env=dict(os.environ, CILQR_SPEC_ROWS="4", CILQR_SPEC_PASS_ENTRIES="8"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["ok"] and rep["accepted_beyond_four_rounds"]["rounds"] == beyond
    print(f"\nfour-row arena: {rep}")


def test_speculative_line_search_is_bit_identical_to_round_by_round(lockstep_only):
    """Small active sets evaluate all 11 step sizes at once; the first passing index must win exactly
    as in the sequential loop (ilqr_optimizer.cc:246-265), so both modes give the same bits."""
    sc = scenario.generate("mix11", 300, seed=95)
    opt = _opt(sc)
    a = opt.plan(sc, max_iter_trajs=3)
    opt.set_option(api.OPT_SPEC_THRESHOLD, 0)
    opt.set_option(api.OPT_SEQ_ROUNDS, 11)
    b = opt.plan(sc, max_iter_trajs=3)
    opt.set_option(api.OPT_SPEC_THRESHOLD, 64)       # switch modes in the middle of the solve
    c = opt.plan(sc, max_iter_trajs=3)
    # re-packing the survivors into dense slots (default on) must not change a bit either
    opt.set_option(api.OPT_COMPACTION, 0)
    d = opt.plan(sc, max_iter_trajs=3)
    opt.set_option(api.OPT_SPEC_THRESHOLD, 8192)
    e = opt.plan(sc, max_iter_trajs=3)
    # hybrid schedules: R rounds one by one, the remaining step sizes at once (R = 11: all sequential)
    outs = [b, c, d, e]
    opt.set_option(api.OPT_SPEC_THRESHOLD, 0)
    for rounds in (1, 2, 3, 5, 11):
        opt.set_option(api.OPT_SEQ_ROUNDS, rounds)
        outs.append(opt.plan(sc, max_iter_trajs=3))
    # ... with one, two or four step sizes costed per round (CILQR_OPT_ROUND_GROUP; the default is two)
    for rounds, group in ((4, 1), (4, 2), (4, 4), (6, 4), (5, 2), (3, 4)):
        opt.set_option(api.OPT_SEQ_ROUNDS, rounds)
        opt.set_option(api.OPT_ROUND_GROUP, group)
        outs.append(opt.plan(sc, max_iter_trajs=3))
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter", "iter_trajs", "n_iter_trajs"):
        for other in outs:
            assert np.array_equal(a[k], other[k]), k
    opt.close()


@pytest.mark.parametrize("family,B,seed", [("mix11", 700, 96), ("ped6", 150, 97), ("dyn20x", 130, 98), ("demo80", 40, 99)])
def test_tail_kernel_is_bit_identical_to_the_lockstep_loop(family, B, seed):
    """CILQR_OPT_TAIL_THRESHOLD: once few problems are left, one workgroup per problem runs all remaining
    iterations (kernels_tail.hip) with the device functions of the lockstep kernels on a private copy of the
    problem.  Every output must be the lockstep loop's, bit for bit, whether the tail takes the batch over from
    the first iteration, in the middle of the solve (after re-packing, or without it), or only for the last
    stragglers -- including the per-iteration records (cost rows, accepted step sizes, iterates)."""
    sc = scenario.generate(family, B, seed=seed)
    opt = _opt(sc)
    opt.set_option(api.OPT_TAIL_THRESHOLD, 0)
    ref = _plan(opt, sc)
    assert ref["n_iter"].max() > 12, "scene set too easy to say anything about a tail"
    keys = ("traj", "cost_hist", "n_cost", "status", "n_iter", "iter_trajs", "n_iter_trajs", "alpha_trace")
    for thr, compaction in ((8192, 1), (B // 3, 1), (B // 3, 0), (16, 1), (1, 1)):
        opt.set_option(api.OPT_TAIL_THRESHOLD, thr)
        opt.set_option(api.OPT_COMPACTION, compaction)
        got = _plan(opt, sc)
        for k in keys:
            assert np.array_equal(ref[k], got[k], equal_nan=True), (family, thr, compaction, k)
    # and without the optional outputs
    opt.set_option(api.OPT_TAIL_THRESHOLD, 1024)
    plain = opt.plan(sc)
    for k in ("traj", "cost_hist", "n_cost", "status", "n_iter"):
        assert np.array_equal(ref[k], plain[k], equal_nan=True), k
    # the tail kernel's quadratisation in its other form (one lane per knot instead of four lanes sharing a knot's planes;
    # CILQR_TAIL_QUAD_SPLIT is read at every launch): the runs above took the split form wherever the horizon's candidate
    # rows offer room for its sums (N = 50, 80), this one never does
    os.environ["CILQR_TAIL_QUAD_SPLIT"] = "0"
    try:
        unsplit = _plan(opt, sc)
    finally:
        del os.environ["CILQR_TAIL_QUAD_SPLIT"]
    for k in keys:
        assert np.array_equal(ref[k], unsplit[k], equal_nan=True), (family, "unsplit", k)
    opt.close()


@pytest.mark.parametrize("types", ["stand-ins", "reference headers"])
def test_cpp_adapter_plan_matches_oracle(tmp_path, types):
    """planning::IlqrOptimizer-shaped C++ adapter (B = 1, host containers) end to end -- compiled against stand-ins for the
    reference's types and against the reference's own headers (the binary built in the build container, oracle/_ref)."""
    import subprocess
    from test_host import build_adapter_test
    exe = build_adapter_test(tmp_path, types)
    g = np.load(os.path.join(HERE, "golden", "mix11_n50.npz"))
    b = 1
    K, cmax = g["coarse"].shape[1], int(g["cmax"])
    scene = tmp_path / "scene.bin"
    with open(scene, "wb") as f:
        np.array([K, cmax, g["left"].shape[0], g["right"].shape[0]], np.int32).tofile(f)
        for a in (g["start"][b], g["coarse"][b]):
            np.ascontiguousarray(a, np.float64).tofile(f)
        np.ascontiguousarray(g["ccount"][b], np.int32).tofile(f)
        for a in (g["corridor"][b], g["left"], g["right"]):
            np.ascontiguousarray(a, np.float64).tofile(f)
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(scene), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, np.uint8)
    n_cost, n_it, flags = np.frombuffer(raw[:12].tobytes(), np.int32)
    assert flags == 3                               # Plan ok + every error path returned false
    body = np.frombuffer(raw[12:].tobytes(), np.float64)
    traj = body[:K * 10].reshape(K, 10)
    cost = body[K * 10:K * 10 + n_cost * 5].reshape(n_cost, 5)
    it0 = body[K * 10 + n_cost * 5:].reshape(K, 10)
    assert n_cost == g["ref_n_cost"][b] and n_it == n_cost - 1
    assert cost_err(cost, g["ref_cost_hist"][b, :n_cost]) < REL_TOL
    assert traj_err(traj, g["ref_traj"][b]) < REL_TOL
    o = orc.Oracle(n_steps=K - 1)
    o.set_problem(g["start"][b], g["coarse"][b], g["corridor"][b], g["ccount"][b], g["left"], g["right"])
    X, U = o.init_guess()
    assert rel_err(it0[:, 1:7], X) < 1e-9           # iter_trajs[0] is the init guess (cc:170)


@pytest.mark.parametrize("over", [
    dict(num_of_disc=3),                                             # run-time disc count (generic kernels)
    dict(num_of_disc=7, safe_margin=0.05),
    dict(w_v=0.2, w_a=0.1, w_delta=0.3, w_theta=0.05, w_jerk=0.3),    # every weight live
    dict(barrier_t=10.0, barrier_eps=0.05),
    dict(dt=0.08, max_velocity=15.0, jerk_max=6.0, jerk_min=-6.0, width=1.6, wheel_base=1.4),
])
def test_nondefault_configuration_parity(over, both_paths):
    """Every live field of IlqrConfig / VehicleParam reaches the kernels (nothing is hard-wired to
    the reference defaults)."""
    sc = scenario.generate("mix11", 72, seed=97)
    opt = _opt(sc, **over)
    g = _plan(opt, sc)
    ref = oracle_reference(sc, oracle_cfg_from(opt.cfg))
    # not reproducible by the oracle itself under these five configurations: 4 / 8 / 6 / 1 / 11 of the 72 scenes
    rep = assert_parity(g, ref, max_unstable_frac=0.16, what=str(over))
    assert_steps(g, sc, oracle_cfg_from(opt.cfg), what=str(over))
    assert rep["n_stable"] >= 60
    # the stages too, on one problem
    opt.stage_load(sc)
    opt.stage_init_guess()
    X, U = opt.read(api.T_X), opt.read(api.T_U)
    opt.stage_quadratize()
    o = orc.Oracle(oracle_cfg_from(opt.cfg))
    o.set_problem(sc["start"][5], sc["coarse"][5], sc["corridor"][5], sc["ccount"][5], sc["left"], sc["right"])
    oq = o.quadratize(X[5], U[5])
    for k, t in dict(A=api.T_A, B=api.T_B, lx=api.T_LX, lu=api.T_LU, lxx=api.T_LXX, luu=api.T_LUU).items():
        got = opt.read(t)[5]
        assert entry_err(got, oq[k])[0] < STAGE_TOL, (k, entry_err(got, oq[k]))
    assert rel_err(opt.stage_total_cost()[5], o.total_cost(X[5], U[5])) < STAGE_TOL
    opt.close()
