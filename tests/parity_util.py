"""Shared helpers for the parity tests: scene -> oracle / HIP outputs, comparison rules.

What "parity" means here (DESIGN.md section 5 has the long form):

1. WHOLE SOLVES.  Status, iteration count, the accepted step size of every iteration, every Cost
   row and the final trajectory of the HIP path agree with the oracle within REL_TOL = 1e-4
   (north_star) -- on every problem whose ORACLE result is itself reproducible: the reference
   iteration amplifies rounding noise (measured: re-running the oracle on inputs scaled by
   1 + 4e-16 N(0,1) changes cost rows of ~5 % of the scenes by more than 1e-4, smoothly, long
   before any accept / reject decision flips), so a problem is *stable* when N_PERTURB = 8 such
   re-runs of the oracle all stay within STABLE_TOL = 1e-5 of the unperturbed oracle.  The unstable
   share is bounded by the tests, never 100 %.

2. EVERY STEP OF EVERY PROBLEM, stable or not (`check_steps`).  The solve is a chain of steps
   "iterate k -> next accepted iterate" (the iterations in between are rejections that only grow
   the regularisation).  The oracle is re-entered at the HIP path's OWN iterate k with the
   regularisation state that follows from the HIP path's decision history (`oracle_replay`), and
   must reproduce: the cost row of iterate k, the reject / accept decisions of the step including
   the accepted step size, the Cost row of the accepted trial, the next iterate and the exit taken.
   One step is not a long chain, so STEP_TOL = 1e-8 relative applies -- except where the step
   itself is discontinuous in the oracle (nearest-lane-segment switches, barrier branch switches,
   tan poles of a wild trial): a step that fails STEP_TOL is excused only if the oracle's own
   result for that step -- the cost of the iterate included -- changes by more than STEP_TOL / 10 (or
   its decisions flip) under a 4e-16 perturbation of the iterate it starts from (8 samples), or if the first decision that differs is one whose
   test sat within KNIFE_EDGE = 1e-9 (relative) of its threshold in the oracle (only seen when both
   cost tolerances are 0 and the solver iterates on in the rounding-noise plateau, where accepted
   cost decreases are ~1e-13 of the cost).  The excused share of steps is bounded.
   A third excuse, `lane_tie`, exists only for solves run with CILQR_OPT_EXACT_LANE_TIES = 0 (the opt-in fast rule;
   `allow_lane_tie=True`): the iterate sits on an exact tie of the reference's own nearest-lane-segment distances,
   where the reference keeps the earlier segment and a search on squared distances the strictly nearer one.  The
   library's default follows the reference there and is checked without it.

Error measure: every trajectory column is scaled by the largest magnitude of that column in the
reference trajectory (theta, delta, kappa, delta_rate are O(0.1) quantities: a floor of 1.0 would
turn "relative" into "absolute"); a Cost row entry is scaled by its own magnitude, with the row's
total magnitude * 1e-3 as the floor (a barrier component can pass through zero).
"""
from __future__ import annotations

import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

REL_TOL = 1e-4        # north_star: per-iteration costs and final trajectories within 1e-4 relative
STABLE_TOL = 1e-5     # a problem is stable when perturbed ORACLE re-runs stay within this of the oracle
PERTURB_EPS = 4e-16   # relative size of the input perturbation (about 2 ulp)
N_PERTURB = 8
STEP_TOL = 1e-8       # one step, oracle re-entered at the HIP path's own iterate
KNIFE_EDGE = 1e-9     # a decision whose test sat this close (relative) to its threshold in the oracle is undecidable
COL_FLOOR = 1e-3      # smallest column scale (a column that is identically ~0)
MAX_UNSTABLE_FRAC = 0.10   # measured: 5.8 % over 13312 scenes (profiles/r01_parity_report.json)


def oracle_cfg_from(cfg) -> "orc.OracleConfig":
    """Build the oracle's config from the product's (same field names, separate structs)."""
    o = orc.OracleConfig()
    for name, _ in orc.OracleConfig._fields_:
        setattr(o, name, getattr(cfg, name))
    return o


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(|b|, floor): for stage tensors whose entries are O(1) or larger."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def traj_err(a, ref):
    """[..., K, C] trajectories: max over entries of |a - ref| / (max_k |ref[:, c]|), per column."""
    a, ref = np.asarray(a, float), np.asarray(ref, float)
    if a.size == 0:
        return 0.0
    scale = np.maximum(np.abs(ref).max(axis=-2, keepdims=True), COL_FLOOR)
    return float(np.max(np.abs(a - ref) / scale))


def cost_err(a, ref):
    """[..., 5] Cost rows (total, target, dynamic, corridor, lane): entry-wise relative error; an
    entry smaller than 1e-3 of the row's mass is measured against that floor."""
    a, ref = np.asarray(a, float), np.asarray(ref, float)
    if a.size == 0:
        return 0.0
    mass = np.abs(ref[..., 1:]).sum(axis=-1, keepdims=True)
    scale = np.maximum(np.abs(ref), 1e-3 * np.maximum(mass, 1e-300))
    return float(np.max(np.abs(a - ref) / scale))


def entry_err(got, ref):
    """Worst error of a stage tensor [knot, ...] per ENTRY: relative to the entry itself, with a floor of 1e-3 of the largest
    entry of the same knot (a sum of barrier terms of 1e5 leaves 1e-11 of absolute rounding in every entry of its knot -- but a
    wrong small entry at a knot without such terms, or within three decades of its knot's largest, does not hide behind the
    tensor's maximum).  Returns (worst, index of the worst entry)."""
    got, ref = np.asarray(got), np.asarray(ref)
    knot_max = np.abs(ref).reshape(ref.shape[0], -1).max(axis=1).reshape((-1,) + (1,) * (ref.ndim - 1))
    scale = np.maximum(np.abs(ref), 1e-3 * np.maximum(knot_max, 1e-3))
    e = np.abs(got - ref) / scale
    return float(e.max()), tuple(int(i) for i in np.unravel_index(int(np.argmax(e)), ref.shape))


def solution_errors(gpu: dict, ref: dict, b: int):
    """(control flow equal?, cost error, trajectory error) of problem b."""
    nc = int(ref["n_cost"][b])
    flow = int(gpu["n_cost"][b]) == nc and int(gpu["status"][b]) == int(ref["status"][b])
    if flow and gpu.get("n_iter") is not None and ref.get("n_iter") is not None:
        flow = int(gpu["n_iter"][b]) == int(ref["n_iter"][b])
    if flow and gpu.get("alpha_trace") is not None and ref.get("alpha_trace") is not None:
        flow = bool(np.array_equal(gpu["alpha_trace"][b], ref["alpha_trace"][b]))
    if not flow:
        return False, np.inf, np.inf
    return True, cost_err(gpu["cost_hist"][b, :nc], ref["cost_hist"][b, :nc]), traj_err(gpu["traj"][b], ref["traj"][b])


def compare_solutions(gpu: dict, ref: dict, tol=REL_TOL):
    """Per-problem comparison of whole solves.  Returns (n_pass, failures[list of (index, reason)])."""
    B = ref["traj"].shape[0]
    n_pass = 0
    fails = []
    for b in range(B):
        flow, e_cost, e_traj = solution_errors(gpu, ref, b)
        if not flow:
            fails.append((b, f"control flow: n_cost {int(gpu['n_cost'][b])} vs {int(ref['n_cost'][b])}, status "
                             f"{int(gpu['status'][b])} vs {int(ref['status'][b])}"))
        elif e_cost > tol:
            fails.append((b, f"cost history rel err {e_cost:.3e}"))
        elif e_traj > tol:
            fails.append((b, f"trajectory rel err {e_traj:.3e}"))
        else:
            n_pass += 1
    return n_pass, fails


def oracle_reference(scene: dict, cfg=None, n_perturb: int = N_PERTURB, eps: float = PERTURB_EPS,
                     stable_tol: float = STABLE_TOL, workers: int | None = None):
    """Oracle solve (with the per-iteration decision trace) + the per-problem stability mask
    (module docstring, point 1).  The perturbed re-runs go through a thread pool: the oracle is a
    C library called through ctypes, which releases the GIL."""
    B = scene["coarse"].shape[0]
    rng = np.random.default_rng(12345)
    noise = [rng.standard_normal(scene["coarse"].shape) for _ in range(n_perturb)]

    def run(i):
        if i < 0:
            return orc.solve_batch(scene, cfg, want_trace=True)
        sc2 = dict(scene)
        sc2["coarse"] = scene["coarse"] * (1.0 + eps * noise[i])
        return orc.solve_batch(sc2, cfg, want_margin=False, want_trace=True)

    workers = workers or min(n_perturb + 1, os.cpu_count() or 1)
    with ThreadPoolExecutor(max(1, workers)) as pool:
        outs = list(pool.map(run, range(-1, n_perturb)))
    r0 = outs[0]
    stable = np.ones(B, bool)
    spread = np.zeros(B)
    for r1 in outs[1:]:
        for b in range(B):
            flow, e_cost, e_traj = solution_errors(r1, r0, b)
            e = max(e_cost, e_traj)
            spread[b] = max(spread[b], e)
            if not flow or e > stable_tol:
                stable[b] = False
    r0["stable"] = stable
    r0["spread"] = spread          # largest deviation of a perturbed oracle run (inf = control flow changed)
    return r0


MIN_FLIPS = 4          # of 64 fresh oracle re-runs: fewer endings elsewhere than this and the problem counts as stable after all


def second_look(scene: dict, cfg, b: int, gpu: dict | None = None, n_perturb: int = 64, eps: float = PERTURB_EPS,
                stable_tol: float = STABLE_TOL, seed: int = 999, max_perturb: int = 256, tol: float = REL_TOL):
    """How often the oracle itself ends elsewhere on problem `b` under `n_perturb` FRESH perturbations of the same size -- and
    whether the library's ending is ONE OF the oracle's.  The stability mask rests on N_PERTURB = 8 re-runs: a problem that
    flips in one run of five is called stable with probability 0.17, and among tens of thousands of problems a few such slip
    through.  A problem that differs from the library although the mask called it stable is looked at again with this.  It is
    excused (`excused`) only if the oracle moves in at least MIN_FLIPS of the re-runs AND -- when the library's result is given --
    that result agrees within `tol` (control flow, cost rows, trajectory) with the ending of at least one perturbed oracle run
    (the search goes on up to `max_perturb` runs while the oracle keeps flipping and no run has matched yet) -- or the oracle's
    own flipped endings are (nearly) all DISTINCT from each other, i.e. there is no set of alternative endings to be one of.
    Anything else is a mismatch: the oracle comparing unequal to ITSELF proves nothing about what the library returned."""
    B = scene["coarse"].shape[0]
    one = {k: (np.ascontiguousarray(v[b:b + 1]) if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in scene.items()}
    g1 = None
    if gpu is not None:
        g1 = {k: (v[b:b + 1] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in gpu.items()}
    r0 = orc.solve_batch(one, cfg, want_trace=True)
    rng = np.random.default_rng(seed + b)
    flips = matches = runs = 0
    endings = []          # one representative per DISTINCT ending among the runs that ended elsewhere (clustered at `tol`)
    while runs < n_perturb or (g1 is not None and matches == 0 and flips * n_perturb >= MIN_FLIPS * runs and runs < max_perturb):
        sc2 = dict(one)
        sc2["coarse"] = one["coarse"] * (1.0 + eps * rng.standard_normal(one["coarse"].shape))
        r1 = orc.solve_batch(sc2, cfg, want_margin=False, want_trace=True)
        runs += 1
        flow, e_cost, e_traj = solution_errors(r1, r0, 0)
        if (not flow) or max(e_cost, e_traj) > stable_tol:
            flips += 1
            for rep in endings:
                f2, c2, t2 = solution_errors(r1, rep, 0)
                if f2 and max(c2, t2) <= tol:
                    break
            else:
                endings.append(r1)
        if g1 is not None:
            gflow, g_cost, g_traj = solution_errors(g1, r1, 0)
            matches += int(gflow and max(g_cost, g_traj) <= tol)
    flips_in_first = flips if runs == n_perturb else None
    enough = flips * n_perturb >= MIN_FLIPS * runs
    # "One of the oracle's alternative endings" presupposes that the oracle HAS alternative endings it returns to.  On the
    # chaotic problems it does not: every perturbed run that leaves the unperturbed ending ends somewhere else again (mix11 #5284
    # of the 8192 report: 31 flips in 128 runs, 31 distinct endings).  Then -- and only then -- there is nothing to match, and the
    # problem is held to the step-by-step replay alone like every oracle-unstable problem.
    no_two_alike = flips >= MIN_FLIPS and len(endings) * 4 >= flips * 3
    return {"problem": int(b), "oracle_reruns": int(runs), "ended_elsewhere": int(flips), "ended_elsewhere_in_the_first_64": flips_in_first,
            "distinct_endings_among_those": len(endings),
            "library_result_equals_a_perturbed_oracle_ending": (int(matches) if g1 is not None else None),
            "excused_because": ("matches an oracle ending" if (enough and g1 is not None and matches > 0) else
                                "the oracle's own perturbed endings do not repeat (chaotic): step replay only" if (enough and no_two_alike) else
                                "the oracle moves (no library result given)" if (enough and g1 is None) else None),
            "excused": bool(enough and (g1 is None or matches > 0 or no_two_alike))}


def assert_parity(gpu: dict, ref: dict, tol=REL_TOL, max_unstable_frac=MAX_UNSTABLE_FRAC, what=""):
    """Whole solves: every oracle-stable problem must match within tol (control flow included); the
    unstable share is bounded.  What the unstable problems are held to is check_steps()."""
    B = ref["traj"].shape[0]
    stable = ref.get("stable")
    if stable is None:
        stable = np.ones(B, bool)
    n_pass, fails = compare_solutions(gpu, ref, tol=tol)
    bad = [(b, why) for b, why in fails if stable[b]]
    n_unstable = int((~stable).sum())
    assert n_unstable <= int(np.ceil(max_unstable_frac * B)), \
        f"{what}: {n_unstable}/{B} scenes are ill-conditioned in the oracle itself (allowed {max_unstable_frac:.0%})"
    assert not bad, f"{what}: {len(bad)} oracle-stable problems differ: {bad[:5]}"
    failed = {b for b, _ in fails}
    return dict(n=B, n_stable=int(stable.sum()), n_match=n_pass, n_unstable=n_unstable,
                n_unstable_matching=int(sum(1 for b in range(B) if not stable[b] and b not in failed)))


# ---------------------------------------------------------------------------------------------
# step-by-step parity (module docstring, point 2)
# ---------------------------------------------------------------------------------------------
def _reg_after(lam, dlam, accepted):
    """Regularisation schedule of Optimize(), ilqr_optimizer.cc:273-275 (accept) / 298-299 (reject)."""
    if accepted:
        dlam = min(dlam / 1.6, 1.0 / 1.6)
        lam = lam * dlam * (1.0 if lam > 1e-8 else 0.0)
    else:
        dlam = max(dlam * 1.6, 1.6)
        lam = max(lam * dlam, 1e-8)
    return lam, dlam


def _xu(traj_pts):
    """[K,10] trajectory points -> X [K,6], U [N,2] (TransformToTrajectory copies them verbatim)."""
    return np.ascontiguousarray(traj_pts[:, 1:7]), np.ascontiguousarray(traj_pts[:-1, 8:10])


def _step_matches(r, seg, gpu_status_after, cost_row0, cost_row1, next_traj, tol):
    """One replayed step against what the HIP path did.  Returns (ok, worst error, reason)."""
    e0 = cost_err(cost_row0, r["cost0"])
    if e0 > tol:
        return False, e0, f"cost of the iterate itself differs by {e0:.2e}"
    if not np.array_equal(r["decisions"], seg):
        return False, np.inf, f"decisions {r['decisions'].tolist()} vs HIP {seg.tolist()}"
    if r["status"] != gpu_status_after:
        return False, np.inf, f"exit {r['status']} vs HIP {gpu_status_after}"
    worst = e0
    if r["cost1"] is not None:
        e1 = cost_err(cost_row1, r["cost1"])
        e2 = traj_err(next_traj, r["traj"])
        worst = max(worst, e1, e2)
        if e1 > tol:
            return False, e1, f"accepted Cost row differs by {e1:.2e}"
        if e2 > tol:
            return False, e2, f"next iterate differs by {e2:.2e}"
    return True, worst, ""


def lane_tie(scene: dict, ocfg, X) -> bool:
    """Does some disc point of trajectory X [K,6] have two lane segments at the SAME distance in the reference's own
    arithmetic (LineSegment2d::DistanceTo: hypot to an end point, |cross| to the foot, line_segment2d.cpp:61-75)?
    The reference's strict '<' then keeps the earlier segment (FindNeastLaneSegment, ilqr_optimizer.cc:605-618) while
    the kernels, which compare squared distances, keep the strictly nearer one: the two lane costs differ by a jump.
    It happens on strips ~1e-7 m wide along the normals through the segment end points -- and iterates do come to
    rest there, because the lane cost is discontinuous exactly there (different segments carry different planes)."""
    L = (ocfg.front_hang + ocfg.wheel_base + ocfg.rear_hang) / ocfg.num_of_disc
    off = np.array([L * (j - 0.5) - ocfg.rear_hang for j in range(ocfg.num_of_disc)])
    px = (X[:, 0, None] + off * np.cos(X[:, 2, None])).ravel()
    py = (X[:, 1, None] + off * np.sin(X[:, 2, None])).ravel()
    for tab in (scene["left"], scene["right"]):
        sx, sy, ex, ey = tab[:, 3], tab[:, 4], tab[:, 5], tab[:, 6]
        ln = np.hypot(ex - sx, ey - sy)
        ux, uy = (ex - sx) / ln, (ey - sy) / ln
        x0, y0 = px[:, None] - sx[None], py[:, None] - sy[None]
        proj = x0 * ux[None] + y0 * uy[None]
        x1, y1 = px[:, None] - ex[None], py[:, None] - ey[None]
        cross = x0 * uy[None] - y0 * ux[None]
        d = np.where(proj <= 0.0, np.hypot(x0, y0), np.where(proj >= ln[None], np.hypot(x1, y1), np.abs(cross)))
        d2 = np.where(proj <= 0.0, x0 * x0 + y0 * y0, np.where(proj >= ln[None], x1 * x1 + y1 * y1, cross * cross))
        # first minimum of the distances (the reference) against first minimum of the squared distances (the kernels)
        if np.any(np.argmin(d, axis=1) != np.argmin(d2, axis=1)):
            return True
    return False


def check_steps(gpu: dict, scene: dict, ocfg, problems=None, tol=STEP_TOL, eps=PERTURB_EPS, n_perturb=8,
                seed=777, allow_lane_tie=False):
    """Replay every step of the listed problems (default: all) in the oracle, starting each step from
    the HIP path's own iterate.  `gpu` must come from plan(..., max_iter_trajs=cap, alpha_trace=True).
    Returns dict(steps, tight, excused, failed[list], worst (among tight), truncated)."""
    B = gpu["traj"].shape[0]
    cap = gpu["iter_trajs"].shape[1]
    rng = np.random.default_rng(seed)
    problems = range(B) if problems is None else problems
    out = dict(steps=0, tight=0, excused=0, knife_edge=0, lane_tie=0, failed=[], worst=0.0, truncated=0, errors=[])
    o = orc.Oracle(ocfg)
    for b in problems:
        assert o.set_problem(scene["start"][b], scene["coarse"][b], scene["corridor"][b], scene["ccount"][b],
                             scene["left"], scene["right"]) == 0
        at = gpu["alpha_trace"][b]
        n_iter = int(gpu["n_iter"][b])
        n_it = int(gpu["n_iter_trajs"][b])
        status = int(gpu["status"][b])
        n_cost = int(gpu["n_cost"][b])
        lam, dlam, it, row = 1.0, 1.0, 0, 0
        for k in range(n_it):
            if k >= cap:
                out["truncated"] += 1
                break
            X, U = _xu(gpu["iter_trajs"][b, k])
            # the HIP path's decisions of this step: rejections, then an accept or the end of the solve
            j = it
            while j < n_iter and at[j] == -1:
                j += 1
            ends_here = j >= n_iter or at[j] == -2
            seg = np.asarray(at[it:min(j + 1, n_iter)], dtype=int)
            accepted = (not ends_here)
            last_step = (j + 1 >= n_iter)
            status_after = status if (last_step or ends_here) else 0
            row1 = gpu["cost_hist"][b, row + 1] if accepted and row + 1 < n_cost else None
            if accepted:
                nxt = gpu["iter_trajs"][b, k + 1] if (k + 1 < n_it and k + 1 < cap) else gpu["traj"][b]
                if k + 1 < n_it and k + 1 >= cap:
                    nxt = None
            else:
                nxt = gpu["traj"][b]
            r = o.replay(X, U, lam, dlam, it)
            out["steps"] += 1
            if accepted and nxt is None:      # the next iterate was not kept: costs and decisions only
                nxt = r["traj"]
            ok, worst, why = _step_matches(r, seg, status_after, gpu["cost_hist"][b, row], row1, nxt, tol)
            if ok:
                out["tight"] += 1
                out["worst"] = max(out["worst"], worst)
                out["errors"].append(worst)
            else:
                # did the first differing decision hang on the last bits of its test in the oracle?
                unstable = False
                nd = min(len(seg), len(r["decisions"]))
                diff = np.nonzero(np.asarray(seg[:nd]) != r["decisions"][:nd])[0]
                jd = int(diff[0]) if diff.size else (nd if nd < len(r["decisions"]) else -1)
                if 0 <= jd < len(r["margins"]) and r["margins"][jd] < KNIFE_EDGE:
                    unstable = True
                    out["knife_edge"] += 1
                # or does the iterate (or the one the step arrives at) sit on a nearest-lane-segment tie?
                # (allow_lane_tie=True only for solves that ran with CILQR_OPT_EXACT_LANE_TIES = 0, the opt-in fast rule; the default
                # follows the reference there and gets no such excuse)
                if not unstable and allow_lane_tie and (lane_tie(scene, ocfg, X) or (nxt is not None and lane_tie(scene, ocfg, np.asarray(nxt)[:, 1:7]))):
                    unstable = True
                    out["lane_tie"] += 1
                # or is the step itself discontinuous in the oracle?
                for _ in range(0 if unstable else n_perturb):
                    rp = o.replay(X * (1.0 + eps * rng.standard_normal(X.shape)),
                                  U * (1.0 + eps * rng.standard_normal(U.shape)), lam, dlam, it)
                    if (not np.array_equal(rp["decisions"], r["decisions"]) or rp["status"] != r["status"]
                            or cost_err(rp["cost0"], r["cost0"]) > tol / 10
                            or (r["cost1"] is not None and (cost_err(rp["cost1"], r["cost1"]) > tol / 10
                                                            or traj_err(rp["traj"], r["traj"]) > tol / 10))):
                        unstable = True
                        break
                if unstable:
                    out["excused"] += 1
                else:
                    out["failed"].append((int(b), k, why))
            # follow the HIP path's own history
            for d in seg:
                lam, dlam = _reg_after(lam, dlam, d >= 0)
            it += len(seg)
            if accepted:
                row += 1
            if ends_here or last_step:
                break
    return out


def assert_steps(gpu, scene, ocfg, what="", problems=None, tol=STEP_TOL, max_excused_frac=0.02, allow_lane_tie=False):
    rep = check_steps(gpu, scene, ocfg, problems=problems, tol=tol, allow_lane_tie=allow_lane_tie)
    assert rep["steps"] > 0, f"{what}: nothing was replayed"
    assert not rep["failed"], f"{what}: {len(rep['failed'])} of {rep['steps']} steps differ from the oracle: {rep['failed'][:5]}"
    assert rep["excused"] <= max(1, int(np.ceil(max_excused_frac * rep["steps"]))), \
        f"{what}: {rep['excused']} of {rep['steps']} steps are discontinuous in the oracle (allowed {max_excused_frac:.0%})"
    rep.pop("errors")
    return rep
