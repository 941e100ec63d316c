"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (cilqr_amd/, include/) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_CFG_FIELDS = [
    ("n_steps", C.c_int), ("dt", C.c_double), ("num_of_disc", C.c_int), ("safe_margin", C.c_double),
    ("w_jerk", C.c_double), ("w_delta_rate", C.c_double), ("w_x", C.c_double), ("w_y", C.c_double),
    ("w_theta", C.c_double), ("w_v", C.c_double), ("w_a", C.c_double), ("w_delta", C.c_double),
    ("max_iter", C.c_int), ("abs_cost_tol", C.c_double), ("rel_cost_tol", C.c_double),
    ("front_hang", C.c_double), ("wheel_base", C.c_double), ("rear_hang", C.c_double),
    ("width", C.c_double), ("max_velocity", C.c_double), ("min_acceleration", C.c_double),
    ("max_acceleration", C.c_double), ("jerk_min", C.c_double), ("jerk_max", C.c_double),
    ("delta_min", C.c_double), ("delta_max", C.c_double), ("delta_rate_min", C.c_double),
    ("delta_rate_max", C.c_double), ("barrier_t", C.c_double), ("barrier_eps", C.c_double),
]


class OracleConfig(C.Structure):
    _fields_ = _CFG_FIELDS


def build(force: bool = False) -> str:
    # CILQR_ORACLE_LIB: another build of the same sources (oracle/Makefile `asan`: liboracle_asan.so, AddressSanitizer +
    # UndefinedBehaviorSanitizer; the process must then run with the sanitizer runtime preloaded, tests/test_oracle.py)
    alt = os.environ.get("CILQR_ORACLE_LIB")
    if alt:
        if not os.path.exists(alt):
            raise RuntimeError(f"CILQR_ORACLE_LIB={alt} does not exist (make -C oracle asan)")
        return alt
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cilqr_oracle.cc", "corridor_oracle.cc", "dp_oracle.cc", "tracker_oracle.cc")]
    if force or not os.path.exists(so) or any(
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so) for src in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


REF_LIB_PATH = os.path.join(_HERE, "_ref", "libcilqr_ref.so")
_REF = [None, False]


def _pin_signatures(L, prefix):
    """argument / result types of the hooks shared by oracle/ref_shim.cc (prefix ref_) and the oracle (prefix oracle_)"""
    D, I, P = C.c_double, C.c_int, C.c_void_p

    def sig(name, res, args):
        f = getattr(L, prefix + name, None)
        if f is not None:
            f.restype, f.argtypes = res, args
    sig("normalize_angle", D, [D])
    sig("segment_distance", D, [P, D, D])
    sig("nearest_segment", I, [P, I, D, D])
    sig("compute_path_profile", I, [D, P, I, P, P, P, P, P])
    sig("polygon_overlaps_aabox", I, [P, I, D, D, D, D])
    sig("polygon_point_in", I, [P, I, D, D])
    sig("trajectory_evaluate_station", None, [P, I, D, P])
    sig("trajectory_evaluate_time", None, [P, I, D, P])
    sig("trajectory_cartesian", None, [P, I, D, D, P])
    sig("tracker_evaluate_time", None, [P, I, D, P])
    sig("tracker_projection", None, [P, I, D, D, P])
    sig("default", I, [C.c_char_p, C.POINTER(D)])
    sig("slerp", D, [D, D, D, D, D])
    sig("tracker_slerp", D, [D, D, D, D, D])
    sig("lin_spaced", I, [I, D, D, P])
    sig("pose_transform", None, [D, D, D, D, D, D, P])
    sig("collision_boxes", None, [D, D, D, D, P])
    sig("vehicle_derived", None, [P])


def ref_lib():
    """oracle/_ref/libcilqr_ref.so: the parts of the REFERENCE itself that build here with g++ alone (oracle/ref_shim.cc,
    oracle/Makefile target `ref`), or None when the file is absent (no reference tree and nothing shipped)."""
    if not _REF[1]:
        _REF[1] = True
        if os.path.isdir("/root/reference/algorithm"):
            subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
        if os.path.exists(REF_LIB_PATH):
            L = C.CDLL(REF_LIB_PATH)
            _pin_signatures(L, "ref_")
            L.ref_trajectory_projection.restype = None
            L.ref_trajectory_projection.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
            _REF[0] = L
    return _REF[0]


_REF_EIGEN = [None, False]


def ref_eigen_lib():
    """oracle/_ref/libcilqr_ref_eigen.so: the reference's own vehicle model and barrier functions, which exist only
    where an image has Eigen (oracle/Makefile target `ref_eigen`, oracle/ref_eigen_shim.cc); None otherwise."""
    if not _REF_EIGEN[1]:
        _REF_EIGEN[1] = True
        ref_lib()                                  # its make target tries the Eigen part too
        path = os.path.join(_HERE, "_ref", "libcilqr_ref_eigen.so")
        if os.path.exists(path):
            L = C.CDLL(path)
            D, I, P = C.c_double, C.c_int, C.c_void_p
            L.ref_model_create.restype, L.ref_model_create.argtypes = P, [D, D, D]
            L.ref_model_destroy.restype, L.ref_model_destroy.argtypes = None, [P]
            L.ref_dynamics.restype, L.ref_dynamics.argtypes = None, [P, P, P, P]
            L.ref_dynamics_jacobian.restype, L.ref_dynamics_jacobian.argtypes = None, [P, P, P, P, P]
            L.ref_barrier_value.restype, L.ref_barrier_value.argtypes = D, [D, D, D]
            L.ref_barrier_jacobian.restype, L.ref_barrier_jacobian.argtypes = None, [D, D, D, P, I, P]
            L.ref_barrier_hessian.restype, L.ref_barrier_hessian.argtypes = None, [D, D, D, P, P, I, P]
            _REF_EIGEN[0] = L
    return _REF_EIGEN[0]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(OracleConfig)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_total_cost.restype = C.c_double
        L.oracle_grad_norm.restype = C.c_double
        L.oracle_normalize_angle.restype = C.c_double
        L.oracle_normalize_angle.argtypes = [C.c_double]
        L.oracle_segment_distance.restype = C.c_double
        L.oracle_barrier_value.restype = C.c_double
        L.oracle_build_corridor.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_lane_constraints.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int]
        _pin_signatures(L, "oracle_")
        L.oracle_trajectory_projection.restype = None
        L.oracle_trajectory_projection.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.oracle_nearest_segment.restype = C.c_int
        L.oracle_nearest_segment.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
        _LIB = L
    return _LIB


DOT_ORDER_SEQUENTIAL, DOT_ORDER_EIGEN_REDUX, DOT_ORDER_EIGEN_SSE2 = 0, 1, 2
DOT_ORDER_DEFAULT = DOT_ORDER_EIGEN_SSE2      # -DCILQR_DOT_ORDER of cilqr_oracle.cc; what the product kernels implement
SEMANTICS_DEFAULT = 0 | DOT_ORDER_DEFAULT << 8   # what set_semantics returns with both switches at their defaults


def set_semantics(dv_eval: int = -1, dot_order: int = -1) -> int:
    """The oracle's two switches for what cannot be checked against Eigen here (cilqr_oracle.cc, top): dv_eval 0 lazy /
    1 eager, dot_order 0 sequential / 1 eigen_redux / 2 eigen_sse2; negative leaves a switch alone.  Process-wide."""
    L = lib()
    L.oracle_set_semantics.restype = C.c_int
    return L.oracle_set_semantics(C.c_int(dv_eval), C.c_int(dot_order))


def reset_semantics() -> int:
    """Both switches back to the defaults (lazy delta_V_, eigen_sse2 dot order)."""
    r = set_semantics(0, DOT_ORDER_DEFAULT)
    assert r == SEMANTICS_DEFAULT, r
    return r


def default_config(n_steps: int = 50, **over) -> OracleConfig:
    c = OracleConfig()
    lib().oracle_default_config(C.byref(c), C.c_int(n_steps))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One problem at a time; mirrors IlqrOptimizer (ilqr_optimizer.h:29-52) + its stages."""

    def __init__(self, cfg: OracleConfig | None = None, n_steps: int = 50):
        self.cfg = cfg or default_config(n_steps)
        self.N = self.cfg.n_steps
        self.K = self.N + 1
        self.L = lib()
        self.h = C.c_void_p(self.L.oracle_create(C.byref(self.cfg)))
        self.cmax = 0
        self.nl = self.nr = 0

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    def set_problem(self, start, coarse, corridor, ccount, left, right) -> int:
        start, coarse, corridor = _f64(start), _f64(coarse), _f64(corridor)
        ccount = np.ascontiguousarray(ccount, dtype=np.int32)
        left, right = _f64(left), _f64(right)
        self.cmax = corridor.shape[1] if corridor.ndim == 3 else 0
        self.nl, self.nr = left.shape[0], right.shape[0]
        return self.L.oracle_set_problem(self.h, _p(start), _p(coarse), C.c_int(coarse.shape[0]),
                                         _p(corridor) if corridor.size else None,
                                         _p(ccount, C.c_int) if ccount.size else None,
                                         C.c_int(self.cmax), _p(left), C.c_int(self.nl), _p(right),
                                         C.c_int(self.nr))

    def plan(self, max_iter_trajs: int = 0, want_trace: bool = False):
        K, M = self.K, self.cfg.max_iter
        traj = np.zeros((K, 10))
        hist = np.zeros((M + 1, 5))
        n_cost, status, n_iter, n_it = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        it = np.zeros((max_iter_trajs, K, 10)) if max_iter_trajs else None
        trace = np.zeros((M, 10)) if want_trace else None
        margin = C.c_double()
        rc = self.L.oracle_plan(self.h, _p(traj), _p(hist), C.byref(n_cost), C.byref(status),
                                C.byref(n_iter), _p(it), C.c_int(max_iter_trajs), C.byref(n_it),
                                _p(trace), C.byref(margin))
        return dict(rc=rc, traj=traj, cost_hist=hist, n_cost=n_cost.value, status=status.value,
                    n_iter=n_iter.value, iter_trajs=it, n_iter_trajs=n_it.value,
                    trace=trace[:n_iter.value] if trace is not None else None,
                    min_margin=margin.value)

    def replay(self, X, U, lam: float, dlam: float, it: int):
        """Optimize() re-entered at iteration `it` from the iterate (X, U) with regularisation state
        (lam, dlam): runs until one iteration is accepted or the solve ends.  Returns dict(cost0 = cost
        row of (X, U), cost1 = accepted row or None, traj = iterate afterwards [K,10], status, n_iter
        = next iteration index, decisions = accepted alpha index per replayed iteration (-1 rejected,
        -2 gradient-norm exit), margins, lam, dlam afterwards)."""
        X, U = _f64(X), _f64(U)
        K, M = self.K, self.cfg.max_iter
        traj = np.zeros((K, 10))
        hist = np.zeros((M + 1, 5))
        n_cost, status, n_iter = C.c_int(), C.c_int(), C.c_int()
        trace = np.zeros((max(1, M - it), 10))
        lo = np.zeros(2)
        rc = self.L.oracle_replay(self.h, _p(X), _p(U), C.c_double(lam), C.c_double(dlam), C.c_int(it), _p(traj),
                                  _p(hist), C.byref(n_cost), C.byref(status), C.byref(n_iter), _p(trace), _p(lo))
        if rc != 0:
            raise RuntimeError("oracle_replay: no problem set")
        n = n_iter.value - it
        return dict(cost0=hist[0].copy(), cost1=hist[1].copy() if n_cost.value > 1 else None, traj=traj,
                    status=status.value, n_iter=n_iter.value, decisions=trace[:n, 0].astype(int),
                    margins=trace[:n, 8].copy(), lam=float(lo[0]), dlam=float(lo[1]))

    # ---- stages ----
    def constraints(self):
        goals = np.zeros((self.K, 6))
        cor = np.zeros((self.K, self.cmax, 3))
        l = np.zeros((self.nl, 3))
        r = np.zeros((self.nr, 3))
        rad = C.c_double()
        self.L.oracle_get_constraints(self.h, _p(goals), _p(cor), _p(l), _p(r), C.byref(rad))
        return goals, cor, l, r, rad.value

    def init_guess(self):
        X, U = np.zeros((self.K, 6)), np.zeros((self.N, 2))
        self.L.oracle_init_guess(self.h, _p(X), _p(U))
        return X, U

    def open_loop_rollout(self, x0, U):
        x0, U = _f64(x0), _f64(U)
        X = np.zeros((self.K, 6))
        self.L.oracle_open_loop_rollout(self.h, _p(x0), _p(U), _p(X))
        return X

    def total_cost(self, X, U):
        X, U = _f64(X), _f64(U)
        c5 = np.zeros(5)
        self.L.oracle_total_cost(self.h, _p(X), _p(U), _p(c5))
        return c5

    def quadratize(self, X, U):
        X, U = _f64(X), _f64(U)
        N, K = self.N, self.K
        out = dict(A=np.zeros((N, 6, 6)), B=np.zeros((N, 6, 2)), lx=np.zeros((K, 6)),
                   lu=np.zeros((N, 2)), lxx=np.zeros((K, 6, 6)), luu=np.zeros((N, 2, 2)))
        self.L.oracle_quadratize(self.h, _p(X), _p(U), _p(out["A"]), _p(out["B"]), _p(out["lx"]),
                                 _p(out["lu"]), _p(out["lxx"]), _p(out["luu"]))
        return out

    def backward(self, lam, q):
        N = self.N
        Kfb, kff, dV = np.zeros((N, 2, 6)), np.zeros((N, 2)), np.zeros(2)
        a = {k: _f64(v) for k, v in q.items()}
        self.L.oracle_backward(self.h, C.c_double(lam), _p(a["A"]), _p(a["B"]), _p(a["lx"]),
                               _p(a["lu"]), _p(a["lxx"]), _p(a["luu"]), _p(Kfb), _p(kff), _p(dV))
        return Kfb, kff, dV

    def grad_norm(self, kff, U):
        kff, U = _f64(kff), _f64(U)
        return self.L.oracle_grad_norm(self.h, _p(kff), _p(U))

    def forward(self, alpha, X, U, Kfb, kff):
        X, U, Kfb, kff = _f64(X), _f64(U), _f64(Kfb), _f64(kff)
        Xn, Un = np.zeros_like(X), np.zeros_like(U)
        self.L.oracle_forward(self.h, C.c_double(alpha), _p(X), _p(U), _p(Kfb), _p(kff), _p(Xn), _p(Un))
        return Xn, Un

    def dynamics(self, x, u):
        x, u = _f64(x), _f64(u)
        xn = np.zeros(6)
        self.L.oracle_dynamics(self.h, _p(x), _p(u), _p(xn))
        return xn

    def dynamics_jacobian(self, x, u):
        x, u = _f64(x), _f64(u)
        A, B = np.zeros((6, 6)), np.zeros((6, 2))
        self.L.oracle_dynamics_jacobian(self.h, _p(x), _p(u), _p(A), _p(B))
        return A, B

    def barrier_value(self, g):
        return self.L.oracle_barrier_value(self.h, C.c_double(g))

    def barrier_jacobian(self, g, dg):
        dg = _f64(dg)
        out = np.zeros_like(dg)
        self.L.oracle_barrier_jacobian(self.h, C.c_double(g), _p(dg), C.c_int(dg.size), _p(out))
        return out

    def barrier_hessian(self, g, dg, ddg=None):
        dg = _f64(dg)
        n = dg.size
        ddg = _f64(ddg) if ddg is not None else None
        out = np.zeros((n, n))
        self.L.oracle_barrier_hessian(self.h, C.c_double(g), _p(dg), _p(ddg), C.c_int(n), _p(out))
        return out


def normalize_angle(a: float) -> float:
    return lib().oracle_normalize_angle(C.c_double(a))


def segment_distance(seg4, px, py) -> float:
    seg4 = _f64(seg4)
    return lib().oracle_segment_distance(_p(seg4), C.c_double(px), C.c_double(py))


def solve_batch_threads(scene: dict, cfg: "OracleConfig | None" = None, threads: int = 0):
    """solve_batch on `threads` host threads inside the C library (0 = all cores): contiguous slices, one Oracle per thread.
    Returns the same arrays as solve_batch (no margins / traces) and `seconds`, the wall time of the threaded loop."""
    start, coarse = _f64(scene["start"]), _f64(scene["coarse"])
    corridor = _f64(scene["corridor"])
    ccount = np.ascontiguousarray(scene["ccount"], dtype=np.int32)
    left, right = _f64(scene["left"]), _f64(scene["right"])
    B, K = coarse.shape[0], coarse.shape[1]
    cfg = cfg or default_config(K - 1)
    assert cfg.n_steps == K - 1
    M = cfg.max_iter
    threads = threads or (os.cpu_count() or 1)
    traj = np.zeros((B, K, 10))
    hist = np.zeros((B, M + 1, 5))
    n_cost, status, n_iter = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    sec = C.c_double(0.0)
    L = lib()
    L.oracle_solve_batch_threads.restype = C.c_int
    rc = L.oracle_solve_batch_threads(C.byref(cfg), C.c_int(B), _p(start), _p(coarse), _p(corridor), _p(ccount),
                                      C.c_int(corridor.shape[2]), _p(left), C.c_int(left.shape[0]), _p(right),
                                      C.c_int(right.shape[0]), _p(traj), _p(hist), _p(n_cost), _p(status), _p(n_iter),
                                      C.c_int(threads), C.byref(sec))
    if rc != 0:
        raise RuntimeError(f"oracle_solve_batch_threads: {rc}")
    return dict(traj=traj, cost_hist=hist, n_cost=n_cost, status=status, n_iter=n_iter, seconds=sec.value, threads=threads)


def solve_batch(scene: dict, cfg: OracleConfig | None = None, want_margin: bool = True, want_trace: bool = False,
                want_times: bool = False):
    """Loop of independent Plan() calls over a problem-major scene dict (scenario.generate).
    want_trace adds alpha_trace [B, max_iter] int8 (accepted alpha index per iteration, -1 all rejected,
    -2 gradient-norm exit, -3 not run) and iter_margin [B, max_iter] (decision margins)."""
    start, coarse = _f64(scene["start"]), _f64(scene["coarse"])
    corridor = _f64(scene["corridor"])
    ccount = np.ascontiguousarray(scene["ccount"], dtype=np.int32)
    left, right = _f64(scene["left"]), _f64(scene["right"])
    B, K = coarse.shape[0], coarse.shape[1]
    cfg = cfg or default_config(K - 1)
    assert cfg.n_steps == K - 1
    M = cfg.max_iter
    traj = np.zeros((B, K, 10))
    hist = np.zeros((B, M + 1, 5))
    n_cost = np.zeros(B, np.int32)
    status = np.zeros(B, np.int32)
    n_iter = np.zeros(B, np.int32)
    margin = np.zeros(B) if want_margin else None
    sec = C.c_double()
    atrace = np.full((B, M), -3, np.int8) if want_trace else None
    imargin = np.zeros((B, M)) if want_trace else None
    ptimes = np.zeros(B) if want_times else None
    rc = lib().oracle_solve_batch_trace(C.byref(cfg), C.c_int(B), _p(start), _p(coarse), _p(corridor),
                                        _p(ccount, C.c_int), C.c_int(corridor.shape[2]), _p(left),
                                        C.c_int(left.shape[0]), _p(right), C.c_int(right.shape[0]), _p(traj),
                                        _p(hist), _p(n_cost, C.c_int), _p(status, C.c_int),
                                        _p(n_iter, C.c_int), _p(margin), C.byref(sec),
                                        _p(atrace, C.c_byte), _p(imargin), _p(ptimes))
    return dict(rc=rc, traj=traj, cost_hist=hist, n_cost=n_cost, status=status, n_iter=n_iter,
                min_margin=margin, seconds=sec.value, alpha_trace=atrace, iter_margin=imargin,
                problem_seconds=ptimes)


CORRIDOR_CFG = (25.0, 25.0, 150.0, 10.0, 10.0, 0.0)   # max_diff_x/y, radius, max_axis_x/y, is_multiple_sample (planner_config.h:75-86)


def build_corridor(ox, oy, theta, pts, cfg=CORRIDOR_CFG, max_out=64, trig=None):
    """Corridor::AddCorridorPoints + BuildCorridor for one knot.  Returns (cons [m,3], poly [m,2]);
    raises ValueError with the oracle's code on failure.  trig=(cos theta, sin theta): test hook, the box corners are
    built from these values instead of this libm's."""
    pts = _f64(np.asarray(pts, dtype=np.float64).reshape(-1, 2))
    cfg = list(cfg) + [0.0] * (6 - len(cfg))
    cfg = _f64(np.asarray(cfg + ([1.0, trig[0], trig[1]] if trig is not None else [0.0, 0.0, 0.0]), dtype=np.float64))
    cons = np.zeros((max_out, 3))
    poly = np.zeros((max_out, 2))
    m = lib().oracle_build_corridor(float(ox), float(oy), float(theta), pts.ctypes.data, pts.shape[0],
                                    cfg.ctypes.data, cons.ctypes.data, poly.ctypes.data, max_out)
    if m < 0:
        raise ValueError(m)
    return cons[:m].copy(), poly[:m].copy()


def lane_constraints(boundary, segment_length=5.0, is_left=True, max_rows=4096):
    """LaneBoundarySample + Cal{Left,Right}LaneConstraints: rows [m,7] = a b c sx sy ex ey."""
    b = _f64(np.asarray(boundary, dtype=np.float64).reshape(-1, 2))
    rows = np.zeros((max_rows, 7))
    m = lib().oracle_lane_constraints(b.ctypes.data, b.shape[0], float(segment_length), int(bool(is_left)),
                                      rows.ctypes.data, max_rows)
    if m < 0:
        raise ValueError(m)
    return rows[:m].copy()


DP_CFG_FIELDS = ("tf", "delta_t", "dp_nominal_velocity", "dp_w_obstacle", "dp_w_lateral", "dp_w_lateral_change",
                 "dp_w_lateral_velocity_change", "dp_w_longitudinal_velocity_bias", "dp_w_longitudinal_velocity_change",
                 "front_hang_length", "wheel_base", "rear_hang_length", "width", "max_velocity")
DP_CFG_DEFAULT = (8.0, 0.1, 10.0, 1000.0, 0.1, 0.5, 1.0, 10.0, 1.0, 0.96, 1.0, 0.929, 1.942, 20.0)   # planner_config.h:94-133


def dp_plan(flat: dict, start3, **over):
    """DpPlanner::Plan, line-by-line restatement (oracle/dp_oracle.cc).  `flat`: the scene flattened as by
    cilqr_amd.scene_io.flatten_scene (plain arrays; this module imports nothing from the product).
    Returns (found, coarse [K, 9] = time s x y theta kappa velocity a delta)."""
    cfg = dict(zip(DP_CFG_FIELDS, DP_CFG_DEFAULT))
    cfg.update(over)
    c = _f64([cfg[k] for k in DP_CFG_FIELDS])
    K = int(cfg["tf"] / cfg["delta_t"] + 1)
    a = {k: np.ascontiguousarray(v) for k, v in flat.items()}
    coarse = np.zeros((K, 9))
    start = _f64(start3)
    L = lib()
    L.oracle_dp_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rc = L.oracle_dp_plan(c.ctypes.data, a["center"].ctypes.data, a["center"].shape[0], a["static_points"].ctypes.data,
                          a["static_counts"].ctypes.data, len(a["static_counts"]), a["dynamic_polygon_points"].ctypes.data,
                          a["dynamic_polygon_counts"].ctypes.data, a["dynamic_trajectories"].ctypes.data,
                          a["dynamic_trajectory_counts"].ctypes.data, len(a["dynamic_polygon_counts"]),
                          start.ctypes.data, coarse.ctypes.data, K)
    if rc < 0:
        raise ValueError(rc)
    return rc == 1, coarse


TRACKER_CFG_FIELDS = ("weight_l", "weight_theta", "weight_delta", "weight_delta_rate", "preview_time", "weight_s", "weight_v",
                      "weight_a", "weight_j", "sumulation_dt", "dt", "tolerance", "max_num_iteration", "wheel_base",
                      "delta_min", "delta_max", "delta_rate_min", "delta_rate_max", "jerk_min", "jerk_max",
                      "min_acceleration", "max_acceleration")
_D40 = 40.0 / 180 * np.pi
TRACKER_CFG_DEFAULT = (1e-1, 1e-12, 1e-12, 0.1, 0.2, 5.0 * 1e-1, 1e-12, 1e-12, 0.1, 0.01, 0.1, 0.01, 150, 1.0,   # planner_config.h:18-43
                       -_D40, _D40, -_D40 / 3.0, _D40 / 3.0, -10.0, 10.0, -5.0, 5.0)                              # vehicle_param.h:31-64


def chord_stations(coarse_xy):
    """Stations of a coarse trajectory when the caller has none: accumulated chord length of its points."""
    d = np.hypot(np.diff(coarse_xy[:, 0]), np.diff(coarse_xy[:, 1]))
    out = np.zeros(len(coarse_xy))
    for i in range(1, len(out)):
        out[i] = out[i - 1] + d[i - 1]
    return out


def tracker_init_guess(start4, coarse, station=None, knot_dt=0.1, **over):
    """IlqrOptimizer::InitGuess through Tracker::Plan (oracle/tracker_oracle.cc).  coarse [K,6] = x y theta v a delta,
    station [K] (default: chord length).  Returns (X [K,6], U [K-1,2], min_margin); raises when the tracker fails."""
    cfg = dict(zip(TRACKER_CFG_FIELDS, TRACKER_CFG_DEFAULT))
    cfg.update(over)
    c = _f64([cfg[k] for k in TRACKER_CFG_FIELDS])
    coarse = _f64(coarse)
    K = coarse.shape[0]
    st = _f64(chord_stations(coarse) if station is None else station)
    s4 = _f64(start4)
    X, U = np.zeros((K, 6)), np.zeros((K - 1, 2))
    m = C.c_double()
    L = lib()
    L.oracle_tracker_init_guess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                                            C.c_void_p, C.POINTER(C.c_double)]
    rc = L.oracle_tracker_init_guess(c.ctypes.data, s4.ctypes.data, coarse.ctypes.data, st.ctypes.data, K, knot_dt,
                                     X.ctypes.data, U.ctypes.data, C.byref(m))
    if rc != 0:
        raise ValueError("tracker failed")
    return X, U, m.value
