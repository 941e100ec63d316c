"""CPU tests of the oracle (oracle/cilqr_oracle.cc) against independently restated formulas and
the committed golden fixtures.  The reference has no tests; these follow SURVEY section 4."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from parity_util import rel_err
from cilqr_amd import scenario
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def o50():
    return orc.Oracle(n_steps=50)


# ---- math_utils.cpp:53-59 ----
def test_normalize_angle():
    for a in [0.0, 1.0, -1.0, np.pi, -np.pi, 3.5, -3.5, 7.0, -7.0, 100.0, -100.0, 1e6]:
        r = orc.normalize_angle(a)
        assert -np.pi <= r < np.pi
        assert abs(np.sin(r) - np.sin(a)) < 1e-9 * max(1.0, abs(a)) and abs(np.cos(r) - np.cos(a)) < 1e-9 * max(1.0, abs(a))
    assert orc.normalize_angle(np.pi) == -np.pi  # [-pi, pi)


# ---- line_segment2d.cpp:61-75 ----
def test_segment_distance():
    seg = [0.0, 0.0, 4.0, 0.0]
    assert orc.segment_distance(seg, 2.0, 3.0) == 3.0          # perpendicular foot inside
    assert orc.segment_distance(seg, -3.0, 4.0) == 5.0         # before the start
    assert orc.segment_distance(seg, 7.0, -4.0) == 5.0         # past the end
    assert orc.segment_distance([1.0, 1.0, 1.0, 1.0], 4.0, 5.0) == 5.0   # degenerate segment
    rng = np.random.default_rng(0)
    for _ in range(200):
        s = rng.normal(size=4) * 5
        p = rng.normal(size=2) * 8
        t = np.clip(np.dot(p - s[:2], s[2:] - s[:2]) / np.dot(s[2:] - s[:2], s[2:] - s[:2]), 0, 1)
        d = np.linalg.norm(p - (s[:2] + t * (s[2:] - s[:2])))
        assert abs(orc.segment_distance(s, *p) - d) < 1e-12 * max(1.0, d)


# ---- barrier_function.h:82-147, against the formulas restated in SURVEY 8(a)-13 ----
def test_relaxed_barrier(o50):
    r, eps = 0.2, 0.01
    for g in [-5.0, -0.5, -0.0100001, -0.01, -0.005, 0.0, 0.02, 1.5]:
        if g < -eps:
            v, jc = -r * np.log(-g), -r / g
        else:
            v, jc = 0.5 * r * (((-g - 2 * eps) / eps) ** 2 - 1) - r * np.log(eps), r * (g + 2 * eps) / eps ** 2
        assert o50.barrier_value(g) == pytest.approx(v, rel=1e-14, abs=1e-14)
        dg = np.array([0.3, -0.7, 1.1, 0.0, 0.0, 0.0])
        ddg = np.zeros((6, 6))
        ddg[2, 2] = 0.45
        assert np.allclose(o50.barrier_jacobian(g, dg), jc * dg, rtol=1e-14, atol=0)
        H = o50.barrier_hessian(g, dg, ddg)
        if g < -eps:
            Href = r / g ** 2 * np.outer(dg, dg) - r / g * ddg
        else:
            # quirk: the relaxed branch reuses the gradient coefficient and drops ddg
            Href = r * (g + 2 * eps) / eps ** 2 * np.outer(dg, dg)
        assert np.allclose(H, Href, rtol=1e-13, atol=1e-300)
    # value and gradient are continuous at g = -eps, the Hessian is not (reference behaviour)
    a, b = o50.barrier_value(-eps - 1e-12), o50.barrier_value(-eps + 1e-12)
    assert abs(a - b) < 1e-9


# ---- vehicle_model.cc:88-138 ----
def _f(x, u, L=1.0):
    wrap = orc.normalize_angle
    th, v, a, de = wrap(x[2]), x[3], x[4], wrap(x[5])
    return np.array([v * np.cos(th), v * np.sin(th), v * np.tan(de) / L, a, u[0], u[1]])


def test_dynamics_midpoint(o50):
    rng = np.random.default_rng(1)
    for _ in range(100):
        x = np.array([rng.normal() * 50, rng.normal() * 50, rng.uniform(-3, 3), rng.uniform(0, 15),
                      rng.uniform(-3, 3), rng.uniform(-0.6, 0.6)])
        u = np.array([rng.uniform(-10, 10), rng.uniform(-0.23, 0.23)])
        k1 = _f(x, u)
        k2 = _f(x + 0.05 * k1, u)
        ref = x + 0.1 * k2
        ref[2], ref[5] = orc.normalize_angle(ref[2]), orc.normalize_angle(ref[5])
        assert np.allclose(o50.dynamics(x, u), ref, rtol=1e-14, atol=1e-14)


def test_open_loop_rollout(o50):
    rng = np.random.default_rng(2)
    x0 = np.array([1.0, 2.0, 0.3, 8.0, 0.1, 0.02])
    U = np.stack([rng.uniform(-2, 2, 50), rng.uniform(-0.1, 0.1, 50)], axis=1)
    X = o50.open_loop_rollout(x0, U)
    x = x0.copy()
    for i in range(50):
        assert np.array_equal(X[i], x)
        x = o50.dynamics(x, U[i])
    assert np.array_equal(X[50], x)


# ---- vehicle_model.cc:21-86: analytic Jacobian vs finite differences ----
def test_dynamics_jacobian_fd(o50):
    rng = np.random.default_rng(3)
    for _ in range(20):
        x = np.array([rng.normal() * 10, rng.normal() * 10, rng.uniform(-1, 1), rng.uniform(2, 12),
                      rng.uniform(-2, 2), rng.uniform(-0.4, 0.4)])
        u = np.array([rng.uniform(-5, 5), rng.uniform(-0.2, 0.2)])
        A, B = o50.dynamics_jacobian(x, u)
        h = 1e-6
        Afd = np.stack([(o50.dynamics(x + h * e, u) - o50.dynamics(x - h * e, u)) / (2 * h) for e in np.eye(6)], axis=1)
        Bfd = np.stack([(o50.dynamics(x, u + h * e) - o50.dynamics(x, u - h * e)) / (2 * h) for e in np.eye(2)], axis=1)
        # A(2,5) and B(2,1) use v instead of v + a*dt/2 (vehicle_model.cc:59,74,82): loose tolerance
        assert np.allclose(A, Afd, rtol=0, atol=2e-2)
        assert np.allclose(B, Bfd, rtol=0, atol=2e-3)
        mask = np.ones((6, 6), bool)
        mask[2, 5] = False
        assert np.allclose(A[mask], Afd[mask], rtol=0, atol=1e-6)
        # structure the HIP kernels rely on: A = I + strictly upper entries, fixed B pattern
        assert np.array_equal(np.diag(A), np.ones(6)) and np.array_equal(np.tril(A, -1), np.zeros((6, 6)))
        assert A[3, 4] == 0.1 and A[3, 5] == 0 and A[4, 5] == 0 and A[0, 1] == 0
        nzB = {(2, 1), (3, 0), (4, 0), (5, 1)}
        assert all((B[r, c] != 0) == ((r, c) in nzB) for r in range(6) for c in range(2))


# ---- Backward (ilqr_optimizer.cc:334-390) against an independent numpy statement ----
def _backward_numpy(lam, q, late_dV=True):
    N = q["A"].shape[0]
    Vx, Vxx = q["lx"][N].copy(), q["lxx"][N].copy()
    Ks, ks, dV = np.zeros((N, 2, 6)), np.zeros((N, 2)), np.zeros(2)
    for i in range(N - 1, -1, -1):
        A, B = q["A"][i], q["B"][i]
        Qx, Qu = q["lx"][i] + A.T @ Vx, q["lu"][i] + B.T @ Vx
        Qxx, Quu, Qux = q["lxx"][i] + A.T @ Vxx @ A, q["luu"][i] + B.T @ Vxx @ B, B.T @ Vxx @ A
        inv = np.linalg.inv(Quu + lam * np.eye(2))
        K, k = -inv @ Qux, -inv @ Qu
        if not late_dV:
            dV += [k @ Qu, 0.5 * k @ Quu @ k]
        Vx = Qx + K.T @ Quu @ k + K.T @ Qu + Qux.T @ k
        Vxx = Qxx + K.T @ Quu @ K + K.T @ Qux + Qux.T @ K
        Vxx = 0.5 * (Vxx + Vxx.T)
        if late_dV:  # the reference's lazy `auto` Qu/Quu see the updated Vx/Vxx
            dV += [k @ (q["lu"][i] + B.T @ Vx), 0.5 * k @ (q["luu"][i] + B.T @ Vxx @ B) @ k]
        Ks[i], ks[i] = K, k
    return Ks, ks, dV


def test_backward_matches_numpy_and_keeps_lazy_dv_quirk():
    sc = scenario.generate("mix11", 2, seed=9)
    o = orc.Oracle(n_steps=50)
    o.set_problem(sc["start"][0], sc["coarse"][0], sc["corridor"][0], sc["ccount"][0], sc["left"], sc["right"])
    X, U = o.init_guess()
    q = o.quadratize(X, U)
    for lam in [1.0, 1e-3, 40.0]:
        K, k, dV = o.backward(lam, q)
        Kn, kn, dVn = _backward_numpy(lam, q, late_dV=True)
        assert rel_err(K, Kn, 1e-9) < 1e-9 and rel_err(k, kn, 1e-9) < 1e-9
        assert rel_err(dV, dVn, 1e-9) < 1e-9
        _, _, dV_textbook = _backward_numpy(lam, q, late_dV=False)
        assert rel_err(dV, dV_textbook, 1e-9) > 1e-6   # the quirk is observable
    g = o.grad_norm(k, U)
    assert g == pytest.approx(np.mean(np.max(np.abs(k) / (np.abs(U) + 1), axis=1)), rel=1e-13)


def test_cost_components_sum_and_shrink(o50):
    sc = scenario.generate("ped6", 1, seed=4)
    o50.set_problem(sc["start"][0], sc["coarse"][0], sc["corridor"][0], sc["ccount"][0], sc["left"], sc["right"])
    goals, cor, la, ra, rad = o50.constraints()
    assert rad == pytest.approx(np.hypot(1.942 / 2, 2.889 / 2 / 5), rel=1e-15)   # cc:97-104
    assert np.array_equal(goals[1:], sc["coarse"][0][1:])
    assert np.array_equal(goals[0], [*sc["start"][0], 0.0, 0.0])                  # cc:151
    raw = sc["corridor"][0][3, 0]
    c = raw[2] - (rad + 0.2) * (raw[0] ** 2 + raw[1] ** 2) / np.hypot(raw[0], raw[1])
    n = np.hypot(np.hypot(raw[0], raw[1]), c)                                     # 3-vector norm quirk
    assert np.allclose(cor[3, 0], [raw[0] / n, raw[1] / n, c / n], rtol=1e-15)
    X, U = o50.init_guess()
    c5 = o50.total_cost(X, U)
    assert c5[0] == c5[1] + c5[2] + c5[3] + c5[4]
    assert np.all(np.abs(U[:, 0]) <= 10.0) and np.all(np.abs(U[:, 1]) <= 40 / 180 * np.pi / 3 + 1e-15)


def test_cost_gradient_matches_finite_differences_of_total_cost(o50):
    """CostJacbian (cc:620-636) must be the gradient of TotalCost (cc:417-436): both are restated
    independently, so this ties them together.  (The Hessian is NOT the second derivative on the
    relaxed branch -- reference quirk -- and is not checked this way.)"""
    sc = scenario.generate("mix11", 3, seed=17)
    for b in range(3):
        o50.set_problem(sc["start"][b], sc["coarse"][b], sc["corridor"][b], sc["ccount"][b], sc["left"], sc["right"])
        X, U = o50.init_guess()
        q = o50.quadratize(X, U)
        h = 1e-6
        for i in (0, 7, 23, 50):
            for e in range(6):
                Xp, Xm = X.copy(), X.copy()
                Xp[i, e] += h
                Xm[i, e] -= h
                fd = (o50.total_cost(Xp, U)[0] - o50.total_cost(Xm, U)[0]) / (2 * h)
                assert fd == pytest.approx(q["lx"][i, e], rel=2e-4, abs=2e-4), (b, i, e)
        for i in (0, 11, 49):
            for e in range(2):
                Up, Um = U.copy(), U.copy()
                Up[i, e] += h
                Um[i, e] -= h
                fd = (o50.total_cost(X, Up)[0] - o50.total_cost(X, Um)[0]) / (2 * h)
                assert fd == pytest.approx(q["lu"][i, e], rel=2e-4, abs=2e-4), (b, i, e)


def test_demo_like_scenes_look_like_the_readme_figures():
    """The only recorded output of the reference is resources/cost.png (README.md:27-28): one 80-step
    demo scene, ~19 cost rows, total ~1e5 at the init guess dominated by the corridor term (8.6e4)
    with target ~1.4e4, lane ~3e2 and dynamic ~0, decreasing monotonically to O(1e2..1e3).  The
    oracle on demo-like scenes (same horizon, start pose, obstacle mix) must show that picture."""
    sc = scenario.generate("demo80", 48, seed=123)
    r = orc.solve_batch(sc, orc.default_config(80))
    n = r["n_cost"]
    init = r["cost_hist"][:, 0]
    fin = np.array([r["cost_hist"][b, n[b] - 1] for b in range(48)])
    for b in range(48):
        assert np.all(np.diff(r["cost_hist"][b, :n[b], 0]) < 0)
    assert 3 <= np.median(n) <= 40 and n.max() <= 120
    assert np.median(init[:, 3] / init[:, 0]) > 0.5            # corridor term dominates the init cost
    assert 1e3 < np.median(init[:, 0]) < 5e6
    assert np.all(init[:, 2] < 50.0)                            # dynamic (bound) barriers inactive at the start
    assert np.median(fin[:, 0]) < 0.5 * np.median(init[:, 0])
    assert np.median(fin[:, 0]) < 5e3
    # typical controls sit well inside the bounds the barriers encode (resources/results.png); the
    # relaxed barrier does let hard scenes (the 5 m U-turn at 10 m/s) exceed them
    jerk, drate = r["traj"][:, :-1, 8], r["traj"][:, :-1, 9]
    assert np.percentile(np.abs(jerk), 90) <= 10.0 and np.percentile(np.abs(drate), 75) <= 0.2327


def test_plan_error_paths(o50):
    sc = scenario.generate("ped6", 1, seed=4)
    args = (sc["start"][0], sc["coarse"][0], sc["corridor"][0], sc["ccount"][0], sc["left"], sc["right"])
    assert o50.set_problem(*args) == 0
    assert o50.set_problem(args[0], args[1][:-1], *args[2:]) == -1               # knot count (cc:75)
    assert o50.set_problem(*args[:4], sc["left"][:0], sc["right"]) == -1        # empty lane list (cc:68)
    assert o50.set_problem(args[0], args[1], np.zeros((0,)), np.zeros((0,), np.int32), args[4], args[5]) == -1


# ---- golden fixtures: the oracle must keep reproducing them bit for bit ----
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))))
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    N = int(g["n_steps"])
    o = orc.Oracle(n_steps=N)
    for b in range(g["start"].shape[0]):
        assert o.set_problem(g["start"][b], g["coarse"][b], g["corridor"][b], g["ccount"][b], g["left"], g["right"]) == 0
        r = o.plan()
        nc = int(g["ref_n_cost"][b])
        assert r["n_cost"] == nc and r["status"] == int(g["ref_status"][b]) and r["n_iter"] == int(g["ref_n_iter"][b])
        assert np.array_equal(r["cost_hist"][:nc], g["ref_cost_hist"][b, :nc])
        assert np.array_equal(r["traj"], g["ref_traj"][b])
        h = r["cost_hist"][:nc, 0]
        assert np.all(np.diff(h) < 0)                     # accepted iterations strictly decrease the cost
        assert np.allclose(r["traj"][:, 0], np.arange(N + 1) * 0.1)
    o.set_problem(g["start"][0], g["coarse"][0], g["corridor"][0], g["ccount"][0], g["left"], g["right"])
    X, U = o.init_guess()
    assert np.array_equal(X, g["st_X"]) and np.array_equal(U, g["st_U"])
    q = o.quadratize(X, U)
    for k in q:
        assert np.array_equal(q[k], g["st_q_" + k])
    K, k, dV = o.backward(1.0, q)
    assert np.array_equal(K, g["st_K"]) and np.array_equal(k, g["st_k"]) and np.array_equal(dV, g["st_dV"])
    r = o.plan(max_iter_trajs=8)
    # iter_trajs = init guess + accepted non-final iterates (cc:170,294)
    assert r["n_iter_trajs"] == int(g["st_n_iter_trajs"]) == r["n_cost"] - 1
    assert np.array_equal(r["iter_trajs"][0, :, 1:7], X)


def test_threaded_batch_driver_equals_the_sequential_one():
    """oracle_solve_batch_threads (bench.py's cpu_baseline.all_cores): contiguous slices on host threads, one solver object each;
    every output array must equal the single-threaded loop's, for thread counts that divide the batch and that do not."""
    from cilqr_amd import scenario
    sc = scenario.generate("ped6", 37, seed=9)
    ref = orc.solve_batch(sc, want_margin=False)
    for threads in (1, 3, 8, 64):
        got = orc.solve_batch_threads(sc, threads=threads)
        for k in ("traj", "cost_hist", "n_cost", "status", "n_iter"):
            assert np.array_equal(got[k], ref[k]), (threads, k)


# ---- the two unverifiable readings of Eigen's semantics, as switches (oracle/cilqr_oracle.cc, top) ----
def test_dot_order_switch_adds_six_terms_in_the_documented_orders():
    """CILQR_DOT_ORDER: 0 sequential; 1 eigen_redux = (t0+(t1+t2)) + (t3+(t4+t5)) on every product; 2 eigen_sse2 =
    (t0+(t2+t4)) + (t1+(t3+t5)) for X.transpose() * Y and sequential for a plain left operand.  Python floats add in IEEE
    double, so the three associations can be written down and compared bit for bit."""
    L = orc.lib()
    L.oracle_dot6.restype = C.c_double
    L.oracle_dot6.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(17)
    n_diff = 0
    try:
        for _ in range(2000):
            a = rng.standard_normal(6) * 10.0 ** rng.integers(-3, 4, 6)
            b = rng.standard_normal(6)
            t = [float(x) * float(y) for x, y in zip(a, b)]
            seq = ((((t[0] + t[1]) + t[2]) + t[3]) + t[4]) + t[5]
            tree = (t[0] + (t[1] + t[2])) + (t[3] + (t[4] + t[5]))
            pack = (t[0] + (t[2] + t[4])) + (t[1] + (t[3] + t[5]))
            n_diff += (seq != tree) + (seq != pack)
            for order, (plain, transposed) in {0: (seq, seq), 1: (tree, tree), 2: (seq, pack)}.items():
                assert orc.set_semantics(-1, order) >> 8 == order
                assert L.oracle_dot6(a.ctypes.data, b.ctypes.data, 0) == plain, order
                assert L.oracle_dot6(a.ctypes.data, b.ctypes.data, 1) == transposed, order
    finally:
        orc.reset_semantics()
    assert n_diff > 1000          # the orders do round differently on such data


def test_semantics_switches_move_only_what_they_should():
    """dv_eval touches delta_V_ and nothing else of a backward pass; a dot order moves gains by rounding only; the defaults
    (lazy, eigen_sse2) are what the golden fixtures were made with (test_oracle_reproduces_golden runs on them)."""
    g = np.load(os.path.join(HERE, "golden", "mix11_n50.npz"))
    o = orc.Oracle(n_steps=int(g["n_steps"]))
    o.set_problem(g["start"][0], g["coarse"][0], g["corridor"][0], g["ccount"][0], g["left"], g["right"])
    X, U = o.init_guess()
    q = o.quadratize(X, U)
    K0, k0, dV0 = o.backward(1.0, q)
    try:
        assert orc.set_semantics(1, -1) == (1 | orc.DOT_ORDER_DEFAULT << 8)
        K1, k1, dV1 = o.backward(1.0, q)
        assert np.array_equal(K0, K1) and np.array_equal(k0, k1)
        assert np.all(np.abs(dV1 - dV0) > 1e-9 * np.abs(dV0))            # a different quantity, not a rounding
        orc.set_semantics(0, -1)
        for order in (orc.DOT_ORDER_SEQUENTIAL, orc.DOT_ORDER_EIGEN_REDUX):
            orc.set_semantics(-1, order)
            K2, k2, dV2 = o.backward(1.0, q)
            assert not np.array_equal(K0, K2)
            assert np.abs(K2 - K0).max() <= 1e-10 * np.abs(K0).max() and np.allclose(dV2, dV0, rtol=1e-9)
            X2, _ = o.init_guess()
            assert np.abs(X2 - X).max() < 1e-9
    finally:
        orc.reset_semantics()
    K3, k3, dV3 = o.backward(1.0, q)
    assert np.array_equal(K3, K0) and np.array_equal(dV3, dV0)


def test_oracle_under_address_and_ub_sanitizers():
    """SURVEY 5 / VERDICT r04 item 8: the restatement built with -fsanitize=address,undefined (`make -C oracle asan`) re-runs
    this file's tests, the parity rules and the CPU part of the reference pins in a python that has the sanitizer runtime
    preloaded: no report, every test green.  (The reference itself has no sanitizer configuration: CMakeLists.txt:9.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    if os.environ.get("CILQR_ORACLE_LIB"):
        pytest.skip("already the sanitizer run")
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "-s", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), f"no libasan.so next to gcc ({asan!r})"
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               CILQR_ORACLE_LIB=os.path.join(root, "oracle", "liboracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(here, "test_oracle.py"), os.path.join(here, "test_parity_rules.py"),
                        os.path.join(here, "test_reference_pins.py")],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error:" not in r.stdout + r.stderr, tail
    assert " passed" in r.stdout, tail
