"""The oracle's restatements held against the REFERENCE ITSELF, bit for bit.

oracle/_ref/libcilqr_ref.so is built by `make -C oracle ref` from the reference's own sources where they lie
(algorithm/math/{math_utils,vec2d,line_segment2d,aabox2d,box2d,polygon2d}.cpp, algorithm/utils/{discrete_points_math.cc,
discretized_trajectory.cpp}) -- the files of the hot path and its callers that need neither Eigen nor ROS nor OpenCV;
oracle/ref_shim.cc only adds extern "C" doors.  The .so travels to the GPU box, /root/reference does not, so nothing
here reads the reference tree at run time.  What these tests pin:

  SURVEY 8(a)-20  NormalizeAngle                               oracle + device (device_math fn 6)
  SURVEY 8(a)-14  LineSegment2d::DistanceTo / nearest lane     oracle + device (CILQR_OPT_EXACT_LANE_TIES)
  SURVEY 8(f)-3   ComputePathProfile, Polygon2d / Box2d / AABox2d collision, DiscretizedTrajectory queries (DP planner)
  SURVEY 8(f)-4   DiscretizedTrajectory time / projection queries as the tracker restates them

Everything else on the path (Eigen-based) stays pinned by the reference's parameter set, its scenario and the analytic
checks of tests/test_oracle.py only -- except where an image has Eigen: then `make -C oracle ref_eigen` also builds the
reference's vehicle_model.cc and barrier_function.h (oracle/ref_eigen_shim.cc), and the two tests marked needs_eigen
below hold the oracle's dynamics, Jacobians and barrier derivatives against them (first check).  Here they
are skipped, and say so.
"""
import ctypes as C

import numpy as np
import pytest

from cilqr_amd import scenario
from oracle import oracle as orc

REF = orc.ref_lib()
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/libcilqr_ref.so is not built (no reference tree here)")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _same_bits(a, b):
    """equal as bit patterns (so -0.0 != 0.0), NaN equal to NaN of the same payload"""
    return np.array_equal(_bits(a), _bits(b))


# ------------------------------------------------------------------------------------------------ NormalizeAngle
def _angles():
    rng = np.random.default_rng(8)
    k = np.arange(-9, 10)
    edges = np.concatenate([k * np.pi + d for d in (0.0, 1e-16, -1e-16, 4e-16, -4e-16, 1e-9, -1e-9)])
    edges = np.concatenate([edges, np.nextafter(k * np.pi, np.inf), np.nextafter(k * np.pi, -np.inf)])
    return np.concatenate([rng.uniform(-np.pi, np.pi, 20000), rng.uniform(-4 * np.pi, 4 * np.pi, 20000),
                           rng.uniform(-1e3, 1e3, 10000), rng.uniform(-1e9, 1e9, 5000), rng.uniform(-1e-6, 1e-6, 2000),
                           edges, np.array([0.0, -0.0, np.pi, -np.pi, 2 * np.pi, -2 * np.pi, 1e300, -1e300, 5e-324])])


def test_normalize_angle_oracle_equals_reference():
    L = orc.lib()
    x = _angles()
    got = np.array([L.oracle_normalize_angle(v) for v in x])
    ref = np.array([REF.ref_normalize_angle(v) for v in x])
    assert _same_bits(got, ref)
    # and the numpy expression the GPU test of tests/test_gpu_parity.py uses is the reference's too
    a = np.fmod(x + np.pi, 2.0 * np.pi)
    assert _same_bits(np.where(a < 0.0, a + 2.0 * np.pi, a) - np.pi, ref)


# ------------------------------------------------------------------------------------- LineSegment2d::DistanceTo
def _segments_and_points(rng, n):
    a = rng.uniform(-50, 50, (n, 2))
    d = rng.uniform(-8, 8, (n, 2))
    d[: n // 50] *= 1e-11                      # shorter than kMathEpsilon: the degenerate branch (line_segment2d.cpp:62)
    d[n // 50: n // 25] = 0.0
    seg = np.concatenate([a, a + d], axis=1)
    t = rng.uniform(-0.6, 1.6, n)
    off = rng.uniform(-5, 5, n)
    nrm = np.stack([-d[:, 1], d[:, 0]], axis=1)
    pts = a + t[:, None] * d + off[:, None] * nrm / np.maximum(np.hypot(nrm[:, 0], nrm[:, 1]), 1e-300)[:, None]
    pts[::7] = a[::7] + d[::7]                 # exactly on an end point
    pts[1::7] = a[1::7]
    pts[2::7] = a[2::7] + 0.5 * d[2::7]        # on the segment
    return np.ascontiguousarray(seg), np.ascontiguousarray(pts)


def test_segment_distance_oracle_equals_reference():
    L = orc.lib()
    seg, pts = _segments_and_points(np.random.default_rng(21), 20000)
    got = np.array([L.oracle_segment_distance(_p(seg[i]), pts[i, 0], pts[i, 1]) for i in range(len(seg))])
    ref = np.array([REF.ref_segment_distance(_p(seg[i]), pts[i, 0], pts[i, 1]) for i in range(len(seg))])
    assert _same_bits(got, ref)
    assert (ref[::7] == 0.0).sum() > len(seg) // 8 - 5    # end points are at distance exactly zero in both


def _tie_points(sc):
    """the point set of test_gpu_parity.test_exact_lane_ties_follow_the_reference_rule: random points in the road band,
    the joints, points on the normals through the joints and their neighbours a few ulp away"""
    rng = np.random.default_rng(11)
    road = scenario.build_road()
    s_ = rng.uniform(0, road.length, 20000)
    x, y = road.cartesian(s_, rng.uniform(-14.0, 10.0, s_.size))
    pts = [np.stack([x, y], axis=1), np.concatenate([sc["left"][:, 3:5], sc["right"][:, 5:7]])]
    for tab in (sc["left"], sc["right"]):
        d = tab[:, 5:7] - tab[:, 3:5]
        u = d / np.hypot(d[:, 0], d[:, 1])[:, None]
        nrm = np.stack([-u[:, 1], u[:, 0]], axis=1)
        for r in (-6.0, -2.5, -0.7, 0.3, 2.0, 6.0):
            for end in (tab[:, 3:5], tab[:, 5:7]):
                base = end + r * nrm
                pts.append(base)
                for k in (-3, -1, 1, 3):
                    pts.append(base + k * np.spacing(np.abs(base)) * np.sign(u))
                pts.append(base + 1e-9 * u)
                pts.append(base - 1e-9 * u)
    return np.ascontiguousarray(np.concatenate(pts))


def _ref_nearest(tab, pts):
    segs = np.ascontiguousarray(tab[:, 3:7])
    return np.array([REF.ref_nearest_segment(_p(segs), len(segs), p[0], p[1]) for p in pts], dtype=np.int32)


def test_nearest_lane_oracle_equals_reference_on_the_tie_strips():
    """FindNeastLaneSegment (ilqr_optimizer.cc:605-618) over the reference's own DistanceTo: the oracle's choice of
    segment is the reference's on every point, exact ties of the two distances included."""
    sc = scenario.generate("ped6", 4, seed=3)
    pts = _tie_points(sc)[::3]
    o = orc.Oracle(orc.default_config(sc["n_steps"]))
    L = orc.lib()
    n_tie = 0
    for tab in (sc["left"], sc["right"]):
        segs = np.ascontiguousarray(tab[:, 3:7])
        got = np.array([L.oracle_nearest_segment(o.h, _p(segs), len(segs), p[0], p[1]) for p in pts], dtype=np.int32)
        ref = _ref_nearest(tab, pts)
        assert np.array_equal(got, ref)
        # the set does hold exact ties of the reference's distances
        for p in pts[:4000]:
            d = np.sort([REF.ref_segment_distance(_p(segs[i]), p[0], p[1]) for i in range(len(segs))])
            n_tie += int(d[0] == d[1])
    assert n_tie > 20


# --------------------------------------------------------------------------------------------- ComputePathProfile
def _profile(fn, dt, xy):
    n = len(xy)
    out = [np.full(n, np.nan) for _ in range(5)]
    ok = fn(C.c_double(dt), _p(xy), n, *[_p(o) for o in out])
    return ok, out


def test_compute_path_profile_oracle_equals_reference():
    """DiscretePointsMath::ComputePathProfile (discrete_points_math.cc:27-176), what turns the DP planner's node chain
    into headings / stations / speeds / accelerations / curvatures of the coarse trajectory."""
    L = orc.lib()
    rng = np.random.default_rng(5)
    for n in (2, 3, 4, 5, 17, 41, 200):
        for trial in range(20):
            t = np.linspace(0, 1, n)
            xy = np.stack([40 * t + rng.normal(0, 0.3, n), 6 * np.sin(3 * t + rng.uniform(0, 6)) + rng.normal(0, 0.2, n)], axis=1)
            if trial % 5 == 4 and n > 3:
                xy[n // 2] = xy[n // 2 - 1]       # a repeated point (zero step: the divisions by ds see it)
            xy = np.ascontiguousarray(xy)
            dt = float(rng.choice([0.1, 0.2, 0.5]))
            ok_o, got = _profile(L.oracle_compute_path_profile, dt, xy)
            ok_r, ref = _profile(REF.ref_compute_path_profile, dt, xy)
            assert ok_o == ok_r == 1
            for g, r, name in zip(got, ref, ("heading", "s", "v", "a", "kappa")):
                assert _same_bits(g, r), (n, trial, name)
    one = np.zeros((1, 2))
    assert _profile(L.oracle_compute_path_profile, 0.1, one)[0] == _profile(REF.ref_compute_path_profile, 0.1, one)[0] == 0


# ------------------------------------------------------------------------------------------- collision geometry
def _polygons(rng, count):
    out = []
    for i in range(count):
        k = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        if i % 3 == 0:                             # a rotated rectangle, what obstacles are
            c, s = np.cos(ang[0]), np.sin(ang[0])
            w, h = rng.uniform(0.3, 3.0, 2)
            base = np.array([[w, h], [-w, h], [-w, -h], [w, -h]])
            pts = base @ np.array([[c, s], [-s, c]])
        else:
            pts = np.stack([np.cos(ang), np.sin(ang)], axis=1) * rng.uniform(0.5, 4.0, (k, 1) if i % 3 == 1 else (1, 1))
        out.append(np.ascontiguousarray(pts + rng.uniform(-6, 6, 2)))
    return out


def test_polygon_box_overlap_and_point_in_oracle_equal_reference():
    """Polygon2d::HasOverlap(Box2d(AABox2d)) and IsPointIn (polygon2d.cpp:120-164, box2d.cpp, aabox2d.cpp): the DP
    planner's collision test.  Convex, star-shaped (non-convex) and rectangular polygons; boxes that miss, touch,
    cross and contain; points inside, outside and on the boundary."""
    L = orc.lib()
    rng = np.random.default_rng(13)
    n_hit = n_in = total = 0
    for poly in _polygons(rng, 400):
        k = len(poly)
        lo, hi = poly.min(axis=0), poly.max(axis=0)
        boxes = [(*rng.uniform(-8, 8, 2), *rng.uniform(-8, 8, 2)) for _ in range(12)]
        boxes.append((lo[0], lo[1], hi[0], hi[1]))                              # the polygon's own bounding box
        boxes.append((hi[0], lo[1], hi[0] + 1.0, hi[1]))                        # touching along an edge of that box
        boxes.append((lo[0] - 3, lo[1] - 3, hi[0] + 3, hi[1] + 3))              # containing the polygon
        c = poly.mean(axis=0)
        boxes.append((c[0] - 1e-3, c[1] - 1e-3, c[0] + 1e-3, c[1] + 1e-3))      # (mostly) inside it
        boxes.append((poly[0, 0], poly[0, 1], poly[0, 0] + 0.5, poly[0, 1] + 0.5))   # corner on a vertex
        for b in boxes:
            b = [float(v) for v in b]
            g = L.oracle_polygon_overlaps_aabox(_p(poly), k, *b)
            r = REF.ref_polygon_overlaps_aabox(_p(poly), k, *b)
            assert g == r, (poly, b)
            n_hit += r
            total += 1
        pts = list(rng.uniform(-8, 8, (10, 2))) + [c, poly[0], 0.5 * (poly[0] + poly[1]), poly[1] + 1e-12]
        for p in pts:
            g = L.oracle_polygon_point_in(_p(poly), k, float(p[0]), float(p[1]))
            r = REF.ref_polygon_point_in(_p(poly), k, float(p[0]), float(p[1]))
            assert g == r, (poly, p)
            n_in += r
    assert total // 5 < n_hit < total and n_in > 400


# -------------------------------------------------------------------------------------- DiscretizedTrajectory
def _trajectory(rng, n, time_step=0.1):
    """rows: time s x y theta kappa velocity left_bound right_bound -- a road-like centre line with bounds"""
    t = np.arange(n) * time_step
    theta = 0.4 * np.sin(np.linspace(0, 2.5, n) + rng.uniform(0, 6)) + rng.uniform(-3, 3)
    step = rng.uniform(0.3, 1.2, n)
    x = np.cumsum(step * np.cos(theta)) + rng.uniform(-20, 20)
    y = np.cumsum(step * np.sin(theta)) + rng.uniform(-20, 20)
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    rows = np.stack([t, s, x, y, theta, np.gradient(theta) / step, step / time_step,
                     rng.uniform(2, 5, n), rng.uniform(2, 5, n)], axis=1)
    return np.ascontiguousarray(rows)


def test_trajectory_station_projection_cartesian_oracle_equal_reference():
    """DiscretizedTrajectory::EvaluateStation / GetProjection / GetCartesian (discretized_trajectory.cpp:117-203) as the
    DP planner's restatement uses them: stations before, on, between and past the knots; points beside, before and
    beyond the line."""
    L = orc.lib()
    rng = np.random.default_rng(17)
    for n in (2, 3, 10, 60, 250):
        for _ in range(6):
            rows = _trajectory(rng, n)
            smax = rows[-1, 1]
            stations = np.concatenate([rng.uniform(-2.0, smax + 2.0, 40), rows[:, 1][:: max(1, n // 7)], [0.0, smax, -1.0, smax + 5.0]])
            for st in stations:
                g, r = np.full(9, np.nan), np.full(9, np.nan)
                L.oracle_trajectory_evaluate_station(_p(rows), n, float(st), _p(g))
                REF.ref_trajectory_evaluate_station(_p(rows), n, float(st), _p(r))
                assert _same_bits(g, r), (n, st)
                for lat in (-1.5, 0.0, 0.7):
                    g2, r2 = np.full(2, np.nan), np.full(2, np.nan)
                    L.oracle_trajectory_cartesian(_p(rows), n, float(st), lat, _p(g2))
                    REF.ref_trajectory_cartesian(_p(rows), n, float(st), lat, _p(r2))
                    assert _same_bits(g2, r2), (n, st, lat)
            i = rng.integers(0, n, 40)
            pts = rows[i, 2:4] + rng.normal(0, 2.0, (40, 2))
            pts = np.concatenate([pts, rows[:: max(1, n // 5), 2:4], rows[:1, 2:4] - 10.0, rows[-1:, 2:4] + 10.0])
            for p in pts:
                g2, r2, r9 = np.full(2, np.nan), np.full(2, np.nan), np.full(9, np.nan)
                L.oracle_trajectory_projection(_p(rows), n, float(p[0]), float(p[1]), _p(g2))
                REF.ref_trajectory_projection(_p(rows), n, float(p[0]), float(p[1]), _p(r2), _p(r9))
                assert _same_bits(g2, r2), (n, p)


def test_tracker_trajectory_queries_equal_reference():
    """The tracker's restatement of EvaluateTime and of the projected point of GetProjection
    (discretized_trajectory.cpp:130-141, 165-197; tracker.cc reads time, s, x, y, theta, kappa, velocity of them).
    The oracle's interpolation carries no lane bounds, so the first seven fields are compared."""
    L = orc.lib()
    rng = np.random.default_rng(19)
    for n in (2, 5, 41, 200):
        for _ in range(6):
            rows = _trajectory(rng, n)
            tmax = rows[-1, 0]
            times = np.concatenate([rng.uniform(-0.3, tmax + 0.3, 40), rows[:: max(1, n // 6), 0], [0.0, tmax, -1.0, tmax + 1.0]])
            for t in times:
                g, r = np.full(9, np.nan), np.full(9, np.nan)
                L.oracle_tracker_evaluate_time(_p(rows), n, float(t), _p(g))
                REF.ref_trajectory_evaluate_time(_p(rows), n, float(t), _p(r))
                assert _same_bits(g[:7], r[:7]), (n, t, g, r)
            i = rng.integers(0, n, 40)
            pts = np.concatenate([rows[i, 2:4] + rng.normal(0, 1.5, (40, 2)), rows[:: max(1, n // 5), 2:4]])
            for p in pts:
                g, r2, r = np.full(9, np.nan), np.full(2, np.nan), np.full(9, np.nan)
                L.oracle_tracker_projection(_p(rows), n, float(p[0]), float(p[1]), _p(g))
                REF.ref_trajectory_projection(_p(rows), n, float(p[0]), float(p[1]), _p(r2), _p(r))
                assert _same_bits(g[:7], r[:7]), (n, p, g, r)


# ------------------------------------------------------- the Eigen part of the reference (absent in this image)
REF_EIGEN = orc.ref_eigen_lib()
needs_eigen = pytest.mark.skipif(REF_EIGEN is None, reason="no Eigen in this image: the reference's vehicle_model.cc / "
                                 "barrier_function.h are not built (make -C oracle ref_eigen)")


@needs_eigen
def test_dynamics_and_jacobian_oracle_equal_reference():
    """VehicleModel::Dynamics / DynamicsJacbian (vehicle_model.cc:20-124) against the oracle's restatement, bit for
    bit: states over the whole box of the barriers and beyond it, headings outside [-pi, pi) (NormalizeAngle acts)."""
    cfg = orc.default_config(50)
    o = orc.Oracle(cfg)
    m = C.c_void_p(REF_EIGEN.ref_model_create(cfg.wheel_base, cfg.n_steps * cfg.dt, cfg.dt))
    rng = np.random.default_rng(29)
    for _ in range(20000):
        x = np.array([rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-7, 7), rng.uniform(-2, 25),
                      rng.uniform(-6, 6), rng.uniform(-0.9, 0.9)])
        u = np.array([rng.uniform(-12, 12), rng.uniform(-0.4, 0.4)])
        xn, A, B = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 2))
        REF_EIGEN.ref_dynamics(m, _p(x), _p(u), _p(xn))
        REF_EIGEN.ref_dynamics_jacobian(m, _p(x), _p(u), _p(A), _p(B))
        assert _same_bits(o.dynamics(x, u), xn), (x, u)
        ga, gb = o.dynamics_jacobian(x, u)
        assert _same_bits(ga, A) and _same_bits(gb, B), (x, u)
    REF_EIGEN.ref_model_destroy(m)


@needs_eigen
def test_barrier_oracle_equals_reference():
    """RelaxBarrierFunction<N> value / Jacbian / Hessian (barrier_function.h:80-146) on both branches and at the
    switch point, with and without the second-derivative term, N = 6 and N = 2."""
    cfg = orc.default_config(50)
    o = orc.Oracle(cfg)
    t, eps = cfg.barrier_t, cfg.barrier_eps
    rng = np.random.default_rng(31)
    gs = np.concatenate([rng.uniform(-30, 3, 4000), -eps + rng.uniform(-1e-6, 1e-6, 500), [-eps, 0.0, -1e-300, np.nextafter(-eps, -1.0)]])
    for g in gs:
        g = float(g)
        assert _same_bits([o.barrier_value(g)], [REF_EIGEN.ref_barrier_value(t, eps, g)]), g
        for n in (6, 2):
            dg = rng.uniform(-3, 3, n)
            ddg = rng.uniform(-2, 2, (n, n))
            ddg = np.ascontiguousarray(ddg + ddg.T)
            rj, rh, rh2 = np.zeros(n), np.zeros((n, n)), np.zeros((n, n))
            REF_EIGEN.ref_barrier_jacobian(t, eps, g, _p(dg), n, _p(rj))
            REF_EIGEN.ref_barrier_hessian(t, eps, g, _p(dg), None, n, _p(rh))
            REF_EIGEN.ref_barrier_hessian(t, eps, g, _p(dg), _p(ddg), n, _p(rh2))
            assert _same_bits(o.barrier_jacobian(g, dg), rj), (g, n)
            assert _same_bits(o.barrier_hessian(g, dg), rh), (g, n)
            assert _same_bits(o.barrier_hessian(g, dg, ddg), rh2), (g, n)


# ------------------------------------------------------------------ configuration: the reference's own structs
def _ref_default(key):
    v = C.c_double()
    assert REF.ref_default(key.encode(), C.byref(v)) == 1, f"oracle/ref_shim.cc does not know {key}"
    return v.value


# cilqr_config field -> member path in the reference's structs (planner_config.h:45-73, vehicle_param.h:26-64)
_ILQR_KEYS = dict(num_of_disc="ilqr.num_of_disc", max_iter="ilqr.max_iter_num", safe_margin="ilqr.safe_margin",
                  w_jerk="ilqr.weights.jerk", w_delta_rate="ilqr.weights.delta_rate", w_x="ilqr.weights.x_target",
                  w_y="ilqr.weights.y_target", w_theta="ilqr.weights.theta", w_v="ilqr.weights.v", w_a="ilqr.weights.a",
                  w_delta="ilqr.weights.delta", abs_cost_tol="ilqr.abs_cost_tol", rel_cost_tol="ilqr.rel_cost_tol",
                  front_hang="vehicle.front_hang_length", wheel_base="vehicle.wheel_base", rear_hang="vehicle.rear_hang_length",
                  width="vehicle.width", max_velocity="vehicle.max_velocity", min_acceleration="vehicle.min_acceleration",
                  max_acceleration="vehicle.max_acceleration", jerk_min="vehicle.jerk_min", jerk_max="vehicle.jerk_max",
                  delta_min="vehicle.delta_min", delta_max="vehicle.delta_max", delta_rate_min="vehicle.delta_rate_min",
                  delta_rate_max="vehicle.delta_rate_max", dt="planner.delta_t")


def test_default_configurations_equal_the_references_own_structs():
    """Every default the C-ABI hands out (cilqr_default_config / _corridor_config / _tracker_config / _dp_config) and
    every default the oracle starts from, against the reference's OWN IlqrConfig, Weights, VehicleParam, CorridorConfig,
    TrackerConfig and PlannerConfig -- instantiated inside oracle/_ref from algorithm/params/planner_config.h:18-188 and
    vehicle_param.h:21-95 (default member initialisers and VehicleParam's constructor), field by field, bit for bit.
    The relaxed barrier's t = 5, epsilon = 0.01 (barrier_function.h:144-145) need Eigen to instantiate and stay literals."""
    from cilqr_amd import api
    # the solver: product and oracle
    tf, dt = _ref_default("planner.tf"), _ref_default("planner.delta_t")
    n_knots = int(np.floor(tf / dt + 1))                                  # ilqr_optimizer.cc:22 with the reference's horizon
    assert n_knots == 81
    c, o = api.default_config(n_knots - 1), orc.default_config(n_knots - 1)
    for field, key in _ILQR_KEYS.items():
        ref = _ref_default(key)
        assert _same_bits(float(getattr(c, field)), ref), (field, key, getattr(c, field), ref)
        assert _same_bits(float(getattr(o, field)), ref), ("oracle", field, key, getattr(o, field), ref)
    assert set(_ILQR_KEYS) | {"n_steps", "init_guess", "barrier_t", "barrier_eps"} == {f for f, _ in api.Config._fields_}
    assert c.init_guess == api.INIT_IQR and (c.barrier_t, c.barrier_eps) == (5.0, 0.01) == (o.barrier_t, o.barrier_eps)
    # the corridor producer
    cc = api.default_corridor_config()
    for field in ("max_diff_x", "max_diff_y", "radius", "max_axis_x", "max_axis_y", "lane_segment_length", "is_multiple_sample"):
        assert _same_bits(float(getattr(cc, field)), _ref_default("corridor." + field)), field
    assert {f for f, _ in api.CorridorConfig._fields_} == {"max_diff_x", "max_diff_y", "radius", "max_axis_x", "max_axis_y",
                                                            "lane_segment_length", "is_multiple_sample", "reserved0"}
    for v, field in zip(orc.CORRIDOR_CFG, ("max_diff_x", "max_diff_y", "radius", "max_axis_x", "max_axis_y", "is_multiple_sample")):
        assert _same_bits(float(v), _ref_default("corridor." + field)), ("oracle", field)
    # the tracker: PlannerConfig::tracker_config and IlqrConfig::tracker_config carry the same defaults
    tc = api.TrackerConfig()
    api.lib().cilqr_default_tracker_config(C.byref(tc))
    names = dict(weight_l="lateral.weight_l", weight_theta="lateral.weight_theta", weight_delta="lateral.weight_delta",
                 weight_delta_rate="lateral.weight_delta_rate", preview_time="lateral.preview_time",
                 weight_s="longitudinal.weight_s", weight_v="longitudinal.weight_v", weight_a="longitudinal.weight_a",
                 weight_j="longitudinal.weight_j", sumulation_dt="sumulation_dt", dt="dt", tolerance="tolerance",
                 max_num_iteration="max_num_iteration")
    ocfg = dict(zip(orc.TRACKER_CFG_FIELDS, orc.TRACKER_CFG_DEFAULT))
    for field, key in names.items():
        for prefix in ("tracker.", "ilqr.tracker."):
            ref = _ref_default(prefix + key)
            assert _same_bits(float(getattr(tc, field)), ref), (field, prefix)
            assert _same_bits(float(ocfg[field]), ref), ("oracle", field, prefix)
    assert _ref_default("tracker.longitudinal.preview_time") == 0.0        # carried by neither: the reference never reads a non-zero value
    for field in ("wheel_base", "delta_min", "delta_max", "delta_rate_min", "delta_rate_max", "jerk_min", "jerk_max",
                  "min_acceleration", "max_acceleration"):
        assert _same_bits(float(ocfg[field]), _ref_default("vehicle." + field)), ("oracle tracker", field)
    # the DP coarse planner
    dc = api.default_dp_config()
    odp = dict(zip(orc.DP_CFG_FIELDS, orc.DP_CFG_DEFAULT))
    for field, _ in api.DpConfig._fields_:
        key = ("vehicle." if field in ("front_hang_length", "wheel_base", "rear_hang_length", "width", "max_velocity") else "planner.") + field
        ref = _ref_default(key)
        assert _same_bits(float(getattr(dc, field)), ref), (field, key)
        assert _same_bits(float(odp[field]), ref), ("oracle", field, key)
    # VehicleParam's constructor (vehicle_param.h:83-88): the collision discs of the DP
    d3 = np.zeros(3)
    orc.lib().oracle_vehicle_derived(_p(d3))
    assert _same_bits(d3, [_ref_default("vehicle.radius"), _ref_default("vehicle.f2x"), _ref_default("vehicle.r2x")])


def test_scene_generator_uses_the_references_vehicle_and_horizon():
    """cilqr_amd/scenario.py (test / bench infrastructure) shrinks nothing itself, but its families quote the reference's
    step and the demo horizon: dt and tf of PlannerConfig."""
    sc = scenario.generate("demo80", 1, seed=1)
    assert sc["n_steps"] + 1 == int(np.floor(_ref_default("planner.tf") / _ref_default("planner.delta_t") + 1))


# ------------------------------------------------------------------ header-only helpers: slerp, LinSpaced, Pose, discs
def test_slerp_equals_reference():
    """math::slerp (math_utils.h:208-225), as the DP oracle and the tracker oracle restate it; the degenerate branch
    (|t1 - t0| <= 1e-10), the +-pi wrap of the difference and arguments outside [t0, t1] included."""
    L = orc.lib()
    rng = np.random.default_rng(31)
    n = 40000
    a0, a1 = rng.uniform(-7, 7, n), rng.uniform(-7, 7, n)
    a1[::5] = a0[::5] + rng.choice([np.pi, -np.pi, 2 * np.pi, 0.0], a0[::5].size) + rng.uniform(-1e-12, 1e-12, a0[::5].size)
    t0 = rng.uniform(0, 50, n)
    t1 = t0 + rng.uniform(0.01, 2.0, n)
    t1[::11] = t0[::11] + rng.uniform(-2e-10, 2e-10, t0[::11].size)
    t1[::13] = t0[::13]
    t = t0 + rng.uniform(-0.3, 1.3, n) * (t1 - t0)
    ref = np.array([REF.ref_slerp(*v) for v in zip(a0, t0, a1, t1, t)])
    assert _same_bits(np.array([L.oracle_slerp(*v) for v in zip(a0, t0, a1, t1, t)]), ref)
    assert _same_bits(np.array([L.oracle_tracker_slerp(*v) for v in zip(a0, t0, a1, t1, t)]), ref)


def test_lin_spaced_equals_reference():
    """math::LinSpaced<N> (math_utils.h:245-254) at the three sizes of DpPlanner's constructor (dp_planner.cpp:31-33), on the
    reference's own arguments (tf / NT .. tf; 0 .. unit_time * max_velocity; 0 .. 1) and on random ones."""
    L = orc.lib()
    tf, vmax = _ref_default("planner.tf"), _ref_default("vehicle.max_velocity")
    rng = np.random.default_rng(5)
    cases = [(5, tf / 5, tf), (7, 0.0, tf / 5 * vmax), (9, 0.0, 1.0)]
    cases += [(int(n), float(a), float(b)) for n, a, b in zip(rng.choice([5, 7, 9], 3000), rng.uniform(-30, 30, 3000), rng.uniform(-30, 200, 3000))]
    for n, a, b in cases:
        r, o = np.zeros(n), np.zeros(n)
        assert REF.ref_lin_spaced(n, a, b, _p(r)) == n and L.oracle_lin_spaced(n, a, b, _p(o)) == n
        assert _same_bits(o, r), (n, a, b)


def test_pose_transform_equals_reference():
    """Pose::transform (pose.h:40-46): body-frame polygon vertices placed along a dynamic obstacle's trajectory
    (planning_node.cc:63-80) -- the oracle's scene builder, and the product's (cilqr_amd/scene_io.py uses numpy on the
    same expression; include/cilqr/scene_file.hpp and dp_planner.hpp the C++ one)."""
    L = orc.lib()
    rng = np.random.default_rng(37)
    n = 20000
    args = np.stack([rng.uniform(-200, 200, n), rng.uniform(-200, 200, n), rng.uniform(-7, 7, n), rng.uniform(-3, 3, n),
                     rng.uniform(-3, 3, n), np.zeros(n)], axis=1)
    r, o = np.zeros(3), np.zeros(3)
    for a in args:
        REF.ref_pose_transform(*a, _p(r))
        L.oracle_pose_transform(*a, _p(o))
        assert _same_bits(o, r), a
        x, y, th, rx, ry, _ = a
        assert _same_bits(np.array([x + rx * np.cos(th) - ry * np.sin(th), y + rx * np.sin(th) + ry * np.cos(th)]), r[:2]), a


def test_collision_boxes_of_the_dp_equal_reference():
    """Environment::CheckOptimizationCollision (environment.cpp:92-104) builds two boxes per pose from
    VehicleParam::GetDiscPositions (with its swapped result names), AABox2d::Shift and Box2d(AABox2d); the oracle's
    restatement against the same statements run on the reference's own classes: disc centres, box centres, half extents,
    min / max and corners, bit for bit."""
    L = orc.lib()
    rng = np.random.default_rng(41)
    r, o = np.zeros(36), np.zeros(36)
    for x, y, th in zip(rng.uniform(-300, 300, 20000), rng.uniform(-300, 300, 20000), rng.uniform(-7, 7, 20000)):
        REF.ref_collision_boxes(x, y, th, 0.0, _p(r))
        L.oracle_collision_boxes(x, y, th, 0.0, _p(o))
        assert _same_bits(o, r), (x, y, th)


# ------------------------------------------------------------------------------------------------- the device
def _opt(sc):
    from cilqr_amd import api
    cfg = api.default_config(sc["n_steps"])
    return api, api.BatchIlqrOptimizer(cfg, batch_capacity=sc["start"].shape[0], cmax=sc["cmax"])


@pytest.mark.gpu
def test_device_normalize_angle_equals_reference():
    sc = scenario.generate("ped6", 4, seed=3)
    api, opt = _opt(sc)
    x = _angles()
    got = opt.device_math(6, x)
    ref = np.array([REF.ref_normalize_angle(v) for v in x])
    assert _same_bits(got, ref), (x[got != ref][:5], got[got != ref][:5], ref[got != ref][:5])
    opt.close()


@pytest.mark.gpu
def test_device_nearest_lane_equals_reference_with_exact_ties():
    """CILQR_OPT_EXACT_LANE_TIES (default 1): the device's nearest lane segment -- grid search and full scan -- is the one the
    reference's own LineSegment2d::DistanceTo loop picks, on the tie strips too."""
    sc = scenario.generate("ped6", 4, seed=3)
    api, opt = _opt(sc)          # nothing set: the reference's tie rule is the library's default
    opt.stage_load(sc)
    pts = _tie_points(sc)
    gl, gr = opt.nearest_lane(pts, use_grid=True)
    sl, sr = opt.nearest_lane(pts, use_grid=False)
    for tab, grid, scan in ((sc["left"], gl, sl), (sc["right"], gr, sr)):
        ref = _ref_nearest(tab, pts)
        bad = np.nonzero((grid != ref) | (scan != ref))[0]
        assert bad.size == 0, (bad[:5], pts[bad[:5]], grid[bad[:5]], scan[bad[:5]], ref[bad[:5]])
    opt.close()


@pytest.mark.gpu
def test_device_segment_distance_equals_reference():
    """device_math fn 9 is hypot_ref(x, 1) -- the glibc hypot kernel LineSegment2d::DistanceTo's end-point branches
    come down to; against the reference's DistanceTo from a degenerate segment at the origin to (x, 1)."""
    sc = scenario.generate("ped6", 4, seed=3)
    api, opt = _opt(sc)
    rng = np.random.default_rng(23)
    x = np.concatenate([rng.uniform(-50, 50, 20000), rng.uniform(-1e-3, 1e-3, 5000), 10.0 ** rng.uniform(-12, 12, 5000)])
    got = opt.device_math(9, x)
    seg = np.zeros(4)
    ref = np.array([REF.ref_segment_distance(_p(seg), float(v), 1.0) for v in x])
    assert _same_bits(got, ref), (x[got != ref][:5], got[got != ref][:5], ref[got != ref][:5])
    opt.close()
