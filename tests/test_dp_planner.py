"""The coarse-trajectory producer: DpPlanner::Plan + ComputePathProfile, host side.

The product's planner (include/cilqr/dp_planner.hpp behind cilqr_dp_plan) against the line-by-line restatement of the
reference in oracle/dp_oracle.cc -- bit for bit: the DP compares costs with '<', so any rounding difference could
pick another cell -- plus properties that hold for any correct planner, and the TrajectoryPlanner-shaped C++ pipeline
(DP -> corridor -> CILQR) on the GPU."""
import dataclasses
import os
import subprocess

import numpy as np
import pytest

from cilqr_amd import api, scenario, scene_io
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module", autouse=True)
def _build(built):
    return built


def _scenes(family, n, seed, keep_all=True):
    spec = scenario.SPECS[family]
    if keep_all:
        spec = dataclasses.replace(spec, min_clearance=-1.0)
    sc = scenario.generate(spec, n, seed=seed, scenarios=True)
    return sc, scene_io.from_generator(sc)


@pytest.mark.parametrize("family,tf,seed", [("demo80", 8.0, 5), ("mix11", 5.0, 6), ("dyn20", 10.0, 7)])
def test_product_planner_equals_the_line_by_line_oracle(family, tf, seed):
    sc, sf = _scenes(family, 6, seed)
    cfg = api.default_dp_config(tf=tf)
    n_found = 0
    for b in range(6):
        flat = scene_io.flatten_scene(sf.center, sf.scenes[b])
        ok, co = api.dp_plan(flat, sc["start"][b, :3], cfg)
        ok2, co2 = orc.dp_plan(flat, sc["start"][b, :3], tf=tf)
        # a plan that stands still for a whole layer (station step 0) has 0 / 0 curvature in the reference too
        assert ok == ok2 and np.array_equal(co, co2, equal_nan=True), b
        assert co.shape == (int(tf / 0.1 + 1), 9)
        n_found += ok
    assert n_found >= 4


def test_non_default_weights_and_vehicle_reach_the_planner():
    sc, sf = _scenes("demo80", 3, 11)
    over = dict(dp_nominal_velocity=6.0, dp_w_lateral=0.8, dp_w_lateral_change=0.1, dp_w_longitudinal_velocity_change=3.0,
                width=2.3, wheel_base=1.6, max_velocity=14.0, dp_w_obstacle=500.0)
    for b in range(3):
        flat = scene_io.flatten_scene(sf.center, sf.scenes[b])
        ok, co = api.dp_plan(flat, sc["start"][b, :3], api.default_dp_config(**over))
        ok2, co2 = orc.dp_plan(flat, sc["start"][b, :3], **over)
        ok3, co3 = api.dp_plan(flat, sc["start"][b, :3])
        assert ok == ok2 and np.array_equal(co, co2)
        assert not np.array_equal(co, co3)
        assert abs(np.median(co[:, 6]) - 6.0) < abs(np.median(co3[:, 6]) - 6.0)      # slower nominal speed is followed


def test_paths_are_well_formed_and_keep_clear_of_the_obstacles():
    """Properties of any found plan: K knots on the time grid, stations never decrease, the two collision discs of
    vehicle_param.h:76-95 (squares of half side `radius`) hold no obstacle corner at any knot, and the profile columns
    are the finite differences ComputePathProfile defines (discrete_points_math.cc:27-176)."""
    sc, sf = _scenes("demo80", 8, 21)
    length = 0.96 + 1.0 + 0.929
    radius, r2x, f2x = np.hypot(0.25 * length, 0.5 * 1.942), 0.25 * length - 0.929, 0.75 * length - 0.929
    checked = 0
    for b in range(8):
        flat = scene_io.flatten_scene(sf.center, sf.scenes[b])
        ok, co = api.dp_plan(flat, sc["start"][b, :3])
        if not ok:
            continue
        checked += 1
        t, s, x, y, th, kap, v, a, dl = co.T
        assert np.allclose(t, np.arange(81) * 0.1) and np.all(np.diff(s) >= -1e-12)
        pts, cnt = scene_io.environment_points(sf.scenes[b], t)
        for i in range(81):
            p = pts[i, :cnt[i]]
            for off in (r2x, f2x):
                cx, cy = x[i] + off * np.cos(th[i]), y[i] + off * np.sin(th[i])
                inside = (np.abs(p[:, 0] - cx) <= radius) & (np.abs(p[:, 1] - cy) <= radius)
                assert not inside.any(), (b, i)
        seg = np.hypot(np.diff(x), np.diff(y))
        acc = np.concatenate([[0.0], np.cumsum(seg)])
        vv = np.diff(acc) / 0.1
        assert np.allclose(v[:-1], vv, rtol=0, atol=1e-9) and v[-1] == v[-2]
        assert np.allclose(a[:-2], np.diff(v[:-1]) / 0.1, atol=1e-7) and a[-1] == a[-2]
        assert np.allclose(dl, np.arctan(kap * 1.0)) and np.abs(kap).max() < 1.0
        assert abs(co[0, 2] - sc["start"][b, 0]) < 0.2 and abs(co[0, 3] - sc["start"][b, 1]) < 0.2   # starts at the ego
    assert checked >= 5


def test_blocked_road_is_reported_as_dp_failed():
    """A wall across the whole road where the ego stands (a wall further ahead is no failure: standing still for all
    five layers is one of the sampled plans): every path collides at its first point, the planner says so (the
    reference's "DP failed", trajectory_planner.cpp:32-35) and still fills the trajectory."""
    sc, sf = _scenes("demo80", 1, 31)
    scene = sf.scenes[0]
    road = sc["road"]
    s0 = 0.5     # demo80 scenes start at station 0.5
    x0, y0, th, _ = road.eval(np.array([s0]))
    c, s_ = np.cos(th[0]), np.sin(th[0])
    wall = np.array([[1.0, 9.0], [1.0, -9.0], [-1.0, -9.0], [-1.0, 9.0]])
    scene.static.append(np.stack([x0[0] + wall[:, 0] * c - wall[:, 1] * s_, y0[0] + wall[:, 0] * s_ + wall[:, 1] * c], 1))
    flat = scene_io.flatten_scene(sf.center, scene)
    ok, co = api.dp_plan(flat, sc["start"][0, :3])
    ok2, co2 = orc.dp_plan(flat, sc["start"][0, :3])
    assert not ok and not ok2 and np.array_equal(co, co2, equal_nan=True)
    assert np.isfinite(co[:, :5]).all()


def test_argument_errors():
    sc, sf = _scenes("demo80", 1, 41)
    flat = scene_io.flatten_scene(sf.center, sf.scenes[0])
    with pytest.raises(api.CilqrError) as e:
        api.dp_plan(flat, sc["start"][0, :3], api.default_dp_config(tf=-1.0))
    assert e.value.code == api.ERR_ARG
    L = api.lib()
    cfg = api.default_dp_config()
    out = np.zeros((81, 9))
    assert L.cilqr_dp_plan(None, None, None, out.ctypes.data, 81) == api.ERR_NULL
    keep = {k: np.ascontiguousarray(v) for k, v in flat.items()}
    scs = api.SceneStruct(keep["center"].ctypes.data, keep["center"].shape[0], 0, None, None, 0, 0, None, None, None, None)
    st = np.ascontiguousarray(sc["start"][0, :3])
    assert L.cilqr_dp_plan(cfg, scs, st.ctypes.data, out.ctypes.data, 51) == api.ERR_KNOTS      # 8 s / 0.1 s is 81 knots
    assert L.cilqr_dp_plan(cfg, scs, st.ctypes.data, out.ctypes.data, 81) == api.OK            # an empty road


def test_road_barriers_follow_the_centre_line():
    """cilqr_road_barriers = Environment::set_reference: one point per 0.1 m of station on either side, at the bound's
    distance from the centre line; on a centre line that is itself sampled at 0.1 m it agrees with shifting the centre
    points (cilqr_amd.scene_io.road_barriers) to rounding."""
    sc, sf = _scenes("demo80", 1, 3)
    left, right = api.road_barriers(sf.center)
    assert left.shape == right.shape == (int((sf.center[-1, 0] - sf.center[0, 0]) / 0.1) + 1, 2)
    l2, r2 = scene_io.road_barriers(sf.center)
    assert np.abs(left - l2[:len(left)]).max() < 1e-9 and np.abs(right - r2[:len(right)]).max() < 1e-9
    d = np.hypot(left[:, 0] - right[:, 0], left[:, 1] - right[:, 1])
    assert np.allclose(d, scenario.LEFT_BOUND + scenario.RIGHT_BOUND, atol=1e-9)


def test_generator_with_dp_coarse_trajectories():
    g = scenario.generate_dp("demo80", 6, seed=51, workers=4)
    assert g["coarse"].shape == (6, 81, 6) and g["found"].dtype == bool and g["found"].sum() >= 4
    assert np.array_equal(g["coarse"][:, :, 0], g["dp"][:, :, 2]) and np.array_equal(g["coarse"][:, :, 5], g["dp"][:, :, 8])
    # kinks: the piecewise-linear (s, l) path bends at the layer boundaries, which the generator's own smooth pick never does
    kappa = g["dp"][g["found"]][:, :, 5]
    assert np.abs(np.diff(kappa, axis=1)).max() > 0.02


def build_planner_test(tmp_path, types="stand-ins"):
    from test_host import build_cpp_test
    return build_cpp_test("planner_test", tmp_path, types)


@pytest.mark.parametrize("types", ["stand-ins", "reference headers"])
def test_trajectory_planner_adapter_compiles_as_cxx14_against_the_c_abi(tmp_path, types):
    """include/cilqr/trajectory_planner.hpp + dp_planner.hpp (the planning::DpPlanner / TrajectoryPlanner call
    surfaces) build with C++14 / g++ and link against the C-ABI only -- against stand-ins of the reference's types and
    against its own headers (PlannerConfig, StartState, TrajectoryPoint, DiscretizedTrajectory, Polygon2d ...)."""
    assert build_planner_test(tmp_path, types).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("types", ["stand-ins", "reference headers"])
def test_cpp_trajectory_planner_pipeline_matches_the_c_abi_stage_by_stage(tmp_path, types):
    """planning::TrajectoryPlanner-shaped C++ pipeline (DP -> Corridor -> IlqrOptimizer, include/cilqr/*.hpp) on scenes
    read from a .cqs file, against the same three stages driven one by one through the C-ABI from Python."""
    exe = build_planner_test(tmp_path, types)
    g = scenario.generate_dp("demo80", 4, seed=61, workers=4)
    path = tmp_path / "scenes.cqs"
    scene_io.save(str(path), g["scene_file"])
    sf = g["scene_file"]
    K = 81
    left_b, right_b = None, None
    n_ok = 0
    for b in range(4):
        out = tmp_path / f"out{b}.bin"
        r = subprocess.run([str(exe), str(path), str(b), "8.0", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = open(out, "rb").read()
        ok, K_, n_cost, n_it = np.frombuffer(raw[:16], np.int32)
        assert K_ == K
        body = np.frombuffer(raw[16:16 + (K * 9 + K * 11 + n_cost * 5) * 8], np.float64)
        coarse = body[:K * 9].reshape(K, 9)
        res = body[K * 9:K * 20].reshape(K, 11)
        cost = body[K * 20:].reshape(n_cost, 5)
        counts = np.frombuffer(raw[16 + body.size * 8:16 + body.size * 8 + K * 4], np.int32)
        if not g["found"][b]:      # "DP failed", or a plan that stands still (0 / 0 curvature): nothing to compare
            assert not ok or not np.isfinite(g["dp"][b]).all()
            continue
        assert ok
        n_ok += 1
        assert np.array_equal(coarse, g["dp"][b])                                  # stage 1: the DP
        # stage 2 + 3 through the C-ABI: corridors from the scene's points at the knot times, lanes, solve
        times = g["dp"][b, :, 0]
        pts, cnt = scene_io.environment_points(sf.scenes[b], times)
        opt = api.BatchIlqrOptimizer(n_steps=K - 1, batch_capacity=1, cmax=64, max_lane_segments=128)
        knots = np.ascontiguousarray(g["dp"][b][None, :, [2, 3, 4]])
        cor, ccnt, nf = opt.build_corridors(knots, pts[None], cnt[None], cmax=64)
        assert nf == 0 and np.array_equal(ccnt[0], counts)
        lb, rb = api.road_barriers(sf.center)
        left, right = api.lane_constraints(lb, 5.0, True), api.lane_constraints(rb, 5.0, False)
        sc1 = dict(start=g["start"][b:b + 1], coarse=g["coarse"][b:b + 1], corridor=cor, ccount=ccnt, left=left, right=right)
        p = opt.plan(sc1)
        opt.close()
        assert n_cost == p["n_cost"][0] and np.array_equal(cost, p["cost_hist"][0, :n_cost])
        tr = p["traj"][0]
        # result columns: time s x y theta kappa velocity a jerk delta delta_rate (trajectory_planner.cpp:101-125)
        assert np.array_equal(res[:, [0, 2, 3, 4, 6, 7, 9]], tr[:, [0, 1, 2, 3, 4, 5, 6]])
        assert np.array_equal(res[:-1, [8, 10]], tr[:-1, [8, 9]])
        assert np.allclose(res[:, 5], np.tan(tr[:, 6]) / 1.0, rtol=1e-15)
        assert np.allclose(res[:, 1], np.concatenate([[0], np.cumsum(np.hypot(np.diff(tr[:, 1]), np.diff(tr[:, 2])))]), rtol=1e-13)
    assert n_ok >= 2
