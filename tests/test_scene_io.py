"""Scene wire format (cilqr_amd/scene_io.py): round trip, and the restated Environment
queries against what the generator computes directly."""
import os

import numpy as np
import pytest

from cilqr_amd import api, scenario, scene_io
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _scene_file(B=6, seed=61, family="mix11"):
    sc = scenario.generate(family, B, seed=seed, obstacle_points=True, scenarios=True)
    return sc, scene_io.from_generator(sc)


def test_round_trip_is_exact(tmp_path):
    sc, f = _scene_file()
    path = tmp_path / "scenes.cqs"
    scene_io.save(str(path), f)
    g = scene_io.load(str(path))
    assert g.dt == f.dt and np.array_equal(g.center, f.center) and len(g.scenes) == len(f.scenes)
    for a, b in zip(f.scenes, g.scenes):
        assert np.array_equal(a.start, b.start) and np.array_equal(a.coarse, b.coarse)
        assert len(a.static) == len(b.static) and len(a.dynamic) == len(b.dynamic)
        assert all(np.array_equal(p, q) for p, q in zip(a.static, b.static))
        for d, e in zip(a.dynamic, b.dynamic):
            assert np.array_equal(d.polygon, e.polygon) and np.array_equal(d.trajectory, e.trajectory)
    raw = path.read_bytes()
    assert raw[:8] == b"CILQRSC1"
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.cqs"
        bad.write_bytes(b"NOTSCENE" + raw[8:])
        scene_io.load(str(bad))
    with pytest.raises(ValueError):
        bad.write_bytes(raw + b"\0")
        scene_io.load(str(bad))


def test_environment_queries_reproduce_the_generator(tmp_path):
    """Obstacle points per knot time from the file (static polygons + dynamic obstacles placed by the
    trajectory sample std::upper_bound picks) are the generator's own per-knot points; the road
    barriers give the generator's lane tables through LaneBoundarySample / HalfPlaneConstraint."""
    sc, f = _scene_file(B=8, seed=62)
    path = tmp_path / "scenes.cqs"
    scene_io.save(str(path), f)
    g = scene_io.load(str(path))
    K = sc["coarse"].shape[1]
    times = np.arange(K) * g.dt
    for b, scene in enumerate(g.scenes):
        pts, cnt = scene_io.environment_points(scene, times)
        assert np.array_equal(cnt, sc["obstacle_count"][b])
        for k in range(K):
            mine = pts[k, :cnt[k]]
            ref = sc["obstacle_points"][b, k, :cnt[k]]
            order_a = np.lexsort((mine[:, 1], mine[:, 0]))
            order_b = np.lexsort((ref[:, 1], ref[:, 0]))
            assert np.allclose(mine[order_a], ref[order_b], rtol=0, atol=1e-12)
    left_b, right_b = scene_io.road_barriers(g.center)
    assert np.allclose(orc.lane_constraints(left_b, 5.0, True), sc["left"], rtol=1e-12, atol=1e-12)
    assert np.allclose(orc.lane_constraints(right_b, 5.0, False), sc["right"], rtol=1e-12, atol=1e-12)


def test_pedestrians_enter_and_leave():
    """A dynamic obstacle only exists inside its trajectory's time span (environment.cpp:139-142)."""
    poly = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, -0.5], [-0.5, 0.5]])
    traj = np.array([[1.0, 10.0, 0.0, 0.0], [1.1, 10.0, 0.1, 0.0], [1.2, 10.0, 0.2, 0.0]])
    s = scene_io.Scene(np.zeros(4), np.zeros((1, 6)), [], [scene_io.DynamicObstacle(poly, traj)])
    pts, cnt = scene_io.environment_points(s, [0.9, 1.0, 1.05, 1.2, 1.3])
    assert cnt.tolist() == [0, 4, 4, 4, 0]
    assert np.allclose(pts[1, :4].mean(0), [10.0, 0.0])      # t = 1.0: the sample at 1.0
    assert np.allclose(pts[2, :4].mean(0), [10.0, 0.1])      # t = 1.05: the first sample later than t
    assert np.allclose(pts[3, :4].mean(0), [10.0, 0.2])


def test_committed_scene_file_is_stable():
    """tests/golden/scenes_mix11_4.cqs pins the format: same bytes, same content, and the generator still
    produces it (a change of either side shows up here)."""
    import hashlib
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(here, "scenes_mix11_4.json")))
    raw = open(os.path.join(here, "scenes_mix11_4.cqs"), "rb").read()
    assert len(raw) == meta["bytes"] and hashlib.sha256(raw).hexdigest() == meta["sha256"]
    g = scene_io.load(os.path.join(here, "scenes_mix11_4.cqs"))
    assert len(g.scenes) == meta["scenes"] and g.dt == meta["dt"]
    assert [len(s.static) for s in g.scenes] == meta["n_static"]
    assert [len(s.dynamic) for s in g.scenes] == meta["n_dynamic"]
    K = meta["knots"]
    for s, want in zip(g.scenes, meta["points_per_knot"]):
        assert scene_io.environment_points(s, np.arange(K) * g.dt)[1].tolist() == want
    sc = scenario.generate("mix11", 4, seed=71, scenarios=True)
    f = scene_io.from_generator(sc)
    assert all(np.array_equal(a.coarse, b.coarse) and np.array_equal(a.start, b.start)
               for a, b in zip(f.scenes, g.scenes))


def test_cpp_reader_agrees_with_the_python_loader(tmp_path):
    """include/cilqr/scene_file.hpp (header-only C++14) reads the same file: counts, checksums of every
    array, the obstacle points of every knot time (its ObstaclePoints restates the same Environment
    queries), the road barriers."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "scene_file_test"
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "scene_file_test.cc"), "-o", str(exe)])
    path = os.path.join(root, "tests", "golden", "scenes_mix11_4.cqs")
    out = subprocess.run([str(exe), path], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    g = scene_io.load(path)
    n, nc, dt = out[0].split()
    assert int(n) == len(g.scenes) and int(nc) == len(g.center) and float(dt) == g.dt
    assert float(out[1]) == pytest.approx(float(g.center.sum()), rel=1e-13)
    for line, sc in zip(out[2:2 + len(g.scenes)], g.scenes):
        tok = line.split()
        K = sc.coarse.shape[0]
        assert [int(tok[0]), int(tok[1]), int(tok[2])] == [K, len(sc.static), len(sc.dynamic)]
        assert float(tok[3]) == pytest.approx(float(sc.start.sum() + sc.coarse.sum()), rel=1e-13)
        pts, cnt = scene_io.environment_points(sc, np.arange(K) * g.dt)
        for k in range(K):
            c, v = tok[4 + k].split(":")
            assert int(c) == cnt[k]
            want = float((pts[k, :cnt[k], 0] + 2.0 * pts[k, :cnt[k], 1]).sum())
            assert float(v) == pytest.approx(want, rel=1e-12, abs=1e-9)
    lb, rb = scene_io.road_barriers(g.center)
    want = float((lb[:, 0] + 2 * lb[:, 1]).sum() - (rb[:, 0] + 2 * rb[:, 1]).sum())
    assert float(out[-1]) == pytest.approx(want, rel=1e-12)
    bad = tmp_path / "bad.cqs"
    bad.write_bytes(open(path, "rb").read()[:-5])
    assert subprocess.run([str(exe), str(bad)], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_replay_from_file_on_the_gpu(built, tmp_path):
    """file -> Environment queries -> cilqr_build_corridors / cilqr_lane_constraints -> solve: the
    corridors equal the ones built from the generator's arrays bit for bit, and so do the solves."""
    sc, f = _scene_file(B=24, seed=63)
    path = tmp_path / "scenes.cqs"
    scene_io.save(str(path), f)
    g = scene_io.load(str(path))
    B, K = len(g.scenes), g.scenes[0].coarse.shape[0]
    times = np.arange(K) * g.dt
    per = [scene_io.environment_points(s, times) for s in g.scenes]
    P = max(p.shape[1] for p, _ in per)
    pts = np.zeros((B, K, P, 2))
    cnt = np.zeros((B, K), np.int32)
    for b, (p, c) in enumerate(per):
        pts[b, :, :p.shape[1]] = p
        cnt[b] = c
    coarse = np.stack([s.coarse for s in g.scenes])
    start = np.stack([s.start for s in g.scenes])
    left_b, right_b = scene_io.road_barriers(g.center)
    scene = dict(start=start, coarse=coarse, left=api.lane_constraints(left_b, 5.0, True),
                 right=api.lane_constraints(right_b, 5.0, False), n_steps=K - 1, dt=g.dt, cmax=sc["cmax"])
    opt = api.BatchIlqrOptimizer(api.default_config(K - 1), batch_capacity=B, cmax=sc["cmax"])
    cor, ccnt, nf = opt.build_corridors(coarse[:, :, :3], pts, cnt, cmax=sc["cmax"])
    cor2, ccnt2, nf2 = opt.build_corridors(sc["coarse"][:, :, :3], sc["obstacle_points"], sc["obstacle_count"],
                                           cmax=sc["cmax"])
    assert nf == 0 and nf2 == 0
    assert np.array_equal(ccnt, ccnt2) and np.array_equal(cor, cor2)
    a = opt.plan(dict(scene, corridor=cor, ccount=ccnt))
    b = opt.plan(dict(sc, corridor=cor2, ccount=ccnt2))
    for k in ("traj", "cost_hist", "n_cost", "status"):
        assert np.array_equal(a[k], b[k]), k
    opt.close()


def test_reference_pickle_import(tmp_path):
    """scene_io.from_reference_pickle reads the reference's own scene artefact -- reference.pickle of
    script/reference_publisher.py:232-236: {"center", "static", "dynamic"} ROS messages in a Python 2 text pickle --
    without rospy (tests/golden/reference_scene.pickle, written by make_reference_pickle.py in that layout), converts
    it to a .cqs that round-trips exactly, refuses pickles that name anything but those message classes, and the DP
    coarse planner plans on it from the state the reference's node starts in."""
    import pickle
    from cilqr_amd import api
    src = os.path.join(HERE, "golden", "reference_scene.pickle")
    sf = scene_io.from_reference_pickle(src)
    assert sf.center.shape[1] == 7 and len(sf.center) == 391 and np.all(np.diff(sf.center[:, 0]) > 0)
    assert np.allclose(sf.center[:, 5], 2.5) and np.allclose(sf.center[:, 6], 6.0)          # reference_publisher.py:25-26
    sc = sf.scenes[0]
    assert len(sc.static) == 1 and len(sc.dynamic) == 8 and sc.coarse.shape == (0, 6)
    assert np.array_equal(sc.start, [0.0, 0.0, 0.0, 10.0])                                   # planning_node.cc:24-30
    assert all(p.shape == (4, 2) for p in sc.static) and all(d.polygon.shape == (4, 2) for d in sc.dynamic)
    assert all(d.trajectory.shape[1] == 4 and np.all(np.diff(d.trajectory[:, 0]) > 0) for d in sc.dynamic)
    # Point32 coordinates are float32 in the message: they arrive as such
    assert all(np.array_equal(p, p.astype(np.float32).astype(np.float64)) for p in sc.static)
    path = tmp_path / "ref.cqs"
    scene_io.save(str(path), sf)
    back = scene_io.load(str(path))
    assert np.array_equal(back.center, sf.center) and len(back.scenes) == 1
    assert all(np.array_equal(a, b) for a, b in zip(back.scenes[0].static, sc.static))
    assert all(np.array_equal(a.trajectory, b.trajectory) and np.array_equal(a.polygon, b.polygon)
               for a, b in zip(back.scenes[0].dynamic, sc.dynamic))
    ok, coarse = api.dp_plan(scene_io.flatten_scene(sf.center, sc), sc.start[:3])
    assert ok and coarse.shape == (81, 9) and np.isfinite(coarse).all() and coarse[-1, 1] > 40.0
    evil = tmp_path / "evil.pickle"
    evil.write_bytes(b"cos\nsystem\n(S'true'\ntR.")
    with pytest.raises(pickle.UnpicklingError):
        scene_io.from_reference_pickle(str(evil))
