"""Scene wire format: a compact binary equivalent of what the reference moves between
its scenario publisher and its planning node -- the six ROS messages under msg/ (CenterLine,
CenterLinePoint, Obstacles, DynamicObstacle, DynamicObstacles, DynamicTrajectoryPoint) and the
pickle {"center", "static", "dynamic"} of script/reference_publisher.py:232-236 -- plus the two
things a replay of the optimizer needs on top (start state, coarse trajectory), so that a scene
can be replayed bit for bit on the CPU oracle and on the GPU.

A file holds ONE road and B scenes on it:

    "CILQRSC1" | u32 version = 1 | u32 B | u32 K | f64 dt
    u32 n_center | center[n_center][7] f64            CenterLinePoint: s x y theta kappa left_bound right_bound
    per scene:
      start[4] f64 (x y theta v) | coarse[K][6] f64 (x y theta v a delta)
      u32 n_static  | per obstacle: u32 m | polygon[m][2] f64 (world frame)           Obstacles.msg
      u32 n_dynamic | per obstacle: u32 m | polygon[m][2] f64 (body frame)            DynamicObstacle.msg
                                  | u32 T | trajectory[T][4] f64 (time x y theta)     DynamicTrajectoryPoint.msg

Everything little-endian.  `environment_points` and `road_barriers` restate the queries the reference's
Environment answers from that data (algorithm/utils/environment.cpp:134-182, planning_node.cc:63-78):
the obstacle corner points valid at a knot's time and the two road barriers -- the inputs of
cilqr_build_corridors / cilqr_lane_constraints.
"""
from __future__ import annotations

import dataclasses
import struct

import numpy as np

MAGIC = b"CILQRSC1"
K_MATH_EPS = 1e-10  # algorithm/math/vec2d.h:33


@dataclasses.dataclass
class DynamicObstacle:
    polygon: np.ndarray      # [m,2] body frame
    trajectory: np.ndarray   # [T,4] time, x, y, theta


@dataclasses.dataclass
class Scene:
    start: np.ndarray                 # [4]
    coarse: np.ndarray                # [K,6]
    static: list                      # of [m,2] world-frame polygons
    dynamic: list                     # of DynamicObstacle


@dataclasses.dataclass
class SceneFile:
    dt: float
    center: np.ndarray                # [n,7]
    scenes: list


def _rect(hl: float, hw: float) -> np.ndarray:
    return np.array([[hl, hw], [hl, -hw], [-hl, -hw], [-hl, hw]], dtype=np.float64)


def from_generator(sc: dict) -> SceneFile:
    """Scenes of cilqr_amd.scenario.generate(..., scenarios=True) in the reference's vocabulary: an
    obstacle that never moves becomes a static polygon (Obstacles.msg), the others a body-frame
    polygon with the trajectory of the knots at which they exist (pedestrians enter and leave)."""
    road = sc["road"]
    from .scenario import LEFT_BOUND, RIGHT_BOUND
    n = len(road.s)
    center = np.stack([road.s, road.x, road.y, road.theta, road.kappa, np.full(n, LEFT_BOUND), np.full(n, RIGHT_BOUND)], 1)
    pose, live, half, kind = sc["obstacle_pose"], sc["obstacle_live"], sc["obstacle_half_size"], sc["obstacle_kind"]
    B, O, K = live.shape
    t = np.arange(K) * sc["dt"]
    scenes = []
    for b in range(B):
        static, dynamic = [], []
        for o in range(O):
            m = live[b, o]
            if not m.any():
                continue
            body = _rect(*half[o])
            if kind[o] == 2:      # static vehicle: one world-frame polygon
                x, y, th = pose[b, o, 0]
                R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
                static.append(body @ R.T + [x, y])
            else:
                traj = np.concatenate([t[m, None], pose[b, o, m]], axis=1)
                dynamic.append(DynamicObstacle(body, traj))
        scenes.append(Scene(sc["start"][b].copy(), sc["coarse"][b].copy(), static, dynamic))
    return SceneFile(float(sc["dt"]), center, scenes)


def save(path: str, f: SceneFile) -> None:
    K = f.scenes[0].coarse.shape[0] if f.scenes else 0
    with open(path, "wb") as o:
        o.write(MAGIC)
        o.write(struct.pack("<IIId", 1, len(f.scenes), K, f.dt))
        c = np.ascontiguousarray(f.center, dtype="<f8")
        o.write(struct.pack("<I", c.shape[0]))
        o.write(c.tobytes())
        for s in f.scenes:
            o.write(np.ascontiguousarray(s.start, dtype="<f8").tobytes())
            o.write(np.ascontiguousarray(s.coarse, dtype="<f8").tobytes())
            o.write(struct.pack("<I", len(s.static)))
            for p in s.static:
                o.write(struct.pack("<I", len(p)))
                o.write(np.ascontiguousarray(p, dtype="<f8").tobytes())
            o.write(struct.pack("<I", len(s.dynamic)))
            for d in s.dynamic:
                o.write(struct.pack("<I", len(d.polygon)))
                o.write(np.ascontiguousarray(d.polygon, dtype="<f8").tobytes())
                o.write(struct.pack("<I", len(d.trajectory)))
                o.write(np.ascontiguousarray(d.trajectory, dtype="<f8").tobytes())


def load(path: str) -> SceneFile:
    raw = open(path, "rb").read()
    if raw[:8] != MAGIC:
        raise ValueError("not a CILQR scene file")
    pos = 8
    version, B, K, dt = struct.unpack_from("<IIId", raw, pos)
    pos += 20
    if version != 1:
        raise ValueError(f"unsupported scene file version {version}")

    def u32():
        nonlocal pos
        v, = struct.unpack_from("<I", raw, pos)
        pos += 4
        return v

    def f64(*shape):
        nonlocal pos
        n = int(np.prod(shape))
        a = np.frombuffer(raw, dtype="<f8", count=n, offset=pos).reshape(shape).copy()
        pos += 8 * n
        return a

    center = f64(u32(), 7)
    scenes = []
    for _ in range(B):
        start, coarse = f64(4), f64(K, 6)
        static = [f64(u32(), 2) for _ in range(u32())]
        dynamic = []
        for _ in range(u32()):
            poly = f64(u32(), 2)
            dynamic.append(DynamicObstacle(poly, f64(u32(), 4)))
        scenes.append(Scene(start, coarse, static, dynamic))
    if pos != len(raw):
        raise ValueError("trailing bytes in scene file")
    return SceneFile(dt, center, scenes)


def sample_points(polygon: np.ndarray) -> np.ndarray:
    """Polygon2d::sample_points (algorithm/math/polygon2d.cpp:259-271): six points per edge (ratio = 0, 0.2, ... with
    the reference's floating-point accumulation), edges in the counter-clockwise order BuildFromPoints leaves
    (polygon2d.cpp:212-220: a clockwise input is reversed first)."""
    p = np.asarray(polygon, float)
    area = sum((p[i - 1, 0] - p[0, 0]) * (p[i, 1] - p[0, 1]) - (p[i - 1, 1] - p[0, 1]) * (p[i, 0] - p[0, 0]) for i in range(1, len(p)))
    if area < 0:
        p = p[::-1]
    out = []
    for i in range(len(p)):
        q = p[(i + 1) % len(p)]
        ratio = 0.0
        while ratio < 1.0 + K_MATH_EPS:
            out.append([p[i, 0] * (1 - ratio) + q[0] * ratio, p[i, 1] * (1 - ratio) + q[1] * ratio])
            ratio += 1.0 / 5.0
    return np.asarray(out)


def environment_points(scene: Scene, times, multiple_sample: bool = False) -> tuple:
    """Environment::QueryStaticObstaclesPoints + QueryDynamicObstaclesPoints (environment.cpp:153-182,
    is_multiple_sample = false) for every knot time: (points [K,P,2] padded with zeros, counts [K]).
    A dynamic obstacle exists at time t when t lies within its trajectory (to 1e-10); its polygon is
    the one of the first trajectory sample later than t - 1e-10 (std::upper_bound, cpp:143-146),
    placed by that sample's pose (planning_node.cc:68-75).  multiple_sample: the polygons' sample points
    (is_multiple_sample, environment.cpp:163,178) instead of their corners."""
    pick = sample_points if multiple_sample else (lambda p: p)
    per_knot = []
    for t in times:
        pts = [pick(p) for p in scene.static]
        for d in scene.dynamic:
            tt = d.trajectory[:, 0]
            if tt[0] > t + K_MATH_EPS or tt[-1] < t - K_MATH_EPS:
                continue
            i = int(np.searchsorted(tt + K_MATH_EPS, t, side="right"))   # first sample with t < time + eps
            i = min(i, len(tt) - 1)
            _, x, y, th = d.trajectory[i]
            c, s = np.cos(th), np.sin(th)     # Pose::transform (pose.h:40-46): x + rx cos - ry sin, in that order
            pts.append(pick(np.stack([x + d.polygon[:, 0] * c - d.polygon[:, 1] * s,
                                      y + d.polygon[:, 0] * s + d.polygon[:, 1] * c], axis=1)))
        per_knot.append(np.concatenate(pts, axis=0) if pts else np.zeros((0, 2)))
    P = max((len(p) for p in per_knot), default=0)
    out = np.zeros((len(per_knot), P, 2))
    cnt = np.zeros(len(per_knot), dtype=np.int32)
    for k, p in enumerate(per_knot):
        out[k, :len(p)] = p
        cnt[k] = len(p)
    return out, cnt


def road_barriers(center: np.ndarray) -> tuple:
    """left_road_barrier / right_road_barrier of the reference's Environment: the centre line shifted by
    +left_bound / -right_bound along its normal (ReferenceLine::GetCartesian at every centre point)."""
    x, y, th, lb, rb = center[:, 1], center[:, 2], center[:, 3], center[:, 5], center[:, 6]
    left = np.stack([x - lb * np.sin(th), y + lb * np.cos(th)], 1)
    right = np.stack([x + rb * np.sin(th), y - rb * np.cos(th)], 1)
    return left, right


def flatten_scene(center: np.ndarray, scene: Scene) -> dict:
    """The scene as the C-ABI's `cilqr_scene` takes it (include/cilqr.h): polygons and trajectories back to back."""
    def cat(arrs, width):
        return np.ascontiguousarray(np.concatenate(arrs, axis=0) if arrs else np.zeros((0, width)), dtype=np.float64)
    return dict(
        center=np.ascontiguousarray(center, dtype=np.float64),
        static_points=cat([np.asarray(p, float).reshape(-1, 2) for p in scene.static], 2),
        static_counts=np.asarray([len(p) for p in scene.static], dtype=np.int32),
        dynamic_polygon_points=cat([np.asarray(d.polygon, float).reshape(-1, 2) for d in scene.dynamic], 2),
        dynamic_polygon_counts=np.asarray([len(d.polygon) for d in scene.dynamic], dtype=np.int32),
        dynamic_trajectories=cat([np.asarray(d.trajectory, float).reshape(-1, 4) for d in scene.dynamic], 4),
        dynamic_trajectory_counts=np.asarray([len(d.trajectory) for d in scene.dynamic], dtype=np.int32))


# ---------------------------------------------------------------------------------------------
# the reference's own scene artefact: reference.pickle
# ---------------------------------------------------------------------------------------------
# script/reference_publisher.py:232-236 dumps {"center": CenterLine, "static": Obstacles | None, "dynamic":
# DynamicObstacles | None} -- ROS message objects (genpy.Message: __slots__ in the field order of msg/*.msg and
# geometry_msgs, pickled as the list of slot values) -- with Python 2's text pickle; script/pickle_publisher.py:24-55
# replays it.  Reading it needs neither rospy nor the generated message modules: the unpickler below maps exactly
# these message classes to slot-ordered stand-ins and refuses everything else.
_MSG_SLOTS = {
    "CenterLine": ("points",),                                                   # msg/CenterLine.msg
    "CenterLinePoint": ("s", "x", "y", "theta", "kappa", "left_bound", "right_bound"),   # msg/CenterLinePoint.msg
    "Obstacles": ("obstacles",),                                                 # msg/Obstacles.msg
    "DynamicObstacles": ("obstacles",),                                          # msg/DynamicObstacles.msg
    "DynamicObstacle": ("polygon", "trajectory"),                                # msg/DynamicObstacle.msg
    "DynamicTrajectoryPoint": ("time", "x", "y", "theta"),                       # msg/DynamicTrajectoryPoint.msg
    "Polygon": ("points",),                                                      # geometry_msgs/Polygon
    "Point32": ("x", "y", "z"),                                                  # geometry_msgs/Point32
}
_MSG_MODULES = ("planning.msg", "geometry_msgs.msg")
_STUBS = {}


def _stub(name):
    if name not in _STUBS:
        slots = _MSG_SLOTS[name]

        def __setstate__(self, state, _slots=slots):           # genpy.Message.__setstate__: slot values in order
            for k, v in zip(_slots, state):
                setattr(self, k, v)

        _STUBS[name] = type(name, (object,), {"__setstate__": __setstate__, "_slots": slots})
    return _STUBS[name]


def _reference_unpickler(f):
    import pickle

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split("._")[0] in _MSG_MODULES and name in _MSG_SLOTS:
                return _stub(name)
            if (module, name) in (("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor")):
                import copyreg
                return copyreg._reconstructor
            if (module, name) in (("__builtin__", "object"), ("builtins", "object")):
                return object
            raise pickle.UnpicklingError(f"{module}.{name} is not part of a reference scene pickle")

    return U(f, encoding="latin1")


def from_reference_pickle(path: str, start=(0.0, 0.0, 0.0, 10.0), dt: float = 0.1) -> SceneFile:
    """The reference's reference.pickle -> a one-scene SceneFile (save() then gives the .cqs that
    include/cilqr/scene_file.hpp and the DP planner read).  `start` defaults to the state PlanningNode hard-codes
    (x = y = theta = 0, v = 10: algorithm/planning_node.cc:24-30); the scene carries no coarse trajectory (K = 0):
    that is what cilqr_dp_plan produces from it."""
    with open(path, "rb") as f:
        ref = _reference_unpickler(f).load()
    if not isinstance(ref, dict) or "center" not in ref:
        raise ValueError("not a reference scene pickle: no 'center'")
    center = np.array([[getattr(p, k) for k in _MSG_SLOTS["CenterLinePoint"]] for p in ref["center"].points], dtype=np.float64)
    static = []
    if ref.get("static") is not None:
        for poly in ref["static"].obstacles:                     # Obstacles.msg: geometry_msgs/Polygon[], world frame
            static.append(np.array([[q.x, q.y] for q in poly.points], dtype=np.float64))
    dynamic = []
    if ref.get("dynamic") is not None:
        for ob in ref["dynamic"].obstacles:
            poly = np.array([[q.x, q.y] for q in ob.polygon.points], dtype=np.float64)      # body frame
            traj = np.array([[t.time, t.x, t.y, t.theta] for t in ob.trajectory], dtype=np.float64).reshape(-1, 4)
            dynamic.append(DynamicObstacle(poly, traj))
    return SceneFile(float(dt), center, [Scene(np.asarray(start, dtype=np.float64), np.zeros((0, 6)), static, dynamic)])
