#!/usr/bin/env python
"""Writes tests/golden/reference_scene.pickle: a scene in the layout of the reference's reference.pickle
(script/reference_publisher.py:232-236) -- {"center", "static", "dynamic"} holding ROS message objects pickled with
the text protocol.  rospy / genpy are not available here, so the message classes are re-created with what pickling
sees of a genpy.Message: module path planning.msg._<Type> / geometry_msgs.msg._<Type>, __slots__ in the field order of
msg/*.msg, __getstate__ = list of slot values.  The scene itself comes from this repository's generator (one "mix11"
scene on the reference road), with float32 polygon coordinates as geometry_msgs/Point32 carries them.
    python tests/golden/make_reference_pickle.py"""
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cilqr_amd import scenario, scene_io  # noqa: E402

SLOTS = {
    ("planning.msg", "CenterLine"): ["points"],
    ("planning.msg", "CenterLinePoint"): ["s", "x", "y", "theta", "kappa", "left_bound", "right_bound"],
    ("planning.msg", "Obstacles"): ["obstacles"],
    ("planning.msg", "DynamicObstacles"): ["obstacles"],
    ("planning.msg", "DynamicObstacle"): ["polygon", "trajectory"],
    ("planning.msg", "DynamicTrajectoryPoint"): ["time", "x", "y", "theta"],
    ("geometry_msgs.msg", "Polygon"): ["points"],
    ("geometry_msgs.msg", "Point32"): ["x", "y", "z"],
}
CLS = {}
for (pkg, name), slots in SLOTS.items():
    modname = f"{pkg}._{name}"
    for part in (pkg.split(".")[0], pkg, modname):
        sys.modules.setdefault(part, types.ModuleType(part))
    cls = type(name, (object,), {"__slots__": slots, "__module__": modname,
                                 "__getstate__": lambda self: [getattr(self, k) for k in self.__slots__],
                                 "__setstate__": lambda self, st: [setattr(self, k, v) for k, v in zip(self.__slots__, st)]})
    setattr(sys.modules[modname], name, cls)
    CLS[name] = cls


def msg(name, **kw):
    m = CLS[name]()
    for k in m.__slots__:
        setattr(m, k, kw[k])
    return m


sc = scenario.generate("mix11", 1, seed=404, scenarios=True)
sf = scene_io.from_generator(sc)
scene = sf.scenes[0]
center = msg("CenterLine", points=[msg("CenterLinePoint", **dict(zip(SLOTS[("planning.msg", "CenterLinePoint")], map(float, row))))
                                   for row in sf.center[::5]])        # every 0.5 m keeps the fixture small
f32 = lambda v: float(np.float32(v))
poly = lambda p: msg("Polygon", points=[msg("Point32", x=f32(q[0]), y=f32(q[1]), z=0.0) for q in p])
static = msg("Obstacles", obstacles=[poly(p) for p in scene.static])
dynamic = msg("DynamicObstacles", obstacles=[
    msg("DynamicObstacle", polygon=poly(d.polygon),
        trajectory=[msg("DynamicTrajectoryPoint", time=float(t[0]), x=float(t[1]), y=float(t[2]), theta=float(t[3])) for t in d.trajectory])
    for d in scene.dynamic])
out = os.path.join(HERE, "reference_scene.pickle")
with open(out, "wb") as f:
    pickle.dump({"center": center, "static": static, "dynamic": dynamic}, f, protocol=0)
print(out, os.path.getsize(out), "bytes;", len(scene.static), "static,", len(scene.dynamic), "dynamic obstacles")
