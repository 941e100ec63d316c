"""Synthetic CILQR scenes in optimizer-input space (numpy, host side).

The reference has no committed scenes: its inputs come from a random ROS publisher
(script/reference_publisher.py) -> DP coarse planner -> corridor builder.  This module is the
build's own generator for the inputs of ``IlqrOptimizer::Plan`` (ilqr_optimizer.h:41-48):

* road: the reference publisher's centre line ``[30, [-90,10], 10, [180,5], 36, [-180,12], 50]``
  with bounds left 2.5 m / right 6.0 m (reference_publisher.py:25-26,200-209), rebuilt
  analytically at 0.1 m;
* lane half-planes + segments: boundary polylines sampled every >=5 m, one half-plane per
  segment, left list walked backwards (corridor.cc:265-331);
* obstacles: 1x1 m pedestrians crossing laterally at 0.4-1.4 m/s, 4x2 m vehicles moving along
  the lane at 4-6 m/s on lateral 0 / -4, static 4x2 m vehicles on lateral {1, 0, -4}
  (reference_publisher.py:116-194);
* coarse trajectory: piecewise-linear (s, l) over 5 time layers like the DP planner's
  interpolation (dp_planner.cpp:226-281), best-clearance pick among a few lateral profiles,
  heading/velocity/acceleration/curvature by finite differences, delta = atan(kappa * L);
* corridor per knot: a 20x20 m heading-aligned box (corridor.cc:89-120 adds those box points)
  intersected with one separating half-plane per obstacle live at that knot's time.  This is a
  simplified convex free region, NOT the reference's sphere-flip/convex-hull construction
  (corridor.cc:122-263, out of scope for the GPU path).

Everything is deterministic in (seed, problem index).  Arrays are problem-major, fp64.
"""
from __future__ import annotations

import dataclasses
import numpy as np

ROAD_CONFIG = (30.0, (-90.0, 10.0), 10.0, (180.0, 5.0), 36.0, (-180.0, 12.0), 50.0)
LEFT_BOUND = 2.5
RIGHT_BOUND = 6.0
WHEEL_BASE = 1.0
NT = 5  # DP time layers (dp_planner.h:27)
MIN_CLEARANCE = 1.6  # m, ego centre line to obstacle boundary


@dataclasses.dataclass
class Road:
    s: np.ndarray
    x: np.ndarray
    y: np.ndarray
    theta: np.ndarray
    kappa: np.ndarray

    @property
    def length(self) -> float:
        return float(self.s[-1])

    def eval(self, s):
        """Linear interpolation of (x, y, theta, kappa) at stations s (any shape)."""
        s = np.clip(s, 0.0, self.length)
        return (np.interp(s, self.s, self.x), np.interp(s, self.s, self.y),
                np.interp(s, self.s, self.theta), np.interp(s, self.s, self.kappa))

    def cartesian(self, s, l):
        x, y, th, _ = self.eval(s)
        return x - l * np.sin(th), y + l * np.cos(th)


def build_road(config=ROAD_CONFIG, resolution=0.1) -> Road:
    """Centre line as straight / arc pieces, sampled every `resolution` metres of arc length."""
    pieces = []  # (s0, length, x0, y0, th0, kappa)
    x = y = th = s = 0.0
    for seg in config:
        if isinstance(seg, (tuple, list)):
            deg, radius = seg
            ang = np.deg2rad(deg)
            kap = np.sign(ang) / radius
            length = abs(ang) * radius
        else:
            kap = 0.0
            length = float(seg)
        pieces.append((s, length, x, y, th, kap))
        if kap == 0.0:
            x += length * np.cos(th)
            y += length * np.sin(th)
        else:
            x += (np.sin(th + kap * length) - np.sin(th)) / kap
            y += -(np.cos(th + kap * length) - np.cos(th)) / kap
            th += kap * length
        s += length
    total = s
    n = int(np.floor(total / resolution)) + 1
    ss = np.arange(n) * resolution
    xs = np.empty(n)
    ys = np.empty(n)
    ths = np.empty(n)
    kps = np.empty(n)
    starts = np.array([p[0] for p in pieces])
    idx = np.clip(np.searchsorted(starts, ss, side="right") - 1, 0, len(pieces) - 1)
    for pi, (s0, length, x0, y0, th0, kap) in enumerate(pieces):
        m = idx == pi
        d = ss[m] - s0
        if kap == 0.0:
            xs[m] = x0 + d * np.cos(th0)
            ys[m] = y0 + d * np.sin(th0)
            ths[m] = th0
        else:
            xs[m] = x0 + (np.sin(th0 + kap * d) - np.sin(th0)) / kap
            ys[m] = y0 - (np.cos(th0 + kap * d) - np.cos(th0)) / kap
            ths[m] = th0 + kap * d
        kps[m] = kap
    return Road(ss, xs, ys, ths, kps)


def lane_constraints(road: Road, seg_len=5.0):
    """Left/right lane tables, rows (a, b, c, start_x, start_y, end_x, end_y), "ax+by<c" inside.

    Follows corridor.cc:265-331: keep a boundary point once it is >= seg_len from the last kept
    one; left segments run backwards (p[i] -> p[i-1]), right forwards (p[i-1] -> p[i]);
    (a, b) = (dy, -dx) of the segment, c = a*sx + b*sy (un-normalised, |(a,b)| = segment length).
    """
    def sample(px, py):
        keep = [0]
        lx, ly = px[0], py[0]
        for i in range(len(px)):
            if np.hypot(px[i] - lx, py[i] - ly) >= seg_len - 1e-10:
                keep.append(i)
                lx, ly = px[i], py[i]
        return px[keep], py[keep]

    def rows(sx, sy, ex, ey):
        a = ey - sy
        b = -(ex - sx)
        c = a * sx + b * sy
        return np.stack([a, b, c, sx, sy, ex, ey], axis=1)

    lx, ly = road.cartesian(road.s, LEFT_BOUND)
    rx, ry = road.cartesian(road.s, -RIGHT_BOUND)
    lx, ly = sample(lx, ly)
    rx, ry = sample(rx, ry)
    left = rows(lx[1:], ly[1:], lx[:-1], ly[:-1])
    right = rows(rx[:-1], ry[:-1], rx[1:], ry[1:])
    return np.ascontiguousarray(left), np.ascontiguousarray(right)


@dataclasses.dataclass
class SceneSpec:
    """What a scene family contains (BASELINE.json configs)."""
    n_steps: int = 50
    dt: float = 0.1
    n_pedestrians: int = 6
    n_dynamic: int = 0
    n_static: int = 0
    cmax: int = 16
    v_range: tuple = (6.0, 12.0)
    s0_range: tuple = (2.0, 80.0)
    n_candidates: int = 8
    # obstacles the chosen coarse path passes closer than this are dropped from the scene (the DP planner
    # only returns collision-free paths); below disc_radius + safe_margin = 1.21 m the shrunk corridor plane
    # of a kept obstacle lies beyond the coarse path: barriers in the relaxed region at the init guess
    min_clearance: float = MIN_CLEARANCE


SPECS = {
    # configs[1]: B=4096 random_pedestrian, 6 obstacles
    "ped6": SceneSpec(n_steps=50, n_pedestrians=6, n_dynamic=0, n_static=0, cmax=16),
    # configs[2]/[3]: 6 pedestrians + 3 moving + 2 static vehicles
    "mix11": SceneSpec(n_steps=50, n_pedestrians=6, n_dynamic=3, n_static=2, cmax=16),
    # the reference's demo: tf = 8 s (81 knots), ego starts at the road origin with v = 10
    # (planning_node.cc:24-30, planner_config.h:94,99), 6 pedestrians + 3 moving + 2 static vehicles
    "demo80": SceneSpec(n_steps=80, n_pedestrians=6, n_dynamic=3, n_static=2, cmax=16,
                        v_range=(10.0, 10.0), s0_range=(0.5, 0.5)),
    # configs[4]: 100-step horizon, 20 dynamic obstacles
    "dyn20": SceneSpec(n_steps=100, n_pedestrians=12, n_dynamic=8, n_static=0, cmax=24,
                       v_range=(5.0, 9.0), s0_range=(2.0, 60.0)),
    # configs[4] as BASELINE words it ("constrained-ILQR barrier active"): the same scenes, but
    # obstacles down to 0.6 m from the coarse path stay -- their corridor planes, shrunk by 1.21 m, cut
    # across the coarse path, so corridor barriers sit in the relaxed region at the init guess
    "dyn20x": SceneSpec(n_steps=100, n_pedestrians=12, n_dynamic=8, n_static=0, cmax=24,
                        v_range=(5.0, 9.0), s0_range=(2.0, 60.0), min_clearance=0.6),
}


def _rect_nearest(ox, oy, cx, cy, cth, hl, hw):
    """Nearest point of an oriented rectangle (centre c, heading cth, half sizes hl/hw) to o."""
    c, s = np.cos(cth), np.sin(cth)
    dx, dy = ox - cx, oy - cy
    lx = c * dx + s * dy
    ly = -s * dx + c * dy
    qx = np.clip(lx, -hl, hl)
    qy = np.clip(ly, -hw, hw)
    inside = (np.abs(lx) <= hl) & (np.abs(ly) <= hw)
    wx = cx + c * qx - s * qy
    wy = cy + s * qx + c * qy
    return wx, wy, inside


def _chunk_job(args):
    spec, n, seed, first, road, want_points, want_scn = args
    return _generate_chunk(spec, n, seed, first, road, want_points, want_scn)


def generate(spec: SceneSpec | str, batch: int, seed: int = 0, first_problem: int = 0,
             road: Road | None = None, chunk: int = 1024, workers: int = 0, obstacle_points: bool = False,
             scenarios: bool = False):
    """Generate `batch` scenes; problem p uses RNG stream (seed, first_problem + p).

    workers > 1 spreads the chunks over a thread pool (results are identical).
    Returns dict(start[B,4], coarse[B,K,6], corridor[B,K,cmax,3], ccount[B,K] int32,
                 left[S,7], right[S,7], n_steps, dt, cmax).
    obstacle_points=True adds what the reference's Environment hands to Corridor::Plan
    (Query{Static,Dynamic}ObstaclesPoints, environment.cpp:153-182): obstacle_points[B,K,4*O,2], the
    corner points of the obstacles alive at each knot's time (packed to the front), and
    obstacle_count[B,K] int32 -- the inputs of BatchIlqrOptimizer.build_corridors.
    scenarios=True adds the scene itself in the vocabulary of the reference's messages (msg/*.msg):
    "obstacle_pose"[B,O,K,3] (x, y, theta of every obstacle at every knot time), "obstacle_live"[B,O,K]
    bool, "obstacle_half_size"[O,2] (half length, half width) -- what cilqr_amd.scene_io turns into
    static polygons and dynamic obstacles with trajectories -- and "road" (the centre line).
    """
    if isinstance(spec, str):
        spec = SPECS[spec]
    road = road or build_road()
    left, right = lane_constraints(road)
    jobs = [(spec, min(chunk, batch - c0), seed, first_problem + c0, road, obstacle_points, scenarios)
            for c0 in range(0, batch, chunk)]
    if workers > 1 and len(jobs) > 1:
        # threads, not processes: numpy releases the GIL inside its loops, and forking a process
        # that has HIP/rocprofv3 loaded is not safe
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(workers, len(jobs))) as pool:
            outs = list(pool.map(_chunk_job, jobs))
    else:
        outs = [_chunk_job(j) for j in jobs]
    shared = {k: outs[0].pop(k) for k in list(outs[0]) if k in ("obstacle_half_size", "obstacle_kind")}
    for o in outs[1:]:
        for k in shared:
            o.pop(k, None)
    out = {k: np.concatenate([o[k] for o in outs], axis=0) for k in outs[0]}
    out.update(shared)
    out.update(left=left, right=right, n_steps=spec.n_steps, dt=spec.dt, cmax=spec.cmax)
    if scenarios:
        out["road"] = road
    return out


def _uniforms(seed, first, n, m):
    """[n, m] uniforms; row p depends only on (seed, first + p)."""
    out = np.empty((n, m))
    for p in range(n):
        out[p] = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, first + p])).random(m)
    return out


def _generate_chunk(spec: SceneSpec, B: int, seed: int, first: int, road: Road, want_points: bool = False,
                    want_scn: bool = False):
    # Array convention inside: obstacle / candidate axes first, (problem, knot) last, so numpy's
    # inner loops run over B*K contiguous elements.
    N, dt = spec.n_steps, spec.dt
    K = N + 1
    T = N * dt
    O = spec.n_pedestrians + spec.n_dynamic + spec.n_static
    C = spec.n_candidates
    n_rand = 8 + 6 * O + C * NT
    R = _uniforms(seed, first, B, n_rand)
    t = np.arange(K) * dt  # [K]

    # ---- ego start and speed profile ----
    s0 = spec.s0_range[0] + R[:, 0] * (spec.s0_range[1] - spec.s0_range[0])
    v0 = spec.v_range[0] + R[:, 1] * (spec.v_range[1] - spec.v_range[0])
    l0 = -3.5 + R[:, 2] * 4.0                       # start lateral in [-3.5, 0.5]
    dv = (R[:, 3] - 0.5) * 2.0                      # end-of-horizon speed change
    th_noise = (R[:, 4] - 0.5) * 0.1
    v_t = np.clip(v0[:, None] + dv[:, None] * (t[None, :] / T), 2.0, 18.0)     # [B,K]
    s_t = s0[:, None] + np.concatenate(
        [np.zeros((B, 1)), np.cumsum(0.5 * (v_t[:, 1:] + v_t[:, :-1]) * dt, axis=1)], axis=1)
    s_t = np.minimum(s_t, road.length - 1.0)

    # ---- obstacles [O,B,(K)]: kind, station, lateral path, speed, timing ----
    ro = np.ascontiguousarray(R[:, 8:8 + 6 * O].reshape(B, O, 6).transpose(1, 2, 0))   # [O,6,B]
    kind = np.concatenate([np.zeros(spec.n_pedestrians, int), np.ones(spec.n_dynamic, int),
                           2 * np.ones(spec.n_static, int)])  # 0 ped, 1 moving vehicle, 2 static
    is_ped = (kind == 0)[:, None, None]
    is_dyn = (kind == 1)[:, None, None]
    span = (s_t[:, -1] - s0)[None, :]
    st_o = s0[None, :] + 8.0 + ro[:, 0] * (span + 6.0)             # [O,B] station at t = 0
    hl = np.where(kind == 0, 0.5, 2.0)[:, None, None]
    hw = np.where(kind == 0, 0.5, 1.0)[:, None, None]
    # pedestrians: lateral sweeps road_ub -> road_lb (or back) at 0.4..1.4 m/s from t_on = ds/20
    road_lb, road_ub = -RIGHT_BOUND - 1.0, LEFT_BOUND + 1.0
    ped_v = 0.4 + ro[:, 1]
    ped_dir = np.where(ro[:, 2] > 0.5, -1.0, 1.0)                  # -1: ub -> lb
    ped_t_on = (st_o - s0[None, :]) / 20.0
    ped_dur = (road_ub - road_lb) / ped_v
    # vehicles: lateral 0 / -4 moving at 4..6 m/s; static: lateral {1, 0, -4}
    veh_l = np.where(ro[:, 2] > 0.5, 0.0, -4.0)
    sta_l = np.choose((ro[:, 2] * 3).astype(int).clip(0, 2), [1.0, 0.0, -4.0])
    veh_v = 4.0 + 2.0 * ro[:, 1]

    tt = np.broadcast_to(t[None, None, :], (O, B, K))
    tau = np.minimum(np.maximum(tt - ped_t_on[:, :, None], 0.0), ped_dur[:, :, None])
    ped_l = np.where(ped_dir[:, :, None] < 0, road_ub - ped_v[:, :, None] * tau,
                     road_lb + ped_v[:, :, None] * tau)
    ped_live = (tt >= ped_t_on[:, :, None] - 1e-10) & (tt <= (ped_t_on + ped_dur)[:, :, None] + 1e-10)
    o_s = np.where(is_dyn, st_o[:, :, None] + veh_v[:, :, None] * tt, st_o[:, :, None] + 0.0 * tt)
    o_s = np.minimum(o_s, road.length - 1.0)
    o_l = np.where(is_ped, ped_l, np.where(is_dyn, veh_l[:, :, None], sta_l[:, :, None]) + 0.0 * tt)
    o_live = np.where(is_ped, ped_live, np.ones_like(ped_live))
    ox_c, oy_c, oth_c, _ = road.eval(o_s)
    o_x = ox_c - o_l * np.sin(oth_c)
    o_y = oy_c + o_l * np.cos(oth_c)
    o_th = np.where(is_ped, 0.0, oth_c)                            # pedestrians axis-aligned (publisher: theta=0)

    # ---- candidate lateral profiles: NT layer targets, piecewise linear in time ----
    rc = R[:, 8 + 6 * O:].reshape(B, C, NT)
    lay = np.empty((B, C, NT + 1))
    lay[:, :, 0] = l0[:, None]
    for j in range(NT):
        step = (rc[:, :, j] - 0.5) * 3.0                           # |dl| <= 1.5 m per layer
        if j == 0:
            step = step * np.linspace(0.0, 1.0, C)[None, :]        # candidate 0 keeps its lane
        lay[:, :, j + 1] = np.clip(lay[:, :, j] + step, -RIGHT_BOUND + 1.6, LEFT_BOUND - 1.6)
    lay[:, 0, 1:] = np.clip(l0[:, None], -RIGHT_BOUND + 1.6, LEFT_BOUND - 1.6)
    tl = np.linspace(0.0, T, NT + 1)
    seg = np.clip(np.searchsorted(tl, t, side="right") - 1, 0, NT - 1)
    w = (t - tl[seg]) / (tl[seg + 1] - tl[seg])
    ex_c, ey_c, eth_c, ekap_c = road.eval(s_t)                     # [B,K]
    sin_e, cos_e = np.sin(eth_c), np.cos(eth_c)
    clear = np.empty((C, B))
    o_clear_c = np.empty((C, O, B))
    l_c = np.empty((C, B, K))
    for c in range(C):
        l_c[c] = lay[:, c, seg] * (1.0 - w)[None, :] + lay[:, c, seg + 1] * w[None, :]
        px = ex_c - l_c[c] * sin_e
        py = ey_c + l_c[c] * cos_e
        qx, qy, inside = _rect_nearest(px[None], py[None], o_x, o_y, o_th, hl, hw)
        dist = np.hypot(qx - px[None], qy - py[None])
        dist = np.where(inside, 0.0, dist)
        dist = np.where(o_live, dist, 1e3)
        o_clear_c[c] = dist.min(axis=2)
        clear[c] = o_clear_c[c].min(axis=0)
    effort = np.abs(np.diff(lay, axis=2)).sum(axis=2).T            # [C,B]
    score = np.minimum(clear, 3.5) - 0.05 * effort
    best = score.argmax(axis=0)                                    # [B]
    bi = np.arange(B)
    # the DP planner only returns collision-free coarse paths (dp_planner.cpp:88-133): an obstacle
    # the chosen path cannot clear by MIN_CLEARANCE is dropped from the scene
    o_clear = o_clear_c[best, :, bi].T                             # [O,B]
    o_live = o_live & (o_clear >= spec.min_clearance)[:, :, None]
    l_t = l_c[best, bi]                                            # [B,K]
    x_t = ex_c - l_t * sin_e
    y_t = ey_c + l_t * cos_e

    # ---- heading / speed / accel / curvature profile (finite differences) ----
    dl = np.gradient(l_t, axis=1)
    ds = np.maximum(np.gradient(s_t, axis=1), 1e-10)
    theta = eth_c + np.arctan((dl / ds) / (1.0 - ekap_c * l_t))
    dxy = np.hypot(np.diff(x_t, axis=1), np.diff(y_t, axis=1))
    acc_s = np.concatenate([np.zeros((B, 1)), np.cumsum(dxy, axis=1)], axis=1)
    vel = np.gradient(acc_s, dt, axis=1)
    acc = np.gradient(vel, dt, axis=1)
    head = np.unwrap(np.arctan2(np.gradient(y_t, axis=1), np.gradient(x_t, axis=1)), axis=1)
    kappa = np.gradient(head, axis=1) / np.maximum(np.gradient(acc_s, axis=1), 1e-6)
    kappa = np.clip(kappa, -0.6, 0.6)
    delta = np.arctan(kappa * WHEEL_BASE)
    coarse = np.stack([x_t, y_t, theta, vel, acc, delta], axis=2)  # [B,K,6]
    start = np.stack([x_t[:, 0], y_t[:, 0], theta[:, 0] + th_noise, vel[:, 0]], axis=1)

    # ---- corridor: box (4 planes) + one separating plane per live obstacle, [P,B,K] ----
    Cm = spec.cmax
    P = 4 + O
    pa = np.zeros((P, B, K))
    pb = np.zeros((P, B, K))
    pc = np.zeros((P, B, K))
    valid = np.zeros((P, B, K), bool)
    ch, sh = np.cos(theta), np.sin(theta)
    half = 10.0
    for j, (nx, ny) in enumerate([(ch, sh), (-ch, -sh), (-sh, ch), (sh, -ch)]):
        scale = 2.0 * half                                         # edge length of the 20 m box
        pa[j] = nx * scale
        pb[j] = ny * scale
        pc[j] = (nx * x_t + ny * y_t + half) * scale
        valid[j] = True
    qx, qy, inside = _rect_nearest(x_t[None], y_t[None], o_x, o_y, o_th, hl, hw)
    ddx, ddy = qx - x_t[None], qy - y_t[None]
    d = np.hypot(ddx, ddy)
    ok = o_live & ~inside & (d > 1e-6) & (np.abs(ddx) <= 25.0) & (np.abs(ddy) <= 25.0)
    dn = np.where(ok, d, 1.0)
    nx, ny = ddx / dn, ddy / dn
    scale = 1.0 + 4.0 * (np.abs(np.sin(3.0 * o_x + o_y)))          # arbitrary |n| like a hull edge length
    pa[4:] = nx * scale
    pb[4:] = ny * scale
    pc[4:] = (nx * qx + ny * qy) * scale
    valid[4:] = ok
    planes = np.stack([pa, pb, pc], axis=-1).transpose(1, 2, 0, 3)  # [B,K,P,3]
    valid = valid.transpose(1, 2, 0)                                # [B,K,P]
    # live planes packed to the front (box first), at most cmax kept
    order = np.argsort(~valid, axis=2, kind="stable")
    planes = np.take_along_axis(planes, order[..., None], axis=2)
    valid = np.take_along_axis(valid, order, axis=2)
    ccount = np.minimum(valid.sum(axis=2), Cm).astype(np.int32)
    if planes.shape[2] < Cm:
        pad = np.zeros((B, K, Cm - planes.shape[2], 3))
        planes = np.concatenate([planes, pad], axis=2)
    planes = planes[:, :, :Cm]
    mask = np.arange(Cm)[None, None, :] < ccount[:, :, None]
    planes = np.where(mask[..., None], planes, 0.0)
    out = dict(start=np.ascontiguousarray(start), coarse=np.ascontiguousarray(coarse),
               corridor=np.ascontiguousarray(planes), ccount=np.ascontiguousarray(ccount))
    if want_points:
        # corners of every live obstacle at every knot time, [O,4,B,K] -> [B,K,4*O,2], live ones first
        co, so = np.cos(o_th), np.sin(o_th)
        cx = np.stack([o_x + sx * hl * co - sy * hw * so for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1))], axis=1)
        cy = np.stack([o_y + sx * hl * so + sy * hw * co for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1))], axis=1)
        pts = np.stack([cx, cy], axis=-1).reshape(O * 4, B, K, 2).transpose(1, 2, 0, 3)     # [B,K,4O,2]
        live = np.repeat(o_live, 4, axis=0).transpose(1, 2, 0)                               # [B,K,4O]
        order = np.argsort(~live, axis=2, kind="stable")
        pts = np.take_along_axis(pts, order[..., None], axis=2)
        cnt = live.sum(axis=2).astype(np.int32)
        pts = np.where((np.arange(O * 4)[None, None, :] < cnt[:, :, None])[..., None], pts, 0.0)
        out.update(obstacle_points=np.ascontiguousarray(pts), obstacle_count=np.ascontiguousarray(cnt))
    if want_scn:
        pose = np.stack([o_x, o_y, np.broadcast_to(o_th, o_x.shape)], axis=-1).transpose(1, 0, 2, 3)   # [B,O,K,3]
        out.update(obstacle_pose=np.ascontiguousarray(pose),
                   obstacle_live=np.ascontiguousarray(np.broadcast_to(o_live, o_x.shape).transpose(1, 0, 2)),
                   obstacle_half_size=np.stack([np.broadcast_to(hl, (O, 1, 1))[:, 0, 0], np.broadcast_to(hw, (O, 1, 1))[:, 0, 0]], axis=1),
                   obstacle_kind=kind.copy())
    return out


def generate_dp(spec: SceneSpec | str, batch: int, seed: int = 0, first_problem: int = 0, workers: int = 8,
                dp_config=None):
    """Scenes whose coarse trajectory comes from the DP coarse planner (the reference's own producer,
    algorithm/planner/dp_planner.cpp, through the C-ABI's cilqr_dp_plan) instead of this module's smooth
    best-clearance pick: every obstacle stays in the scene (the planner avoids them itself), the start state is the
    generator's, and the coarse trajectory carries the kinks of the 5-layer piecewise-linear (s, l) path.

    Returns dict(start[B,4], coarse[B,K,6] (x, y, theta, v, a, delta), dp[B,K,9] (the planner's full rows), found[B]
    bool ("DP failed" where False), obstacle_points[B,K,P,2], obstacle_count[B,K], left, right, n_steps, dt, cmax,
    scene_file).  There is no `corridor`: it is built from the obstacle points by cilqr_build_corridors
    (BatchIlqrOptimizer.build_corridors), as Corridor::Plan does behind the reference's DP."""
    from concurrent.futures import ThreadPoolExecutor
    from . import api, scene_io
    if isinstance(spec, str):
        spec = SPECS[spec]
    spec = dataclasses.replace(spec, min_clearance=-1.0)
    sc = generate(spec, batch, seed=seed, first_problem=first_problem, workers=workers, obstacle_points=True,
                  scenarios=True)
    sf = scene_io.from_generator(sc)
    cfg = dp_config or api.default_dp_config(tf=spec.n_steps * spec.dt, delta_t=spec.dt)

    def one(b):
        flat = scene_io.flatten_scene(sf.center, sf.scenes[b])
        return api.dp_plan(flat, sc["start"][b, :3], cfg)

    with ThreadPoolExecutor(max(1, workers)) as pool:
        outs = list(pool.map(one, range(batch)))
    dp = np.stack([o[1] for o in outs])
    found = np.array([o[0] for o in outs], dtype=bool)
    # a plan that stands still for a whole layer has 0 / 0 curvature (ComputePathProfile divides by the station
    # difference of neighbouring points): as unusable as a failed plan
    found &= np.isfinite(dp).all(axis=(1, 2))
    coarse = np.ascontiguousarray(dp[:, :, [2, 3, 4, 6, 7, 8]])        # x, y, theta, velocity, a, delta (cc:148)
    for b in range(batch):                                              # the scene file replays what was solved
        sf.scenes[b].coarse = coarse[b].copy()
    return dict(start=sc["start"], coarse=coarse, dp=dp, found=found, obstacle_points=sc["obstacle_points"],
                obstacle_count=sc["obstacle_count"], left=sc["left"], right=sc["right"], n_steps=spec.n_steps,
                dt=spec.dt, cmax=spec.cmax, scene_file=sf)
