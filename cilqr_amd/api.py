"""ctypes binding of the C-ABI in include/cilqr.h (cilqr_amd/lib/libcilqr_hip.so).

Host-side mirror of the reference's ``planning::IlqrOptimizer`` (algorithm/ilqr/
ilqr_optimizer.h:29-52) for a batch of problems.  There is no CPU path: if the HIP library is
missing, or no MI355X is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CILQR_LIB") or os.path.join(_HERE, "lib", "libcilqr_hip.so")   # CILQR_LIB: development override
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cilqr.h")

OK = 0
ERR_NULL, ERR_CONSTRAINTS, ERR_KNOTS, ERR_CAPACITY, ERR_DEVICE, ERR_ARG, ERR_STATE, ERR_NO_PATH = -1, -2, -3, -4, -5, -6, -7, -8
MEM_HOST, MEM_DEVICE = 0, 1
OPT_SPEC_THRESHOLD = 1
OPT_COMPACTION = 2
OPT_SEQ_ROUNDS = 3
OPT_TEAM_THRESHOLD = 4
OPT_TAIL_THRESHOLD = 5
OPT_WAVE_THRESHOLD = 6
OPT_ROUND_GROUP = 7
OPT_FINISH_THRESHOLD = 8
OPT_EXACT_LANE_TIES = 9
ST_RUNNING, ST_CONVERGED_ABS, ST_CONVERGED_REL, ST_GNORM, ST_UNSOLVED, ST_MAX_ITER, ST_NO_CORRIDOR = range(7)
ABI_VERSION = 6

T_GOALS, T_CORRIDOR, T_LANES, T_X, T_U, T_XCAND, T_UCAND, T_A, T_B, T_LX, T_LU, T_LXX, T_LUU, \
    T_KFB, T_KFF, T_DV, T_GNORM = range(17)

# dense fp64 scalars moved per problem-step / per problem by the backward pass
DENSE_DOUBLES_PER_STEP = 110
DENSE_DOUBLES_TERMINAL = 44
REAL_BYTES_PER_STEP = (18 + 7) * 16   # what k_backward really moves per problem-step: 17 pairs of `lin` + 1 of U in, 7 pairs of gains out


class Config(C.Structure):
    _fields_ = [
        ("n_steps", C.c_int32), ("num_of_disc", C.c_int32), ("max_iter", C.c_int32),
        ("init_guess", C.c_int32), ("dt", C.c_double), ("safe_margin", C.c_double),
        ("w_jerk", C.c_double), ("w_delta_rate", C.c_double), ("w_x", C.c_double),
        ("w_y", C.c_double), ("w_theta", C.c_double), ("w_v", C.c_double), ("w_a", C.c_double),
        ("w_delta", C.c_double), ("abs_cost_tol", C.c_double), ("rel_cost_tol", C.c_double),
        ("front_hang", C.c_double), ("wheel_base", C.c_double), ("rear_hang", C.c_double),
        ("width", C.c_double), ("max_velocity", C.c_double), ("min_acceleration", C.c_double),
        ("max_acceleration", C.c_double), ("jerk_min", C.c_double), ("jerk_max", C.c_double),
        ("delta_min", C.c_double), ("delta_max", C.c_double), ("delta_rate_min", C.c_double),
        ("delta_rate_max", C.c_double), ("barrier_t", C.c_double), ("barrier_eps", C.c_double),
    ]


class ProblemBatch(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n_knots", C.c_int32), ("cmax", C.c_int32), ("memory", C.c_int32),
        ("start", C.c_void_p), ("coarse", C.c_void_p), ("corridor", C.c_void_p),
        ("corridor_count", C.c_void_p), ("n_left", C.c_int32), ("n_right", C.c_int32),
        ("left_lane", C.c_void_p), ("right_lane", C.c_void_p),
        ("n_lane_groups", C.c_int32), ("reserved1", C.c_int32), ("lane_group_start", C.c_void_p),
        ("lane_group_left", C.c_void_p), ("lane_group_right", C.c_void_p), ("coarse_station", C.c_void_p),
    ]


class SolutionBatch(C.Structure):
    _fields_ = [
        ("memory", C.c_int32), ("max_iter_trajs", C.c_int32), ("traj", C.c_void_p),
        ("cost_hist", C.c_void_p), ("n_cost", C.c_void_p), ("status", C.c_void_p),
        ("n_iter", C.c_void_p), ("iter_trajs", C.c_void_p), ("n_iter_trajs", C.c_void_p),
        ("alpha_trace", C.c_void_p),
    ]


class CorridorConfig(C.Structure):
    """CorridorConfig of the reference (algorithm/params/planner_config.h:75-86)."""
    _fields_ = [("max_diff_x", C.c_double), ("max_diff_y", C.c_double), ("radius", C.c_double),
                ("max_axis_x", C.c_double), ("max_axis_y", C.c_double), ("lane_segment_length", C.c_double),
                ("is_multiple_sample", C.c_int32), ("reserved0", C.c_int32)]


class DpConfig(C.Structure):
    """Live fields of PlannerConfig / VehicleParam for the DP coarse planner (include/cilqr.h)."""
    _fields_ = [(n, C.c_double) for n in (
        "tf", "delta_t", "dp_nominal_velocity", "dp_w_obstacle", "dp_w_lateral", "dp_w_lateral_change",
        "dp_w_lateral_velocity_change", "dp_w_longitudinal_velocity_bias", "dp_w_longitudinal_velocity_change",
        "front_hang_length", "wheel_base", "rear_hang_length", "width", "max_velocity")]


class SceneStruct(C.Structure):
    _fields_ = [("center", C.c_void_p), ("n_center", C.c_int32), ("n_static", C.c_int32),
                ("static_points", C.c_void_p), ("static_counts", C.c_void_p), ("n_dynamic", C.c_int32),
                ("reserved0", C.c_int32), ("dynamic_polygon_points", C.c_void_p), ("dynamic_polygon_counts", C.c_void_p),
                ("dynamic_trajectories", C.c_void_p), ("dynamic_trajectory_counts", C.c_void_p)]


COARSE_FIELDS = 9   # time, s, x, y, theta, kappa, velocity, a, delta


class TrackerConfig(C.Structure):
    """TrackerConfig of the reference (algorithm/params/planner_config.h:18-43) for init_guess = INIT_TRACKER."""
    _fields_ = [(n, C.c_double) for n in ("weight_l", "weight_theta", "weight_delta", "weight_delta_rate", "preview_time",
                                          "weight_s", "weight_v", "weight_a", "weight_j", "sumulation_dt", "dt", "tolerance")] + \
               [("max_num_iteration", C.c_int32), ("reserved0", C.c_int32)]


INIT_IQR, INIT_TRACKER = 0, 1


class Profile(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("backward_launches", C.c_int32), ("backward_ms", C.c_double),
        ("quadratize_ms", C.c_double), ("linesearch_ms", C.c_double), ("other_ms", C.c_double),
        ("total_ms", C.c_double), ("backward_problem_steps", C.c_int64),
        ("backward_full_launches", C.c_int32), ("tail_problems", C.c_int32), ("backward_full_ms", C.c_double),
        ("tail_ms", C.c_double),
    ]


EXPORTS = [
    "cilqr_abi_version", "cilqr_build_id", "cilqr_default_config", "cilqr_create", "cilqr_destroy", "cilqr_set_stream",
    "cilqr_set_option", "cilqr_get_option", "cilqr_set_profiling", "cilqr_get_profile", "cilqr_device_bytes", "cilqr_solve_batch",
    "cilqr_submit", "cilqr_wait", "cilqr_device_math", "cilqr_stage_load", "cilqr_stage_init_guess", "cilqr_stage_set_trajectory",
    "cilqr_stage_total_cost", "cilqr_stage_quadratize", "cilqr_stage_backward", "cilqr_stage_forward",
    "cilqr_stage_read", "cilqr_stage_nearest_lane", "cilqr_open_loop_rollout", "cilqr_error_string",
    "cilqr_default_corridor_config", "cilqr_build_corridors", "cilqr_lane_constraints",
    "cilqr_default_dp_config", "cilqr_dp_plan", "cilqr_road_barriers", "cilqr_default_tracker_config",
    "cilqr_set_tracker_config",
    "cilqr_multi_create", "cilqr_multi_destroy", "cilqr_multi_solve", "cilqr_multi_set_option", "cilqr_multi_shards",
    "cilqr_multi_device_bytes",
    "cilqr_pool_create", "cilqr_pool_destroy", "cilqr_pool_submit", "cilqr_pool_wait", "cilqr_pool_depth", "cilqr_pool_handle_at",
    "cilqr_pool_set_option", "cilqr_pool_get_profile", "cilqr_pool_device_bytes",
    "cilqr_comm_unique_id", "cilqr_comm_create", "cilqr_comm_destroy", "cilqr_comm_info", "cilqr_gather_results",
]
UNIQUE_ID_BYTES = 128

_LIB = None


class CilqrError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().cilqr_error_string(code).decode() if _LIB is not None else str(code)
        super().__init__(f"cilqr error {code} ({msg}) {what}")


def lib():
    """Load the HIP library; raises if it was not built (no fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                               "(make -C cilqr_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.cilqr_error_string.restype = C.c_char_p
        L.cilqr_device_bytes.restype = C.c_int64
        L.cilqr_device_bytes.argtypes = [C.c_void_p]
        L.cilqr_create.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_void_p)]
        L.cilqr_destroy.argtypes = [C.c_void_p]
        L.cilqr_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.cilqr_set_profiling.argtypes = [C.c_void_p, C.c_int32]
        L.cilqr_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.cilqr_get_option.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cilqr_get_option.restype = C.c_int
        L.cilqr_get_profile.argtypes = [C.c_void_p, C.POINTER(Profile)]
        L.cilqr_solve_batch.argtypes = [C.c_void_p, C.POINTER(ProblemBatch), C.POINTER(SolutionBatch)]
        L.cilqr_submit.argtypes = [C.c_void_p, C.POINTER(ProblemBatch), C.POINTER(SolutionBatch)]
        L.cilqr_wait.argtypes = [C.c_void_p]
        L.cilqr_default_corridor_config.argtypes = [C.POINTER(CorridorConfig)]
        L.cilqr_default_corridor_config.restype = None
        L.cilqr_build_corridors.argtypes = [C.c_void_p, C.POINTER(CorridorConfig), C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
        L.cilqr_lane_constraints.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_int32]
        L.cilqr_device_math.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.cilqr_stage_load.argtypes = [C.c_void_p, C.POINTER(ProblemBatch)]
        L.cilqr_stage_init_guess.argtypes = [C.c_void_p]
        L.cilqr_stage_set_trajectory.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.cilqr_stage_total_cost.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.cilqr_stage_quadratize.argtypes = [C.c_void_p]
        L.cilqr_stage_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.cilqr_stage_forward.argtypes = [C.c_void_p, C.c_double]
        L.cilqr_stage_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.cilqr_stage_nearest_lane.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int32, C.c_int32]
        L.cilqr_open_loop_rollout.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int32]
        L.cilqr_default_dp_config.argtypes = [C.POINTER(DpConfig)]
        L.cilqr_default_dp_config.restype = None
        L.cilqr_dp_plan.argtypes = [C.POINTER(DpConfig), C.POINTER(SceneStruct), C.c_void_p, C.c_void_p, C.c_int32]
        L.cilqr_default_tracker_config.argtypes = [C.POINTER(TrackerConfig)]
        L.cilqr_default_tracker_config.restype = None
        L.cilqr_set_tracker_config.argtypes = [C.c_void_p, C.POINTER(TrackerConfig)]
        L.cilqr_road_barriers.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.cilqr_multi_create.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_void_p)]
        L.cilqr_multi_destroy.argtypes = [C.c_void_p]
        L.cilqr_multi_solve.argtypes = [C.c_void_p, C.POINTER(ProblemBatch), C.POINTER(SolutionBatch)]
        L.cilqr_multi_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.cilqr_multi_shards.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.cilqr_multi_device_bytes.argtypes = [C.c_void_p]
        L.cilqr_multi_device_bytes.restype = C.c_int64
        L.cilqr_pool_create.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(C.c_void_p)]
        L.cilqr_pool_destroy.argtypes = [C.c_void_p]
        L.cilqr_pool_submit.argtypes = [C.c_void_p, C.POINTER(ProblemBatch), C.POINTER(SolutionBatch)]
        L.cilqr_pool_wait.argtypes = [C.c_void_p]
        L.cilqr_pool_depth.argtypes = [C.c_void_p]
        L.cilqr_pool_depth.restype = C.c_int32
        L.cilqr_pool_handle_at.argtypes = [C.c_void_p, C.c_int32]
        L.cilqr_pool_handle_at.restype = C.c_void_p
        L.cilqr_pool_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.cilqr_pool_get_profile.argtypes = [C.c_void_p, C.POINTER(Profile)]
        L.cilqr_pool_device_bytes.argtypes = [C.c_void_p]
        L.cilqr_pool_device_bytes.restype = C.c_int64
        L.cilqr_comm_unique_id.argtypes = [C.c_void_p]
        L.cilqr_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.cilqr_comm_destroy.argtypes = [C.c_void_p]
        L.cilqr_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.cilqr_gather_results.argtypes = [C.c_void_p, C.c_int32, C.POINTER(SolutionBatch), C.c_int32,
                                           C.POINTER(SolutionBatch)]
        _LIB = L
    return _LIB


def default_config(n_steps: int = 50, **over) -> Config:
    c = Config()
    rc = lib().cilqr_default_config(C.byref(c), C.c_int32(n_steps))
    if rc != OK:
        raise CilqrError(rc)
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class BatchIlqrOptimizer:
    """Batched drop-in of IlqrOptimizer: ``plan`` = Plan + cost() for B problems on one GPU."""

    def __init__(self, cfg: Config | None = None, n_steps: int = 50, device: int = 0,
                 batch_capacity: int = 1, cmax: int = 16, max_lane_segments: int = 64, adopt=None):
        """adopt: an existing cilqr_handle (HandlePool.handle_at) to wrap instead of creating one; never destroyed here."""
        self.cfg = cfg or default_config(n_steps)
        self.N = self.cfg.n_steps
        self.K = self.N + 1
        self.cmax = cmax
        self.capacity = batch_capacity
        self.L = lib()
        self.owned = adopt is None
        if adopt is not None:
            self.h = C.c_void_p(adopt)
        else:
            self.h = C.c_void_p()
            rc = self.L.cilqr_create(C.byref(self.cfg), device, batch_capacity, cmax, max_lane_segments,
                                     C.byref(self.h))
            if rc != OK:
                self.h = C.c_void_p()
                raise CilqrError(rc, "in cilqr_create")
        self.B = 0
        self.nl = self.nr = 0

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            if self.owned:
                self.L.cilqr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- raw-pointer interface (device or host memory) ----
    def make_problem(self, B, start, coarse, corridor, ccount, cmax, left, right, n_left, n_right,
                     memory) -> ProblemBatch:
        return ProblemBatch(B, self.K, cmax, memory, start, coarse, corridor, ccount, n_left, n_right,
                            left, right)

    def set_stream(self, stream_ptr):
        rc = self.L.cilqr_set_stream(self.h, C.c_void_p(stream_ptr))
        if rc != OK:
            raise CilqrError(rc)

    def set_option(self, option: int, value: int):
        rc = self.L.cilqr_set_option(self.h, option, value)
        if rc != OK:
            raise CilqrError(rc, "in cilqr_set_option")

    def get_option(self, option: int):
        """(value for cilqr_solve_batch, value for submitted solves) of a cilqr_set_option code"""
        a, b = C.c_int64(0), C.c_int64(0)
        rc = self.L.cilqr_get_option(self.h, option, C.byref(a), C.byref(b))
        if rc != OK:
            raise CilqrError(rc, "in cilqr_get_option")
        return int(a.value), int(b.value)

    def set_tracker_config(self, **over):
        """Override fields of the tracker init guess's configuration (reference defaults otherwise)."""
        c = TrackerConfig()
        self.L.cilqr_default_tracker_config(C.byref(c))
        for k, v in over.items():
            setattr(c, k, v)
        self._chk(self.L.cilqr_set_tracker_config(self.h, C.byref(c)), "set_tracker_config")

    def set_profiling(self, on):
        """False / 0: off; True / 1: every phase of every iteration; 2: the backward launches only."""
        self.L.cilqr_set_profiling(self.h, int(on))

    def profile(self) -> Profile:
        p = Profile()
        self.L.cilqr_get_profile(self.h, C.byref(p))
        return p

    def device_bytes(self) -> int:
        return int(self.L.cilqr_device_bytes(self.h))

    def solve_raw(self, prob: ProblemBatch, sol: SolutionBatch) -> int:
        return self.L.cilqr_solve_batch(self.h, C.byref(prob), C.byref(sol))

    def submit_raw(self, prob: ProblemBatch, sol: SolutionBatch) -> int:
        """Asynchronous solve_raw; collect the result code with wait()."""
        return self.L.cilqr_submit(self.h, C.byref(prob), C.byref(sol))

    def wait(self) -> int:
        return self.L.cilqr_wait(self.h)

    def device_math(self, fn: int, x) -> np.ndarray:
        """Test hook: the kernels' lean log (fn 0, 2) / reciprocal (fn 1) on a host array."""
        x = _f64(x).ravel()
        out = np.empty_like(x)
        self._chk(self.L.cilqr_device_math(self.h, fn, x.size, _ptr(x), _ptr(out)), "device_math")
        return out

    # ---- multi-GPU: one process per GPU, one RCCL gather of the results (include/cilqr.h) ----
    def comm_create(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._chk(self.L.cilqr_comm_create(self.h, buf, rank, world), "comm_create")

    def comm_destroy(self):
        self._chk(self.L.cilqr_comm_destroy(self.h), "comm_destroy")

    def gather_results_raw(self, batch: int, local: SolutionBatch, root: int, gathered: "SolutionBatch | None") -> int:
        return self.L.cilqr_gather_results(self.h, batch, C.byref(local), root,
                                           C.byref(gathered) if gathered is not None else None)

    # ---- numpy (host memory) interface ----
    def _host_problem(self, scene: dict):
        a = dict(start=_f64(scene["start"]), coarse=_f64(scene["coarse"]),
                 corridor=_f64(scene["corridor"]),
                 ccount=np.ascontiguousarray(scene["ccount"], dtype=np.int32),
                 left=_f64(scene["left"]), right=_f64(scene["right"]))
        B = a["coarse"].shape[0]
        K = a["coarse"].shape[1] if a["coarse"].ndim == 3 else 0
        cmax = a["corridor"].shape[2] if a["corridor"].ndim == 4 else 0
        prob = ProblemBatch(B, K, cmax, MEM_HOST, _ptr(a["start"]), _ptr(a["coarse"]),
                            _ptr(a["corridor"]) if a["corridor"].size else None,
                            _ptr(a["ccount"]) if a["ccount"].size else None,
                            a["left"].shape[0], a["right"].shape[0],
                            _ptr(a["left"]) if a["left"].size else None,
                            _ptr(a["right"]) if a["right"].size else None)
        if scene.get("coarse_station") is not None:      # stations of the coarse points (tracker init guess)
            a["station"] = _f64(scene["coarse_station"])
            prob.coarse_station = a["station"].ctypes.data
        groups = scene.get("lane_groups")
        if groups is not None:
            # per-problem lane tables: [(first problem, left rows, right rows), ...]; scene["left"] / ["right"] hold the
            # groups' tables back to back
            a["g_start"] = np.asarray([g[0] for g in groups] + [B], dtype=np.int32)
            a["g_left"] = np.asarray([g[1] for g in groups], dtype=np.int32)
            a["g_right"] = np.asarray([g[2] for g in groups], dtype=np.int32)
            prob.n_lane_groups = len(groups)
            prob.lane_group_start = a["g_start"].ctypes.data
            prob.lane_group_left = a["g_left"].ctypes.data
            prob.lane_group_right = a["g_right"].ctypes.data
        return prob, a

    def plan(self, scene: dict, max_iter_trajs: int = 0, check: bool = True, alpha_trace: bool = False):
        """scene: dict from cilqr_amd.scenario.generate (problem-major numpy arrays).
        alpha_trace=True adds "alpha_trace" [B, max_iter] int8: the accepted step-size index of every
        iteration (-1 all rejected, -2 left before the line search, -3 not run)."""
        prob, keep = self._host_problem(scene)
        B, K, M = prob.batch, self.K, self.cfg.max_iter
        traj = np.zeros((B, K, 10))
        hist = np.zeros((B, M + 1, 5))
        n_cost = np.zeros(B, np.int32)
        status = np.zeros(B, np.int32)
        n_iter = np.zeros(B, np.int32)
        it = np.zeros((B, max_iter_trajs, K, 10)) if max_iter_trajs else None
        n_it = np.zeros(B, np.int32) if max_iter_trajs else None
        at = np.full((B, M), -3, np.int8) if alpha_trace else None
        sol = SolutionBatch(MEM_HOST, max_iter_trajs, _ptr(traj), _ptr(hist), _ptr(n_cost), _ptr(status),
                            _ptr(n_iter), _ptr(it), _ptr(n_it), _ptr(at))
        rc = self.solve_raw(prob, sol)
        del keep
        if rc != OK:
            if check:
                raise CilqrError(rc, "in cilqr_solve_batch")
            return dict(rc=rc)
        self.B = B
        return dict(rc=rc, traj=traj, cost_hist=hist, n_cost=n_cost, status=status, n_iter=n_iter,
                    iter_trajs=it, n_iter_trajs=n_it, alpha_trace=at)

    # ---- stages ----
    def _chk(self, rc, what):
        if rc != OK:
            raise CilqrError(rc, what)

    def stage_load(self, scene: dict):
        prob, keep = self._host_problem(scene)
        self._chk(self.L.cilqr_stage_load(self.h, C.byref(prob)), "stage_load")
        self.B = prob.batch
        self.nl, self.nr = prob.n_left, prob.n_right

    def stage_init_guess(self):
        self._chk(self.L.cilqr_stage_init_guess(self.h), "stage_init_guess")

    def stage_set_trajectory(self, X, U):
        X, U = _f64(X), _f64(U)
        self._chk(self.L.cilqr_stage_set_trajectory(self.h, _ptr(X), _ptr(U), MEM_HOST), "set_trajectory")

    def stage_total_cost(self):
        c = np.zeros((self.B, 5))
        self._chk(self.L.cilqr_stage_total_cost(self.h, _ptr(c), MEM_HOST), "total_cost")
        return c

    def stage_quadratize(self):
        self._chk(self.L.cilqr_stage_quadratize(self.h), "quadratize")

    def stage_backward(self, lam=None):
        if lam is not None:
            lam = _f64(np.broadcast_to(lam, (self.B,)))
        self._chk(self.L.cilqr_stage_backward(self.h, _ptr(lam), MEM_HOST), "backward")

    def stage_forward(self, alpha: float):
        self._chk(self.L.cilqr_stage_forward(self.h, C.c_double(alpha)), "forward")

    def read(self, tensor: int):
        B, N, K = self.B, self.N, self.K
        shape = {
            T_GOALS: (B, K, 6), T_CORRIDOR: (B, K, self.cmax, 3), T_LANES: (self.nl + self.nr, 3),
            T_X: (B, K, 6), T_U: (B, N, 2), T_XCAND: (B, K, 6), T_UCAND: (B, N, 2),
            T_A: (B, N, 6, 6), T_B: (B, N, 6, 2), T_LX: (B, K, 6), T_LU: (B, N, 2),
            T_LXX: (B, K, 6, 6), T_LUU: (B, N, 2, 2), T_KFB: (B, N, 2, 6), T_KFF: (B, N, 2),
            T_DV: (B, 2), T_GNORM: (B,),
        }[tensor]
        out = np.zeros(shape)
        self._chk(self.L.cilqr_stage_read(self.h, tensor, _ptr(out), MEM_HOST), f"read({tensor})")
        return out

    def nearest_lane(self, xy, use_grid: bool = True):
        xy = _f64(xy)
        n = xy.shape[0]
        left, right = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._chk(self.L.cilqr_stage_nearest_lane(self.h, n, _ptr(xy), _ptr(left), _ptr(right),
                                                  1 if use_grid else 0, MEM_HOST), "nearest_lane")
        return left, right

    def build_corridors(self, knots, points, point_count, cmax: int = 16, cfg: "CorridorConfig | None" = None,
                        want_polygons: bool = False):
        """Corridor::BuildCorridorConstraints for a batch (corridor.cc:58-87): knots [B,K,3] = x, y, theta,
        points [B,K,P,2] obstacle corner points per knot, point_count [B,K].  Returns (corridor [B,K,cmax,3],
        corridor_count [B,K] int32 -- negative where a corridor could not be built --, n_failed)."""
        knots = _f64(knots)
        B, K = knots.shape[:2]
        points = _f64(points).reshape(B, K, -1, 2)
        P = points.shape[2]
        cnt = np.ascontiguousarray(point_count, dtype=np.int32).reshape(B, K)
        cfg = cfg or default_corridor_config()
        cor = np.zeros((B, K, cmax, 3))
        ccnt = np.zeros((B, K), dtype=np.int32)
        nf = C.c_int32(0)
        poly = np.zeros((B, K, cmax, 2)) if want_polygons else None
        self._chk(self.L.cilqr_build_corridors(self.h, C.byref(cfg), B, K, _ptr(knots), _ptr(points) if P else None,
                                               cnt.ctypes.data, P, _ptr(cor), ccnt.ctypes.data, cmax, MEM_HOST,
                                               C.byref(nf), _ptr(poly) if want_polygons else None), "build_corridors")
        if want_polygons:
            return cor, ccnt, int(nf.value), poly
        return cor, ccnt, int(nf.value)

    def build_corridors_raw(self, cfg, batch, n_knots, knots_ptr, points_ptr, count_ptr, max_points, corridor_ptr,
                            ccount_ptr, cmax, memory, polygons_ptr=None):
        """Pointer-level form (device or host memory); returns (rc, n_failed)."""
        nf = C.c_int32(0)
        rc = self.L.cilqr_build_corridors(self.h, C.byref(cfg), batch, n_knots, knots_ptr, points_ptr, count_ptr,
                                          max_points, corridor_ptr, ccount_ptr, cmax, memory, C.byref(nf),
                                          polygons_ptr)
        return rc, int(nf.value)

    def open_loop_rollout(self, x0, U):
        x0, U = _f64(x0), _f64(U)
        B = x0.shape[0]
        X = np.zeros((B, self.K, 6))
        self._chk(self.L.cilqr_open_loop_rollout(self.h, B, _ptr(x0), _ptr(U), _ptr(X), MEM_HOST), "rollout")
        return X


def default_dp_config(**over) -> DpConfig:
    c = DpConfig()
    lib().cilqr_default_dp_config(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def dp_plan(flat: dict, start3, cfg: "DpConfig | None" = None):
    """DpPlanner::Plan through the C-ABI (host only).  `flat` = cilqr_amd.scene_io.flatten_scene(center, scene);
    returns (found, coarse [K, 9] = time s x y theta kappa velocity a delta); found = False is the
    reference's "DP failed" (every sampled path collides), coarse is filled all the same."""
    cfg = cfg or default_dp_config()
    K = max(1, int(cfg.tf / cfg.delta_t + 1)) if cfg.delta_t > 0 else 1
    keep = {k: np.ascontiguousarray(v) for k, v in flat.items()}
    sc = SceneStruct(keep["center"].ctypes.data, keep["center"].shape[0], len(keep["static_counts"]),
                     keep["static_points"].ctypes.data, keep["static_counts"].ctypes.data,
                     len(keep["dynamic_polygon_counts"]), 0, keep["dynamic_polygon_points"].ctypes.data,
                     keep["dynamic_polygon_counts"].ctypes.data, keep["dynamic_trajectories"].ctypes.data,
                     keep["dynamic_trajectory_counts"].ctypes.data)
    start = _f64(start3)
    coarse = np.zeros((K, COARSE_FIELDS))
    rc = lib().cilqr_dp_plan(C.byref(cfg), C.byref(sc), start.ctypes.data, coarse.ctypes.data, K)
    if rc not in (OK, ERR_NO_PATH):
        raise CilqrError(rc, "in cilqr_dp_plan")
    return rc == OK, coarse


def road_barriers(center):
    """Environment::set_reference (environment.cpp:20-43): left / right road barriers [n, 2] sampled every 0.1 m of
    station from the centre line [m, 7] -- what Corridor::Plan builds its lane constraints from."""
    c = _f64(center)
    cap = int((c[-1, 0] - c[0, 0]) / 0.1) + 8
    left, right = np.zeros((cap, 2)), np.zeros((cap, 2))
    n = lib().cilqr_road_barriers(c.ctypes.data, c.shape[0], left.ctypes.data, right.ctypes.data, cap)
    if n < 0:
        raise CilqrError(n, "in cilqr_road_barriers")
    return left[:n].copy(), right[:n].copy()


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C-ABI (rank 0; ship the 128 bytes to the other ranks)."""
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    rc = lib().cilqr_comm_unique_id(buf)
    if rc != OK:
        raise CilqrError(rc, "in cilqr_comm_unique_id")
    return bytes(buf)


def default_corridor_config() -> CorridorConfig:
    c = CorridorConfig()
    lib().cilqr_default_corridor_config(C.byref(c))
    return c


def lane_constraints(boundary, segment_length: float = 5.0, is_left: bool = True) -> np.ndarray:
    """LaneBoundarySample + Cal{Left,Right}LaneConstraints (corridor.cc:265-321), host only:
    boundary [n,2] -> rows [m,7] = a b c sx sy ex ey (the left_lane / right_lane layout)."""
    b = _f64(np.asarray(boundary, dtype=np.float64).reshape(-1, 2))
    rows = np.zeros((max(1, b.shape[0]), 7))
    m = lib().cilqr_lane_constraints(b.ctypes.data, b.shape[0], float(segment_length), int(bool(is_left)),
                                     rows.ctypes.data, rows.shape[0])
    if m < 0:
        raise CilqrError(m, "lane_constraints")
    return rows[:m].copy()


class HandlePool:
    """cilqr_pool_*: a stream of batches on ONE GPU through several handles dealt out round-robin (include/cilqr.h).
    submit_raw() up to depth() solves, wait() collects the oldest; results bit-identical to BatchIlqrOptimizer."""

    def __init__(self, cfg: "Config | None" = None, device: int = 0, handles: int = 3, batch_capacity: int = 1, cmax: int = 16,
                 max_lane_segments: int = 64, n_steps: int = 50):
        self.L = lib()
        self.cfg = cfg if cfg is not None else default_config(n_steps)
        self.K = self.cfg.n_steps + 1
        self.h = C.c_void_p()
        self._adopted = []
        rc = self.L.cilqr_pool_create(C.byref(self.cfg), device, handles, batch_capacity, cmax, max_lane_segments, C.byref(self.h))
        if rc != OK:
            self.h = C.c_void_p()
            raise CilqrError(rc, "in cilqr_pool_create")

    def close(self):
        """Destroys the pool (n_handles x the device memory of one handle) and invalidates the wrappers handle_at() gave out."""
        if getattr(self, "h", None) is not None and self.h.value:
            for w in self._adopted:
                w.h = C.c_void_p()       # the handle dies with the pool: a stale wrapper must not touch it
            self._adopted = []
            self.L.cilqr_pool_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def depth(self) -> int:
        return int(self.L.cilqr_pool_depth(self.h))

    def handle_at(self, k: int, **kw) -> "BatchIlqrOptimizer":
        """handle k of the pool as a BatchIlqrOptimizer (not owned; valid while the pool lives and nothing is in flight on it)"""
        h = self.L.cilqr_pool_handle_at(self.h, k)
        if not h:
            raise CilqrError(ERR_STATE, "in cilqr_pool_handle_at")
        w = BatchIlqrOptimizer(self.cfg, adopt=h, **kw)
        self._adopted.append(w)
        return w

    def set_option(self, option: int, value: int):
        rc = self.L.cilqr_pool_set_option(self.h, option, value)
        if rc != OK:
            raise CilqrError(rc, "in cilqr_pool_set_option")

    def device_bytes(self) -> int:
        return int(self.L.cilqr_pool_device_bytes(self.h))

    def submit_raw(self, prob: ProblemBatch, sol: SolutionBatch) -> int:
        return self.L.cilqr_pool_submit(self.h, C.byref(prob), C.byref(sol))

    def wait(self) -> int:
        return self.L.cilqr_pool_wait(self.h)

    def profile(self) -> Profile:
        """of the solve the last wait() collected (CilqrError ERR_STATE before the first wait)"""
        p = Profile()
        rc = self.L.cilqr_pool_get_profile(self.h, C.byref(p))
        if rc != OK:
            raise CilqrError(rc, "in cilqr_pool_get_profile")
        return p


class MultiDeviceOptimizer:
    """cilqr_multi_*: ONE host process, several GPUs (include/cilqr.h).  The batch of a plan() call is cut into
    contiguous shards, one per entry of `devices` (an entry may repeat: logical shards on one GPU), solved
    concurrently, and every shard writes its rows of the caller's arrays: results in problem order, bit-identical to
    one BatchIlqrOptimizer.plan over the whole batch."""

    def __init__(self, cfg: "Config | None" = None, devices=(0,), batch_capacity: int = 1, cmax: int = 16,
                 max_lane_segments: int = 64, n_steps: int = 50):
        self.L = lib()
        self.cfg = cfg if cfg is not None else default_config(n_steps)
        self.K = self.cfg.n_steps + 1
        self.devices = np.asarray(list(devices), dtype=np.int32)
        self.h = C.c_void_p()
        rc = self.L.cilqr_multi_create(C.byref(self.cfg), self.devices.ctypes.data, len(self.devices), batch_capacity, cmax,
                                       max_lane_segments, C.byref(self.h))
        if rc != OK:
            self.h = C.c_void_p()
            raise CilqrError(rc, "in cilqr_multi_create")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.cilqr_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def set_option(self, option: int, value: int):
        rc = self.L.cilqr_multi_set_option(self.h, option, value)
        if rc != OK:
            raise CilqrError(rc, "in cilqr_multi_set_option")

    def shards(self, batch: int):
        n = len(self.devices)
        first, dev = np.zeros(n, np.int32), np.zeros(n, np.int32)
        got = self.L.cilqr_multi_shards(self.h, batch, first.ctypes.data, dev.ctypes.data, n)
        return got, first, dev

    def device_bytes(self) -> int:
        return int(self.L.cilqr_multi_device_bytes(self.h))

    def solve_raw(self, prob: ProblemBatch, sol: SolutionBatch) -> int:
        return self.L.cilqr_multi_solve(self.h, C.byref(prob), C.byref(sol))

    def plan(self, scene: dict, max_iter_trajs: int = 0, check: bool = True, alpha_trace: bool = False):
        prob, keep = BatchIlqrOptimizer._host_problem(self, scene)
        B, K, M = prob.batch, self.K, self.cfg.max_iter
        traj = np.zeros((B, K, 10))
        hist = np.zeros((B, M + 1, 5))
        n_cost = np.zeros(B, np.int32)
        status = np.zeros(B, np.int32)
        n_iter = np.zeros(B, np.int32)
        it = np.zeros((B, max_iter_trajs, K, 10)) if max_iter_trajs else None
        n_it = np.zeros(B, np.int32) if max_iter_trajs else None
        at = np.full((B, M), -3, np.int8) if alpha_trace else None
        sol = SolutionBatch(MEM_HOST, max_iter_trajs, _ptr(traj), _ptr(hist), _ptr(n_cost), _ptr(status),
                            _ptr(n_iter), _ptr(it), _ptr(n_it), _ptr(at))
        rc = self.solve_raw(prob, sol)
        del keep
        if rc != OK:
            if check:
                raise CilqrError(rc, "in cilqr_multi_solve")
            return dict(rc=rc)
        return dict(rc=rc, traj=traj, cost_hist=hist, n_cost=n_cost, status=status, n_iter=n_iter,
                    iter_trajs=it, n_iter_trajs=n_it, alpha_trace=at)
