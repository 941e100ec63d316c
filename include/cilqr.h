/*
 * include/cilqr.h -- C-ABI of the MI355X-native batched CILQR trajectory optimiser.
 *
 * Drop-in boundary for the reference's `planning::IlqrOptimizer`
 * (algorithm/ilqr/ilqr_optimizer.h:29-52) and its stages.  Plain pointers and sizes only; no
 * C++/torch types cross this boundary.  Every entry point returns 0 (CILQR_OK) or a negative
 * error code and never throws.
 *
 * The reference has no FFI/plugin layer; what each entry replaces:
 *   cilqr_default_config    IlqrConfig/Weights/VehicleParam default member initialisers
 *                           (algorithm/params/planner_config.h:45-73, vehicle_param.h:21-64),
 *                           RelaxBarrierFunction t/epsilon (algorithm/ilqr/barrier_function.h:144-145)
 *   cilqr_create/destroy    IlqrOptimizer::IlqrOptimizer / Init   (ilqr_optimizer.cc:13-51)
 *   cilqr_solve_batch       IlqrOptimizer::Plan + cost()          (ilqr_optimizer.cc:53-95, .h:50-52),
 *                           B independent problems per call
 *   cilqr_stage_*           the private stages of Optimize()      (ilqr_optimizer.cc:154-320):
 *       load            TransformGoals cc:141, ShrinkConstraints cc:438, NormalizeHalfPlane cc:475
 *       init_guess      iqr cc:793-842
 *       total_cost      TotalCost cc:417-436
 *       quadratize      DynamicsJacbian vehicle_model.cc:21 + CostJacbian cc:620 + CostHessian cc:638
 *       backward        Backward cc:334-390 (+ CalGradientNorm cc:322)
 *       forward         Forward cc:392-415
 *       nearest_lane    FindNeastLaneSegment cc:605-618 + LineSegment2d::DistanceTo (algorithm/math/line_segment2d.cpp:61-75)
 *   cilqr_open_loop_rollout ilqr::iLQR::OpenLoopRollout (algorithm/slover/ilqr.h:363-370) on
 *                           VehicleModel::Dynamics (vehicle_model.cc:88-121)
 *
 * Layout convention: every per-problem array is problem-major ("[B][...]"), IEEE fp64,
 * in host or device memory as flagged by `memory`.
 */
#ifndef CILQR_H_
#define CILQR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CILQR_ABI_VERSION 6

#define CILQR_NX 6  /* state  (x, y, theta, v, a, delta)   vehicle_model.h:11 */
#define CILQR_NU 2  /* control (jerk, delta_rate)           vehicle_model.h:12 */
#define CILQR_TRAJ_FIELDS 10 /* time,x,y,theta,v,a,delta,kappa,jerk,delta_rate  (cc:771-791) */
#define CILQR_COST_FIELDS 5  /* total,target,dynamic,corridor,lane_boundary     (h:14-27)    */
#define CILQR_LANE_FIELDS 7  /* a,b,c,start_x,start_y,end_x,end_y               (corridor.h:24-25) */
#define CILQR_MAX_DISCS 16
#define CILQR_MAX_LANE_SEGMENTS 256

/* error codes */
#define CILQR_OK 0
#define CILQR_ERR_NULL (-1)         /* null output / handle            (cc:64-66)  */
#define CILQR_ERR_CONSTRAINTS (-2)  /* empty corridor or lane list     (cc:68-73)  */
#define CILQR_ERR_KNOTS (-3)        /* n_knots != floor(horizon/dt+1)  (cc:75-78)  */
#define CILQR_ERR_CAPACITY (-4)     /* batch / cmax / lane segments above what create() sized */
#define CILQR_ERR_DEVICE (-5)       /* HIP runtime error or no gfx950 device */
#define CILQR_ERR_ARG (-6)          /* invalid argument value */
#define CILQR_ERR_STATE (-7)        /* stage called before the stage it depends on */
#define CILQR_ERR_NO_PATH (-8)      /* cilqr_dp_plan: every sampled path collides ("DP failed", trajectory_planner.cpp:32-35) */

/* per-problem termination status (exits of Optimize(), cc:154-320) */
#define CILQR_ST_RUNNING 0
#define CILQR_ST_CONVERGED_ABS 1 /* dcost < abs_cost_tol                  cc:281,287 */
#define CILQR_ST_CONVERGED_REL 2 /* dcost / cost_old < rel_cost_tol      cc:282     */
#define CILQR_ST_GNORM 3         /* gnorm < 1e-6 && lambda < 1e-5        cc:236     */
#define CILQR_ST_UNSOLVED 4      /* lambda > 1e11                        cc:302     */
#define CILQR_ST_MAX_ITER 5      /* iter == max_iter_num                 cc:312     */
#define CILQR_ST_NO_CORRIDOR 6   /* corridor_count < 0 at some knot: the corridor producer failed there
                                    (cilqr_build_corridors codes); the reference aborts the whole Plan
                                    (corridor.cc:78-81, trajectory_planner.cpp:49-57).  The problem is not
                                    optimised: traj = the init guess, n_cost = 1, n_iter = 1 */

#define CILQR_INIT_IQR 0
#define CILQR_INIT_TRACKER 1

#define CILQR_MEM_HOST 0
#define CILQR_MEM_DEVICE 1

/* Live configuration fields only (dead ones -- IlqrConfig::t/t_rate/alpha/gamma/rho,
 * planner_config.h:60-61,68-70 -- are not carried). */
typedef struct cilqr_config {
  int32_t n_steps;       /* N; knots K = N+1 = floor(horizon/dt + 1)  (cc:22) */
  int32_t num_of_disc;   /* planner_config.h:58 */
  int32_t max_iter;      /* :63 */
  int32_t init_guess;    /* CILQR_INIT_IQR (0): iqr, what the reference runs (cc:169, 793-842); CILQR_INIT_TRACKER (1):
                            InitGuess through the closed-loop Tracker (cc:107-139, tracker.cc), the alternative the
                            reference keeps commented out at cc:168 and recommends in README.md:61-67 */
  double dt;             /* delta_t, :94 */
  double safe_margin;    /* :59 */
  double w_jerk, w_delta_rate, w_x, w_y, w_theta, w_v, w_a, w_delta; /* Weights :45-55 */
  double abs_cost_tol, rel_cost_tol;                                  /* :65-66 */
  double front_hang, wheel_base, rear_hang, width;                    /* vehicle_param.h:26-41 */
  double max_velocity, min_acceleration, max_acceleration;            /* :46-52 */
  double jerk_min, jerk_max, delta_min, delta_max, delta_rate_min, delta_rate_max; /* :57-64 */
  double barrier_t, barrier_eps;                                      /* barrier_function.h:144-145 */
} cilqr_config;

typedef struct cilqr_solver* cilqr_handle;

/* Inputs of IlqrOptimizer::Plan (ilqr_optimizer.h:41-48) for B problems. */
typedef struct cilqr_problem_batch {
  int32_t batch;                 /* B */
  int32_t n_knots;               /* K; must equal n_steps + 1 */
  int32_t cmax;                  /* planes stored per knot in `corridor` (<= create() cmax) */
  int32_t memory;                /* CILQR_MEM_* of start/coarse/corridor/corridor_count */
  const double* start;           /* [B][4]  x, y, theta, velocity  (cc:151) */
  const double* coarse;          /* [B][K][6]  x, y, theta, velocity, a, delta  (cc:148) */
  const double* corridor;        /* [B][K][cmax][3]  a, b, c with "a x + b y < c"  (corridor.h:19-21) */
  const int32_t* corridor_count; /* [B][K]  live planes per knot; negative = no corridor at that knot
                                    (the problem ends with CILQR_ST_NO_CORRIDOR) */
  int32_t n_left, n_right;       /* lane segments, shared by the whole batch */
  const double* left_lane;       /* [n_left][7]  HOST memory */
  const double* right_lane;      /* [n_right][7] HOST memory */
  /* Per-problem lane tables (every Plan call of the reference carries its own lane constraints,
   * ilqr_optimizer.h:41-48).  n_lane_groups = 0 or 1: the one table above serves the whole batch.  Otherwise the
   * problems come grouped by table: group g is problems [lane_group_start[g], lane_group_start[g + 1]) and uses the
   * next lane_group_left[g] rows of left_lane and lane_group_right[g] rows of right_lane (the tables lie back to
   * back; n_left / n_right are then ignored).  The groups are solved one after the other on the handle. */
  int32_t n_lane_groups;
  int32_t reserved1;
  const int32_t* lane_group_start;   /* [n_lane_groups + 1], HOST memory, lane_group_start[0] = 0, last = batch */
  const int32_t* lane_group_left;    /* [n_lane_groups] rows */
  const int32_t* lane_group_right;   /* [n_lane_groups] rows */
  /* CILQR_INIT_TRACKER only: stations of the coarse trajectory's points ([B][K], same memory as `coarse`; TrajectoryPoint::s,
   * which the tracker's projection interpolates along: discretized_trajectory.cpp:165-197).  NULL: the accumulated
   * chord length of the coarse points is used. */
  const double* coarse_station;
} cilqr_problem_batch;

/* Outputs of Plan + cost().  iter_trajs is optional (NULL to skip). */
typedef struct cilqr_solution_batch {
  int32_t memory;                /* CILQR_MEM_* of every pointer below */
  int32_t max_iter_trajs;        /* capacity per problem of iter_trajs */
  double* traj;                  /* [B][K][10] */
  double* cost_hist;             /* [B][max_iter+1][5]; rows >= n_cost[b]: zero (CILQR_MEM_HOST) or left
                                    untouched (CILQR_MEM_DEVICE) */
  int32_t* n_cost;               /* [B] */
  int32_t* status;               /* [B] CILQR_ST_* */
  int32_t* n_iter;               /* [B] iterations started */
  double* iter_trajs;            /* [B][max_iter_trajs][K][10]: init guess + accepted non-final iterates (cc:170,294);
                                    entries >= n_iter_trajs[b] are unspecified */
  int32_t* n_iter_trajs;         /* [B] number that would have been produced (may exceed the capacity) */
  int8_t* alpha_trace;           /* optional (NULL to skip) [B][max_iter]: per iteration of Optimize() the index
                                    into the step-size list (cc:197) that the line search accepted, -1 = all
                                    eleven rejected (cc:296-308), -2 = left before the line search
                                    (gradient-norm exit cc:235-241, or status 6), -3 = iteration not run */
} cilqr_solution_batch;

/* Per-solve kernel timing, filled when profiling is on (cilqr_set_profiling). */
typedef struct cilqr_profile {
  int32_t iterations;            /* outer iterations of the last solve (the slowest problem's count) */
  int32_t backward_launches;
  double backward_ms;            /* sum of HIP-event durations of the backward kernel */
  double quadratize_ms;
  double linesearch_ms;          /* forward + cost + reduce kernels */
  double other_ms;               /* load, init guess, update, export */
  double total_ms;               /* first kernel start -> last kernel end */
  int64_t backward_problem_steps;/* sum over launches of (active problems x N) */
  int32_t backward_full_launches;/* launches whose active set was the whole batch */
  int32_t tail_problems;         /* problems handed to the per-problem tail kernel (upper bound), 0 = not used */
  double backward_full_ms;       /* their summed duration */
  double tail_ms;                /* duration of the tail kernel (not part of the four phase sums above) */
} cilqr_profile;

int cilqr_abi_version(void);
/* first 32 hex digits of the SHA-256 over the library's sources (cilqr_amd/csrc/Makefile): which sources this binary was built from */
const char* cilqr_build_id(void);
int cilqr_default_config(cilqr_config* cfg, int32_t n_steps);

/* device: HIP ordinal.  batch_capacity/cmax/max_lane_segments size the HBM arena once. */
int cilqr_create(const cilqr_config* cfg, int32_t device, int32_t batch_capacity, int32_t cmax,
                 int32_t max_lane_segments, cilqr_handle* out);
/* waits for the solves that were submitted and not collected (cilqr_submit), then frees everything */
int cilqr_destroy(cilqr_handle h);
/* hipStream_t to launch on (NULL = the handle's own stream). */
int cilqr_set_stream(cilqr_handle h, void* hip_stream);
/* Tuning knobs that never change results.  CILQR_OPT_SPEC_THRESHOLD: lockstep iterations with at
 * most this many active problems evaluate all 11 line-search step sizes concurrently instead of
 * round by round (0 disables).  Default: 8192 for cilqr_solve_batch (the shortest solve when the GPU is the caller's
 * alone), 1024 for solves submitted with cilqr_submit / cilqr_pool_submit (eleven candidates where two or three
 * would do is throughput taken from the other solves in flight: +4 % on a pool of two); setting the option sets both. */
#define CILQR_OPT_SPEC_THRESHOLD 1
/* CILQR_OPT_SEQ_ROUNDS (default 4, 1..11): with more active problems than the threshold above, this
 * many step sizes are tried round by round; the problems that rejected all of them evaluate the
 * remaining ones concurrently.  11 = fully sequential. */
#define CILQR_OPT_SEQ_ROUNDS 3
/* CILQR_OPT_COMPACTION (default 1): re-pack the surviving problems into dense slots whenever they
 * fill at most 75 % of the occupied slots, so later iterations keep reading coalesced rows.
 * 0 = never, 2..100 = re-pack at that occupancy percentage. */
#define CILQR_OPT_COMPACTION 2
/* CILQR_OPT_TEAM_THRESHOLD (default 4096): backward passes over at most this many problems spread each
 * problem over eight lanes (column-wise) instead of one, which shortens the chain of dependent
 * steps a small launch waits on; 0 = always one lane per problem.  Bit-identical results. */
#define CILQR_OPT_TEAM_THRESHOLD 4
/* CILQR_OPT_ROUND_GROUP (default 2; 1, 2 or 4): step sizes costed together in one sequential round.  The candidates
 * of a problem sit in neighbouring lanes and share its corridor / goal reads; a round may cost a candidate the
 * sequential loop would not have reached, the first passing index still wins.  Bit-identical results. */
#define CILQR_OPT_ROUND_GROUP 7
/* CILQR_OPT_WAVE_THRESHOLD (default 3072): backward passes over at most this many problems give every problem a
 * whole wavefront (operands in LDS, one output element per lane): the shortest chain of dependent work per step.
 * 0 = never.  Bit-identical results. */
#define CILQR_OPT_WAVE_THRESHOLD 6
/* CILQR_OPT_TAIL_THRESHOLD (default 1024 for cilqr_solve_batch, 128 for solves submitted with cilqr_submit /
 * cilqr_pool_submit -- beside other solves the tail's workgroups take CUs from the neighbours' bulk kernels; setting the
 * option sets both; at most 8192): once at most this many problems are still iterating they
 * leave the lockstep loop; one workgroup per problem runs all its remaining iterations in a single launch
 * (kernels_tail.hip), so the stragglers of a batch no longer cost nine launches per iteration.  0 = lockstep to
 * the end.  Bit-identical results. */
#define CILQR_OPT_TAIL_THRESHOLD 5
/* CILQR_OPT_FINISH_THRESHOLD (default min(batch capacity, 8192), which is also the most): once at most this many
 * problems are still iterating, the survivors move into a small finishing arena of the handle and the main arena is free:
 * with cilqr_submit, the next solve starts there while this one's stragglers finish on a second stream.  0 = never
 * (a submitted solve then runs start to end before the next begins).  Bit-identical results. */
#define CILQR_OPT_FINISH_THRESHOLD 8
/* CILQR_OPT_EXACT_LANE_TIES (default 1; 0 or 1) -- the one option that CAN change results.  FindNeastLaneSegment
 * (ilqr_optimizer.cc:605-618) keeps the first segment whose DistanceTo (line_segment2d.cpp:61-75: hypot to an end
 * point, |cross| to the foot) is strictly smaller.  The kernels compare squared distances, which order the same way
 * except on strips ~1e-7 m wide along the normals through segment end points, where two squares one or two ulp apart
 * have the same root: the reference sees a tie and keeps the earlier segment, a search on squares alone keeps the
 * strictly nearer one.  With 1 (the default: the reference's rule), whenever the two best squared distances are within
 * 1e-13 (relative) of each other the pair is decided by the reference's own distance values (libm-identical hypot)
 * and its strict '<', out of line, after the search loop; candidates that share an end point (the wedge outside every
 * joint) are the same distance in any arithmetic and are not re-examined.  0 = squares only: the opt-in fast rule,
 * 3 % more throughput on a pool of two handles, 6 % on a single solve; the step-replay tests then need the `lane_tie`
 * excuse (tests/parity_util.py). */
#define CILQR_OPT_EXACT_LANE_TIES 9
int cilqr_set_option(cilqr_handle h, int32_t option, int64_t value);
/* the value in force (for the options with two defaults -- CILQR_OPT_SPEC_THRESHOLD, CILQR_OPT_TAIL_THRESHOLD -- the one of
 * cilqr_solve_batch in *value and, if value_submitted is not NULL, the one of submitted solves there) */
int cilqr_get_option(cilqr_handle h, int32_t option, int64_t* value, int64_t* value_submitted);
/* enable = 1: HIP events around every phase of every lockstep iteration (cilqr_profile complete; about 4 %
 * slower: ~800 event records per solve); enable = 2: around the backward launches only (backward_* fields;
 * under 1 %); 0: none. */
int cilqr_set_profiling(cilqr_handle h, int32_t enable);
int cilqr_get_profile(cilqr_handle h, cilqr_profile* out);
/* bytes of device memory held by the handle */
int64_t cilqr_device_bytes(cilqr_handle h);

int cilqr_solve_batch(cilqr_handle h, const cilqr_problem_batch* in, cilqr_solution_batch* out);

/* TrackerConfig / LateralTrackerConfig / LongitudinalTrackerConfig (algorithm/params/planner_config.h:18-43) for
 * CILQR_INIT_TRACKER; cilqr_create starts from the reference's defaults. */
typedef struct cilqr_tracker_config {
  double weight_l, weight_theta, weight_delta, weight_delta_rate, preview_time;   /* lateral  :18-25 */
  double weight_s, weight_v, weight_a, weight_j;                                  /* longitudinal :27-34 */
  double sumulation_dt, dt, tolerance;                                            /* :37-39 */
  int32_t max_num_iteration;                                                      /* :40 */
  int32_t reserved0;
} cilqr_tracker_config;
void cilqr_default_tracker_config(cilqr_tracker_config* cfg);
int cilqr_set_tracker_config(cilqr_handle h, const cilqr_tracker_config* cfg);

/* Asynchronous form of cilqr_solve_batch: submit returns at once (the structs are copied, the
 * arrays they point to -- inputs and outputs -- must stay valid and distinct per solve until its wait),
 * wait blocks for the result code of the OLDEST submitted solve.  A handle accepts up to THREE solves before the
 * oldest is collected (a fourth submit returns CILQR_ERR_STATE): TWO are in flight -- the handle iterates the bulk of
 * solve i+1 in its main arena while the last few thousand problems of solve i (the latency-bound part of a solve: lockstep
 * iterations over few problems, then the per-problem tail kernel) finish in a small second arena on a second stream
 * (CILQR_OPT_FINISH_THRESHOLD) -- and the third is queued behind them.  Keep the handle fed:
 *     submit(A); submit(B); submit(C); wait(); submit(D); wait(); submit(E); ...
 * Host arrays (CILQR_MEM_HOST) of more than 4 MB take a path of their own on a submitted solve: the inputs of the queued
 * solve are uploaded on a separate stream while the solve in front of it iterates (which is what the third place is for:
 * an upload is as long as most of a solve); the results come back on another one -- trajectories straight into the caller's
 * array, the LIVE Cost rows packed (24 MB instead of the dense 527 MB on the bench workload) and scattered into the
 * caller's dense cost_hist on the host, whose other rows the library clears.  Pinned or pageable memory alike (57 / 51 GB/s
 * measured); from the moment of the submit until the wait returns the library reads the inputs and writes the outputs.
 * Results are bit-identical to cilqr_solve_batch.  cilqr_get_profile reports the solve the last wait
 * collected.  A handle is not re-entrant: between a submit and the wait that collects it, call only
 * cilqr_submit / cilqr_wait on it (cilqr_solve_batch and cilqr_stage_load return CILQR_ERR_STATE). */
int cilqr_submit(cilqr_handle h, const cilqr_problem_batch* in, cilqr_solution_batch* out);
int cilqr_wait(cilqr_handle h);

/* ---- stage entry points (operate on the handle's device state, whole batch) ----
 * A cilqr_solve_batch / cilqr_submit on the same handle invalidates the staged state: call
 * cilqr_stage_load again afterwards (CILQR_ERR_STATE otherwise). */
int cilqr_stage_load(cilqr_handle h, const cilqr_problem_batch* in);
int cilqr_stage_init_guess(cilqr_handle h);
/* overwrite the current iterate: X [B][K][6], U [B][N][2] */
int cilqr_stage_set_trajectory(cilqr_handle h, const double* X, const double* U, int32_t memory);
/* cost of the current iterate: cost5 [B][5] */
int cilqr_stage_total_cost(cilqr_handle h, double* cost5, int32_t memory);
int cilqr_stage_quadratize(cilqr_handle h);
/* lambda: [B] regularisation per problem, or NULL to use the solver state */
int cilqr_stage_backward(cilqr_handle h, const double* lambda, int32_t memory);
/* roll out with step alpha into the candidate buffers */
int cilqr_stage_forward(cilqr_handle h, double alpha);

/* tensors readable with cilqr_stage_read, all returned problem-major fp64 */
#define CILQR_T_GOALS 0      /* [B][K][6] */
#define CILQR_T_CORRIDOR 1   /* [B][K][cmax][3] shrunk + normalised (create() cmax) */
#define CILQR_T_LANES 2      /* [n_left+n_right][3] shrunk + normalised a,b,c (left rows first) */
#define CILQR_T_X 3          /* [B][K][6] current iterate */
#define CILQR_T_U 4          /* [B][N][2] */
#define CILQR_T_XCAND 5      /* [B][K][6] candidate of the last forward */
#define CILQR_T_UCAND 6      /* [B][N][2] */
#define CILQR_T_A 7          /* [B][N][6][6] */
#define CILQR_T_B 8          /* [B][N][6][2] */
#define CILQR_T_LX 9         /* [B][K][6] */
#define CILQR_T_LU 10        /* [B][N][2] */
#define CILQR_T_LXX 11       /* [B][K][6][6] */
#define CILQR_T_LUU 12       /* [B][N][2][2] */
#define CILQR_T_KFB 13       /* [B][N][2][6] feedback gains K */
#define CILQR_T_KFF 14       /* [B][N][2] feedforward k */
#define CILQR_T_DV 15        /* [B][2] delta_V_ */
#define CILQR_T_GNORM 16     /* [B] */
int cilqr_stage_read(cilqr_handle h, int32_t tensor, double* dst, int32_t memory);

/* FindNeastLaneSegment (cc:605-618) for n arbitrary points xy[n][2]: index of the nearest left /
 * right lane segment.  use_grid = 1: the accelerated lookup the solver uses; 0: the reference's
 * linear scan.  Both must agree everywhere (test hook). */
int cilqr_stage_nearest_lane(cilqr_handle h, int32_t n, const double* xy, int32_t* left, int32_t* right,
                             int32_t use_grid, int32_t memory);

/* ---- corridor producer: the inputs of cilqr_problem_batch from obstacle points ---- */

/* CorridorConfig, algorithm/params/planner_config.h:75-86 */
typedef struct cilqr_corridor_config {
  double max_diff_x, max_diff_y;   /* obstacle points farther than this from the knot are ignored */
  double radius;                   /* sphere-flip radius */
  double max_axis_x, max_axis_y;   /* half extents of the box added around every knot */
  double lane_segment_length;      /* LaneBoundarySample spacing */
  int32_t is_multiple_sample;      /* 0: both ends of every box edge (8 points); 1: six samples per box edge (24
                                      points, corridor.cc:110-118) -- the caller then passes the obstacles' sample
                                      points (Polygon2d::sample_points, polygon2d.cpp:259-271) instead of their corners */
  int32_t reserved0;
} cilqr_corridor_config;
void cilqr_default_corridor_config(cilqr_corridor_config* cfg);

/* Corridor::BuildCorridorConstraints (algorithm/ilqr/corridor.cc:58-87) for `batch` trajectories of
 * `n_knots` knots: per knot AddCorridorPoints (cc:89-120) + BuildCorridor (cc:122-263), with
 * cv::convexHull replaced by an own float32 hull (see kernels_corridor.hip).
 *   knots          [batch][n_knots][3]              x, y, theta of the coarse trajectory point
 *   points         [batch][n_knots][max_points][2]  obstacle corner points valid at the knot's time
 *                                                   (Environment::Query{Static,Dynamic}ObstaclesPoints)
 *   point_count    [batch][n_knots]
 *   corridor       [batch][n_knots][cmax][3]        out: a, b, c with a x + b y <= c -- the layout of
 *   corridor_count [batch][n_knots]                 cilqr_problem_batch::corridor / corridor_count
 * A knot whose corridor cannot be built gets a negative count (-2: fewer than 4 usable points,
 * -3: more than cmax half-planes, -4: degenerate hull) and is counted in *n_failed; the reference
 * fails the whole Plan in that case (cc:78-81).  max_points + 8 (24 with is_multiple_sample) <= 320.  `memory`
 * applies to all arrays.
 *   polygons       [batch][n_knots][cmax][2]        optional out (NULL to skip): the vertices of each
 *                                                   corridor polygon (`convex_polygons`, cc:244-249),
 *                                                   vertex i being the start of half-plane i */
int cilqr_build_corridors(cilqr_handle h, const cilqr_corridor_config* cfg, int32_t batch, int32_t n_knots,
                          const double* knots, const double* points, const int32_t* point_count,
                          int32_t max_points, double* corridor, int32_t* corridor_count, int32_t cmax,
                          int32_t memory, int32_t* n_failed, double* polygons);

/* LaneBoundarySample (corridor.cc:298-311) + CalLeftLaneConstraints / CalRightLaneConstraints
 * (cc:265-296) + HalfPlaneConstraint (cc:313-321), host only: boundary [n][2] -> rows [.][7] in the
 * layout of cilqr_problem_batch::left_lane / right_lane.  Returns the number of rows (>= 1) or a
 * negative error code. */
int cilqr_lane_constraints(const double* boundary, int32_t n, double segment_length, int32_t is_left,
                           double* rows, int32_t max_rows);

/* Test hook for the kernels' own fp64 routines (host arrays of n doubles):
 * fn 0: log(x) for normal finite x > 0;  fn 1: 1 / x for normal finite x != 0;
 * fn 2: log(x) with mantissa and exponent handed over separately (the long-product path);
 * fn 3 / 4 / 5: sin / cos / tan(x) for |x| <= 1e5;  fn 6: NormalizeAngle(x) (math_utils.cpp:53-59);
 * fn 7 / 8: the device library's cos / sin as the corridor producer uses them for the box corners (corridor.cc:96-99);
 * fn 9: hypot(x, 1) by the libm-identical routine of the lane search and the constraint normalisation;
 * fn 10: NormalizeAngle(x) as the rollouts evaluate it (short paths as one straight line; the complete function for the
 *        wavefront when a lane's argument lies outside (-3 pi, 3 pi)) -- the same bits as fn 6 on every input. */
int cilqr_device_math(cilqr_handle h, int32_t fn, int32_t n, const double* in, double* out);

/* X[b][0] = x0[b]; X[b][i+1] = Dynamics(X[b][i], U[b][i]).  x0 [B][6], U [B][N][2], X [B][K][6] */
int cilqr_open_loop_rollout(cilqr_handle h, int32_t batch, const double* x0, const double* U,
                            double* X, int32_t memory);

/* ---- coarse trajectory: the producer of `coarse` / `start`, host only ----
 * DpPlanner::Plan (algorithm/planner/dp_planner.cpp:135-281): 5 x 7 x 10 (time, station, lateral) sampling DP in
 * the Frenet frame of the centre line with collision checks against the scene, then ComputePathProfile
 * (algorithm/utils/discrete_points_math.cc:27-176).  C++ callers use include/cilqr/dp_planner.hpp directly;
 * include/cilqr/trajectory_planner.hpp chains it with the corridor producer and the solve
 * (TrajectoryPlanner::Plan, algorithm/planner/trajectory_planner.cpp:28-162). */
#define CILQR_COARSE_FIELDS 9   /* time, s, x, y, theta, kappa, velocity, a, delta */
typedef struct cilqr_dp_config {   /* PlannerConfig planner_config.h:94-133 + VehicleParam vehicle_param.h:26-46 */
  double tf, delta_t;
  double dp_nominal_velocity, dp_w_obstacle, dp_w_lateral, dp_w_lateral_change, dp_w_lateral_velocity_change;
  double dp_w_longitudinal_velocity_bias, dp_w_longitudinal_velocity_change;
  double front_hang_length, wheel_base, rear_hang_length, width, max_velocity;
} cilqr_dp_config;
void cilqr_default_dp_config(cilqr_dp_config* cfg);
/* A scene in the vocabulary of the reference's messages (the files under msg/), flattened; HOST memory. */
typedef struct cilqr_scene {
  const double* center;                 /* [n_center][7]  CenterLinePoint: s x y theta kappa left_bound right_bound */
  int32_t n_center;
  int32_t n_static;
  const double* static_points;          /* [sum static_counts][2]  world-frame polygons, back to back (Obstacles.msg) */
  const int32_t* static_counts;         /* [n_static] */
  int32_t n_dynamic;
  int32_t reserved0;
  const double* dynamic_polygon_points;     /* [sum dynamic_polygon_counts][2]  body-frame polygons (DynamicObstacle.msg) */
  const int32_t* dynamic_polygon_counts;    /* [n_dynamic] */
  const double* dynamic_trajectories;       /* [sum dynamic_trajectory_counts][4]  time x y theta (DynamicTrajectoryPoint.msg) */
  const int32_t* dynamic_trajectory_counts; /* [n_dynamic] */
} cilqr_scene;
/* Environment::set_reference (algorithm/utils/environment.cpp:20-43), host only: the left / right road barriers -- the
 * centre line evaluated every 0.1 m of station and shifted by +left_bound / -right_bound along its normal -- that
 * Corridor::Plan samples its lane constraints from (corridor.cc:43-51; the `boundary` input of
 * cilqr_lane_constraints).  left / right [max_points][2]; returns the number of points or a negative error code. */
int cilqr_road_barriers(const double* center, int32_t n_center, double* left, double* right, int32_t max_points);
/* start3 = x, y, theta (dp_planner.cpp:135-141); coarse [n_knots][CILQR_COARSE_FIELDS], n_knots = tf / delta_t + 1.
 * Returns CILQR_OK, or CILQR_ERR_NO_PATH when every sampled path collides (coarse is filled all the same, as in
 * the reference, whose caller then stops: trajectory_planner.cpp:32-35). */
int cilqr_dp_plan(const cilqr_dp_config* cfg, const cilqr_scene* scene, const double* start3, double* coarse,
                  int32_t n_knots);

/* ---- several GPUs from ONE host process (the reference's caller is one process:
 * algorithm/planning_node.cc:9-31) ----
 * A multi handle owns one solver handle per listed device.  cilqr_multi_solve cuts the batch into contiguous shards
 * (the first batch % n shards hold one problem more), solves them concurrently -- each shard on its own device,
 * stream and host threads -- and every shard writes its results into the caller's arrays at its offset: problem order
 * is preserved, nothing is copied between devices, results are bit-identical to one cilqr_solve_batch over the whole
 * batch.  `devices` may list a device more than once (logical shards on one GPU).  The arrays of `in` / `out` must be
 * reachable from every listed device: host memory (CILQR_MEM_HOST), or device memory when all shards share the
 * device that holds it.  One lane table per call (n_lane_groups <= 1).  batch_capacity is for all shards together. */
typedef struct cilqr_multi* cilqr_multi_handle;
int cilqr_multi_create(const cilqr_config* cfg, const int32_t* devices, int32_t n_devices, int32_t batch_capacity,
                       int32_t cmax, int32_t max_lane_segments, cilqr_multi_handle* out);
int cilqr_multi_destroy(cilqr_multi_handle m);
int cilqr_multi_solve(cilqr_multi_handle m, const cilqr_problem_batch* in, cilqr_solution_batch* out);
/* cilqr_set_option on every shard */
int cilqr_multi_set_option(cilqr_multi_handle m, int32_t option, int64_t value);
/* the split of a batch: first problem and device of every shard (arrays of max_shards entries, NULL to skip);
 * returns the number of shards */
int cilqr_multi_shards(cilqr_multi_handle m, int32_t batch, int32_t* first_problem, int32_t* device, int32_t max_shards);
int64_t cilqr_multi_device_bytes(cilqr_multi_handle m);

/* ---- a stream of batches on ONE GPU: several handles dealt out round-robin ----
 * cilqr_submit overlaps the latency-bound end of a solve with the bulk of the next one.  What stays idle then is inside
 * the bulk itself (a backward pass or a rollout is one lane per problem: N dependent steps on a fraction of the chip once
 * the active set has shrunk); other solves' cost kernels fit there.  A pool owns n_handles handles on one device
 * (n_handles x the memory of one; 1..16) and runs submitted solve s on handle s % n_handles: up to 3 x n_handles solves
 * submitted (two in flight and one queued per handle, see cilqr_submit), cilqr_pool_wait collects the OLDEST.  Keep it fed:
 *     for (i = 0; i < depth; ++i) submit(batch[i]);   then   wait(); submit(next); wait(); submit(next); ...
 * The rules of cilqr_submit hold per solve (structs copied, arrays valid and distinct until the wait that collects them; a
 * submit beyond the depth returns CILQR_ERR_STATE).  Results are bit-identical to cilqr_solve_batch.  Measured on the
 * bench workload (65536 problems per batch): 1.69-1.72 M solves/s with one handle, 1.94-1.99 M with two, 1.97-2.00 M with three.
 * cilqr_pool_set_option: cilqr_set_option on every handle (nothing in flight); cilqr_pool_get_profile: of the solve
 * the last wait collected; cilqr_pool_destroy waits for whatever is still in flight.  Solves may also be submitted to a
 * handle of the pool directly (cilqr_pool_handle_at) as long as the pool itself is empty meanwhile.  Like a handle, a pool
 * is driven by one host thread at a time (its handles run their own worker threads). */
typedef struct cilqr_pool* cilqr_pool_handle;
int cilqr_pool_create(const cilqr_config* cfg, int32_t device, int32_t n_handles, int32_t batch_capacity, int32_t cmax,
                      int32_t max_lane_segments, cilqr_pool_handle* out);
int cilqr_pool_destroy(cilqr_pool_handle p);
int cilqr_pool_submit(cilqr_pool_handle p, const cilqr_problem_batch* in, cilqr_solution_batch* out);
int cilqr_pool_wait(cilqr_pool_handle p);
int32_t cilqr_pool_depth(cilqr_pool_handle p);   /* 3 x n_handles */
/* handle k (0 <= k < n_handles) for what the pool has no call of its own for -- cilqr_set_profiling, cilqr_set_stream, a
 * synchronous cilqr_solve_batch, the stage entry points; NULL while solves are in flight on the pool.  Owned by the pool. */
cilqr_handle cilqr_pool_handle_at(cilqr_pool_handle p, int32_t k);
int cilqr_pool_set_option(cilqr_pool_handle p, int32_t option, int64_t value);
int cilqr_pool_get_profile(cilqr_pool_handle p, cilqr_profile* out);
int64_t cilqr_pool_device_bytes(cilqr_pool_handle p);

/* ---- multi-GPU (nothing in the single-process reference to replace) ----
 * One process per GPU.  Problems are independent: rank r solves a contiguous block of `batch` problems with
 * its own handle and no communication; ONE gather then moves the results to a root rank through RCCL
 * (grouped ncclSend / ncclRecv, point-to-point over xGMI).  librccl.so.1 is loaded on the first call --
 * libcilqr_hip.so does not link against it.
 *   rank 0:      cilqr_comm_unique_id(id)  ->  ship the 128 bytes to every rank (any transport)
 *   every rank:  cilqr_comm_create(h, id, rank, world)          (collective: ncclCommInitRank)
 *   every step:  cilqr_solve_batch(h, ...)  then  cilqr_gather_results(h, batch, &local, root, &gathered)
 * Travelling per rank: 8 of the 10 trajectory columns (time and kappa are rebuilt on the root with the
 * expressions of TransformToTrajectory, ilqr_optimizer.cc:771-791 -- bit-identical to a local export),
 * the LIVE Cost rows only (n_cost[b] of the max_iter + 1), n_cost, status, n_iter.
 * `local` and `gathered` must be CILQR_MEM_DEVICE; `gathered` (root only, NULL elsewhere) holds
 * world * batch problems in rank order: traj [world*batch][K][10], cost_hist [world*batch][max_iter+1][5]
 * (rows >= n_cost untouched), n_cost, status, n_iter (optional).  iter_trajs / alpha_trace do not travel. */
#define CILQR_UNIQUE_ID_BYTES 128
int cilqr_comm_unique_id(uint8_t* id /* [CILQR_UNIQUE_ID_BYTES] */);
int cilqr_comm_create(cilqr_handle h, const uint8_t* id, int32_t rank, int32_t world);
int cilqr_comm_destroy(cilqr_handle h);
int cilqr_comm_info(cilqr_handle h, int32_t* rank, int32_t* world);
int cilqr_gather_results(cilqr_handle h, int32_t batch, const cilqr_solution_batch* local, int32_t root,
                         cilqr_solution_batch* gathered);

const char* cilqr_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* CILQR_H_ */
