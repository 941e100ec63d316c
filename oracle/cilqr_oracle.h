/*
 * oracle/cilqr_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cilqr_amd/ or include/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (mpt0816/Cilqr) holds no tests, golden vectors
 * or fixtures for this path and cannot be built here (Eigen 3.4, ROS and OpenCV
 * are absent).  This is a scalar-fp64 restatement of
 *   algorithm/ilqr/ilqr_optimizer.cc, vehicle_model.cc, barrier_function.h,
 *   algorithm/math/math_utils.cpp:53-59, line_segment2d.cpp:38-75
 * written by reading those files; see cilqr_oracle.cc for per-function cites.
 */
#ifndef CILQR_ORACLE_H_
#define CILQR_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Live configuration fields (planner_config.h:45-73, vehicle_param.h:21-64,
 * barrier_function.h:144-145).  Defaults via oracle_default_config(). */
typedef struct oracle_config {
  int n_steps;          /* N = num_of_knots_-1; K = floor(horizon/dt + 1) (cc:22) */
  double dt;
  int num_of_disc;
  double safe_margin;
  double w_jerk, w_delta_rate, w_x, w_y, w_theta, w_v, w_a, w_delta;
  int max_iter;
  double abs_cost_tol, rel_cost_tol;
  double front_hang, wheel_base, rear_hang, width;
  double max_velocity, min_acceleration, max_acceleration;
  double jerk_min, jerk_max, delta_min, delta_max, delta_rate_min, delta_rate_max;
  double barrier_t, barrier_eps;
} oracle_config;

/* status codes of a finished solve (ilqr_optimizer.cc:154-320 exits) */
enum {
  ORACLE_ST_RUNNING = 0,
  ORACLE_ST_CONVERGED_ABS = 1, /* dcost < abs_cost_tol (cc:281,287) */
  ORACLE_ST_CONVERGED_REL = 2, /* dcost/cost_old < rel_cost_tol (cc:282) */
  ORACLE_ST_GNORM = 3,         /* gnorm < 1e-6 && lambda < 1e-5 (cc:236) */
  ORACLE_ST_UNSOLVED = 4,      /* lambda > 1e11 (cc:302) */
  ORACLE_ST_MAX_ITER = 5       /* iter == max_iter_num (cc:312) */
};

void oracle_default_config(oracle_config* c, int n_steps);

/* The two unverifiable readings of Eigen's semantics as switches (cilqr_oracle.cc, top): dv_eval 0 = lazy (default),
 * 1 = eager; dot_order 0 = sequential, 1 = eigen_redux, 2 = eigen_sse2 (default); negative = unchanged.  Process-wide,
 * between solves only.  Returns dv_eval | dot_order << 8 as now set. */
int oracle_set_semantics(int dv_eval, int dot_order);
/* test hook: a 6-term dot product through this file's MatTMul (transposed_lhs = 1) / MatMul (0) with the order now set */
double oracle_dot6(const double* a, const double* b, int transposed_lhs);

void* oracle_create(const oracle_config* c);
void oracle_destroy(void* h);

/* Plan() input (cc:53-95): TransformGoals + ShrinkConstraints + NormalizeHalfPlane.
 * start = (x, y, theta, v); coarse = K x (x, y, theta, v, a, delta);
 * corridor = K x cmax x (a, b, c) with ccount[i] live planes at knot i ("ax+by<c");
 * lanes = S x (a, b, c, start_x, start_y, end_x, end_y).
 * Returns 0, or -1 on the reference's `return false` cases (cc:64-78). */
int oracle_set_problem(void* h, const double* start4, const double* coarse, int n_coarse,
                       const double* corridor, const int* ccount, int cmax,
                       const double* left, int n_left, const double* right, int n_right);

/* Optimize() (cc:154-320).
 * traj: K x (time,x,y,theta,v,a,delta,kappa,jerk,delta_rate)  (cc:771-791)
 * cost_hist: (max_iter+1) x (total,target,dynamic,corridor,lane) (h:14-27)
 * iter_trajs (nullable): up to max_iter_trajs x K x 10, init guess + accepted non-final iterates
 * trace (nullable): max_iter x 10: (accepted alpha index, -1 = all eleven rejected, -2 = gradient-norm
 *   exit; lambda used in Backward; dV0; dV1; cost_new of last trial; dcost; z; gnorm; smallest relative
 *   distance of this iteration's accept / converge / exit tests to their thresholds; trials evaluated)
 * min_margin (nullable): smallest relative distance of any accept/converge decision to its threshold */
int oracle_plan(void* h, double* traj, double* cost_hist, int* n_cost, int* status, int* n_iter,
                double* iter_trajs, int max_iter_trajs, int* n_iter_trajs, double* trace,
                double* min_margin);

/* Step-by-step replay for the parity tests: Optimize() re-entered at iteration `iter` from the iterate
 * X [K][6], U [N][2] with regularisation state (lambda, dlambda); runs until ONE iteration is accepted or
 * the solve ends (status stays ORACLE_ST_RUNNING when it stopped after a non-converging accept).
 * cost_hist row 0 = TotalCost(X, U), row 1 = the accepted trial; n_iter = index of the next iteration;
 * trace rows (10 columns, see oracle_plan) count from the first replayed iteration;
 * traj = the iterate afterwards; lambda_out[2] = (lambda, dlambda) afterwards. */
int oracle_replay(void* h, const double* X, const double* U, double lambda, double dlambda, int iter,
                  double* traj, double* cost_hist, int* n_cost, int* status, int* n_iter, double* trace,
                  double* lambda_out);

/* ---- stage entry points (same arithmetic as inside oracle_plan) ---- */
void oracle_get_constraints(void* h, double* goals /*K*6*/, double* corridor /*K*cmax*3*/,
                            double* left_abc /*SL*3*/, double* right_abc /*SR*3*/, double* disc_radius);
void oracle_init_guess(void* h, double* X /*K*6*/, double* U /*N*2*/);               /* iqr cc:793 */
void oracle_open_loop_rollout(void* h, const double* x0, const double* U, double* X); /* slover/ilqr.h:363 */
double oracle_total_cost(void* h, const double* X, const double* U, double* cost5);   /* cc:417 */
void oracle_quadratize(void* h, const double* X, const double* U, double* A /*N*36*/, double* B /*N*12*/,
                       double* lx /*K*6*/, double* lu /*N*2*/, double* lxx /*K*36*/, double* luu /*N*4*/);
void oracle_backward(void* h, double lambda, const double* A, const double* B, const double* lx,
                     const double* lu, const double* lxx, const double* luu, double* Kfb /*N*12*/,
                     double* kff /*N*2*/, double* dV2);                               /* cc:334 */
double oracle_grad_norm(void* h, const double* kff, const double* U);                 /* cc:322 */
void oracle_forward(void* h, double alpha, const double* X, const double* U, const double* Kfb,
                    const double* kff, double* Xn, double* Un);                       /* cc:392 */
void oracle_dynamics(void* h, const double* x, const double* u, double* xn);          /* vm.cc:88 */
void oracle_dynamics_jacobian(void* h, const double* x, const double* u, double* A, double* B); /* vm.cc:21 */
double oracle_normalize_angle(double a);                                              /* math_utils.cpp:53 */
double oracle_segment_distance(const double* seg4, double px, double py);             /* line_segment2d.cpp:61 */
double oracle_barrier_value(void* h, double g);                                       /* barrier_function.h:104 */
void oracle_barrier_jacobian(void* h, double g, const double* dg, int n, double* out);    /* :115 */
void oracle_barrier_hessian(void* h, double g, const double* dg, const double* ddg, int n, double* out); /* :127 */

/* B independent solves in a loop, single thread; inputs problem-major:
 * start[B][4], coarse[B][K][6], corridor[B][K][cmax][3], ccount[B][K]; lanes shared.
 * cost_hist[B][max_iter+1][5]; traj[B][K][10]. */
int oracle_solve_batch(const oracle_config* c, int B, const double* start, const double* coarse,
                       const double* corridor, const int* ccount, int cmax, const double* left,
                       int n_left, const double* right, int n_right, double* traj, double* cost_hist,
                       int* n_cost, int* status, int* n_iter, double* min_margin, double* seconds);
/* Same, plus per iteration ([B][max_iter], nullable): the accepted alpha index (-1 = all eleven
 * rejected, -2 = gradient-norm exit, -3 = iteration not run) and the smallest relative distance of
 * that iteration's tests to their thresholds (trace column 8); problem_seconds ([B], nullable): the
 * steady_clock time of every single solve. */
int oracle_solve_batch_trace(const oracle_config* c, int B, const double* start, const double* coarse,
                             const double* corridor, const int* ccount, int cmax, const double* left,
                             int n_left, const double* right, int n_right, double* traj, double* cost_hist,
                             int* n_cost, int* status, int* n_iter, double* min_margin, double* seconds,
                             signed char* alpha_trace, double* iter_margin, double* problem_seconds);

#ifdef __cplusplus
}
#endif
#endif
