/*
 * oracle/tracker_oracle.cc -- CPU oracle of the alternative init guess.
 *
 * TEST INFRASTRUCTURE ONLY (see cilqr_oracle.h).  PARITY UNPINNED for the tracker as a whole (no reference vectors;
 * tracker.cc needs Eigen).  PINNED against the reference's own code (oracle/_ref, tests/test_reference_pins.py): the
 * EvaluateTime / GetProjection interpolation of the followed trajectory (discretized_trajectory.cpp:130-197).
 * Restates, with plain 3x3 arrays instead of Eigen::MatrixXd,
 *   Tracker::Plan / lqr                 algorithm/ilqr/tracker.cc:12-17, 169-215
 *   Tracker::CalcaulateInitState        tracker.cc:19-55
 *   Tracker::LateralControl / LongitudinalControl   tracker.cc:57-81, InitMatrix :137-167
 *   Tracker::VehicleDynamic (RK4)       tracker.cc:83-135, vehicle_mode tracker.h:72-87
 *   math::SolveLQRProblem               algorithm/math/linear_quadratic_regulator.cc:30-78 (iterative DARE)
 *   IlqrOptimizer::InitGuess            algorithm/ilqr/ilqr_optimizer.cc:107-139 (the call site the reference
 *                                       keeps commented out at cc:168, README.md:61-67)
 *   DiscretizedTrajectory::GetProjection / EvaluateTime   algorithm/utils/discretized_trajectory.cpp:86-141, 165-197
 * Eigen semantics restated by hand: matrix products are coefficient-wise dot products accumulated k = 0, 1, 2
 * (dynamic 3x3 operands are below Eigen's GEMM threshold), `AT*P*A - (AT*P*B + M)*inv*(BT*P*A + MT) + Q` groups as
 * ((AT*P)*A - (((AT*P)*B + M)*inv)*((BT*P)*A + MT)) + Q with M = 0, the 1x1 inverse is 1 / x.
 */
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr double kMathEpsilon = 1e-10;   // vec2d.h:33

double NormalizeAngle(const double angle) {   // math_utils.cpp:53-59
  double a = std::fmod(angle + M_PI, 2.0 * M_PI);
  if (a < 0.0) a += (2.0 * M_PI);
  return a - M_PI;
}
double slerp(const double a0, const double t0, const double a1, const double t1, const double t) {   // math_utils.h:208-225
  if (std::abs(t1 - t0) <= kMathEpsilon) return NormalizeAngle(a0);
  const double a0_n = NormalizeAngle(a0);
  const double a1_n = NormalizeAngle(a1);
  double d = a1_n - a0_n;
  if (d > M_PI) {
    d = d - 2 * M_PI;
  } else if (d < -M_PI) {
    d = d + 2 * M_PI;
  }
  const double r = (t - t0) / (t1 - t0);
  const double a = a0_n + d * r;
  return NormalizeAngle(a);
}

struct TrajectoryPoint {   // discretized_trajectory.h:26-43
  double time = 0.0, s = 0.0, x = 0.0, y = 0.0, theta = 0.0, kappa = 0.0, velocity = 0.0;
  double a = 0.0, jerk = 0.0, delta = 0.0, delta_rate = 0.0, left_bound = 0.0, right_bound = 0.0;
};

TrajectoryPoint LinearInterpolateTrajectory(const TrajectoryPoint& p0, const TrajectoryPoint& p1, const double s) {   // cpp:66-89
  double s0 = p0.s, s1 = p1.s;
  if (std::abs(s1 - s0) < kMathEpsilon) return p0;
  TrajectoryPoint pt;
  double weight = (s - s0) / (s1 - s0);
  pt.time = (1 - weight) * p0.time + weight * p1.time;
  pt.s = s;
  pt.x = (1 - weight) * p0.x + weight * p1.x;
  pt.y = (1 - weight) * p0.y + weight * p1.y;
  pt.theta = slerp(p0.theta, p0.s, p1.theta, p1.s, s);
  pt.kappa = (1 - weight) * p0.kappa + weight * p1.kappa;
  pt.velocity = (1 - weight) * p0.velocity + weight * p1.velocity;
  return pt;
}
TrajectoryPoint LinearInterpolateTrajectoryWithTime(const TrajectoryPoint& p0, const TrajectoryPoint& p1, const double time) {   // cpp:91-115
  double time0 = p0.time, time1 = p1.time;
  if (std::fabs(time1 - time0) < kMathEpsilon) return p0;
  TrajectoryPoint pt;
  double weight = (time - time0) / (time1 - time0);
  pt.time = time;
  pt.s = (1 - weight) * p0.s + weight * p1.s;
  pt.x = (1 - weight) * p0.x + weight * p1.x;
  pt.y = (1 - weight) * p0.y + weight * p1.y;
  pt.theta = slerp(p0.theta, p0.time, p1.theta, p1.time, time);
  pt.kappa = (1 - weight) * p0.kappa + weight * p1.kappa;
  pt.velocity = (1 - weight) * p0.velocity + weight * p1.velocity;
  return pt;
}

struct Follow {   // the DiscretizedTrajectory queries the tracker uses
  std::vector<TrajectoryPoint> tr;
  TrajectoryPoint EvaluateTime(const double time) const {   // cpp:130-141 with :49-64
    size_t it;
    if (time >= tr.back().time) {
      it = tr.size() - 1;
    } else if (time < tr.front().time) {
      it = 0;
    } else {
      it = std::lower_bound(tr.begin(), tr.end(), time, [](const TrajectoryPoint& tp, const double t) { return tp.time < t; }) - tr.begin();
    }
    if (it == 0) it = 1;
    return LinearInterpolateTrajectoryWithTime(tr[it - 1], tr[it], time);
  }
  void GetProjection(double px, double py, TrajectoryPoint* project_point_ptr) const {   // cpp:165-197
    long point_idx = 0;
    double nearest_distance = std::numeric_limits<double>::max();
    for (size_t i = 0; i < tr.size(); ++i) {
      double dx = tr[i].x - px, dy = tr[i].y - py;
      double distance = dx * dx + dy * dy;
      if (distance < nearest_distance) {
        point_idx = (long)i;
        nearest_distance = distance;
      }
    }
    TrajectoryPoint project_point = tr[point_idx];
    long index_start = std::max(0l, point_idx - 1);
    unsigned long index_end = std::min(tr.size() - 1, (unsigned long)point_idx + 1);
    if ((unsigned long)index_start < index_end) {
      double v0x = px - tr[index_start].x, v0y = py - tr[index_start].y;
      double v1x = tr[index_end].x - tr[index_start].x, v1y = tr[index_end].y - tr[index_start].y;
      double v1_norm = std::sqrt(v1x * v1x + v1y * v1y);
      double dot = v0x * v1x + v0y * v1y;
      double delta_s = dot / v1_norm;
      project_point = LinearInterpolateTrajectory(tr[index_start], tr[index_end], tr[index_start].s + delta_s);
    }
    *project_point_ptr = project_point;
  }
};

struct Mat3 {
  double m[9];
};
Mat3 Mul(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = a.m[i * 3 + 0] * b.m[0 * 3 + j];
      for (int k = 1; k < 3; ++k) s += a.m[i * 3 + k] * b.m[k * 3 + j];
      r.m[i * 3 + j] = s;
    }
  return r;
}
Mat3 Transpose(const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[j * 3 + i];
  return r;
}

// linear_quadratic_regulator.cc:30-78 with B 3x1, R 1x1, M = 0; K out: 1x3.  *margin: smallest relative distance
// of `diff` to the tolerance over the iterations (the loop's stopping test is the only decision in here)
void SolveLQRProblem(const Mat3& A, const double* B, const Mat3& Q, double R, double tolerance, unsigned max_num_iteration,
                     double* K, double* margin) {
  const Mat3 AT = Transpose(A);
  Mat3 P = Q;
  unsigned num_iteration = 0;
  double diff = std::numeric_limits<double>::max();
  auto row_times = [](const double* v, const Mat3& m, double* out) {   // (1x3) * (3x3)
    for (int j = 0; j < 3; ++j) {
      double s = v[0] * m.m[0 * 3 + j];
      for (int k = 1; k < 3; ++k) s += v[k] * m.m[k * 3 + j];
      out[j] = s;
    }
  };
  while (num_iteration++ < max_num_iteration && diff > tolerance) {
    const Mat3 ATP = Mul(AT, P);
    const Mat3 ATPA = Mul(ATP, A);
    double ATPB[3];                                  // (AT*P)*B + M
    for (int i = 0; i < 3; ++i) {
      double s = ATP.m[i * 3 + 0] * B[0];
      for (int k = 1; k < 3; ++k) s += ATP.m[i * 3 + k] * B[k];
      ATPB[i] = s + 0.0;
    }
    double BTP[3], BTPA[3];
    row_times(B, P, BTP);                            // BT*P
    double BTPB = BTP[0] * B[0];
    for (int k = 1; k < 3; ++k) BTPB += BTP[k] * B[k];
    const double inv = 1.0 / (R + BTPB);             // (R + BT*P*B).inverse()
    row_times(BTP, A, BTPA);                         // (BT*P)*A
    for (int j = 0; j < 3; ++j) BTPA[j] = BTPA[j] + 0.0;   // + MT
    Mat3 P_next;
    double maxc = -std::numeric_limits<double>::max();
    for (int i = 0; i < 3; ++i) {
      const double left = ATPB[i] * inv;             // ((AT*P*B + M) * inv)(i)
      for (int j = 0; j < 3; ++j) {
        P_next.m[i * 3 + j] = (ATPA.m[i * 3 + j] - left * BTPA[j]) + Q.m[i * 3 + j];
        maxc = std::max(maxc, P_next.m[i * 3 + j] - P.m[i * 3 + j]);
      }
    }
    diff = std::fabs(maxc);                          // fabs((P_next - P).maxCoeff())
    if (margin) *margin = std::min(*margin, std::fabs(diff - tolerance) / tolerance);
    P = P_next;
  }
  double BTP[3], BTPA[3];
  row_times(B, P, BTP);
  double BTPB = BTP[0] * B[0];
  for (int k = 1; k < 3; ++k) BTPB += BTP[k] * B[k];
  const double inv = 1.0 / (R + BTPB);
  row_times(BTP, A, BTPA);
  for (int j = 0; j < 3; ++j) K[j] = inv * (BTPA[j] + 0.0);
}

struct TrackerCfg {   // planner_config.h:18-43 + the VehicleParam fields the tracker reads
  double weight_l, weight_theta, weight_delta, weight_delta_rate, lat_preview_time;
  double weight_s, weight_v, weight_a, weight_j;
  double sumulation_dt, dt, tolerance;
  unsigned max_num_iteration;
  double wheel_base, delta_min, delta_max, delta_rate_min, delta_rate_max, jerk_min, jerk_max, min_acceleration, max_acceleration;
};

struct VehicleState {
  double x, y, theta, v, delta, a;
};

class Tracker {
 public:
  explicit Tracker(const TrackerCfg& c) : cfg(c) {   // InitMatrix tracker.cc:137-167
    std::memset(&lateral_A_, 0, sizeof(Mat3));
    std::memset(&lateral_Q_, 0, sizeof(Mat3));
    std::memset(&longitudinal_A_, 0, sizeof(Mat3));
    std::memset(&longitudinal_Q_, 0, sizeof(Mat3));
    const double dt = cfg.dt;
    lateral_A_.m[0] = 1.0; lateral_A_.m[4] = 1.0; lateral_A_.m[8] = 1.0;
    lateral_B_[0] = 0.0; lateral_B_[1] = 0.0; lateral_B_[2] = 1.0 * dt;
    lateral_Q_.m[0] = cfg.weight_l; lateral_Q_.m[4] = cfg.weight_theta; lateral_Q_.m[8] = cfg.weight_delta;
    lateral_R_ = cfg.weight_delta_rate;
    longitudinal_A_.m[0] = 1.0; longitudinal_A_.m[4] = 1.0; longitudinal_A_.m[8] = 1.0;
    longitudinal_A_.m[1] = dt;
    longitudinal_A_.m[5] = -dt;
    longitudinal_B_[0] = 0.0; longitudinal_B_[1] = 0.0; longitudinal_B_[2] = 1.0 * dt;
    longitudinal_Q_.m[0] = cfg.weight_s; longitudinal_Q_.m[4] = cfg.weight_v; longitudinal_Q_.m[8] = cfg.weight_a;
    longitudinal_R_ = cfg.weight_j;
  }

  bool lqr(const TrajectoryPoint& start_state, const Follow& follow, std::vector<TrajectoryPoint>* out, double* margin) {   // cc:169-215
    follow_ = &follow;
    std::vector<TrajectoryPoint> trajectory;
    TrajectoryPoint cur_state = start_state;
    trajectory.push_back(cur_state);
    double start_time = follow.tr.front().time;
    double end_time = follow.tr.back().time;
    cur_state.time = start_time;
    cur_state.s = 0.0;
    size_t i = 1;
    for (double t = start_time; t < end_time + kMathEpsilon; t += cfg.sumulation_dt) {
      double lat[3], lon[3];
      CalcaulateInitState(cur_state, lat, lon);
      double delta_rate = LateralControl(lat, cur_state.velocity, margin);
      double jerk = LongitudinalControl(lon, margin);
      delta_rate = std::fmax(cfg.delta_rate_min, std::fmin(cfg.delta_rate_max, delta_rate));
      jerk = std::fmax(cfg.jerk_min, std::fmin(cfg.jerk_max, jerk));
      trajectory.back().delta_rate = delta_rate;
      trajectory.back().jerk = jerk;
      cur_state = VehicleDynamic(cur_state, delta_rate, jerk);
      cur_state.time = t;
      if (i < follow.tr.size() && cur_state.time > follow.tr[i].time - kMathEpsilon) {   // .at(i): in range whenever it pushes
        trajectory.push_back(cur_state);
        ++i;
      }
    }
    if (trajectory.size() != follow.tr.size()) return false;   // "tacker failed."
    *out = trajectory;
    return true;
  }

 private:
  void CalcaulateInitState(const TrajectoryPoint& current_state, double* lateral_state, double* longitudinal_state) const {   // cc:19-55
    double preveiw_x = current_state.x + std::cos(current_state.theta) * current_state.velocity * cfg.lat_preview_time;
    double preveiw_y = current_state.y + std::sin(current_state.theta) * current_state.velocity * cfg.lat_preview_time;
    TrajectoryPoint project_pt;
    follow_->GetProjection(preveiw_x, preveiw_y, &project_pt);
    double dx = current_state.x - project_pt.x;
    double dy = current_state.y - project_pt.y;
    double l = std::sin(project_pt.theta) * dx - std::cos(project_pt.theta) * dy;
    double theta_error = NormalizeAngle(project_pt.theta - current_state.theta);
    lateral_state[0] = l; lateral_state[1] = theta_error; lateral_state[2] = current_state.delta;
    TrajectoryPoint match_pt = follow_->EvaluateTime(current_state.time + 0.0);
    double v_error = match_pt.velocity - current_state.velocity;
    longitudinal_state[0] = match_pt.s - project_pt.s; longitudinal_state[1] = v_error; longitudinal_state[2] = current_state.a;
  }
  double LateralControl(const double* state, const double v, double* margin) {   // cc:57-72
    double v_amend = std::fmax(2, v);
    double dt = 0.1;
    lateral_A_.m[0 * 3 + 1] = v_amend * dt;
    lateral_A_.m[1 * 3 + 2] = -v_amend / cfg.wheel_base * dt;
    double K[3];
    SolveLQRProblem(lateral_A_, lateral_B_, lateral_Q_, lateral_R_, cfg.tolerance, cfg.max_num_iteration, K, margin);
    double s = K[0] * state[0];
    for (int k = 1; k < 3; ++k) s += K[k] * state[k];
    return -s;
  }
  double LongitudinalControl(const double* state, double* margin) {   // cc:74-81
    double K[3];
    SolveLQRProblem(longitudinal_A_, longitudinal_B_, longitudinal_Q_, longitudinal_R_, cfg.tolerance, cfg.max_num_iteration, K, margin);
    double s = K[0] * state[0];
    for (int k = 1; k < 3; ++k) s += K[k] * state[k];
    return -s;
  }
  VehicleState vehicle_mode(double theta, double v, double delta, double a, double j, double delta_rate) const {   // tracker.h:72-87
    VehicleState d;
    d.x = v * std::cos(theta);
    d.y = v * std::sin(theta);
    d.theta = v * std::tan(delta) / cfg.wheel_base;
    d.v = a;
    d.a = j;
    d.delta = delta_rate;
    return d;
  }
  TrajectoryPoint VehicleDynamic(const TrajectoryPoint& cur_state, const double delta_rate, const double jerk) const {   // cc:83-135
    const double dt = cfg.sumulation_dt;
    const double dt_2 = dt / 2.0;
    VehicleState k1 = vehicle_mode(cur_state.theta, cur_state.velocity, cur_state.delta, cur_state.a, jerk, delta_rate);
    VehicleState k2 = vehicle_mode(cur_state.theta + k1.theta * dt_2, cur_state.velocity + k1.v * dt_2, cur_state.delta + k1.delta * dt_2,
                                   cur_state.a + k1.a * dt_2, jerk, delta_rate);
    VehicleState k3 = vehicle_mode(cur_state.theta + k2.theta * dt_2, cur_state.velocity + k2.v * dt_2, cur_state.delta + k2.delta * dt_2,
                                   cur_state.a + k2.a * dt_2, jerk, delta_rate);
    VehicleState k4 = vehicle_mode(cur_state.theta + k3.theta * dt, cur_state.velocity + k3.v * dt, cur_state.delta + k3.delta * dt,
                                   cur_state.a + k3.a * dt, jerk, delta_rate);
    TrajectoryPoint next_state;
    next_state.time = cur_state.time + dt;
    next_state.x = cur_state.x + (k1.x + k2.x * 2.0 + k3.x * 2.0 + k4.x) / 6.0 * dt;
    next_state.y = cur_state.y + (k1.y + k2.y * 2.0 + k3.y * 2.0 + k4.y) / 6.0 * dt;
    next_state.theta = NormalizeAngle(cur_state.theta + (k1.theta + k2.theta * 2.0 + k3.theta * 2.0 + k4.theta) / 6.0 * dt);
    next_state.velocity = std::fmax(0.0, cur_state.velocity + (k1.v + k2.v * 2.0 + k3.v * 2.0 + k4.v) / 6.0 * dt);
    next_state.delta = NormalizeAngle(std::fmin(
        cfg.delta_max, std::fmax(cfg.delta_min, cur_state.delta + (k1.delta + k2.delta * 2.0 + k3.delta * 2.0 + k4.delta) / 6.0 * dt)));
    next_state.a = std::fmin(cfg.max_acceleration,
                             std::fmax(cfg.min_acceleration, cur_state.a + (k1.a + k2.a * 2.0 + k3.a * 2.0 + k4.a) / 6.0 * dt));
    next_state.kappa = std::tan(next_state.delta) / cfg.wheel_base;
    double ds = std::hypot(next_state.x - cur_state.x, next_state.y - cur_state.y);
    next_state.s = cur_state.s + ds;
    return next_state;
  }

  TrackerCfg cfg;
  const Follow* follow_ = nullptr;
  Mat3 lateral_A_, lateral_Q_, longitudinal_A_, longitudinal_Q_;
  double lateral_B_[3], longitudinal_B_[3];
  double lateral_R_ = 0.0, longitudinal_R_ = 0.0;
};

}  // namespace

extern "C" {

/* IlqrOptimizer::InitGuess (ilqr_optimizer.cc:107-139) through Tracker::Plan.
 * cfg (22 doubles): weight_l, weight_theta, weight_delta, weight_delta_rate, lateral preview_time, weight_s, weight_v,
 *   weight_a, weight_j, sumulation_dt, dt, tolerance, max_num_iteration, wheel_base, delta_min, delta_max,
 *   delta_rate_min, delta_rate_max, jerk_min, jerk_max, min_acceleration, max_acceleration
 * start4 = x, y, theta, velocity (start_state_; a, delta, time, s = 0); coarse [K][6] = x y theta v a delta with
 * time_i = i * knot_dt and station[K]; X [K][6], U [K-1][2] out; min_margin (nullable): smallest relative distance of a
 * DARE stopping test to its tolerance.  Returns 0, or -1 when the tracker produced another number of knots. */
int oracle_tracker_init_guess(const double* cfg, const double* start4, const double* coarse, const double* station, int K,
                              double knot_dt, double* X, double* U, double* min_margin) {
  TrackerCfg c;
  c.weight_l = cfg[0]; c.weight_theta = cfg[1]; c.weight_delta = cfg[2]; c.weight_delta_rate = cfg[3]; c.lat_preview_time = cfg[4];
  c.weight_s = cfg[5]; c.weight_v = cfg[6]; c.weight_a = cfg[7]; c.weight_j = cfg[8];
  c.sumulation_dt = cfg[9]; c.dt = cfg[10]; c.tolerance = cfg[11]; c.max_num_iteration = (unsigned)cfg[12];
  c.wheel_base = cfg[13]; c.delta_min = cfg[14]; c.delta_max = cfg[15]; c.delta_rate_min = cfg[16]; c.delta_rate_max = cfg[17];
  c.jerk_min = cfg[18]; c.jerk_max = cfg[19]; c.min_acceleration = cfg[20]; c.max_acceleration = cfg[21];
  Follow follow;
  follow.tr.resize(K);
  for (int i = 0; i < K; ++i) {
    TrajectoryPoint& p = follow.tr[i];
    p.time = knot_dt * i;                        // dp_planner.cpp:236
    p.s = station[i];
    p.x = coarse[i * 6 + 0]; p.y = coarse[i * 6 + 1]; p.theta = coarse[i * 6 + 2];
    p.velocity = coarse[i * 6 + 3]; p.a = coarse[i * 6 + 4]; p.delta = coarse[i * 6 + 5];
  }
  TrajectoryPoint start;
  start.x = start4[0]; start.y = start4[1]; start.theta = start4[2]; start.velocity = start4[3];
  double margin = std::numeric_limits<double>::infinity();
  Tracker tracker(c);
  std::vector<TrajectoryPoint> out;
  const bool ok = tracker.lqr(start, follow, &out, &margin);
  if (min_margin) *min_margin = margin;
  if (!ok) return -1;
  for (int i = 0; i < K; ++i) {                   // cc:121-138
    const TrajectoryPoint& p = out[i];
    X[i * 6 + 0] = p.x; X[i * 6 + 1] = p.y; X[i * 6 + 2] = p.theta; X[i * 6 + 3] = p.velocity; X[i * 6 + 4] = p.a; X[i * 6 + 5] = p.delta;
    if (i < K - 1) {
      U[i * 2 + 0] = p.jerk;
      U[i * 2 + 1] = p.delta_rate;
    }
  }
  return 0;
}


/* test hooks with the signatures of oracle/ref_shim.cc (tests/test_reference_pins.py): the tracker's trajectory queries */
double oracle_tracker_slerp(double a0, double t0, double a1, double t1, double t) { return slerp(a0, t0, a1, t1, t); }
void oracle_tracker_evaluate_time(const double* rows, int n, double time, double* out9) {
  Follow f;
  f.tr.resize(n);
  for (int i = 0; i < n; ++i) {
    const double* r = rows + 9 * i;
    f.tr[i].time = r[0]; f.tr[i].s = r[1]; f.tr[i].x = r[2]; f.tr[i].y = r[3]; f.tr[i].theta = r[4];
    f.tr[i].kappa = r[5]; f.tr[i].velocity = r[6]; f.tr[i].left_bound = r[7]; f.tr[i].right_bound = r[8];
  }
  const TrajectoryPoint p = f.EvaluateTime(time);
  out9[0] = p.time; out9[1] = p.s; out9[2] = p.x; out9[3] = p.y; out9[4] = p.theta; out9[5] = p.kappa; out9[6] = p.velocity;
  out9[7] = p.left_bound; out9[8] = p.right_bound;
}
void oracle_tracker_projection(const double* rows, int n, double px, double py, double* out9) {
  Follow f;
  f.tr.resize(n);
  for (int i = 0; i < n; ++i) {
    const double* r = rows + 9 * i;
    f.tr[i].time = r[0]; f.tr[i].s = r[1]; f.tr[i].x = r[2]; f.tr[i].y = r[3]; f.tr[i].theta = r[4];
    f.tr[i].kappa = r[5]; f.tr[i].velocity = r[6]; f.tr[i].left_bound = r[7]; f.tr[i].right_bound = r[8];
  }
  TrajectoryPoint p;
  f.GetProjection(px, py, &p);
  out9[0] = p.time; out9[1] = p.s; out9[2] = p.x; out9[3] = p.y; out9[4] = p.theta; out9[5] = p.kappa; out9[6] = p.velocity;
  out9[7] = p.left_bound; out9[8] = p.right_bound;
}

}  // extern "C"
