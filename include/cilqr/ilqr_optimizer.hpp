// include/cilqr/ilqr_optimizer.hpp -- header-only C++ adapter with the call surface of the
// reference's `planning::IlqrOptimizer` (algorithm/ilqr/ilqr_optimizer.h:29-52), implemented on
// the C-ABI of include/cilqr.h (HIP kernels on MI355X).  It is what "drops into the existing ROS
// node": TrajectoryPlanner (algorithm/planner/trajectory_planner.h:49, .cpp:26,80-86,97) keeps
// constructing it with (ilqr_config, vehicle, tf, delta_t), calling Plan(...) and cost().
//
// The adapter is a template over the reference's own types so that this repository does not have
// to carry (or copy) them.  In the reference tree:
//
//     // algorithm/ilqr/ilqr_optimizer.h
//     #include <cilqr/ilqr_optimizer.hpp>
//     namespace planning {
//     struct Cost { ... unchanged (ilqr_optimizer.h:14-27) ... };
//     using IlqrOptimizer = cilqr::IlqrOptimizerT<TrajectoryPoint, DiscretizedTrajectory,
//                                                 CorridorConstraints, LaneConstraints,
//                                                 IlqrConfig, VehicleParam, Cost>;
//     }
//
// What the template needs from those types (all true for the reference):
//   TrajectoryPoint        public doubles time,x,y,theta,kappa,velocity,a,jerk,delta,delta_rate
//   DiscretizedTrajectory  explicit ctor from std::vector<TrajectoryPoint>; trajectory() -> vector
//   CorridorConstraints    corridor[i][c][0..2]   ("a x + b y < c")
//   LaneConstraints        lane[k].first[0..2]; lane[k].second.start()/.end() with .x()/.y()
//   IlqrConfig             num_of_disc, safe_margin, max_iter_num, abs_cost_tol, rel_cost_tol,
//                          weights.{jerk,delta_rate,x_target,y_target,theta,v,a,delta}
//   VehicleParam           front_hang_length, wheel_base, rear_hang_length, width, max_velocity,
//                          min/max_acceleration, jerk_min/max, delta_min/max, delta_rate_min/max
//   Cost                   ctor (total, target, dynamic, corridor, lane_boundary)
//
// Behavioural notes (kept from the reference):
//   * Plan returns false for null outputs, empty constraints or a knot-count mismatch
//     (ilqr_optimizer.cc:64-78) and true otherwise (the reference falls off the end of the
//     function there; its caller only tests opt_trajectory.empty()).
//   * iter_trajs is appended to (init guess + accepted non-final iterates, cc:170,294), cost()
//     is cleared per Plan (cc:62).
//   * One handle per optimizer, sized for one problem; not re-entrant, like the reference.
#ifndef CILQR_ILQR_OPTIMIZER_HPP_
#define CILQR_ILQR_OPTIMIZER_HPP_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../cilqr.h"

namespace cilqr {

template <class TrajectoryPoint, class DiscretizedTrajectory, class CorridorConstraints,
          class LaneConstraints, class IlqrConfig, class VehicleParam, class Cost>
class IlqrOptimizerT {
 public:
  IlqrOptimizerT() = default;

  IlqrOptimizerT(const IlqrConfig& config, const VehicleParam& param, const double horizon, const double dt) {
    Init(config, param, horizon, dt);
  }

  ~IlqrOptimizerT() { Release(); }

  IlqrOptimizerT(const IlqrOptimizerT& o) { CopyFrom(o); }
  IlqrOptimizerT& operator=(const IlqrOptimizerT& o) {
    if (this != &o) {
      Release();
      CopyFrom(o);
    }
    return *this;
  }

  // ilqr_optimizer.cc:37-51
  void Init(const IlqrConfig& config, const VehicleParam& param, const double horizon, const double dt) {
    Release();
    cilqr_default_config(&cfg_, 1);
    num_of_knots_ = static_cast<int>(std::floor(horizon / dt + 1));   // cc:22
    cfg_.n_steps = num_of_knots_ - 1;
    cfg_.dt = dt;
    cfg_.num_of_disc = config.num_of_disc;
    cfg_.safe_margin = config.safe_margin;
    cfg_.max_iter = config.max_iter_num;
    cfg_.abs_cost_tol = config.abs_cost_tol;
    cfg_.rel_cost_tol = config.rel_cost_tol;
    cfg_.w_jerk = config.weights.jerk;
    cfg_.w_delta_rate = config.weights.delta_rate;
    cfg_.w_x = config.weights.x_target;
    cfg_.w_y = config.weights.y_target;
    cfg_.w_theta = config.weights.theta;
    cfg_.w_v = config.weights.v;
    cfg_.w_a = config.weights.a;
    cfg_.w_delta = config.weights.delta;
    cfg_.front_hang = param.front_hang_length;
    cfg_.wheel_base = param.wheel_base;
    cfg_.rear_hang = param.rear_hang_length;
    cfg_.width = param.width;
    cfg_.max_velocity = param.max_velocity;
    cfg_.min_acceleration = param.min_acceleration;
    cfg_.max_acceleration = param.max_acceleration;
    cfg_.jerk_min = param.jerk_min;
    cfg_.jerk_max = param.jerk_max;
    cfg_.delta_min = param.delta_min;
    cfg_.delta_max = param.delta_max;
    cfg_.delta_rate_min = param.delta_rate_min;
    cfg_.delta_rate_max = param.delta_rate_max;
    configured_ = true;
    cost_.clear();
  }

  // ilqr_optimizer.cc:53-95
  bool Plan(const TrajectoryPoint& start_state, const DiscretizedTrajectory& coarse_traj,
            const CorridorConstraints& corridor, const LaneConstraints& left_lane_cons,
            const LaneConstraints& right_lane_cons, DiscretizedTrajectory* const opt_trajectory,
            std::vector<DiscretizedTrajectory>* const iter_trajs) {
    cost_.clear();
    if (opt_trajectory == nullptr || iter_trajs == nullptr) return false;          // cc:64-66
    if (corridor.size() == 0 || left_lane_cons.size() == 0 || right_lane_cons.size() == 0) {
      std::fprintf(stderr, "ilqr input constraints error\n");                      // cc:68-73
      return false;
    }
    if (!configured_ || static_cast<size_t>(num_of_knots_) != coarse_traj.trajectory().size()) {
      std::fprintf(stderr, "ilqr input coarse_traj error\n");                      // cc:75-78
      return false;
    }
    const auto t_enter = std::chrono::steady_clock::now();
    const int K = num_of_knots_;
    int cmax = 1;
    for (size_t i = 0; i < corridor.size(); ++i) cmax = std::max<int>(cmax, static_cast<int>(corridor[i].size()));
    const int smax = static_cast<int>(std::max(left_lane_cons.size(), right_lane_cons.size()));
    if (!EnsureHandle(cmax, smax)) return false;

    // flatten the reference containers into the problem-major arrays of the C-ABI
    const double start[4] = {start_state.x, start_state.y, start_state.theta, start_state.velocity};
    std::vector<double> coarse(static_cast<size_t>(K) * 6);
    {
      size_t i = 0;
      for (const auto& pt : coarse_traj.trajectory()) {                            // cc:147-150
        double* g = &coarse[i * 6];
        g[0] = pt.x; g[1] = pt.y; g[2] = pt.theta; g[3] = pt.velocity; g[4] = pt.a; g[5] = pt.delta;
        ++i;
      }
    }
    std::vector<double> planes(static_cast<size_t>(K) * cmax_ * 3, 0.0);
    std::vector<int32_t> counts(K, 0);
    for (int i = 0; i < K && i < static_cast<int>(corridor.size()); ++i) {
      counts[i] = static_cast<int32_t>(corridor[i].size());
      for (int c = 0; c < counts[i]; ++c)
        for (int e = 0; e < 3; ++e) planes[(static_cast<size_t>(i) * cmax_ + c) * 3 + e] = corridor[i][c][e];
    }
    std::vector<double> left, right;
    FlattenLane(left_lane_cons, &left);
    FlattenLane(right_lane_cons, &right);

    cilqr_problem_batch in;
    in.batch = 1;
    in.n_knots = K;
    in.cmax = cmax_;
    in.memory = CILQR_MEM_HOST;
    in.start = start;
    in.coarse = coarse.data();
    in.corridor = planes.data();
    in.corridor_count = counts.data();
    in.n_left = static_cast<int32_t>(left_lane_cons.size());
    in.n_right = static_cast<int32_t>(right_lane_cons.size());
    in.left_lane = left.data();
    in.right_lane = right.data();
    in.n_lane_groups = 0;
    in.reserved1 = 0;
    in.lane_group_start = nullptr;
    in.lane_group_left = nullptr;
    in.lane_group_right = nullptr;

    const int max_it = cfg_.max_iter + 1;
    // result buffers live with the object: 0.8 MB of iterates need not be allocated and cleared per call
    std::vector<double>& traj = traj_buf_;
    std::vector<double>& hist = hist_buf_;
    std::vector<double>& iters = iters_buf_;
    traj.resize(static_cast<size_t>(K) * CILQR_TRAJ_FIELDS);
    hist.resize(static_cast<size_t>(max_it) * CILQR_COST_FIELDS);
    iters.resize(static_cast<size_t>(max_it) * K * CILQR_TRAJ_FIELDS);
    int32_t n_cost = 0, status = 0, n_iter = 0, n_it = 0;
    cilqr_solution_batch out;
    out.memory = CILQR_MEM_HOST;
    out.max_iter_trajs = max_it;
    out.traj = traj.data();
    out.cost_hist = hist.data();
    out.n_cost = &n_cost;
    out.status = &status;
    out.n_iter = &n_iter;
    out.iter_trajs = iters.data();
    out.n_iter_trajs = &n_it;
    out.alpha_trace = nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    const int rc = cilqr_solve_batch(handle_, &in, &out);
    const auto t_back = std::chrono::steady_clock::now();
    if (rc != CILQR_OK) {
      std::fprintf(stderr, "cilqr_solve_batch failed: %s\n", cilqr_error_string(rc));
      return false;
    }
    status_ = status;
    if (status == CILQR_ST_NO_CORRIDOR) {   // a knot without corridor: the reference's pipeline stops before
      std::fprintf(stderr, "ilqr input constraints error\n");   // Optimize (trajectory_planner.cpp:49-57)
      return false;
    }
    for (int r = 0; r < n_cost; ++r) {
      const double* c = &hist[static_cast<size_t>(r) * CILQR_COST_FIELDS];
      cost_.push_back(Cost(c[0], c[1], c[2], c[3], c[4]));
    }
    for (int t = 0; t < n_it && t < max_it; ++t)
      iter_trajs->emplace_back(ToTrajectory(&iters[static_cast<size_t>(t) * K * CILQR_TRAJ_FIELDS], K));
    *opt_trajectory = ToTrajectory(traj.data(), K);
    const auto t_leave = std::chrono::steady_clock::now();
    timing_.flatten_ms = std::chrono::duration<double, std::milli>(t_call - t_enter).count();
    timing_.solve_ms = std::chrono::duration<double, std::milli>(t_back - t_call).count();
    timing_.unflatten_ms = std::chrono::duration<double, std::milli>(t_leave - t_back).count();
    return true;
  }

  std::vector<Cost> cost() { return cost_; }            // ilqr_optimizer.h:50-52

  // not in the reference: CILQR_ST_* of the last Plan
  int status() const { return status_; }
  // not in the reference (measurement plumbing, tests/cpp/latency_bench.cc): where the last successful Plan spent its wall
  // time on the host side of the C-ABI, and the handle behind this object (cilqr_set_profiling / cilqr_get_profile)
  struct PlanTiming {
    double flatten_ms = 0.0;     // reference containers -> problem-major arrays (+ the handle, on the first call)
    double solve_ms = 0.0;       // cilqr_solve_batch: transfers, kernels, launch gaps
    double unflatten_ms = 0.0;   // arrays -> DiscretizedTrajectory / Cost objects
  };
  const PlanTiming& last_timing() const { return timing_; }
  cilqr_handle native_handle() const { return handle_; }

 private:
  static DiscretizedTrajectory ToTrajectory(const double* t, int K) {               // cc:771-791
    std::vector<TrajectoryPoint> pts(K);
    for (int i = 0; i < K; ++i) {
      const double* r = t + static_cast<size_t>(i) * CILQR_TRAJ_FIELDS;
      pts[i].time = r[0];
      pts[i].x = r[1];
      pts[i].y = r[2];
      pts[i].theta = r[3];
      pts[i].velocity = r[4];
      pts[i].a = r[5];
      pts[i].delta = r[6];
      pts[i].kappa = r[7];
      if (i < K - 1) {
        pts[i].jerk = r[8];
        pts[i].delta_rate = r[9];
      }
    }
    return DiscretizedTrajectory(pts);
  }

  static void FlattenLane(const LaneConstraints& lane, std::vector<double>* out) {
    out->resize(lane.size() * CILQR_LANE_FIELDS);
    for (size_t k = 0; k < lane.size(); ++k) {
      double* r = &(*out)[k * CILQR_LANE_FIELDS];
      r[0] = lane[k].first[0];
      r[1] = lane[k].first[1];
      r[2] = lane[k].first[2];
      r[3] = lane[k].second.start().x();
      r[4] = lane[k].second.start().y();
      r[5] = lane[k].second.end().x();
      r[6] = lane[k].second.end().y();
    }
  }

  bool EnsureHandle(int cmax, int smax) {
    if (handle_ != nullptr && cmax <= cmax_ && smax <= smax_) return true;
    if (handle_ != nullptr) cilqr_destroy(handle_);
    handle_ = nullptr;
    cmax_ = std::max(cmax, 16);
    smax_ = std::max(smax, 64);
    if (smax_ > CILQR_MAX_LANE_SEGMENTS) return false;
    const int rc = cilqr_create(&cfg_, /*device=*/0, /*batch_capacity=*/1, cmax_, smax_, &handle_);
    if (rc != CILQR_OK) {
      std::fprintf(stderr, "cilqr_create failed: %s\n", cilqr_error_string(rc));
      handle_ = nullptr;
      return false;
    }
    return true;
  }

  void Release() {
    if (handle_ != nullptr) cilqr_destroy(handle_);
    handle_ = nullptr;
  }

  void CopyFrom(const IlqrOptimizerT& o) {   // the reference class is copyable (used by value)
    cfg_ = o.cfg_;
    num_of_knots_ = o.num_of_knots_;
    configured_ = o.configured_;
    cost_ = o.cost_;
    status_ = o.status_;
    handle_ = nullptr;   // device state is re-created lazily
    cmax_ = smax_ = 0;
  }

  cilqr_config cfg_{};
  int num_of_knots_ = 0;
  bool configured_ = false;
  cilqr_handle handle_ = nullptr;
  int cmax_ = 0, smax_ = 0;
  int status_ = 0;
  std::vector<Cost> cost_;
  std::vector<double> traj_buf_, hist_buf_, iters_buf_;
  PlanTiming timing_;
};

}  // namespace cilqr

#endif  // CILQR_ILQR_OPTIMIZER_HPP_
