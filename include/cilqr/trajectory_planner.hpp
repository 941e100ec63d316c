// include/cilqr/trajectory_planner.hpp -- header-only C++14 adapters with the call surfaces of the reference's
// `planning::DpPlanner` (algorithm/planner/dp_planner.h:31-38) and `planning::TrajectoryPlanner`
// (algorithm/planner/trajectory_planner.h:22-43): the pipeline DP coarse planner -> safe corridor -> CILQR that
// PlanningNode drives (algorithm/planning_node.cc:11,86), minus the plotting.
//
// Templates over the reference's own types, like include/cilqr/ilqr_optimizer.hpp and corridor.hpp (this repository
// neither carries nor copies them).  In the reference tree:
//
//     // algorithm/planner/dp_planner.h
//     #include <cilqr/trajectory_planner.hpp>
//     namespace planning {
//     using DpPlanner = cilqr::DpPlannerT<PlannerConfig, Env, DiscretizedTrajectory, TrajectoryPoint>;
//     }
//     // algorithm/planner/trajectory_planner.h
//     namespace planning {
//     using TrajectoryPlanner = cilqr::TrajectoryPlannerT<PlannerConfig, Env, StartState, TrajectoryPoint,
//                                                         DiscretizedTrajectory, DpPlanner, Corridor, IlqrOptimizer,
//                                                         math::Vec2d, math::LineSegment2d>;
//     }
//
// What the templates need (all true for the reference):
//   PlannerConfig          tf, delta_t, dp_nominal_velocity, dp_w_obstacle, dp_w_lateral, dp_w_lateral_change,
//                          dp_w_lateral_velocity_change, dp_w_longitudinal_velocity_bias,
//                          dp_w_longitudinal_velocity_change, vehicle.{front_hang_length, wheel_base, rear_hang_length,
//                          width, max_velocity}, corridor_config, ilqr_config
//   Env                    pointer-like; ->reference().trajectory() (points with s x y theta kappa left_bound
//                          right_bound), ->obstacles() (polygons with points() -> x(), y()), ->dynamic_obstacles()
//                          (vector of vector<pair<time, polygon>>)
//   StartState             x, y, theta, v
//   TrajectoryPoint        the thirteen public doubles of discretized_trajectory.h:26-43
//   Corridor               Corridor(corridor_config, env); Plan(...) corridor.h:33-38; points_for_corridors();
//                          the nested typedefs of cilqr::CorridorT (or the reference's corridor.h:18-25 at namespace scope)
//   IlqrOptimizer          (ilqr_config, vehicle, tf, delta_t); Plan(...) ilqr_optimizer.h:41-48
//
// Behaviour kept from the reference: Plan returns false when the DP finds no collision-free path ("DP failed"), when
// the corridor cannot be built, or when the optimiser leaves opt_trajectory empty (trajectory_planner.cpp:32-35,
// 49-57, 91-94); the bool the optimiser returns is ignored; the result carries re-accumulated stations and
// kappa = tan(delta) / wheel_base (cpp:101-125).
#ifndef CILQR_TRAJECTORY_PLANNER_HPP_
#define CILQR_TRAJECTORY_PLANNER_HPP_

#include <cmath>
#include <vector>

#include "dp_planner.hpp"

namespace cilqr {

template <class PlannerConfig, class Env, class DiscretizedTrajectory, class TrajectoryPoint>
class DpPlannerT {
 public:
  DpPlannerT(const PlannerConfig& config, const Env& env) : env_(env), config_(config) {}

  // dp_planner.cpp:135-281
  bool Plan(const double start_x, const double start_y, const double start_theta, DiscretizedTrajectory& result) {
    DpConfig c;
    c.tf = config_.tf; c.delta_t = config_.delta_t; c.dp_nominal_velocity = config_.dp_nominal_velocity;
    c.dp_w_obstacle = config_.dp_w_obstacle; c.dp_w_lateral = config_.dp_w_lateral;
    c.dp_w_lateral_change = config_.dp_w_lateral_change;
    c.dp_w_lateral_velocity_change = config_.dp_w_lateral_velocity_change;
    c.dp_w_longitudinal_velocity_bias = config_.dp_w_longitudinal_velocity_bias;
    c.dp_w_longitudinal_velocity_change = config_.dp_w_longitudinal_velocity_change;
    c.front_hang_length = config_.vehicle.front_hang_length; c.wheel_base = config_.vehicle.wheel_base;
    c.rear_hang_length = config_.vehicle.rear_hang_length; c.width = config_.vehicle.width;
    c.max_velocity = config_.vehicle.max_velocity;
    // the scene as the reference's Environment holds it at this moment (its callbacks may have changed it)
    std::vector<std::array<double, 7>> center;
    for (const auto& p : env_->reference().trajectory())
      center.push_back(std::array<double, 7>{{p.s, p.x, p.y, p.theta, p.kappa, p.left_bound, p.right_bound}});
    if (center.size() < 2) return false;
    DpEnvironment scene(c, ReferenceLine(center));
    for (auto& obstacle : env_->obstacles()) {
      std::vector<DpPoint2> poly;
      for (const auto& v : obstacle.points()) poly.push_back(DpPoint2{v.x(), v.y()});
      scene.AddStatic(poly);
    }
    for (auto& obstacle : env_->dynamic_obstacles()) {
      std::vector<double> times;
      std::vector<std::vector<DpPoint2>> polys;
      for (auto& sample : obstacle) {
        times.push_back(sample.first);
        std::vector<DpPoint2> poly;
        for (const auto& v : sample.second.points()) poly.push_back(DpPoint2{v.x(), v.y()});
        polys.push_back(poly);
      }
      scene.AddDynamicPlaced(times, polys);
    }
    DpPlanner dp(c, &scene);
    std::vector<CoarsePoint> coarse;
    const bool ok = dp.Plan(start_x, start_y, start_theta, &coarse);
    std::vector<TrajectoryPoint> data(coarse.size());
    for (size_t i = 0; i < coarse.size(); ++i) {
      data[i].time = coarse[i].time; data[i].s = coarse[i].s; data[i].x = coarse[i].x; data[i].y = coarse[i].y;
      data[i].theta = coarse[i].theta; data[i].kappa = coarse[i].kappa; data[i].delta = coarse[i].delta;
      data[i].velocity = coarse[i].velocity; data[i].a = coarse[i].a;
      data[i].jerk = 0.0; data[i].delta_rate = 0.0;
    }
    result = DiscretizedTrajectory(data);
    return ok;
  }

 private:
  Env env_;
  PlannerConfig config_;
};

template <class PlannerConfig, class Env, class StartState, class TrajectoryPoint, class DiscretizedTrajectory,
          class DpPlanner, class Corridor, class IlqrOptimizer, class Vec2d, class LineSegment2d>
class TrajectoryPlannerT {
 public:
  using CorridorConstraints = typename Corridor::CorridorConstraints;
  using ConvexPolygons = typename Corridor::ConvexPolygons;
  using LaneConstraints = typename Corridor::LaneConstraints;

  TrajectoryPlannerT(const PlannerConfig& config, const Env& env)                        // trajectory_planner.cpp:22-26
      : config_(config), dp_(config, env), corridor_(config.corridor_config, env),
        ilqr_optimizer_(config.ilqr_config, config.vehicle, config.tf, config.delta_t) {}

  bool Plan(const StartState& state, DiscretizedTrajectory& result) {                    // cpp:28-162
    DiscretizedTrajectory coarse_trajectory;
    if (!dp_.Plan(state.x, state.y, state.theta, coarse_trajectory)) return false;       // "DP failed"
    CorridorConstraints corridor_constraints;
    ConvexPolygons convex_polygons;
    LaneConstraints left_lane_constraints, right_lane_constraints;
    if (!corridor_.Plan(coarse_trajectory, &corridor_constraints, &convex_polygons, &left_lane_constraints,
                        &right_lane_constraints))
      return false;                                                                      // "Corridor failed"
    TrajectoryPoint start_state;
    start_state.x = state.x; start_state.y = state.y;
    start_state.velocity = state.v; start_state.theta = state.theta;
    DiscretizedTrajectory opt_trajectory;
    iter_trajs_.clear();
    (void)ilqr_optimizer_.Plan(start_state, coarse_trajectory, corridor_constraints, left_lane_constraints,
                               right_lane_constraints, &opt_trajectory, &iter_trajs_);
    if (opt_trajectory.empty()) return false;                                            // "ilqr failed"
    coarse_trajectory_ = coarse_trajectory;
    std::vector<TrajectoryPoint> result_data;
    double incremental_s = 0.0;
    const int nfe = config_.tf / config_.delta_t + 1;
    const auto& opt = opt_trajectory.trajectory();
    for (int i = 0; i < nfe; i++) {                                                      // cpp:101-125
      TrajectoryPoint tp;
      tp.time = config_.delta_t * i;
      incremental_s += i > 0 ? std::hypot(opt[i].x - opt[i - 1].x, opt[i].y - opt[i - 1].y) : 0.0;
      tp.s = incremental_s;
      tp.x = opt[i].x;
      tp.y = opt[i].y;
      tp.theta = opt[i].theta;
      tp.velocity = opt[i].velocity;
      tp.kappa = std::tan(opt[i].delta) / config_.vehicle.wheel_base;
      tp.a = opt[i].a;
      tp.jerk = opt[i].jerk;
      tp.delta = opt[i].delta;
      tp.delta_rate = opt[i].delta_rate;
      result_data.push_back(tp);
    }
    result = DiscretizedTrajectory(result_data);
    convex_polygons_ = convex_polygons;
    left_lane_boundary_.clear();
    for (const auto& seg : left_lane_constraints) left_lane_boundary_.push_back(seg.second);
    right_lane_boundary_.clear();
    for (const auto& seg : right_lane_constraints) right_lane_boundary_.push_back(seg.second);
    return true;
  }

  ConvexPolygons SafeCorridors() { return convex_polygons_; }                            // trajectory_planner.h:27-41
  std::vector<std::vector<Vec2d>> points_for_corridors() { return corridor_.points_for_corridors(); }
  std::vector<LineSegment2d> left_lane_boundary() { return left_lane_boundary_; }
  std::vector<LineSegment2d> right_lane_boundary() { return right_lane_boundary_; }

  // not in the reference (its Plan hands these to the plots, cpp:96-97): the stages' own outputs of the last Plan
  const DiscretizedTrajectory& coarse_trajectory() const { return coarse_trajectory_; }
  const std::vector<DiscretizedTrajectory>& iteration_trajectories() const { return iter_trajs_; }
  IlqrOptimizer& ilqr_optimizer() { return ilqr_optimizer_; }

 private:
  PlannerConfig config_;
  DpPlanner dp_;
  Corridor corridor_;
  IlqrOptimizer ilqr_optimizer_;
  ConvexPolygons convex_polygons_;
  std::vector<LineSegment2d> left_lane_boundary_, right_lane_boundary_;
  DiscretizedTrajectory coarse_trajectory_;
  std::vector<DiscretizedTrajectory> iter_trajs_;
};

}  // namespace cilqr

#endif  // CILQR_TRAJECTORY_PLANNER_HPP_
