// Minimal stand-ins for the reference's own types, shared by the C++ programs under tests/cpp that drive
// include/cilqr/ilqr_optimizer.hpp the way the reference's TrajectoryPlanner drives planning::IlqrOptimizer
// (algorithm/planner/trajectory_planner.cpp:26,80-97).  Only the members the adapter touches; these are this
// repository's definitions, the reference headers are not copied.
#pragma once
#include <array>
#include <cmath>
#include <utility>
#include <vector>

#include "cilqr/ilqr_optimizer.hpp"

namespace planning {

struct TrajectoryPoint {
  double time = 0.0, s = 0.0, x = 0.0, y = 0.0, theta = 0.0, kappa = 0.0, velocity = 0.0;
  double a = 0.0, jerk = 0.0, delta = 0.0, delta_rate = 0.0, left_bound = 0.0, right_bound = 0.0;
};
class DiscretizedTrajectory {
 public:
  DiscretizedTrajectory() = default;
  explicit DiscretizedTrajectory(const std::vector<TrajectoryPoint>& p) : pts_(p) {}
  const std::vector<TrajectoryPoint>& trajectory() const { return pts_; }
  bool empty() const { return pts_.empty(); }

 private:
  std::vector<TrajectoryPoint> pts_;
};
struct Vec2 {
  double x_, y_;
  double x() const { return x_; }
  double y() const { return y_; }
};
struct Segment {
  Vec2 s, e;
  const Vec2& start() const { return s; }
  const Vec2& end() const { return e; }
};
using Vector3 = std::array<double, 3>;
using Constraints = std::vector<Vector3>;
using CorridorConstraints = std::vector<Constraints>;
using LaneConstraints = std::vector<std::pair<Vector3, Segment>>;
struct Weights {
  double jerk = 1, delta_rate = 1, x_target = 0.5, y_target = 0.5, theta = 1e-3, v = 0.0, a = 0.0, delta = 0.0;
};
struct IlqrConfig {
  int num_of_disc = 5;
  double safe_margin = 0.2;
  Weights weights;
  int max_iter_num = 200;
  double abs_cost_tol = 1e-2, rel_cost_tol = 1e-2;
};
struct VehicleParam {
  double front_hang_length = 0.96, wheel_base = 1.0, rear_hang_length = 0.929, width = 1.942;
  double max_velocity = 20.0, min_acceleration = -5.0, max_acceleration = 5.0;
  double jerk_min = -10.0, jerk_max = 10.0;
  double delta_min = -40.0 / 180 * M_PI, delta_max = 40.0 / 180 * M_PI;
  double delta_rate_min = delta_min / 3.0, delta_rate_max = delta_max / 3.0;
};
struct Cost {
  double total_cost = 0, target_cost = 0, dynamic_cost = 0, corridor_cost = 0, lane_boundary_cost = 0;
  Cost() = default;
  Cost(double c0, double c1, double c2, double c3, double c4)
      : total_cost(c0), target_cost(c1), dynamic_cost(c2), corridor_cost(c3), lane_boundary_cost(c4) {}
};
using IlqrOptimizer = cilqr::IlqrOptimizerT<TrajectoryPoint, DiscretizedTrajectory, CorridorConstraints,
                                            LaneConstraints, IlqrConfig, VehicleParam, Cost>;

}  // namespace planning
