// The reference's types as the C++ programs under tests/cpp see them, in two forms.
//
// -DCILQR_TEST_REFERENCE_HEADERS (build container only: `-I/root/reference`, link oracle/_ref/libcilqr_ref.so, which
// holds the reference's own vec2d / line_segment2d / polygon2d / discretized_trajectory objects):
//     planning::TrajectoryPoint, StartState, DiscretizedTrajectory   algorithm/utils/discretized_trajectory.h:22-43,50
//     planning::math::Vec2d, LineSegment2d, Polygon2d                algorithm/math/*.h
//     planning::Weights, IlqrConfig, CorridorConfig, TrackerConfig, PlannerConfig, VehicleParam
//                                                                    algorithm/params/planner_config.h, vehicle_param.h
//   are the REFERENCE'S OWN declarations: the adapters of include/cilqr/*.hpp are compiled against exactly what
//   trajectory_planner.cpp:26,80-97 hands them.  What stays a stand-in even then: Eigen::Vector3d / Vector2d (no Eigen
//   in this image), planning::Cost (ilqr_optimizer.h:14-27 pulls in Eigen, ROS and OpenCV) and planning::Environment
//   (environment.cpp includes the ROS plotting header).
// default (what travels to the GPU box, where /root/reference does not exist): minimal stand-ins with the same member
//   names, this repository's own definitions.
#pragma once
#include <array>
#include <cmath>
#include <utility>
#include <vector>

#ifdef CILQR_TEST_REFERENCE_HEADERS
#include "algorithm/math/line_segment2d.h"
#include "algorithm/math/polygon2d.h"
#include "algorithm/math/vec2d.h"
#include "algorithm/params/planner_config.h"
#include "algorithm/params/vehicle_param.h"
#include "algorithm/utils/discretized_trajectory.h"
#define CILQR_TEST_TYPES "reference headers"
#else
#define CILQR_TEST_TYPES "stand-ins"
namespace planning {

struct StartState {
  double x, y, theta, v, phi, a, omega;
};
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
namespace math {
struct Vec2d {
  double x_ = 0, y_ = 0;
  Vec2d() = default;
  Vec2d(double x, double y) : x_(x), y_(y) {}
  double x() const { return x_; }
  double y() const { return y_; }
};
struct LineSegment2d {
  Vec2d s, e;
  LineSegment2d(const Vec2d& a, const Vec2d& b) : s(a), e(b) {}
  const Vec2d& start() const { return s; }
  const Vec2d& end() const { return e; }
};
struct Polygon2d {
  std::vector<Vec2d> pts;
  Polygon2d() = default;
  explicit Polygon2d(std::vector<Vec2d> p) : pts(std::move(p)) {}
  const std::vector<Vec2d>& points() const { return pts; }
};
}  // namespace math
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
struct CorridorConfig {
  bool is_multiple_sample = false;
  double max_diff_x = 25.0, max_diff_y = 25.0, radius = 150.0, max_axis_x = 10.0, max_axis_y = 10.0;
  double lane_segment_length = 5.0;
};
struct VehicleParam {
  double front_hang_length = 0.96, wheel_base = 1.0, rear_hang_length = 0.929, width = 1.942;
  double max_velocity = 20.0, min_acceleration = -5.0, max_acceleration = 5.0;
  double jerk_min = -10.0, jerk_max = 10.0;
  double delta_min = -40.0 / 180 * M_PI, delta_max = 40.0 / 180 * M_PI;
  double delta_rate_min = delta_min / 3.0, delta_rate_max = delta_max / 3.0;
};
struct PlannerConfig {
  double delta_t = 0.1, tf = 8;
  double dp_nominal_velocity = 10.0, dp_w_obstacle = 1000, dp_w_lateral = 0.1, dp_w_lateral_change = 0.5;
  double dp_w_lateral_velocity_change = 1.0, dp_w_longitudinal_velocity_bias = 10.0, dp_w_longitudinal_velocity_change = 1.0;
  VehicleParam vehicle;
  CorridorConfig corridor_config;
  IlqrConfig ilqr_config;
};

}  // namespace planning
#endif

namespace planning {
// Stand-ins in BOTH forms (see the header comment): the two Eigen vectors the boundary carries, and Cost.
struct Vector3d {   // Eigen::Vector3d as corridor.h:18-25 uses it: three doubles, operator[]
  double v[3];
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double operator[](int i) const { return v[i]; }
};
struct Vector2d {
  double v[2];
  Vector2d(double a, double b) : v{a, b} {}
  double operator[](int i) const { return v[i]; }
};
struct Cost {       // ilqr_optimizer.h:14-27
  double total_cost = 0, target_cost = 0, dynamic_cost = 0, corridor_cost = 0, lane_boundary_cost = 0;
  Cost() = default;
  Cost(double c0, double c1, double c2, double c3, double c4)
      : total_cost(c0), target_cost(c1), dynamic_cost(c2), corridor_cost(c3), lane_boundary_cost(c4) {}
};
using Constraints = std::vector<Vector3d>;                                         // corridor.h:19-21
using CorridorConstraints = std::vector<Constraints>;
using LaneConstraints = std::vector<std::pair<Vector3d, math::LineSegment2d>>;     // corridor.h:24-25
}  // namespace planning
