/*
 * oracle/dp_oracle.cc -- CPU oracle of the coarse-trajectory producer.
 *
 * TEST INFRASTRUCTURE ONLY (see cilqr_oracle.h).  PARITY UNPINNED for DpPlanner and Environment: the reference
 * holds no tests or vectors for this stage and dp_planner.cpp / environment.cpp cannot be built here (ROS headers).
 * PINNED against the reference's own code, bit for bit (oracle/_ref built from the reference's files by g++ alone;
 * tests/test_reference_pins.py): ComputePathProfile, the DiscretizedTrajectory station / projection / cartesian
 * queries, Polygon2d::HasOverlap(Box2d(AABox2d)) and Polygon2d::IsPointIn.  This is a line-by-line restatement, with the
 * reference's own structure (one small class per reference class, every helper re-evaluated where the
 * reference re-evaluates it, no caching), of
 *   algorithm/planner/dp_planner.{h,cpp}                      DpPlanner
 *   algorithm/utils/discrete_points_math.cc:27-176            ComputePathProfile
 *   algorithm/utils/discretized_trajectory.cpp:33-203         station / projection queries
 *   algorithm/utils/environment.cpp:20-130                    road barriers, collision checks
 *   algorithm/math/{polygon2d,box2d,aabox2d}.cpp, math_utils.{h,cpp}   the geometry those use
 * The product's planner (include/cilqr/dp_planner.hpp) is held to it bit for bit (tests/test_dp_planner.py).
 */
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

namespace {

namespace math {
constexpr double kMathEpsilon = 1e-10;   // vec2d.h:33

struct Vec2d {
  double x_ = 0.0, y_ = 0.0;
  Vec2d() = default;
  Vec2d(double x, double y) : x_(x), y_(y) {}
  double x() const { return x_; }
  double y() const { return y_; }
};

double NormalizeAngle(const double angle) {   // math_utils.cpp:53-59
  double a = std::fmod(angle + M_PI, 2.0 * M_PI);
  if (a < 0.0) a += (2.0 * M_PI);
  return a - M_PI;
}

double slerp(const double a0, const double t0, const double a1, const double t1, const double t) {   // math_utils.h:208
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

template <int N>
std::array<double, N> LinSpaced(double start, double end) {   // math_utils.h:245
  std::array<double, N> res;
  double step = (end - start) / (N - 1);
  for (int i = 0; i < N; i++) res[i] = start + step * i;
  return res;
}

double CrossProd(const Vec2d& start_point, const Vec2d& end_point_1, const Vec2d& end_point_2) {   // math_utils.cpp:28
  const double ax = end_point_1.x() - start_point.x(), ay = end_point_1.y() - start_point.y();
  const double bx = end_point_2.x() - start_point.x(), by = end_point_2.y() - start_point.y();
  return ax * by - ay * bx;   // Vec2d::CrossProd
}

// Box2d built from an AABox2d (box2d.cpp:93-105): heading 0
struct Box2d {
  Vec2d center_;
  double half_length_ = 0.0, half_width_ = 0.0;
  double cos_heading_ = 1.0, sin_heading_ = 0.0;
  double min_x_ = 0, min_y_ = 0, max_x_ = 0, max_y_ = 0;
  std::vector<Vec2d> corners_;
  // AABox2d(one_corner, opposite_corner) (aabox2d.cpp:38-41) shifted (aabox2d.cpp:120)
  static Box2d FromAABox(const Vec2d& one_corner, const Vec2d& opposite_corner, const Vec2d& shift) {
    Box2d b;
    Vec2d center((one_corner.x() + opposite_corner.x()) / 2.0, (one_corner.y() + opposite_corner.y()) / 2.0);
    const double length = std::abs(one_corner.x() - opposite_corner.x());
    const double width = std::abs(one_corner.y() - opposite_corner.y());
    center = Vec2d(center.x() + shift.x(), center.y() + shift.y());
    b.center_ = center;
    b.half_length_ = length / 2.0;
    b.half_width_ = width / 2.0;
    b.min_x_ = center.x() - b.half_length_;   // AABox2d::min_x() etc. (aabox2d.h)
    b.max_x_ = center.x() + b.half_length_;
    b.min_y_ = center.y() - b.half_width_;
    b.max_y_ = center.y() + b.half_width_;
    b.corners_.emplace_back(center.x() + b.half_length_, center.y() - b.half_width_);   // aabox2d.cpp:63-71
    b.corners_.emplace_back(center.x() + b.half_length_, center.y() + b.half_width_);
    b.corners_.emplace_back(center.x() - b.half_length_, center.y() + b.half_width_);
    b.corners_.emplace_back(center.x() - b.half_length_, center.y() - b.half_width_);
    return b;
  }
  double min_x() const { return min_x_; }
  double max_x() const { return max_x_; }
  double min_y() const { return min_y_; }
  double max_y() const { return max_y_; }
  const std::vector<Vec2d>& corners() const { return corners_; }
  bool IsPointIn(const Vec2d& point) const {   // box2d.cpp:123-129
    const double x0 = point.x() - center_.x();
    const double y0 = point.y() - center_.y();
    const double dx = std::abs(x0 * cos_heading_ + y0 * sin_heading_);
    const double dy = std::abs(-x0 * sin_heading_ + y0 * cos_heading_);
    return dx <= half_length_ + kMathEpsilon && dy <= half_width_ + kMathEpsilon;
  }
};

class Polygon2d {
 public:
  Polygon2d() = default;
  explicit Polygon2d(std::vector<Vec2d> points) : points_(std::move(points)) { BuildFromPoints(); }
  double min_x() const { return min_x_; }
  double max_x() const { return max_x_; }
  double min_y() const { return min_y_; }
  double max_y() const { return max_y_; }

  bool IsPointIn(const Vec2d& point) const {   // polygon2d.cpp:120-140
    if (point.x() < min_x() || point.x() > max_x() || point.y() < min_y() || point.y() > max_y()) return false;
    int j = num_points_ - 1;
    int c = 0;
    for (int i = 0; i < num_points_; ++i) {
      if ((points_[i].y() > point.y()) != (points_[j].y() > point.y())) {
        const double side = CrossProd(point, points_[i], points_[j]);
        if (points_[i].y() < points_[j].y() ? side > 0.0 : side < 0.0) ++c;
      }
      j = i;
    }
    return c & 1;
  }

  bool HasOverlap(const Box2d& box) const {   // polygon2d.cpp:150-164
    if (box.max_x() < min_x() || box.min_x() > max_x() || box.max_y() < min_y() || box.min_y() > max_y()) return false;
    for (auto& pt : points_)
      if (box.IsPointIn(pt)) return true;
    for (auto& corner : box.corners())
      if (IsPointIn(corner)) return true;
    return false;
  }

 private:
  void BuildFromPoints() {   // polygon2d.cpp:200-257 (the parts a collision test reads)
    num_points_ = static_cast<int>(points_.size());
    double area = 0.0;
    for (int i = 1; i < num_points_; ++i) area += CrossProd(points_[0], points_[i - 1], points_[i]);
    if (area < 0) std::reverse(points_.begin(), points_.end());
    min_x_ = points_[0].x();
    max_x_ = points_[0].x();
    min_y_ = points_[0].y();
    max_y_ = points_[0].y();
    for (const auto& point : points_) {
      min_x_ = std::min(min_x_, point.x());
      max_x_ = std::max(max_x_, point.x());
      min_y_ = std::min(min_y_, point.y());
      max_y_ = std::max(max_y_, point.y());
    }
  }
  std::vector<Vec2d> points_;
  int num_points_ = 0;
  double min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0;
};
}  // namespace math

using math::Vec2d;

struct TrajectoryPoint {   // discretized_trajectory.h:26-43
  double time = 0.0, s = 0.0, x = 0.0, y = 0.0, theta = 0.0, kappa = 0.0, velocity = 0.0;
  double a = 0.0, jerk = 0.0, delta = 0.0, delta_rate = 0.0, left_bound = 0.0, right_bound = 0.0;
};
typedef std::vector<TrajectoryPoint> Trajectory;

TrajectoryPoint LinearInterpolateTrajectory(const TrajectoryPoint& p0, const TrajectoryPoint& p1, const double s) {   // cpp:66
  double s0 = p0.s;
  double s1 = p1.s;
  if (std::abs(s1 - s0) < math::kMathEpsilon) return p0;
  TrajectoryPoint pt;
  double weight = (s - s0) / (s1 - s0);
  pt.time = (1 - weight) * p0.time + weight * p1.time;
  pt.s = s;
  pt.x = (1 - weight) * p0.x + weight * p1.x;
  pt.y = (1 - weight) * p0.y + weight * p1.y;
  pt.theta = math::slerp(p0.theta, p0.s, p1.theta, p1.s, s);
  pt.kappa = (1 - weight) * p0.kappa + weight * p1.kappa;
  pt.velocity = (1 - weight) * p0.velocity + weight * p1.velocity;
  pt.left_bound = (1 - weight) * p0.left_bound + weight * p1.left_bound;
  pt.right_bound = (1 - weight) * p0.right_bound + weight * p1.right_bound;
  return pt;
}

class DiscretizedTrajectory {
 public:
  DiscretizedTrajectory() = default;
  explicit DiscretizedTrajectory(const std::vector<TrajectoryPoint>& points) : trajectory_(points) {}
  const Trajectory& trajectory() const { return trajectory_; }

  Trajectory::const_iterator QueryLowerBoundStationPoint(const double station) const {   // cpp:33-47
    if (station >= trajectory_.back().s) {
      return trajectory_.end() - 1;
    } else if (station < trajectory_.front().s) {
      return trajectory_.begin();
    }
    return std::lower_bound(trajectory_.begin(), trajectory_.end(), station,
                            [](const TrajectoryPoint& t, const double station) { return t.s < station; });
  }
  TrajectoryPoint EvaluateStation(const double station) const {   // cpp:117-128
    auto iter = QueryLowerBoundStationPoint(station);
    if (iter == trajectory_.begin()) iter = std::next(iter);
    auto prev = std::prev(iter, 1);
    return LinearInterpolateTrajectory(*prev, *iter, station);
  }
  Trajectory::const_iterator QueryNearestPoint(const Vec2d& point) const {   // cpp:143-163
    auto nearest_iter = trajectory_.begin();
    double nearest_distance = std::numeric_limits<double>::max();
    for (auto iter = trajectory_.begin(); iter != trajectory_.end(); iter++) {
      double dx = iter->x - point.x(), dy = iter->y - point.y();
      double distance = dx * dx + dy * dy;
      if (distance < nearest_distance) {
        nearest_iter = iter;
        nearest_distance = distance;
      }
    }
    return nearest_iter;
  }
  Vec2d GetProjection(const Vec2d& xy) const {   // cpp:165-197
    long point_idx = std::distance(trajectory_.begin(), QueryNearestPoint(xy));
    auto project_point = trajectory_[point_idx];
    auto index_start = std::max(0l, point_idx - 1);
    auto index_end = std::min(trajectory_.size() - 1, (unsigned long)point_idx + 1);
    if ((unsigned long)index_start < index_end) {
      double v0x = xy.x() - trajectory_[index_start].x;
      double v0y = xy.y() - trajectory_[index_start].y;
      double v1x = trajectory_[index_end].x - trajectory_[index_start].x;
      double v1y = trajectory_[index_end].y - trajectory_[index_start].y;
      double v1_norm = std::sqrt(v1x * v1x + v1y * v1y);
      double dot = v0x * v1x + v0y * v1y;
      double delta_s = dot / v1_norm;
      project_point = LinearInterpolateTrajectory(trajectory_[index_start], trajectory_[index_end],
                                                  trajectory_[index_start].s + delta_s);
    }
    double nr_x = xy.x() - project_point.x, nr_y = xy.y() - project_point.y;
    double lateral = copysign(hypot(nr_x, nr_y), nr_y * std::cos(project_point.theta) - nr_x * std::sin(project_point.theta));
    return {project_point.s, lateral};
  }
  Vec2d GetCartesian(const double station, const double lateral) const {   // cpp:199-203
    auto ref = EvaluateStation(station);
    return {ref.x - lateral * std::sin(ref.theta), ref.y + lateral * std::cos(ref.theta)};
  }

 private:
  Trajectory trajectory_;
};

struct VehicleParam {   // vehicle_param.h:26-95
  double front_hang_length = 0.96, wheel_base = 1.0, rear_hang_length = 0.929, width = 1.942, max_velocity = 20.0;
  double radius = 0.0, f2x = 0.0, r2x = 0.0;
  void Finish() {
    double length = (wheel_base + rear_hang_length + front_hang_length);
    radius = hypot(0.25 * length, 0.5 * width);
    r2x = 0.25 * length - rear_hang_length;
    f2x = 0.75 * length - rear_hang_length;
  }
  std::tuple<double, double, double, double> GetDiscPositions(const double& x, const double& y, const double& theta) const {
    auto xf = x + f2x * cos(theta);
    auto xr = x + r2x * cos(theta);
    auto yf = y + f2x * sin(theta);
    auto yr = y + r2x * sin(theta);
    return std::make_tuple(xf, yf, xr, yr);
  }
};

struct PlannerConfig {   // planner_config.h:88-133 (the fields DpPlanner reads)
  double delta_t = 0.1, tf = 8;
  double dp_nominal_velocity = 10.0, dp_w_obstacle = 1000, dp_w_lateral = 0.1, dp_w_lateral_change = 0.5;
  double dp_w_lateral_velocity_change = 1.0, dp_w_longitudinal_velocity_bias = 10.0, dp_w_longitudinal_velocity_change = 1.0;
  VehicleParam vehicle;
};

class Environment {   // environment.{h,cpp}
 public:
  using DynamicObstacle = std::vector<std::pair<double, math::Polygon2d>>;
  explicit Environment(const PlannerConfig& config) : config_(config) {}
  std::vector<math::Polygon2d>& obstacles() { return obstacles_; }
  std::vector<DynamicObstacle>& dynamic_obstacles() { return dynamic_obstacles_; }
  const DiscretizedTrajectory& reference() const { return reference_; }

  void set_reference(const DiscretizedTrajectory& reference) {   // cpp:20-43
    constexpr double kSampleStep = 0.1;
    reference_ = reference;
    road_barrier_.clear();
    double start_s = reference_.trajectory().front().s;
    double back_s = reference_.trajectory().back().s;
    int sample_points = int((back_s - start_s) / kSampleStep);
    for (int i = 0; i <= sample_points; i++) {
      double s = start_s + i * kSampleStep;
      auto ref = reference_.EvaluateStation(s);
      road_barrier_.push_back(reference_.GetCartesian(s, ref.left_bound));
      road_barrier_.push_back(reference_.GetCartesian(s, -ref.right_bound));
    }
    std::sort(road_barrier_.begin(), road_barrier_.end(), [](const Vec2d& a, const Vec2d& b) { return a.x() < b.x(); });
  }

  bool CheckOptimizationCollision(const double time, const double px, const double py, const double ptheta) {   // cpp:92-111
    const double collision_buffer = 0.0;
    const Vec2d c0(-config_.vehicle.radius - collision_buffer, -config_.vehicle.radius - collision_buffer);
    const Vec2d c1(config_.vehicle.radius + collision_buffer, config_.vehicle.radius + collision_buffer);
    double xr, yr, xf, yf;
    std::tie(xr, yr, xf, yf) = config_.vehicle.GetDiscPositions(px, py, ptheta);
    const math::Box2d f_box = math::Box2d::FromAABox(c0, c1, Vec2d(xf, yf));
    const math::Box2d r_box = math::Box2d::FromAABox(c0, c1, Vec2d(xr, yr));
    if (CheckStaticCollision(f_box) || CheckStaticCollision(r_box) || CheckDynamicCollision(time, f_box) ||
        CheckDynamicCollision(time, r_box)) {
      return true;
    }
    return false;
  }

 private:
  bool CheckStaticCollision(const math::Box2d& rect) {   // cpp:45-80
    for (auto& obstacle : obstacles_)
      if (obstacle.HasOverlap(rect)) return true;
    if (road_barrier_.empty()) return false;
    if (rect.max_x() < road_barrier_.front().x() || rect.min_x() > road_barrier_.back().x()) return false;
    auto comp = [](const double val, const Vec2d& a) { return val < a.x(); };
    auto check_start = std::upper_bound(road_barrier_.begin(), road_barrier_.end(), rect.min_x(), comp);
    auto check_end = std::upper_bound(road_barrier_.begin(), road_barrier_.end(), rect.max_x(), comp);
    if (check_start > road_barrier_.begin()) std::advance(check_start, -1);
    for (auto iter = check_start; iter != check_end; iter++)
      if (rect.IsPointIn(*iter)) return true;
    return false;
  }
  bool CheckDynamicCollision(const double time, const math::Box2d& rect) {   // cpp:113-130
    for (auto& obstacle : dynamic_obstacles_) {
      if (obstacle.front().first > time || obstacle.back().first < time) continue;
      auto result = std::upper_bound(obstacle.begin(), obstacle.end(), time,
                                     [](const double val, const std::pair<double, math::Polygon2d>& ob) { return val < ob.first; });
      if (result == obstacle.end()) result = obstacle.end() - 1;   // the reference dereferences end(): last sample here
      if (result->second.HasOverlap(rect)) return true;
    }
    return false;
  }
  PlannerConfig config_;
  std::vector<DynamicObstacle> dynamic_obstacles_;
  std::vector<math::Polygon2d> obstacles_;
  std::vector<Vec2d> road_barrier_;
  DiscretizedTrajectory reference_;
};
using Env = std::shared_ptr<Environment>;

bool ComputePathProfile(const double dt, const std::vector<std::pair<double, double>>& xy_points, std::vector<double>* headings,
                        std::vector<double>* accumulated_s, std::vector<double>* speeds, std::vector<double>* accelerations,
                        std::vector<double>* kappas) {   // discrete_points_math.cc:27-176
  headings->clear();
  accumulated_s->clear();
  speeds->clear();
  accelerations->clear();
  kappas->clear();
  if (xy_points.size() < 2) return false;
  std::vector<double> dxs, dys, y_over_s_first_derivatives, x_over_s_first_derivatives, y_over_s_second_derivatives,
      x_over_s_second_derivatives;
  std::size_t points_size = xy_points.size();
  for (std::size_t i = 0; i < points_size; ++i) {
    double x_delta = 0.0, y_delta = 0.0;
    if (i == 0) {
      x_delta = (xy_points[i + 1].first - xy_points[i].first);
      y_delta = (xy_points[i + 1].second - xy_points[i].second);
    } else if (i == points_size - 1) {
      x_delta = (xy_points[i].first - xy_points[i - 1].first);
      y_delta = (xy_points[i].second - xy_points[i - 1].second);
    } else {
      x_delta = 0.5 * (xy_points[i + 1].first - xy_points[i - 1].first);
      y_delta = 0.5 * (xy_points[i + 1].second - xy_points[i - 1].second);
    }
    dxs.push_back(x_delta);
    dys.push_back(y_delta);
  }
  for (std::size_t i = 0; i < points_size; ++i) headings->push_back(std::atan2(dys[i], dxs[i]));
  double distance = 0.0;
  accumulated_s->push_back(distance);
  double fx = xy_points[0].first, fy = xy_points[0].second, nx = 0.0, ny = 0.0;
  for (std::size_t i = 1; i < points_size; ++i) {
    nx = xy_points[i].first;
    ny = xy_points[i].second;
    double end_segment_s = std::sqrt((fx - nx) * (fx - nx) + (fy - ny) * (fy - ny));
    accumulated_s->push_back(end_segment_s + distance);
    distance += end_segment_s;
    fx = nx;
    fy = ny;
  }
  for (std::size_t i = 1; i < accumulated_s->size(); ++i) speeds->push_back((accumulated_s->at(i) - accumulated_s->at(i - 1)) / dt);
  double v = speeds->back();
  speeds->push_back(v);
  for (std::size_t i = 1; i < speeds->size(); ++i) accelerations->push_back((speeds->at(i) - speeds->at(i - 1)) / dt);
  double a = accelerations->back();
  accelerations->push_back(a);
  for (std::size_t i = 0; i < points_size; ++i) {
    double xds = 0.0, yds = 0.0;
    if (i == 0) {
      xds = (xy_points[i + 1].first - xy_points[i].first) / (accumulated_s->at(i + 1) - accumulated_s->at(i));
      yds = (xy_points[i + 1].second - xy_points[i].second) / (accumulated_s->at(i + 1) - accumulated_s->at(i));
    } else if (i == points_size - 1) {
      xds = (xy_points[i].first - xy_points[i - 1].first) / (accumulated_s->at(i) - accumulated_s->at(i - 1));
      yds = (xy_points[i].second - xy_points[i - 1].second) / (accumulated_s->at(i) - accumulated_s->at(i - 1));
    } else {
      xds = (xy_points[i + 1].first - xy_points[i - 1].first) / (accumulated_s->at(i + 1) - accumulated_s->at(i - 1));
      yds = (xy_points[i + 1].second - xy_points[i - 1].second) / (accumulated_s->at(i + 1) - accumulated_s->at(i - 1));
    }
    x_over_s_first_derivatives.push_back(xds);
    y_over_s_first_derivatives.push_back(yds);
  }
  for (std::size_t i = 0; i < points_size; ++i) {
    double xdds = 0.0, ydds = 0.0;
    if (i == 0) {
      xdds = (x_over_s_first_derivatives[i + 1] - x_over_s_first_derivatives[i]) / (accumulated_s->at(i + 1) - accumulated_s->at(i));
      ydds = (y_over_s_first_derivatives[i + 1] - y_over_s_first_derivatives[i]) / (accumulated_s->at(i + 1) - accumulated_s->at(i));
    } else if (i == points_size - 1) {
      xdds = (x_over_s_first_derivatives[i] - x_over_s_first_derivatives[i - 1]) / (accumulated_s->at(i) - accumulated_s->at(i - 1));
      ydds = (y_over_s_first_derivatives[i] - y_over_s_first_derivatives[i - 1]) / (accumulated_s->at(i) - accumulated_s->at(i - 1));
    } else {
      xdds = (x_over_s_first_derivatives[i + 1] - x_over_s_first_derivatives[i - 1]) / (accumulated_s->at(i + 1) - accumulated_s->at(i - 1));
      ydds = (y_over_s_first_derivatives[i + 1] - y_over_s_first_derivatives[i - 1]) / (accumulated_s->at(i + 1) - accumulated_s->at(i - 1));
    }
    x_over_s_second_derivatives.push_back(xdds);
    y_over_s_second_derivatives.push_back(ydds);
  }
  for (std::size_t i = 0; i < points_size; ++i) {
    double xds = x_over_s_first_derivatives[i], yds = y_over_s_first_derivatives[i];
    double xdds = x_over_s_second_derivatives[i], ydds = y_over_s_second_derivatives[i];
    double kappa = (xds * ydds - yds * xdds) / (std::sqrt(xds * xds + yds * yds) * (xds * xds + yds * yds) + 1e-6);
    kappas->push_back(kappa);
  }
  return true;
}

constexpr int NT = 5, NS = 7, NL = 10;   // dp_planner.h:27-29
constexpr double kMathEpsilon = 1e-3;    // dp_planner.cpp:25 (shadows math::kMathEpsilon inside the planner)

class DpPlanner {
 public:
  DpPlanner(const PlannerConfig& config, const Env& env) : env_(env), config_(config), unit_time_(config.tf / NT) {   // cpp:27-34
    time_ = math::LinSpaced<NT>(unit_time_, config.tf);
    station_ = math::LinSpaced<NS>(0, unit_time_ * config_.vehicle.max_velocity);
    lateral_ = math::LinSpaced<NL - 1>(0, 1);
    safe_margin_ = config_.vehicle.width / 2 * 1.5;
  }

  bool Plan(double start_x, double start_y, double start_theta, Trajectory& data) {   // cpp:135-281
    auto sl = env_->reference().GetProjection({start_x, start_y});
    state_.start_s = sl.x();
    state_.start_l = sl.y();
    state_.start_theta = start_theta;
    for (int i = 0; i < NT; i++)
      for (int j = 0; j < NS; j++)
        for (int k = 0; k < NL; k++) state_space_[i][j][k] = StateCell();
    for (int i = 0; i < NS; i++)
      for (int j = 0; j < NL; j++) {
        auto tup = GetCost(StateIndex(-1, -1, -1), StateIndex(0, i, j));
        state_space_[0][i][j].current_s = tup.first;
        state_space_[0][i][j].cost = tup.second;
      }
    for (int i = 0; i < NT - 1; i++)
      for (int j = 0; j < NS; j++)
        for (int k = 0; k < NL; k++) {
          StateIndex parent_ind(i, j, k);
          for (int m = 0; m < NS; m++)
            for (int n = 0; n < NL; n++) {
              StateIndex current_ind(i + 1, m, n);
              auto tup = GetCost(parent_ind, current_ind);
              double delta_cost = tup.second;
              double cur_s = tup.first;
              double cur_cost = state_space_[i][j][k].cost + delta_cost;
              if (cur_cost < state_space_[i + 1][m][n].cost) state_space_[i + 1][m][n] = StateCell(cur_cost, cur_s, j, k);
            }
        }
    double min_cost = std::numeric_limits<double>::max();
    int min_s_ind = 0, min_l_ind = 0;
    for (int i = 0; i < NS; i++)
      for (int j = 0; j < NL; j++) {
        double cost = state_space_[NT - 1][i][j].cost;
        if (cost < min_cost) {
          min_s_ind = i;
          min_l_ind = j;
          min_cost = cost;
        }
      }
    std::vector<std::pair<StateIndex, StateCell>> waypoints(NT);
    for (int i = NT - 1; i >= 0; i--) {
      auto& cell = state_space_[i][min_s_ind][min_l_ind];
      waypoints[i] = std::make_pair(StateIndex(i, min_s_ind, min_l_ind), cell);
      min_s_ind = cell.parent_s_ind;
      min_l_ind = cell.parent_l_ind;
    }
    data.clear();
    data.resize(config_.tf / config_.delta_t + 1);
    double last_l = state_.start_l, last_s = state_.start_s;
    std::vector<std::pair<double, double>> xy_points;
    size_t n = 0;
    for (int i = 0; i < NT; i++) {
      double parent_s = i > 0 ? waypoints[i - 1].second.current_s : state_.start_s;
      auto segment = InterpolateLinearly(parent_s, waypoints[i].second.parent_l_ind, i, waypoints[i].first.s, waypoints[i].first.l);
      for (size_t j = 0; j < segment.size(); j++) {
        auto dl = segment[j].y() - last_l;
        auto ds = std::max(segment[j].x() - last_s, kMathEpsilon);
        last_l = segment[j].y();
        last_s = segment[j].x();
        auto xy = env_->reference().GetCartesian(segment[j].x(), segment[j].y());
        auto tp = env_->reference().EvaluateStation(segment[j].x());
        if (n < data.size()) {   // the reference writes data[n] unchecked; the sizes agree for its configurations
          data[n].time = config_.delta_t * n;
          data[n].s = segment[j].x();
          data[n].x = xy.x();
          data[n].y = xy.y();
          data[n].theta = tp.theta + atan((dl / ds) / (1 - tp.kappa * segment[j].y()));
        }
        xy_points.emplace_back(xy.x(), xy.y());
        ++n;
      }
    }
    std::vector<double> headings, accumulated_s, speeds, accelerations, kappas;
    ComputePathProfile(config_.delta_t, xy_points, &headings, &accumulated_s, &speeds, &accelerations, &kappas);
    for (size_t i = 0; i < xy_points.size() && i < data.size(); ++i) {
      data[i].kappa = kappas[i];
      data[i].delta = std::atan(data[i].kappa * config_.vehicle.wheel_base);
      data[i].velocity = speeds[i];
      data[i].a = accelerations[i];
      data[i].jerk = 0.0;
      data[i].delta_rate = 0.0;
    }
    return min_cost < config_.dp_w_obstacle;
  }

 private:
  struct StateCell {   // dp_planner.h:44-55
    double cost = std::numeric_limits<double>::max();
    double current_s = std::numeric_limits<double>::min();
    int parent_s_ind = -1;
    int parent_l_ind = -1;
    StateCell() = default;
    StateCell(double cost, double cur_s, int parent_s_ind, int parent_l_ind)
        : cost(cost), current_s(cur_s), parent_s_ind(parent_s_ind), parent_l_ind(parent_l_ind) {}
  };
  struct StateIndex {
    int t = -1, s = -1, l = -1;
    StateIndex() = default;
    StateIndex(int tt, int ss, int ll) : t(tt), s(ss), l(ll) {}
  };
  struct StartState {
    double start_s = 0, start_l = 0, start_theta = 0;
  };

  double GetLateralOffset(double s, int l_ind) {   // dp_planner.h:84-92
    if (l_ind == NL - 1) return 0.0;
    auto ref = env_->reference().EvaluateStation(s);
    double lb = -ref.right_bound + safe_margin_;
    double ub = ref.left_bound - safe_margin_;
    return lb + (ub - lb) * lateral_[l_ind];
  }

  std::vector<Vec2d> InterpolateLinearly(double parent_s, int parent_l_ind, int cur_t_ind, int cur_s_ind, int cur_l_ind) {   // cpp:283
    int nseg = 0;
    for (double t = 0.0; t < config_.tf + config_.delta_t - math::kMathEpsilon; t += config_.delta_t) {
      if (cur_t_ind == 0) {
        if (t > 0.0 - kMathEpsilon && t < unit_time_ + kMathEpsilon) ++nseg;
      } else {
        if (t > time_[cur_t_ind] - unit_time_ + math::kMathEpsilon && t < time_[cur_t_ind] + math::kMathEpsilon) ++nseg;
      }
    }
    std::vector<Vec2d> result(nseg);
    double p_l = state_.start_l;
    double p_s = state_.start_s;
    if (parent_l_ind >= 0) {
      p_s = parent_s;
      p_l = GetLateralOffset(p_s, parent_l_ind);
    }
    double cur_s = p_s + station_[cur_s_ind];
    double cur_l = GetLateralOffset(cur_s, cur_l_ind);
    double s_step = station_[cur_s_ind] / nseg;
    double l_step = (cur_l - p_l) / nseg;
    for (int i = 0; i < nseg; i++) result[i] = Vec2d(p_s + i * s_step, p_l + i * l_step);
    return result;
  }

  double GetCollisionCost(StateIndex parent_ind, StateIndex cur_ind) {   // cpp:44-86
    double parent_s = state_.start_s, grandparent_s = state_.start_s;
    double last_l = state_.start_l, last_s = state_.start_s;
    if (parent_ind.t >= 0) {
      auto& cell = state_space_[parent_ind.t][parent_ind.s][parent_ind.l];
      parent_s = cell.current_s;
      if (parent_ind.t > 0) {
        auto& parent_cell = state_space_[parent_ind.t - 1][cell.parent_s_ind][cell.parent_l_ind];
        grandparent_s = parent_cell.current_s;
      }
      auto prev_path = InterpolateLinearly(grandparent_s, cell.parent_l_ind, parent_ind.t, parent_ind.s, parent_ind.l);
      last_l = prev_path.back().y();
      last_s = prev_path.back().x();
    }
    auto path = InterpolateLinearly(parent_s, parent_ind.l, cur_ind.t, cur_ind.s, cur_ind.l);
    int nseg = path.size();
    for (size_t i = 0; i < path.size(); i++) {
      auto& pt = path[i];
      double dl = pt.y() - last_l;
      double ds = std::max(pt.x() - last_s, kMathEpsilon);
      last_l = pt.y();
      last_s = pt.x();
      auto cart = env_->reference().GetCartesian(pt.x(), pt.y());
      auto ref = env_->reference().EvaluateStation(pt.x());
      double lb = std::min(0.0, -ref.right_bound + safe_margin_);
      double ub = std::max(0.0, ref.left_bound - safe_margin_);
      if (pt.y() < lb - kMathEpsilon || pt.y() > ub + kMathEpsilon) return config_.dp_w_obstacle;
      double heading = ref.theta + atan((dl / ds) / (1 - ref.kappa * pt.y()));
      double parent_time = parent_ind.t < 0 ? 0.0 : time_[parent_ind.t];
      double time = parent_time + i * (unit_time_ / nseg);
      if (env_->CheckOptimizationCollision(time, cart.x(), cart.y(), heading)) return config_.dp_w_obstacle;
    }
    return 0.0;
  }

  std::pair<double, double> GetCost(StateIndex parent_ind, StateIndex cur_ind) {   // cpp:88-133
    double parent_s = state_.start_s, grandparent_s = state_.start_s;
    double parent_l = state_.start_l, grandparent_l = state_.start_l;
    if (parent_ind.t >= 0) {
      auto& cell = state_space_[parent_ind.t][parent_ind.s][parent_ind.l];
      int grandparent_s_ind = cell.parent_s_ind;
      int grandparent_l_ind = cell.parent_l_ind;
      parent_s = cell.current_s;
      parent_l = GetLateralOffset(parent_s, parent_ind.l);
      if (parent_ind.t >= 1) {
        grandparent_s = state_space_[parent_ind.t - 1][grandparent_s_ind][grandparent_l_ind].current_s;
        grandparent_l = GetLateralOffset(grandparent_s, grandparent_l_ind);
      }
    }
    double cur_s = parent_s + station_[cur_ind.s];
    double cur_l = GetLateralOffset(cur_s, cur_ind.l);
    double ds1 = cur_s - parent_s;
    double dl1 = cur_l - parent_l;
    double ds0 = parent_s - grandparent_s;
    double dl0 = parent_l - grandparent_l;
    double cost_obstacle = GetCollisionCost(parent_ind, cur_ind);
    if (cost_obstacle >= config_.dp_w_obstacle) return std::make_pair(cur_s, config_.dp_w_obstacle);
    double cost_lateral = fabs(cur_l);
    double cost_lateral_change = fabs(parent_l - cur_l) / (station_[cur_ind.s] + kMathEpsilon);
    double cost_lateral_change_t = fabs(dl1 - dl0) / unit_time_;
    double cost_longitudinal_velocity = fabs(ds1 / unit_time_ - config_.dp_nominal_velocity);
    double cost_longitudinal_velocity_change = fabs((ds1 - ds0) / unit_time_);
    double delta_cost = (config_.dp_w_lateral * cost_lateral + config_.dp_w_lateral_change * cost_lateral_change +
                         config_.dp_w_lateral_velocity_change * cost_lateral_change_t +
                         config_.dp_w_longitudinal_velocity_bias * cost_longitudinal_velocity +
                         config_.dp_w_longitudinal_velocity_change * cost_longitudinal_velocity_change);
    return std::make_pair(cur_s, delta_cost);
  }

  Env env_;
  PlannerConfig config_;
  double unit_time_;
  std::array<double, NT> time_;
  std::array<double, NS> station_;
  std::array<double, NL - 1> lateral_;
  StartState state_;
  StateCell state_space_[NT][NS][NL];
  double safe_margin_;
};

}  // namespace

extern "C" {

/* cfg: tf, delta_t, dp_nominal_velocity, dp_w_obstacle, dp_w_lateral, dp_w_lateral_change,
 *      dp_w_lateral_velocity_change, dp_w_longitudinal_velocity_bias, dp_w_longitudinal_velocity_change,
 *      front_hang_length, wheel_base, rear_hang_length, width, max_velocity                      (14 doubles)
 * center [n_center][7] s x y theta kappa left_bound right_bound; static polygons flattened (world frame);
 * dynamic obstacles: body-frame polygons flattened + trajectories [.][4] time x y theta flattened;
 * start3 = x, y, theta; coarse out [K][9] = time s x y theta kappa velocity a delta.
 * Returns 1 (path found), 0 (DP failed: every path collides; coarse still filled), negative on bad input. */
int oracle_dp_plan(const double* cfg, const double* center, int n_center, const double* static_pts, const int* static_counts,
                   int n_static, const double* dyn_poly_pts, const int* dyn_poly_counts, const double* dyn_traj,
                   const int* dyn_traj_counts, int n_dyn, const double* start3, double* coarse, int K) {
  if (cfg == nullptr || center == nullptr || start3 == nullptr || coarse == nullptr || n_center < 2) return -1;
  PlannerConfig pc;
  pc.tf = cfg[0]; pc.delta_t = cfg[1]; pc.dp_nominal_velocity = cfg[2]; pc.dp_w_obstacle = cfg[3]; pc.dp_w_lateral = cfg[4];
  pc.dp_w_lateral_change = cfg[5]; pc.dp_w_lateral_velocity_change = cfg[6]; pc.dp_w_longitudinal_velocity_bias = cfg[7];
  pc.dp_w_longitudinal_velocity_change = cfg[8];
  pc.vehicle.front_hang_length = cfg[9]; pc.vehicle.wheel_base = cfg[10]; pc.vehicle.rear_hang_length = cfg[11];
  pc.vehicle.width = cfg[12]; pc.vehicle.max_velocity = cfg[13];
  pc.vehicle.Finish();
  Env env = std::make_shared<Environment>(pc);
  Trajectory data;
  for (int i = 0; i < n_center; ++i) {   // PlanningNode::CenterLineCallback, planning_node.cc:33-49
    const double* c = center + (size_t)i * 7;
    TrajectoryPoint tp;
    tp.s = c[0]; tp.x = c[1]; tp.y = c[2]; tp.theta = c[3]; tp.kappa = c[4]; tp.left_bound = c[5]; tp.right_bound = c[6];
    data.push_back(tp);
  }
  env->set_reference(DiscretizedTrajectory(data));
  size_t at = 0;
  for (int o = 0; o < n_static; ++o) {   // ObstaclesCallback, planning_node.cc:51-61
    std::vector<Vec2d> points;
    for (int k = 0; k < static_counts[o]; ++k, ++at) points.emplace_back(static_pts[at * 2], static_pts[at * 2 + 1]);
    env->obstacles().emplace_back(points);
  }
  size_t pa = 0, ta = 0;
  for (int o = 0; o < n_dyn; ++o) {   // DynamicObstaclesCallback, planning_node.cc:63-80
    Environment::DynamicObstacle dynamic_obstacle;
    for (int t = 0; t < dyn_traj_counts[o]; ++t) {
      const double* tp = dyn_traj + (ta + t) * 4;
      std::vector<Vec2d> points;
      for (int k = 0; k < dyn_poly_counts[o]; ++k) {   // Pose::transform
        const double x = dyn_poly_pts[(pa + k) * 2], y = dyn_poly_pts[(pa + k) * 2 + 1];
        points.emplace_back(tp[1] + x * cos(tp[3]) - y * sin(tp[3]), tp[2] + x * sin(tp[3]) + y * cos(tp[3]));   // pose.h:40-46
      }
      dynamic_obstacle.emplace_back(tp[0], math::Polygon2d(points));
    }
    pa += dyn_poly_counts[o];
    ta += dyn_traj_counts[o];
    if (!dynamic_obstacle.empty()) env->dynamic_obstacles().push_back(dynamic_obstacle);
  }
  DpPlanner dp(pc, env);
  Trajectory out;
  const bool ok = dp.Plan(start3[0], start3[1], start3[2], out);
  if ((int)out.size() != K) return -2;
  for (int i = 0; i < K; ++i) {
    double* r = coarse + (size_t)i * 9;
    const TrajectoryPoint& p = out[i];
    r[0] = p.time; r[1] = p.s; r[2] = p.x; r[3] = p.y; r[4] = p.theta; r[5] = p.kappa; r[6] = p.velocity; r[7] = p.a; r[8] = p.delta;
  }
  return ok ? 1 : 0;
}


/* ---- test hooks: the restated geometry / trajectory pieces one by one, with the signatures of oracle/ref_shim.cc, so
 * that tests/test_reference_pins.py can hold each against the reference's own code (oracle/_ref) bit for bit ---- */
namespace {
DiscretizedTrajectory hook_trajectory(const double* rows, int n) {
  std::vector<TrajectoryPoint> pts(n);
  for (int i = 0; i < n; ++i) {
    const double* r = rows + 9 * i;
    pts[i].time = r[0]; pts[i].s = r[1]; pts[i].x = r[2]; pts[i].y = r[3]; pts[i].theta = r[4];
    pts[i].kappa = r[5]; pts[i].velocity = r[6]; pts[i].left_bound = r[7]; pts[i].right_bound = r[8];
  }
  return DiscretizedTrajectory(pts);
}
void hook_put(const TrajectoryPoint& p, double* o) {
  o[0] = p.time; o[1] = p.s; o[2] = p.x; o[3] = p.y; o[4] = p.theta; o[5] = p.kappa; o[6] = p.velocity;
  o[7] = p.left_bound; o[8] = p.right_bound;
}
}  // namespace

int oracle_compute_path_profile(double dt, const double* xy, int n, double* headings, double* s, double* v, double* a,
                                double* kappa) {
  std::vector<std::pair<double, double>> pts(n);
  for (int i = 0; i < n; ++i) pts[i] = {xy[2 * i], xy[2 * i + 1]};
  std::vector<double> h, ss, vv, aa, kk;
  if (!ComputePathProfile(dt, pts, &h, &ss, &vv, &aa, &kk)) return 0;
  for (int i = 0; i < n; ++i) {
    headings[i] = h[i]; s[i] = ss[i]; v[i] = vv[i]; a[i] = aa[i]; kappa[i] = kk[i];
  }
  return 1;
}
int oracle_polygon_overlaps_aabox(const double* poly, int n, double x0, double y0, double x1, double y1) {
  using namespace math;
  std::vector<Vec2d> pts;
  for (int i = 0; i < n; ++i) pts.emplace_back(poly[2 * i], poly[2 * i + 1]);
  const Polygon2d polygon(pts);
  const Box2d box = Box2d::FromAABox(Vec2d(x0, y0), Vec2d(x1, y1), Vec2d(0.0, 0.0));
  return polygon.HasOverlap(box) ? 1 : 0;
}
int oracle_polygon_point_in(const double* poly, int n, double px, double py) {
  using namespace math;
  std::vector<Vec2d> pts;
  for (int i = 0; i < n; ++i) pts.emplace_back(poly[2 * i], poly[2 * i + 1]);
  return Polygon2d(pts).IsPointIn(Vec2d(px, py)) ? 1 : 0;
}
void oracle_trajectory_evaluate_station(const double* rows, int n, double station, double* out9) {
  hook_put(hook_trajectory(rows, n).EvaluateStation(station), out9);
}
void oracle_trajectory_projection(const double* rows, int n, double px, double py, double* out2) {
  const auto sl = hook_trajectory(rows, n).GetProjection(math::Vec2d(px, py));
  out2[0] = sl.x();
  out2[1] = sl.y();
}
void oracle_trajectory_cartesian(const double* rows, int n, double station, double lateral, double* out2) {
  const auto p = hook_trajectory(rows, n).GetCartesian(station, lateral);
  out2[0] = p.x();
  out2[1] = p.y();
}

// the header-only helpers of the reference the planner leans on, one by one (signatures of oracle/ref_shim.cc)
double oracle_slerp(double a0, double t0, double a1, double t1, double t) { return math::slerp(a0, t0, a1, t1, t); }
int oracle_lin_spaced(int n, double start, double end, double* out) {   // the three sizes of DpPlanner's constructor
  if (n == 5) { const auto r = math::LinSpaced<5>(start, end); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  else if (n == 7) { const auto r = math::LinSpaced<7>(start, end); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  else if (n == 9) { const auto r = math::LinSpaced<9>(start, end); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  else return -1;
  return n;
}
// the placement of a dynamic obstacle's body-frame vertex as oracle_dp_plan does it above (Pose::transform, pose.h:40-46)
void oracle_pose_transform(double x, double y, double theta, double rx, double ry, double rtheta, double* out3) {
  out3[0] = x + rx * cos(theta) - ry * sin(theta);
  out3[1] = y + rx * sin(theta) + ry * cos(theta);
  out3[2] = theta + rtheta;
}
// the two boxes of Environment::CheckOptimizationCollision as restated above (environment.cpp:92-104); layout of
// ref_collision_boxes: xr yr xf yf | per box (f, r): centre, half extents, min/max x, min/max y, four corners
void oracle_collision_boxes(double x, double y, double theta, double collision_buffer, double* out36) {
  using math::Vec2d;
  VehicleParam vehicle;
  vehicle.Finish();
  const Vec2d c0(-vehicle.radius - collision_buffer, -vehicle.radius - collision_buffer);
  const Vec2d c1(vehicle.radius + collision_buffer, vehicle.radius + collision_buffer);
  double xr, yr, xf, yf;
  std::tie(xr, yr, xf, yf) = vehicle.GetDiscPositions(x, y, theta);
  out36[0] = xr; out36[1] = yr; out36[2] = xf; out36[3] = yf;
  const math::Box2d boxes[2] = {math::Box2d::FromAABox(c0, c1, Vec2d(xf, yf)), math::Box2d::FromAABox(c0, c1, Vec2d(xr, yr))};
  for (int k = 0; k < 2; ++k) {
    double* o = out36 + 4 + 16 * k;
    const math::Box2d& b = boxes[k];
    o[0] = b.center_.x(); o[1] = b.center_.y(); o[2] = b.half_length_; o[3] = b.half_width_;
    o[4] = b.min_x(); o[5] = b.max_x(); o[6] = b.min_y(); o[7] = b.max_y();
    for (int i = 0; i < 4; ++i) { o[8 + 2 * i] = b.corners()[i].x(); o[9 + 2 * i] = b.corners()[i].y(); }
  }
}
// VehicleParam's derived members (vehicle_param.h:83-88) from the defaults: radius, f2x, r2x
void oracle_vehicle_derived(double* out3) {
  VehicleParam vehicle;
  vehicle.Finish();
  out3[0] = vehicle.radius; out3[1] = vehicle.f2x; out3[2] = vehicle.r2x;
}

}  // extern "C"
