// Drives include/cilqr/trajectory_planner.hpp the way the reference's PlanningNode drives
// planning::TrajectoryPlanner (algorithm/planning_node.cc:11,86; trajectory_planner.cpp:22-162): DP coarse planner ->
// safe corridor -> CILQR, on a scene read from a .cqs file (include/cilqr/scene_file.hpp), with minimal stand-ins for
// the reference's own types (this test's definitions; the reference headers are not copied).
//
//   planner_test <scenes.cqs> <scene index> <tf> <out.bin>
// out.bin: int32 plan_ok, K, n_cost, n_iter_trajs | coarse[K][9] (time s x y theta kappa velocity a delta) |
//          result[K][11] (time s x y theta kappa velocity a jerk delta delta_rate) | cost[n_cost][5] |
//          int32 corridor plane counts[K] | int32 n_left, n_right
#include <array>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

#include "cilqr/corridor.hpp"
#include "cilqr/ilqr_optimizer.hpp"
#include "cilqr/scene_file.hpp"
#include "cilqr/trajectory_planner.hpp"

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
  const std::vector<Vec2d>& points() const { return pts; }
};
}  // namespace math
struct Vector3d {
  double v[3];
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double operator[](int i) const { return v[i]; }
};
struct Vector2d {
  double v[2];
  Vector2d(double a, double b) : v{a, b} {}
};
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
struct Cost {
  double c[5] = {0, 0, 0, 0, 0};
  Cost() = default;
  Cost(double c0, double c1, double c2, double c3, double c4) : c{c0, c1, c2, c3, c4} {}
};

// what PlanningNode's callbacks build from the messages (planning_node.cc:33-80) and what Environment answers
// (environment.cpp:20-43, 133-182), from a scene file
class Environment {
 public:
  using DynamicObstacle = std::vector<std::pair<double, math::Polygon2d>>;
  Environment(const cilqr::SceneFile& file, const cilqr::Scene& scene) {
    std::vector<TrajectoryPoint> data;
    for (const auto& c : file.center) {
      TrajectoryPoint tp;
      tp.s = c[0]; tp.x = c[1]; tp.y = c[2]; tp.theta = c[3]; tp.kappa = c[4]; tp.left_bound = c[5]; tp.right_bound = c[6];
      data.push_back(tp);
    }
    reference_ = DiscretizedTrajectory(data);
    const cilqr::ReferenceLine ref(file.center);
    const double start_s = file.center.front()[0], back_s = file.center.back()[0];
    const int sample_points = int((back_s - start_s) / 0.1);
    for (int i = 0; i <= sample_points; i++) {
      const double s = start_s + i * 0.1;
      const cilqr::RefPoint r = ref.EvaluateStation(s);
      const cilqr::DpPoint2 l = ref.GetCartesian(s, r.left_bound), q = ref.GetCartesian(s, -r.right_bound);
      left_.push_back(math::Vec2d(l.x, l.y));
      right_.push_back(math::Vec2d(q.x, q.y));
    }
    for (const auto& p : scene.statics) {
      math::Polygon2d poly;
      for (const auto& v : p) poly.pts.push_back(math::Vec2d(v.x, v.y));
      obstacles_.push_back(poly);
    }
    for (const auto& d : scene.dynamics) {
      DynamicObstacle ob;
      for (const auto& tp : d.trajectory) {
        math::Polygon2d poly;
        const double c = std::cos(tp[3]), s = std::sin(tp[3]);
        for (const auto& v : d.polygon) poly.pts.push_back(math::Vec2d(tp[1] + v.x * c - v.y * s, tp[2] + v.x * s + v.y * c));
        ob.emplace_back(tp[0], poly);
      }
      if (!ob.empty()) dynamic_obstacles_.push_back(ob);
    }
  }
  const DiscretizedTrajectory& reference() const { return reference_; }
  std::vector<math::Polygon2d>& obstacles() { return obstacles_; }
  std::vector<DynamicObstacle>& dynamic_obstacles() { return dynamic_obstacles_; }
  const std::vector<math::Vec2d>& left_road_barrier() { return left_; }
  const std::vector<math::Vec2d>& right_road_barrier() { return right_; }
  bool QueryStaticObstaclesPoints(std::vector<math::Vec2d>* const points, bool) {
    for (const auto& o : obstacles_) points->insert(points->end(), o.pts.begin(), o.pts.end());
    return true;
  }
  bool QueryDynamicObstaclesPoints(const double time, std::vector<math::Vec2d>* const points, bool) {
    constexpr double kEps = 1e-10;
    for (auto& ob : dynamic_obstacles_) {
      if (ob.front().first > time + kEps || ob.back().first < time - kEps) continue;
      size_t i = 0;
      while (i + 1 < ob.size() && !(time < ob[i].first + kEps)) ++i;
      points->insert(points->end(), ob[i].second.pts.begin(), ob[i].second.pts.end());
    }
    return true;
  }

 private:
  DiscretizedTrajectory reference_;
  std::vector<math::Polygon2d> obstacles_;
  std::vector<DynamicObstacle> dynamic_obstacles_;
  std::vector<math::Vec2d> left_, right_;
};
using Env = std::shared_ptr<Environment>;

using Corridor = cilqr::CorridorT<CorridorConfig, Env, DiscretizedTrajectory, math::Vec2d, math::LineSegment2d, Vector3d, Vector2d>;
using IlqrOptimizer = cilqr::IlqrOptimizerT<TrajectoryPoint, DiscretizedTrajectory, Corridor::CorridorConstraints,
                                            Corridor::LaneConstraints, IlqrConfig, VehicleParam, Cost>;
using DpPlanner = cilqr::DpPlannerT<PlannerConfig, Env, DiscretizedTrajectory, TrajectoryPoint>;
using TrajectoryPlanner = cilqr::TrajectoryPlannerT<PlannerConfig, Env, StartState, TrajectoryPoint, DiscretizedTrajectory,
                                                    DpPlanner, Corridor, IlqrOptimizer, math::Vec2d, math::LineSegment2d>;

}  // namespace planning

int main(int argc, char** argv) {
  using namespace planning;
  if (argc < 5) return 2;
  cilqr::SceneFile file;
  const std::string err = cilqr::LoadSceneFile(argv[1], &file);
  if (!err.empty()) {
    std::fprintf(stderr, "%s\n", err.c_str());
    return 3;
  }
  const size_t idx = (size_t)std::atoi(argv[2]);
  if (idx >= file.scenes.size()) return 4;
  const cilqr::Scene& scene = file.scenes[idx];
  PlannerConfig config;
  config.tf = std::atof(argv[3]);
  Env env = std::make_shared<Environment>(file, scene);
  TrajectoryPlanner planner(config, env);                    // planning_node.cc:11
  StartState state{scene.start[0], scene.start[1], scene.start[2], scene.start[3], 0.0, 0.0, 0.0};
  DiscretizedTrajectory result;
  const bool ok = planner.Plan(state, result);               // planning_node.cc:86
  const int K = (int)(config.tf / config.delta_t + 1);
  FILE* o = std::fopen(argv[4], "wb");
  if (!o) return 5;
  const auto costs = planner.ilqr_optimizer().cost();
  int32_t hdr[4] = {ok ? 1 : 0, K, (int32_t)costs.size(), (int32_t)planner.iteration_trajectories().size()};
  std::fwrite(hdr, sizeof(int32_t), 4, o);
  std::vector<double> coarse((size_t)K * 9, 0.0), res((size_t)K * 11, 0.0);
  const auto& ct = planner.coarse_trajectory().trajectory();
  for (size_t i = 0; i < ct.size() && i < (size_t)K; ++i) {
    const double row[9] = {ct[i].time, ct[i].s, ct[i].x, ct[i].y, ct[i].theta, ct[i].kappa, ct[i].velocity, ct[i].a, ct[i].delta};
    for (int e = 0; e < 9; ++e) coarse[i * 9 + e] = row[e];
  }
  const auto& rt = result.trajectory();
  for (size_t i = 0; i < rt.size() && i < (size_t)K; ++i) {
    const double row[11] = {rt[i].time, rt[i].s, rt[i].x, rt[i].y, rt[i].theta, rt[i].kappa, rt[i].velocity, rt[i].a,
                            rt[i].jerk, rt[i].delta, rt[i].delta_rate};
    for (int e = 0; e < 11; ++e) res[i * 11 + e] = row[e];
  }
  std::fwrite(coarse.data(), sizeof(double), coarse.size(), o);
  std::fwrite(res.data(), sizeof(double), res.size(), o);
  for (const auto& c : costs) std::fwrite(c.c, sizeof(double), 5, o);
  const auto polys = planner.SafeCorridors();
  for (int i = 0; i < K; ++i) {
    const int32_t n = i < (int)polys.size() ? (int32_t)polys[i].size() : -1;
    std::fwrite(&n, sizeof(int32_t), 1, o);
  }
  int32_t nl[2] = {(int32_t)planner.left_lane_boundary().size(), (int32_t)planner.right_lane_boundary().size()};
  std::fwrite(nl, sizeof(int32_t), 2, o);
  std::fclose(o);
  return 0;
}
