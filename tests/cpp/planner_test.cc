// Drives include/cilqr/trajectory_planner.hpp the way the reference's PlanningNode drives
// planning::TrajectoryPlanner (algorithm/planning_node.cc:11,86; trajectory_planner.cpp:22-162): DP coarse planner ->
// safe corridor -> CILQR, on a scene read from a .cqs file (include/cilqr/scene_file.hpp).  The reference's types come
// from tests/cpp/reference_types.hpp: its OWN headers with -DCILQR_TEST_REFERENCE_HEADERS (build container), minimal
// stand-ins otherwise (GPU box); Environment (environment.cpp needs ROS) is this test's own in both.
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
#include "reference_types.hpp"

namespace planning {

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
      std::vector<math::Vec2d> pts;
      for (const auto& v : p) pts.push_back(math::Vec2d(v.x, v.y));
      obstacles_.push_back(math::Polygon2d(pts));
    }
    for (const auto& d : scene.dynamics) {
      DynamicObstacle ob;
      for (const auto& tp : d.trajectory) {
        std::vector<math::Vec2d> pts;
        const double c = std::cos(tp[3]), s = std::sin(tp[3]);
        for (const auto& v : d.polygon) pts.push_back(math::Vec2d(tp[1] + v.x * c - v.y * s, tp[2] + v.x * s + v.y * c));
        ob.emplace_back(tp[0], math::Polygon2d(pts));
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
    for (const auto& o : obstacles_) points->insert(points->end(), o.points().begin(), o.points().end());
    return true;
  }
  bool QueryDynamicObstaclesPoints(const double time, std::vector<math::Vec2d>* const points, bool) {
    constexpr double kEps = 1e-10;
    for (auto& ob : dynamic_obstacles_) {
      if (ob.front().first > time + kEps || ob.back().first < time - kEps) continue;
      size_t i = 0;
      while (i + 1 < ob.size() && !(time < ob[i].first + kEps)) ++i;
      points->insert(points->end(), ob[i].second.points().begin(), ob[i].second.points().end());
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
  for (const auto& c : costs) {
    const double row[5] = {c.total_cost, c.target_cost, c.dynamic_cost, c.corridor_cost, c.lane_boundary_cost};
    std::fwrite(row, sizeof(double), 5, o);
  }
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
