// Drives include/cilqr/corridor.hpp the way the reference's TrajectoryPlanner drives
// planning::Corridor (algorithm/ilqr/corridor.h:27-44).  The reference's types come from tests/cpp/reference_types.hpp:
// its OWN headers with -DCILQR_TEST_REFERENCE_HEADERS (build container), minimal stand-ins otherwise (GPU box);
// Environment (environment.cpp needs ROS) is this test's own in both.
//
//   corridor_adapter_test <scene.bin> <out.bin>
// scene.bin: int32 K, P, nlb, nrb | knots[K][4] (time, x, y, theta) | counts[K] (int32) |
//            points[K][P][2] | left_barrier[nlb][2] | right_barrier[nrb][2]
//            (every obstacle point is served as a "dynamic" point of its knot's time)
// out.bin:   int32 plan_ok, n_left, n_right, errors_ok | counts[K] (int32) | per knot: planes[count][3],
//            polygon[count][2] | left[n_left][7] | right[n_right][7] | int32 stored point counts[K]
#include <array>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

#include "cilqr/corridor.hpp"
#include "reference_types.hpp"

namespace planning {

using math::LineSegment2d;
using math::Vec2d;
class Environment {
 public:
  std::vector<double> times;
  std::vector<std::vector<Vec2d>> per_time;
  std::vector<Vec2d> left, right;
  bool QueryStaticObstaclesPoints(std::vector<Vec2d>* const, bool) { return true; }
  bool QueryDynamicObstaclesPoints(double time, std::vector<Vec2d>* const pts, bool) {
    for (size_t i = 0; i < times.size(); ++i)
      if (times[i] == time) pts->insert(pts->end(), per_time[i].begin(), per_time[i].end());
    return true;
  }
  const std::vector<Vec2d>& left_road_barrier() const { return left; }
  const std::vector<Vec2d>& right_road_barrier() const { return right; }
};
using Env = std::shared_ptr<Environment>;
using Corridor = cilqr::CorridorT<CorridorConfig, Env, DiscretizedTrajectory, Vec2d, LineSegment2d, Vector3d, Vector2d>;

}  // namespace planning

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return n == 0 || std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  using namespace planning;
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[4];
  if (!rd(f, hdr, 4)) return 4;
  const int K = hdr[0], P = hdr[1], nlb = hdr[2], nrb = hdr[3];
  std::vector<double> knots((size_t)K * 4), points((size_t)K * P * 2), lb((size_t)nlb * 2), rb((size_t)nrb * 2);
  std::vector<int32_t> counts(K);
  if (!rd(f, knots.data(), knots.size()) || !rd(f, counts.data(), counts.size()) || !rd(f, points.data(), points.size()) ||
      !rd(f, lb.data(), lb.size()) || !rd(f, rb.data(), rb.size()))
    return 5;
  std::fclose(f);
  Env env = std::make_shared<Environment>();
  std::vector<TrajectoryPoint> pts(K);
  for (int i = 0; i < K; ++i) {
    pts[i].time = knots[4 * i]; pts[i].x = knots[4 * i + 1]; pts[i].y = knots[4 * i + 2]; pts[i].theta = knots[4 * i + 3];
    env->times.push_back(pts[i].time);
    std::vector<Vec2d> v;
    for (int k = 0; k < counts[i]; ++k)
      v.push_back(Vec2d(points[((size_t)i * P + k) * 2], points[((size_t)i * P + k) * 2 + 1]));
    env->per_time.push_back(v);
  }
  for (int i = 0; i < nlb; ++i) env->left.push_back(Vec2d(lb[2 * i], lb[2 * i + 1]));
  for (int i = 0; i < nrb; ++i) env->right.push_back(Vec2d(rb[2 * i], rb[2 * i + 1]));

  CorridorConfig config;
  Corridor corridor(config, env);                 // trajectory_planner.cpp constructs it like this
  Corridor by_value = corridor;
  Corridor::CorridorConstraints cons;
  Corridor::ConvexPolygons polys;
  Corridor::LaneConstraints left, right;
  int errors_ok = 1;                               // corridor.cc:24-35
  errors_ok &= by_value.Plan(DiscretizedTrajectory(), &cons, &polys, &left, &right) == false;
  errors_ok &= by_value.Plan(DiscretizedTrajectory(pts), nullptr, &polys, &left, &right) == false;
  errors_ok &= by_value.Plan(DiscretizedTrajectory(pts), &cons, &polys, &left, nullptr) == false;
  errors_ok &= cons.empty() && polys.empty();
  const bool ok = by_value.Plan(DiscretizedTrajectory(pts), &cons, &polys, &left, &right);
  const auto stored = by_value.points_for_corridors();

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 6;
  int32_t oh[4] = {ok ? 1 : 0, (int32_t)left.size(), (int32_t)right.size(), errors_ok};
  std::fwrite(oh, sizeof(int32_t), 4, o);
  for (int i = 0; i < K; ++i) {
    const int32_t c = (i < (int)cons.size()) ? (int32_t)cons[i].size() : -1;
    std::fwrite(&c, sizeof(int32_t), 1, o);
  }
  for (size_t i = 0; i < cons.size(); ++i) {
    for (const auto& c : cons[i]) std::fwrite(c.v, sizeof(double), 3, o);
    for (const auto& q : polys[i]) std::fwrite(q.v, sizeof(double), 2, o);
  }
  auto dump_lane = [&](const Corridor::LaneConstraints& l) {
    for (const auto& e : l) {
      const double row[7] = {e.first.v[0], e.first.v[1], e.first.v[2], e.second.start().x(), e.second.start().y(),
                             e.second.end().x(), e.second.end().y()};
      std::fwrite(row, sizeof(double), 7, o);
    }
  };
  dump_lane(left);
  dump_lane(right);
  for (int i = 0; i < K; ++i) {
    const int32_t c = (i < (int)stored.size()) ? (int32_t)stored[i].size() : -1;
    std::fwrite(&c, sizeof(int32_t), 1, o);
  }
  std::fclose(o);
  return 0;
}
