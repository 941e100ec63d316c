// Drives include/cilqr/ilqr_optimizer.hpp the way the reference's TrajectoryPlanner drives
// planning::IlqrOptimizer (algorithm/planner/trajectory_planner.cpp:26,80-97).  The reference's types come from
// tests/cpp/reference_types.hpp: its OWN headers with -DCILQR_TEST_REFERENCE_HEADERS (build container), minimal
// stand-ins otherwise (GPU box).
//
//   adapter_test <scene.bin> <out.bin>
// scene.bin: int32 K, cmax, nl, nr | start[4] | coarse[K][6] | counts[K] (int32) |
//            corridor[K][cmax][3] | left[nl][7] | right[nr][7]
// out.bin:   int32 n_cost, n_iter_trajs, plan_ok | traj[K][10] | cost[n_cost][5] | iter0[K][10]
#include <cstdio>
#include <cstdlib>

#include "cilqr/ilqr_optimizer.hpp"
#include "reference_types.hpp"

namespace planning {
using IlqrOptimizer = cilqr::IlqrOptimizerT<TrajectoryPoint, DiscretizedTrajectory, CorridorConstraints, LaneConstraints,
                                            IlqrConfig, VehicleParam, Cost>;
}

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  using namespace planning;
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[4];
  if (!rd(f, hdr, 4)) return 4;
  const int K = hdr[0], cmax = hdr[1], nl = hdr[2], nr = hdr[3];
  std::vector<double> start(4), coarse((size_t)K * 6), cor((size_t)K * cmax * 3), left((size_t)nl * 7), right((size_t)nr * 7);
  std::vector<int32_t> counts(K);
  if (!rd(f, start.data(), 4) || !rd(f, coarse.data(), coarse.size()) || !rd(f, counts.data(), counts.size()) ||
      !rd(f, cor.data(), cor.size()) || !rd(f, left.data(), left.size()) || !rd(f, right.data(), right.size()))
    return 5;
  std::fclose(f);

  TrajectoryPoint st;
  st.x = start[0]; st.y = start[1]; st.theta = start[2]; st.velocity = start[3];
  std::vector<TrajectoryPoint> pts(K);
  for (int i = 0; i < K; ++i) {
    pts[i].time = 0.1 * i;
    pts[i].x = coarse[i * 6 + 0]; pts[i].y = coarse[i * 6 + 1]; pts[i].theta = coarse[i * 6 + 2];
    pts[i].velocity = coarse[i * 6 + 3]; pts[i].a = coarse[i * 6 + 4]; pts[i].delta = coarse[i * 6 + 5];
  }
  DiscretizedTrajectory coarse_traj(pts);
  CorridorConstraints corridor(K);
  for (int i = 0; i < K; ++i)
    for (int c = 0; c < counts[i]; ++c) {
      const double* p = &cor[((size_t)i * cmax + c) * 3];
      corridor[i].push_back(Vector3d(p[0], p[1], p[2]));
    }
  auto lanes = [](const std::vector<double>& t, int n) {
    LaneConstraints out;
    for (int k = 0; k < n; ++k) {
      const double* r = &t[(size_t)k * 7];
      out.push_back({Vector3d(r[0], r[1], r[2]), math::LineSegment2d(math::Vec2d(r[3], r[4]), math::Vec2d(r[5], r[6]))});
    }
    return out;
  };
  const LaneConstraints l = lanes(left, nl), r = lanes(right, nr);

  IlqrConfig config;
  VehicleParam vehicle;
  // same construction as trajectory_planner.cpp:26: (ilqr_config, vehicle, tf, delta_t)
  IlqrOptimizer opt(config, vehicle, 0.1 * (K - 1), 0.1);
  IlqrOptimizer by_value = opt;                 // the reference holds it by value and copies it
  DiscretizedTrajectory result;
  std::vector<DiscretizedTrajectory> iter_trajs;
  // error paths first (ilqr_optimizer.cc:64-78)
  int errors_ok = 1;
  errors_ok &= by_value.Plan(st, coarse_traj, corridor, l, r, nullptr, &iter_trajs) == false;
  errors_ok &= by_value.Plan(st, coarse_traj, corridor, LaneConstraints(), r, &result, &iter_trajs) == false;
  errors_ok &= by_value.Plan(st, DiscretizedTrajectory(std::vector<TrajectoryPoint>(K - 1)), corridor, l, r, &result, &iter_trajs) == false;
  errors_ok &= result.empty() && iter_trajs.empty() && by_value.cost().empty();
  const bool ok = by_value.Plan(st, coarse_traj, corridor, l, r, &result, &iter_trajs);
  const std::vector<Cost> costs = by_value.cost();

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 6;
  int32_t oh[3] = {(int32_t)costs.size(), (int32_t)iter_trajs.size(), (int32_t)((ok ? 1 : 0) | (errors_ok ? 2 : 0))};
  std::fwrite(oh, sizeof(int32_t), 3, o);
  auto dump = [&](const DiscretizedTrajectory& t) {
    for (const auto& p : t.trajectory()) {
      const double row[10] = {p.time, p.x, p.y, p.theta, p.velocity, p.a, p.delta, p.kappa, p.jerk, p.delta_rate};
      std::fwrite(row, sizeof(double), 10, o);
    }
  };
  dump(result);
  for (const auto& c : costs) {
    const double row[5] = {c.total_cost, c.target_cost, c.dynamic_cost, c.corridor_cost, c.lane_boundary_cost};
    std::fwrite(row, sizeof(double), 5, o);
  }
  if (!iter_trajs.empty()) dump(iter_trajs[0]);
  std::fclose(o);
  return (ok && errors_ok) ? 0 : 1;
}
