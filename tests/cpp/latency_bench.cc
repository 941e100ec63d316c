// Latency of the drop-in call: planning::IlqrOptimizer::Plan (include/cilqr/ilqr_optimizer.hpp, batch of ONE, host
// containers in and out -- what the reference's ROS node issues, trajectory_planner.cpp:79-89) and cilqr_solve_batch
// with a small batch of host arrays, timed per call with steady_clock like the reference times its own Plan
// (ilqr_optimizer.cc:82-94).
//
//   latency_bench <scenes.bin> <batch>
// scenes.bin: int32 n, K, cmax, nl, nr | left[nl][7] | right[nr][7] | n x { start[4] | coarse[K][6] | counts[K] int32 |
//             corridor[K][cmax][3] }
// stdout: one JSON object.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "cilqr/ilqr_optimizer.hpp"
#include "reference_types.hpp"

namespace planning {
using IlqrOptimizer = cilqr::IlqrOptimizerT<TrajectoryPoint, DiscretizedTrajectory, CorridorConstraints, LaneConstraints,
                                            IlqrConfig, VehicleParam, Cost>;
}

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

static void stats(std::vector<double> v, const char* name) {
  std::sort(v.begin(), v.end());
  const double mean = std::accumulate(v.begin(), v.end(), 0.0) / v.size();
  std::printf("\"%s\": {\"calls\": %zu, \"mean_ms\": %.4f, \"median_ms\": %.4f, \"p95_ms\": %.4f, \"min_ms\": %.4f, \"max_ms\": %.4f}",
              name, v.size(), mean, v[v.size() / 2], v[(size_t)(0.95 * (v.size() - 1))], v.front(), v.back());
}

int main(int argc, char** argv) {
  using namespace planning;
  using clk = std::chrono::steady_clock;
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  const int batch = std::atoi(argv[2]);
  int32_t hdr[5];
  if (!rd(f, hdr, 5)) return 4;
  const int n = hdr[0], K = hdr[1], cmax = hdr[2], nl = hdr[3], nr = hdr[4];
  std::vector<double> left((size_t)nl * 7), right((size_t)nr * 7);
  if (!rd(f, left.data(), left.size()) || !rd(f, right.data(), right.size())) return 5;
  std::vector<double> start((size_t)n * 4), coarse((size_t)n * K * 6), cor((size_t)n * K * cmax * 3);
  std::vector<int32_t> counts((size_t)n * K);
  for (int b = 0; b < n; ++b)
    if (!rd(f, &start[(size_t)b * 4], 4) || !rd(f, &coarse[(size_t)b * K * 6], (size_t)K * 6) ||
        !rd(f, &counts[(size_t)b * K], K) || !rd(f, &cor[(size_t)b * K * cmax * 3], (size_t)K * cmax * 3))
      return 6;
  std::fclose(f);

  auto lanes = [](const std::vector<double>& t, int m) {
    LaneConstraints out;
    for (int k = 0; k < m; ++k) {
      const double* r = &t[(size_t)k * 7];
      out.push_back({Vector3d(r[0], r[1], r[2]), math::LineSegment2d(math::Vec2d(r[3], r[4]), math::Vec2d(r[5], r[6]))});
    }
    return out;
  };
  const LaneConstraints l = lanes(left, nl), r = lanes(right, nr);
  IlqrConfig config;
  VehicleParam vehicle;
  IlqrOptimizer opt(config, vehicle, 0.1 * (K - 1), 0.1);     // trajectory_planner.cpp:26

  // ---- Plan, one scene per call ----
  // pass 0: the first calls create the handle and load code objects (untimed); pass 1: the timed calls; pass 2: the same
  // calls once more with the library's phase profiling on (HIP events around every phase: adds a little, so not the
  // timed pass) -- where a batch-of-one call spends its time
  std::vector<double> t_plan, iters;
  struct Phases { double wall = 0, flatten = 0, solve = 0, unflatten = 0, dev_span = 0, tail = 0, other = 0, lockstep = 0, iters = 0; int n = 0; } ph;
  int failed = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int m = pass == 0 ? std::min(n, 8) : n;
    if (pass == 2 && opt.native_handle() != nullptr) cilqr_set_profiling(opt.native_handle(), 1);
    for (int b = 0; b < m; ++b) {
      TrajectoryPoint st;
      st.x = start[(size_t)b * 4]; st.y = start[(size_t)b * 4 + 1]; st.theta = start[(size_t)b * 4 + 2]; st.velocity = start[(size_t)b * 4 + 3];
      std::vector<TrajectoryPoint> pts(K);
      for (int i = 0; i < K; ++i) {
        const double* c = &coarse[((size_t)b * K + i) * 6];
        pts[i].time = 0.1 * i;
        pts[i].x = c[0]; pts[i].y = c[1]; pts[i].theta = c[2]; pts[i].velocity = c[3]; pts[i].a = c[4]; pts[i].delta = c[5];
      }
      DiscretizedTrajectory coarse_traj(pts);
      CorridorConstraints corridor(K);
      for (int i = 0; i < K; ++i)
        for (int c = 0; c < counts[(size_t)b * K + i]; ++c) {
          const double* p = &cor[(((size_t)b * K + i) * cmax + c) * 3];
          corridor[i].push_back(Vector3d(p[0], p[1], p[2]));
        }
      DiscretizedTrajectory result;
      std::vector<DiscretizedTrajectory> iter_trajs;
      const auto t0 = clk::now();
      const bool ok = opt.Plan(st, coarse_traj, corridor, l, r, &result, &iter_trajs);
      const auto t1 = clk::now();
      if (pass == 1) {
        t_plan.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        iters.push_back((double)opt.cost().size() - 1.0);
        if (!ok || result.empty()) ++failed;
      }
      if (pass == 2 && ok) {
        cilqr_profile pr;
        if (cilqr_get_profile(opt.native_handle(), &pr) == CILQR_OK) {
          ph.wall += std::chrono::duration<double, std::milli>(t1 - t0).count();
          ph.flatten += opt.last_timing().flatten_ms; ph.solve += opt.last_timing().solve_ms; ph.unflatten += opt.last_timing().unflatten_ms;
          ph.dev_span += pr.total_ms; ph.tail += pr.tail_ms; ph.other += pr.other_ms;
          ph.lockstep += pr.quadratize_ms + pr.backward_ms + pr.linesearch_ms; ph.iters += pr.iterations; ph.n += 1;
        }
      }
    }
  }
  if (opt.native_handle() != nullptr) cilqr_set_profiling(opt.native_handle(), 0);

  // ---- cilqr_solve_batch, `batch` scenes per call, host arrays ----
  std::vector<double> t_batch, t_submit;
  if (batch > 1 && n >= batch) {
    cilqr_config cfg;
    cilqr_default_config(&cfg, K - 1);
    cilqr_handle h = nullptr;
    if (cilqr_create(&cfg, 0, batch, cmax, std::max(nl, nr), &h) != CILQR_OK) return 7;
    const int M1 = cfg.max_iter + 1;
    std::vector<double> traj((size_t)batch * K * CILQR_TRAJ_FIELDS), hist((size_t)batch * M1 * CILQR_COST_FIELDS);
    std::vector<int32_t> nc(batch), stt(batch), ni(batch);
    for (int pass = 0; pass < 2; ++pass)
      for (int b0 = 0; b0 + batch <= (pass == 0 ? batch : n); b0 += batch) {
        cilqr_problem_batch in{};
        in.batch = batch; in.n_knots = K; in.cmax = cmax; in.memory = CILQR_MEM_HOST;
        in.start = &start[(size_t)b0 * 4]; in.coarse = &coarse[(size_t)b0 * K * 6];
        in.corridor = &cor[(size_t)b0 * K * cmax * 3]; in.corridor_count = &counts[(size_t)b0 * K];
        in.n_left = nl; in.n_right = nr; in.left_lane = left.data(); in.right_lane = right.data();
        cilqr_solution_batch out{};
        out.memory = CILQR_MEM_HOST; out.traj = traj.data(); out.cost_hist = hist.data();
        out.n_cost = nc.data(); out.status = stt.data(); out.n_iter = ni.data();
        const auto t0 = clk::now();
        const int rc = cilqr_solve_batch(h, &in, &out);
        const auto t1 = clk::now();
        if (rc != CILQR_OK) return 8;
        if (pass == 1) t_batch.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        // the same batch through cilqr_submit + cilqr_wait (a worker thread drives it and its host waits nap between polls:
        // ADVICE r05 -- what that costs a small batch, whose iterations are ~100 us)
        const auto t2 = clk::now();
        const int rs = cilqr_submit(h, &in, &out);
        const int rw = rs == CILQR_OK ? cilqr_wait(h) : rs;
        const auto t3 = clk::now();
        if (rw != CILQR_OK) return 9;
        if (pass == 1) t_submit.push_back(std::chrono::duration<double, std::milli>(t3 - t2).count());
      }
    cilqr_destroy(h);
  }
  std::printf("{\"scenes\": %d, \"n_steps\": %d, \"plan_failed\": %d, \"mean_accepted_iterations\": %.2f, ", n, K - 1, failed,
              std::accumulate(iters.begin(), iters.end(), 0.0) / std::max<size_t>(1, iters.size()));
  stats(t_plan, "plan_b1");
  if (ph.n > 0) {
    const double k = 1.0 / ph.n;
    // iterations of Optimize() per call: all of them run inside the tail kernel for a batch of one (one workgroup, all
    // remaining iterations in one launch); load / init guess / first cost / export are `load_initguess_export_ms`
    std::printf(", \"plan_b1_phases\": {\"calls\": %d, \"wall_ms\": %.4f, \"flatten_ms\": %.4f, \"solve_call_ms\": %.4f, \"unflatten_ms\": %.4f, "
                "\"device_span_ms\": %.4f, \"tail_kernel_ms\": %.4f, \"load_initguess_export_ms\": %.4f, \"lockstep_ms\": %.4f, "
                "\"host_transfers_and_launch_gaps_ms\": %.4f, \"iterations\": %.2f, \"tail_us_per_iteration\": %.1f, "
                "\"note\": \"means over the calls of an extra pass with HIP events around every phase (not the timed pass)\"}",
                ph.n, ph.wall * k, ph.flatten * k, ph.solve * k, ph.unflatten * k, ph.dev_span * k, ph.tail * k, ph.other * k, ph.lockstep * k,
                (ph.solve - ph.dev_span) * k, ph.iters * k, ph.iters > 0 ? 1e3 * ph.tail / ph.iters : 0.0);
  }
  if (!t_batch.empty()) {
    std::printf(", \"batch\": %d, ", batch);
    stats(t_batch, "solve_batch");
    std::printf(", ");
    stats(t_submit, "submit_wait");
  }
  std::printf("}\n");
  return 0;
}
