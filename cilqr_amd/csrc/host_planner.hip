// Host-side producers of the solve's inputs behind the C-ABI (include/cilqr.h, "coarse trajectory"):
// cilqr_dp_plan wraps the header-only planner of include/cilqr/dp_planner.hpp (DpPlanner::Plan,
// algorithm/planner/dp_planner.cpp:135-281 with ComputePathProfile, discrete_points_math.cc:27-176) for
// callers that are not C++ (the ctypes tests, the scene generator).  No device code in this file.
#include <vector>

#include "../../include/cilqr.h"
#include "../../include/cilqr/dp_planner.hpp"

extern "C" {

void cilqr_default_dp_config(cilqr_dp_config* c) {
  if (c == nullptr) return;
  const cilqr::DpConfig d;   // default member initialisers = planner_config.h:94-133, vehicle_param.h:26-46
  c->tf = d.tf; c->delta_t = d.delta_t; c->dp_nominal_velocity = d.dp_nominal_velocity; c->dp_w_obstacle = d.dp_w_obstacle;
  c->dp_w_lateral = d.dp_w_lateral; c->dp_w_lateral_change = d.dp_w_lateral_change;
  c->dp_w_lateral_velocity_change = d.dp_w_lateral_velocity_change;
  c->dp_w_longitudinal_velocity_bias = d.dp_w_longitudinal_velocity_bias;
  c->dp_w_longitudinal_velocity_change = d.dp_w_longitudinal_velocity_change;
  c->front_hang_length = d.front_hang_length; c->wheel_base = d.wheel_base; c->rear_hang_length = d.rear_hang_length;
  c->width = d.width; c->max_velocity = d.max_velocity;
}

int cilqr_road_barriers(const double* center, int32_t n_center, double* left, double* right, int32_t max_points) {
  if (center == nullptr || left == nullptr || right == nullptr) return CILQR_ERR_NULL;
  if (n_center < 2 || max_points < 1) return CILQR_ERR_ARG;
  std::vector<std::array<double, 7>> c(n_center);
  for (int i = 0; i < n_center; ++i)
    for (int e = 0; e < 7; ++e) c[i][e] = center[(size_t)i * 7 + e];
  const cilqr::ReferenceLine ref(c);
  constexpr double kSampleStep = 0.1;                                   // environment.cpp:18
  const double start_s = c.front()[0], back_s = c.back()[0];
  const int sample_points = int((back_s - start_s) / kSampleStep);      // cpp:29-31
  if (sample_points + 1 > max_points) return CILQR_ERR_CAPACITY;
  for (int i = 0; i <= sample_points; ++i) {
    const double s = start_s + i * kSampleStep;
    const cilqr::RefPoint r = ref.EvaluateStation(s);
    const cilqr::DpPoint2 l = ref.GetCartesian(s, r.left_bound), q = ref.GetCartesian(s, -r.right_bound);   // cpp:38-39
    left[2 * i] = l.x; left[2 * i + 1] = l.y;
    right[2 * i] = q.x; right[2 * i + 1] = q.y;
  }
  return sample_points + 1;
}

int cilqr_dp_plan(const cilqr_dp_config* cfg, const cilqr_scene* scene, const double* start3, double* coarse,
                  int32_t n_knots) {
  if (cfg == nullptr || scene == nullptr || start3 == nullptr || coarse == nullptr || scene->center == nullptr)
    return CILQR_ERR_NULL;
  if (scene->n_center < 2 || scene->n_static < 0 || scene->n_dynamic < 0 || !(cfg->delta_t > 0.0) || !(cfg->tf > 0.0))
    return CILQR_ERR_ARG;
  if ((scene->n_static > 0 && (scene->static_points == nullptr || scene->static_counts == nullptr)) ||
      (scene->n_dynamic > 0 && (scene->dynamic_polygon_points == nullptr || scene->dynamic_polygon_counts == nullptr ||
                                scene->dynamic_trajectories == nullptr || scene->dynamic_trajectory_counts == nullptr)))
    return CILQR_ERR_NULL;
  cilqr::DpConfig d;
  d.tf = cfg->tf; d.delta_t = cfg->delta_t; d.dp_nominal_velocity = cfg->dp_nominal_velocity; d.dp_w_obstacle = cfg->dp_w_obstacle;
  d.dp_w_lateral = cfg->dp_w_lateral; d.dp_w_lateral_change = cfg->dp_w_lateral_change;
  d.dp_w_lateral_velocity_change = cfg->dp_w_lateral_velocity_change;
  d.dp_w_longitudinal_velocity_bias = cfg->dp_w_longitudinal_velocity_bias;
  d.dp_w_longitudinal_velocity_change = cfg->dp_w_longitudinal_velocity_change;
  d.front_hang_length = cfg->front_hang_length; d.wheel_base = cfg->wheel_base; d.rear_hang_length = cfg->rear_hang_length;
  d.width = cfg->width; d.max_velocity = cfg->max_velocity;
  if ((int32_t)(d.tf / d.delta_t + 1) != n_knots) return CILQR_ERR_KNOTS;
  std::vector<std::array<double, 7>> center(scene->n_center);
  for (int i = 0; i < scene->n_center; ++i)
    for (int e = 0; e < 7; ++e) center[i][e] = scene->center[(size_t)i * 7 + e];
  const cilqr::ReferenceLine ref(center);
  cilqr::DpEnvironment env(d, ref);
  size_t at = 0;
  for (int o = 0; o < scene->n_static; ++o) {
    if (scene->static_counts[o] < 1) return CILQR_ERR_ARG;
    std::vector<cilqr::DpPoint2> poly(scene->static_counts[o]);
    for (auto& p : poly) {
      p = cilqr::DpPoint2{scene->static_points[at * 2], scene->static_points[at * 2 + 1]};
      ++at;
    }
    env.AddStatic(poly);
  }
  size_t pa = 0, ta = 0;
  for (int o = 0; o < scene->n_dynamic; ++o) {
    const int m = scene->dynamic_polygon_counts[o], T = scene->dynamic_trajectory_counts[o];
    if (m < 1 || T < 0) return CILQR_ERR_ARG;
    std::vector<cilqr::DpPoint2> poly(m);
    for (int k = 0; k < m; ++k) poly[k] = cilqr::DpPoint2{scene->dynamic_polygon_points[(pa + k) * 2], scene->dynamic_polygon_points[(pa + k) * 2 + 1]};
    std::vector<std::array<double, 4>> traj(T);
    for (int t = 0; t < T; ++t)
      for (int e = 0; e < 4; ++e) traj[t][e] = scene->dynamic_trajectories[(ta + t) * 4 + e];
    env.AddDynamic(poly, traj);
    pa += m;
    ta += T;
  }
  cilqr::DpPlanner dp(d, &env);
  std::vector<cilqr::CoarsePoint> out;
  const bool ok = dp.Plan(start3[0], start3[1], start3[2], &out);
  if ((int32_t)out.size() != n_knots) return CILQR_ERR_KNOTS;
  for (int i = 0; i < n_knots; ++i) {
    double* r = coarse + (size_t)i * CILQR_COARSE_FIELDS;
    const cilqr::CoarsePoint& p = out[i];
    r[0] = p.time; r[1] = p.s; r[2] = p.x; r[3] = p.y; r[4] = p.theta; r[5] = p.kappa; r[6] = p.velocity; r[7] = p.a; r[8] = p.delta;
  }
  return ok ? CILQR_OK : CILQR_ERR_NO_PATH;
}

}  // extern "C"
