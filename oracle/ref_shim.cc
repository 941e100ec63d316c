// oracle/_ref: the REAL reference code of the parts of mpt0816/Cilqr that build in this image.
//
// Most of the hot path cannot be compiled here (Eigen 3.4, ROS and OpenCV are absent: ilqr_optimizer.cc,
// vehicle_model.cc, barrier_function.h, corridor.cc, tracker.cc, dp_planner.cpp).  But the Apollo geometry and
// trajectory utilities the path calls are plain C++14 and compile from the reference's own files with g++ alone
// (oracle/Makefile, target `ref`; sources stay where they lie under /root/reference, nothing is copied):
//   algorithm/math/math_utils.cpp          NormalizeAngle                     SURVEY 8(a)-20
//   algorithm/math/line_segment2d.cpp      LineSegment2d::DistanceTo          SURVEY 8(a)-14
//   algorithm/math/{vec2d,box2d,aabox2d,polygon2d}.cpp   the collision geometry of the DP planner    8(f)-3
//   algorithm/utils/discrete_points_math.cc              ComputePathProfile                          8(f)-3
//   algorithm/utils/discretized_trajectory.cpp           station / time / projection queries         8(f)-3, 8(f)-4
// and, header-only (nothing but the reference's own headers is compiled):
//   algorithm/params/{planner_config,vehicle_param}.h    every default member initialiser             8(a)-21, 8(f)-1,3,4
//   algorithm/math/math_utils.h                          slerp, LinSpaced<N>                          8(f)-3, 8(f)-4
//   algorithm/math/pose.h                                Pose::transform                              8(f)-2, 8(f)-3
//   VehicleParam::GetDiscPositions + AABox2d::Shift + Box2d(AABox2d): the two collision boxes of the DP's cost
// This file is the only code of this repository in that library: extern "C" entry points that CALL the reference's
// functions, so that tests/test_reference_pins.py can hold the oracle's restatements (and, on the GPU box, the device
// code) against the reference itself, bit for bit.  Test infrastructure only.
#include <cstring>
#include <tuple>
#include <utility>
#include <vector>

#include "algorithm/math/aabox2d.h"
#include "algorithm/math/box2d.h"
#include "algorithm/math/line_segment2d.h"
#include "algorithm/math/math_utils.h"
#include "algorithm/math/polygon2d.h"
#include "algorithm/math/pose.h"
#include "algorithm/math/vec2d.h"
#include "algorithm/params/planner_config.h"
#include "algorithm/params/vehicle_param.h"
#include "algorithm/utils/discrete_points_math.h"
#include "algorithm/utils/discretized_trajectory.h"

using planning::math::AABox2d;
using planning::math::Box2d;
using planning::math::LineSegment2d;
using planning::math::Polygon2d;
using planning::math::Vec2d;

namespace {
// rows: time s x y theta kappa velocity left_bound right_bound (what the interpolation reads and writes)
planning::DiscretizedTrajectory make_trajectory(const double* rows, int n) {
  std::vector<planning::TrajectoryPoint> pts(n);
  for (int i = 0; i < n; ++i) {
    const double* r = rows + 9 * i;
    pts[i].time = r[0]; pts[i].s = r[1]; pts[i].x = r[2]; pts[i].y = r[3]; pts[i].theta = r[4];
    pts[i].kappa = r[5]; pts[i].velocity = r[6]; pts[i].left_bound = r[7]; pts[i].right_bound = r[8];
  }
  return planning::DiscretizedTrajectory(pts);
}
void put_point(const planning::TrajectoryPoint& p, double* o) {
  o[0] = p.time; o[1] = p.s; o[2] = p.x; o[3] = p.y; o[4] = p.theta; o[5] = p.kappa; o[6] = p.velocity;
  o[7] = p.left_bound; o[8] = p.right_bound;
}
}  // namespace

extern "C" {

double ref_normalize_angle(double a) { return planning::math::NormalizeAngle(a); }   // math_utils.cpp:53-59

// seg4 = start x, y, end x, y
double ref_segment_distance(const double* seg4, double px, double py) {              // line_segment2d.cpp:38-75
  const LineSegment2d seg(Vec2d(seg4[0], seg4[1]), Vec2d(seg4[2], seg4[3]));
  return seg.DistanceTo(Vec2d(px, py));
}

// The loop of FindNeastLaneSegment (ilqr_optimizer.cc:605-618, not buildable here: strict '<' on DistanceTo, first
// index wins) over the reference's own LineSegment2d::DistanceTo.  segs: n x (start x, y, end x, y).
int ref_nearest_segment(const double* segs, int n, double px, double py) {
  double min_dis = 1e300;
  int best = 0;
  for (int i = 0; i < n; ++i) {
    const LineSegment2d seg(Vec2d(segs[4 * i], segs[4 * i + 1]), Vec2d(segs[4 * i + 2], segs[4 * i + 3]));
    const double d = seg.DistanceTo(Vec2d(px, py));
    if (d < min_dis) {
      min_dis = d;
      best = i;
    }
  }
  return best;
}

// DiscretePointsMath::ComputePathProfile, discrete_points_math.cc:27-176; out arrays of n doubles each
int ref_compute_path_profile(double dt, const double* xy, int n, double* headings, double* s, double* v, double* a,
                             double* kappa) {
  std::vector<std::pair<double, double>> pts(n);
  for (int i = 0; i < n; ++i) pts[i] = {xy[2 * i], xy[2 * i + 1]};
  std::vector<double> h, ss, vv, aa, kk;
  const bool ok = planning::DiscretePointsMath::ComputePathProfile(dt, pts, &h, &ss, &vv, &aa, &kk);
  if (!ok) return 0;
  for (int i = 0; i < n; ++i) {
    headings[i] = h[i]; s[i] = ss[i]; v[i] = vv[i]; a[i] = aa[i]; kappa[i] = kk[i];
  }
  return 1;
}

// Polygon2d(points).HasOverlap(Box2d(AABox2d(corner, opposite corner))): polygon2d.cpp:150-164 with box2d.cpp:93-105,
// aabox2d.cpp:38-41 -- the collision test of the DP planner's cost
int ref_polygon_overlaps_aabox(const double* poly, int n, double x0, double y0, double x1, double y1) {
  std::vector<Vec2d> pts;
  for (int i = 0; i < n; ++i) pts.emplace_back(poly[2 * i], poly[2 * i + 1]);
  const Polygon2d polygon(pts);
  const Box2d box(AABox2d(Vec2d(x0, y0), Vec2d(x1, y1)));
  return polygon.HasOverlap(box) ? 1 : 0;
}
int ref_polygon_point_in(const double* poly, int n, double px, double py) {          // polygon2d.cpp:120-140
  std::vector<Vec2d> pts;
  for (int i = 0; i < n; ++i) pts.emplace_back(poly[2 * i], poly[2 * i + 1]);
  return Polygon2d(pts).IsPointIn(Vec2d(px, py)) ? 1 : 0;
}

// DiscretizedTrajectory queries (discretized_trajectory.cpp); rows: n x 9 (see make_trajectory); out9 = one point
void ref_trajectory_evaluate_station(const double* rows, int n, double station, double* out9) {   // cpp:117-128
  put_point(make_trajectory(rows, n).EvaluateStation(station), out9);
}
void ref_trajectory_evaluate_time(const double* rows, int n, double time, double* out9) {         // cpp:130-141
  put_point(make_trajectory(rows, n).EvaluateTime(time), out9);
}
// GetProjection cpp:165-197: out2 = (station, lateral), out9 = the projected point
void ref_trajectory_projection(const double* rows, int n, double px, double py, double* out2, double* out9) {
  planning::TrajectoryPoint pp;
  const Vec2d sl = make_trajectory(rows, n).GetProjection(Vec2d(px, py), &pp);
  out2[0] = sl.x();
  out2[1] = sl.y();
  put_point(pp, out9);
}
void ref_trajectory_cartesian(const double* rows, int n, double station, double lateral, double* out2) {   // cpp:199-203
  const Vec2d p = make_trajectory(rows, n).GetCartesian(station, lateral);
  out2[0] = p.x();
  out2[1] = p.y();
}

// ---- the reference's configuration, read from its OWN structs (default member initialisers + VehicleParam's
// constructor): planner_config.h:18-188, vehicle_param.h:21-95.  One value per key; 1 = known key.  Keys are the
// reference's member paths: "ilqr.weights.x_target", "vehicle.delta_rate_max", "corridor.radius", "planner.tf",
// "tracker.lateral.weight_l", "ilqr.tracker.max_num_iteration", ...
int ref_default(const char* key, double* out) {
  static const planning::PlannerConfig pc;     // holds vehicle, corridor_config, ilqr_config, tracker_config
  struct Entry { const char* key; double value; };
  const planning::IlqrConfig& ic = pc.ilqr_config;
  const planning::VehicleParam& v = pc.vehicle;
  const planning::CorridorConfig& cc = pc.corridor_config;
  const Entry table[] = {
      {"ilqr.num_of_disc", (double)ic.num_of_disc}, {"ilqr.safe_margin", ic.safe_margin}, {"ilqr.t", ic.t},
      {"ilqr.t_rate", ic.t_rate}, {"ilqr.max_iter_num", (double)ic.max_iter_num}, {"ilqr.abs_cost_tol", ic.abs_cost_tol},
      {"ilqr.rel_cost_tol", ic.rel_cost_tol}, {"ilqr.alpha", ic.alpha}, {"ilqr.gamma", ic.gamma}, {"ilqr.rho", ic.rho},
      {"ilqr.weights.jerk", ic.weights.jerk}, {"ilqr.weights.delta_rate", ic.weights.delta_rate},
      {"ilqr.weights.x_target", ic.weights.x_target}, {"ilqr.weights.y_target", ic.weights.y_target},
      {"ilqr.weights.theta", ic.weights.theta}, {"ilqr.weights.v", ic.weights.v}, {"ilqr.weights.a", ic.weights.a},
      {"ilqr.weights.delta", ic.weights.delta},
      {"vehicle.front_hang_length", v.front_hang_length}, {"vehicle.wheel_base", v.wheel_base},
      {"vehicle.rear_hang_length", v.rear_hang_length}, {"vehicle.width", v.width}, {"vehicle.max_velocity", v.max_velocity},
      {"vehicle.min_acceleration", v.min_acceleration}, {"vehicle.max_acceleration", v.max_acceleration},
      {"vehicle.jerk_min", v.jerk_min}, {"vehicle.jerk_max", v.jerk_max}, {"vehicle.delta_min", v.delta_min},
      {"vehicle.delta_max", v.delta_max}, {"vehicle.delta_rate_min", v.delta_rate_min},
      {"vehicle.delta_rate_max", v.delta_rate_max}, {"vehicle.phi_max", v.phi_max}, {"vehicle.omega_max", v.omega_max},
      {"vehicle.radius", v.radius}, {"vehicle.f2x", v.f2x}, {"vehicle.r2x", v.r2x},
      {"corridor.is_multiple_sample", cc.is_multiple_sample ? 1.0 : 0.0}, {"corridor.max_diff_x", cc.max_diff_x},
      {"corridor.max_diff_y", cc.max_diff_y}, {"corridor.radius", cc.radius}, {"corridor.max_axis_x", cc.max_axis_x},
      {"corridor.max_axis_y", cc.max_axis_y}, {"corridor.lane_segment_length", cc.lane_segment_length},
      {"planner.nfe", (double)pc.nfe}, {"planner.delta_t", pc.delta_t}, {"planner.tf", pc.tf},
      {"planner.dp_nominal_velocity", pc.dp_nominal_velocity}, {"planner.dp_w_obstacle", pc.dp_w_obstacle},
      {"planner.dp_w_lateral", pc.dp_w_lateral}, {"planner.dp_w_lateral_change", pc.dp_w_lateral_change},
      {"planner.dp_w_lateral_velocity_change", pc.dp_w_lateral_velocity_change},
      {"planner.dp_w_longitudinal_velocity_bias", pc.dp_w_longitudinal_velocity_bias},
      {"planner.dp_w_longitudinal_velocity_change", pc.dp_w_longitudinal_velocity_change},
  };
  for (const Entry& e : table)
    if (std::strcmp(e.key, key) == 0) {
      *out = e.value;
      return 1;
    }
  // the two TrackerConfig instances (PlannerConfig::tracker_config, IlqrConfig::tracker_config: planner_config.h:72,187)
  for (int which = 0; which < 2; ++which) {
    const char* prefix = which ? "ilqr.tracker." : "tracker.";
    const planning::TrackerConfig& tc = which ? ic.tracker_config : pc.tracker_config;
    if (std::strncmp(key, prefix, std::strlen(prefix)) != 0) continue;
    const char* k = key + std::strlen(prefix);
    const Entry t[] = {
        {"sumulation_dt", tc.sumulation_dt}, {"dt", tc.dt}, {"tolerance", tc.tolerance},
        {"max_num_iteration", (double)tc.max_num_iteration},
        {"lateral.weight_l", tc.lateral_config.weight_l}, {"lateral.weight_theta", tc.lateral_config.weight_theta},
        {"lateral.weight_delta", tc.lateral_config.weight_delta},
        {"lateral.weight_delta_rate", tc.lateral_config.weight_delta_rate},
        {"lateral.preview_time", tc.lateral_config.preview_time},
        {"longitudinal.weight_s", tc.longitudinal_config.weight_s}, {"longitudinal.weight_v", tc.longitudinal_config.weight_v},
        {"longitudinal.weight_a", tc.longitudinal_config.weight_a}, {"longitudinal.weight_j", tc.longitudinal_config.weight_j},
        {"longitudinal.preview_time", tc.longitudinal_config.preview_time},
    };
    for (const Entry& e : t)
      if (std::strcmp(e.key, k) == 0) {
        *out = e.value;
        return 1;
      }
  }
  return 0;
}

// math::slerp, math_utils.h:208-225 (EvaluateStation / EvaluateTime interpolate headings with it)
double ref_slerp(double a0, double t0, double a1, double t1, double t) { return planning::math::slerp(a0, t0, a1, t1, t); }

// math::LinSpaced<N>, math_utils.h:245-254, for the three sizes DpPlanner's constructor uses (dp_planner.cpp:31-33 with
// NT = 5, NS = 7, NL - 1 = 9: dp_planner.h:27-29); other n: the run-time overload math_utils.h:256-265.  Returns n.
int ref_lin_spaced(int n, double start, double end, double* out) {
  if (n == 5) { const auto r = planning::math::LinSpaced<5>(start, end); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  else if (n == 7) { const auto r = planning::math::LinSpaced<7>(start, end); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  else if (n == 9) { const auto r = planning::math::LinSpaced<9>(start, end); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  else { const auto r = planning::math::LinSpaced(start, end, n); for (int i = 0; i < n; ++i) out[i] = r[i]; }
  return n;
}

// Pose(x, y, theta).transform(Pose(rx, ry, rtheta)), pose.h:40-46: how the node places a dynamic obstacle's body-frame
// polygon along its trajectory (planning_node.cc:63-80).  out3 = x, y, theta
void ref_pose_transform(double x, double y, double theta, double rx, double ry, double rtheta, double* out3) {
  const planning::math::Pose p = planning::math::Pose(x, y, theta).transform(planning::math::Pose(rx, ry, rtheta));
  out3[0] = p.x(); out3[1] = p.y(); out3[2] = p.theta();
}

// The two boxes Environment::CheckOptimizationCollision tests (environment.cpp:92-104; that file includes the ROS
// plotting header and does not build here): the statements of :94-104 on the reference's own VehicleParam (defaults),
// AABox2d and Box2d.  out[0..3] = xr yr xf yf as the caller NAMES them (std::tie(xr, yr, xf, yf) = GetDiscPositions,
// which returns (xf, yf, xr, yr): the names are swapped in the reference and that is what it computes with);
// then per box (f_box, r_box): centre x y, half_length, half_width, min_x max_x min_y max_y, 4 corners x y = 16 doubles.
void ref_collision_boxes(double x, double y, double theta, double collision_buffer, double* out36) {
  static const planning::VehicleParam vehicle;
  AABox2d initial_box({-vehicle.radius - collision_buffer, -vehicle.radius - collision_buffer},
                      {vehicle.radius + collision_buffer, vehicle.radius + collision_buffer});
  double xr, yr, xf, yf;
  std::tie(xr, yr, xf, yf) = vehicle.GetDiscPositions(x, y, theta);
  out36[0] = xr; out36[1] = yr; out36[2] = xf; out36[3] = yf;
  auto f_box = initial_box, r_box = initial_box;
  f_box.Shift({xf, yf});
  r_box.Shift({xr, yr});
  const Box2d boxes[2] = {Box2d(f_box), Box2d(r_box)};
  for (int k = 0; k < 2; ++k) {
    double* o = out36 + 4 + 16 * k;
    const Box2d& b = boxes[k];
    o[0] = b.center_x(); o[1] = b.center_y(); o[2] = b.half_length(); o[3] = b.half_width();
    o[4] = b.min_x(); o[5] = b.max_x(); o[6] = b.min_y(); o[7] = b.max_y();
    std::vector<Vec2d> c;
    b.GetAllCorners(&c);
    for (int i = 0; i < 4; ++i) { o[8 + 2 * i] = c[i].x(); o[9 + 2 * i] = c[i].y(); }
  }
}

}  // extern "C"
