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
// This file is the only code of this repository in that library: extern "C" entry points that CALL the reference's
// functions, so that tests/test_reference_pins.py can hold the oracle's restatements (and, on the GPU box, the device
// code) against the reference itself, bit for bit.  Test infrastructure only.
#include <utility>
#include <vector>

#include "algorithm/math/aabox2d.h"
#include "algorithm/math/box2d.h"
#include "algorithm/math/line_segment2d.h"
#include "algorithm/math/math_utils.h"
#include "algorithm/math/polygon2d.h"
#include "algorithm/math/vec2d.h"
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

}  // extern "C"
