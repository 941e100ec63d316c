// include/cilqr/corridor.hpp -- header-only C++ adapter with the call surface of the reference's
// `planning::Corridor` (algorithm/ilqr/corridor.h:27-44), implemented on the C-ABI of
// include/cilqr.h: cilqr_build_corridors (HIP kernel on MI355X) for BuildCorridorConstraints and
// cilqr_lane_constraints (host) for the lane boundaries.  TrajectoryPlanner keeps constructing it
// with (config.corridor_config, env) and calling Plan(...) (algorithm/planner/trajectory_planner.cpp).
//
// A template over the reference's own types, so that this repository neither carries nor copies
// them.  In the reference tree:
//
//     // algorithm/ilqr/corridor.h  (OpenCV no longer needed)
//     #include <cilqr/corridor.hpp>
//     namespace planning {
//     using ConvexPolygon = std::vector<Eigen::Vector2d>;     ... typedefs unchanged (corridor.h:18-25)
//     using Corridor = cilqr::CorridorT<CorridorConfig, Env, DiscretizedTrajectory, math::Vec2d,
//                                       math::LineSegment2d, Eigen::Vector3d, Eigen::Vector2d>;
//     }
//
// What the template needs from those types (all true for the reference):
//   CorridorConfig         is_multiple_sample, max_diff_x/y, radius, max_axis_x/y, lane_segment_length
//   Env                    pointer-like; ->QueryStaticObstaclesPoints(std::vector<Vec2d>*, bool),
//                          ->QueryDynamicObstaclesPoints(double time, std::vector<Vec2d>*, bool),
//                          ->left_road_barrier(), ->right_road_barrier() -> std::vector<Vec2d>
//   DiscretizedTrajectory  empty(); trajectory() -> vector of points with .time .x .y .theta
//   Vec2d                  ctor (x, y); x(), y()
//   LineSegment2d          ctor (Vec2d start, Vec2d end)
//   Vec3 / Vec2            ctor (a, b, c) / (x, y)          (Eigen::Vector3d / Eigen::Vector2d)
//
// Behavioural notes (kept from the reference):
//   * Plan returns false for an empty trajectory, null outputs, a knot whose corridor cannot be
//     built, or a lane barrier with fewer than two sampled points (corridor.cc:24-53, 78-81).
//   * points_for_corridors() keeps the per-knot point lists INCLUDING the eight box points that
//     AddCorridorPoints appends (cc:76-77).
//   * is_multiple_sample = true: the environment is asked for the obstacles' sample points and the box gets
//     six samples per edge (cc:110-118), as in the reference; up to 296 obstacle points per knot.
//   * cv::convexHull is replaced by the library's own float32 hull: the SET of half-planes of a
//     knot is the reference's wherever no three points are collinear to float32 rounding; their
//     ORDER may start at a different vertex.
#ifndef CILQR_CORRIDOR_HPP_
#define CILQR_CORRIDOR_HPP_

#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "../cilqr.h"

namespace cilqr {

template <class CorridorConfig, class Env, class DiscretizedTrajectory, class Vec2d, class LineSegment2d,
          class Vec3, class Vec2>
class CorridorT {
 public:
  using ConvexPolygon = std::vector<Vec2>;
  using ConvexPolygons = std::vector<ConvexPolygon>;
  using Constraints = std::vector<Vec3>;
  using CorridorConstraints = std::vector<Constraints>;
  using LaneConstraints = std::vector<std::pair<Vec3, LineSegment2d>>;

  CorridorT() = default;
  CorridorT(const CorridorConfig& config, const Env& env) : config_(config), env_(env) {}
  ~CorridorT() { Release(); }
  CorridorT(const CorridorT& o) : config_(o.config_), env_(o.env_), points_for_corridors_(o.points_for_corridors_) {}
  CorridorT& operator=(const CorridorT& o) {
    if (this != &o) {
      config_ = o.config_;
      env_ = o.env_;
      points_for_corridors_ = o.points_for_corridors_;
    }
    return *this;
  }

  void Init(const CorridorConfig& config, const Env& env) {
    config_ = config;
    env_ = env;
  }

  bool Plan(const DiscretizedTrajectory& trajectory, CorridorConstraints* const corridor_constraints,
            ConvexPolygons* const convex_polygons, LaneConstraints* const left_lane_constraints,
            LaneConstraints* const right_lane_constraints) {
    if (trajectory.empty()) return false;                                              // cc:24-27
    if (corridor_constraints == nullptr || convex_polygons == nullptr || left_lane_constraints == nullptr ||
        right_lane_constraints == nullptr)
      return false;                                                                    // cc:29-35
    if (!BuildCorridorConstraints(trajectory, corridor_constraints, convex_polygons)) return false;
    if (!LaneSide(env_->left_road_barrier(), true, left_lane_constraints)) return false;     // cc:43-46
    if (!LaneSide(env_->right_road_barrier(), false, right_lane_constraints)) return false;  // cc:48-51
    return true;
  }

  std::vector<std::vector<Vec2d>> points_for_corridors() { return points_for_corridors_; }

 private:
  static constexpr int kMaxPlanes = 64;   // half-planes kept per knot

  bool BuildCorridorConstraints(const DiscretizedTrajectory& trajectory, CorridorConstraints* const corridor_constraints,
                                ConvexPolygons* const convex_polygons) {
    points_for_corridors_.clear();
    corridor_constraints->clear();
    convex_polygons->clear();
    std::vector<Vec2d> static_pts;
    env_->QueryStaticObstaclesPoints(&static_pts, config_.is_multiple_sample);              // cc:66-68
    const auto& pts = trajectory.trajectory();
    const int K = (int)pts.size();
    // gather the obstacle points of every knot (cc:73-76), padded to the longest list
    std::vector<std::vector<Vec2d>> per_knot(K);
    size_t pmax = 0;
    for (int i = 0; i < K; ++i) {
      per_knot[i] = static_pts;
      env_->QueryDynamicObstaclesPoints(pts[i].time, &per_knot[i], config_.is_multiple_sample);
      if (per_knot[i].size() > pmax) pmax = per_knot[i].size();
    }
    std::vector<double> knots((size_t)K * 3), flat((size_t)K * pmax * 2, 0.0);
    std::vector<int32_t> counts(K);
    for (int i = 0; i < K; ++i) {
      knots[3 * i] = pts[i].x; knots[3 * i + 1] = pts[i].y; knots[3 * i + 2] = pts[i].theta;
      counts[i] = (int32_t)per_knot[i].size();
      for (size_t k = 0; k < per_knot[i].size(); ++k) {
        flat[((size_t)i * pmax + k) * 2] = per_knot[i][k].x();
        flat[((size_t)i * pmax + k) * 2 + 1] = per_knot[i][k].y();
      }
    }
    if (!Acquire()) return false;
    cilqr_corridor_config cc;
    cc.max_diff_x = config_.max_diff_x; cc.max_diff_y = config_.max_diff_y; cc.radius = config_.radius;
    cc.max_axis_x = config_.max_axis_x; cc.max_axis_y = config_.max_axis_y;
    cc.lane_segment_length = config_.lane_segment_length;
    cc.is_multiple_sample = config_.is_multiple_sample ? 1 : 0;
    cc.reserved0 = 0;
    std::vector<double> cor((size_t)K * kMaxPlanes * 3), poly((size_t)K * kMaxPlanes * 2);
    std::vector<int32_t> ccount(K);
    int32_t failed = 0;
    const int rc = cilqr_build_corridors(h_, &cc, 1, K, knots.data(), pmax ? flat.data() : nullptr, counts.data(),
                                         (int32_t)pmax, cor.data(), ccount.data(), kMaxPlanes, CILQR_MEM_HOST, &failed,
                                         poly.data());
    if (rc != CILQR_OK || failed != 0) return false;                                          // cc:78-81
    for (int i = 0; i < K; ++i) {
      // the eight box points the reference appends to the stored lists (AddCorridorPoints cc:89-120)
      AppendBoxPoints(pts[i].x, pts[i].y, pts[i].theta, &per_knot[i]);
      Constraints cons;
      ConvexPolygon polygon;
      for (int c = 0; c < ccount[i]; ++c) {
        const double* p = &cor[((size_t)i * kMaxPlanes + c) * 3];
        const double* q = &poly[((size_t)i * kMaxPlanes + c) * 2];
        cons.push_back(Vec3(p[0], p[1], p[2]));
        polygon.push_back(Vec2(q[0], q[1]));
      }
      corridor_constraints->push_back(cons);
      convex_polygons->push_back(polygon);
    }
    points_for_corridors_ = per_knot;
    return true;
  }

  void AppendBoxPoints(double x, double y, double theta, std::vector<Vec2d>* const points) const {
    const double ch = std::cos(theta), sh = std::sin(theta);
    const double dx1 = ch * config_.max_axis_x, dy1 = sh * config_.max_axis_x;
    const double dx2 = sh * config_.max_axis_y, dy2 = -ch * config_.max_axis_y;
    const double cx[4] = {x + dx1 + dx2, x + dx1 - dx2, x - dx1 - dx2, x - dx1 + dx2};
    const double cy[4] = {y + dy1 + dy2, y + dy1 - dy2, y - dy1 - dy2, y - dy1 + dy2};
    const double kSampleMultiple = config_.is_multiple_sample ? 5.0 : 1.0;              // cc:110-118
    const double ratio_step = 1.0 / kSampleMultiple;
    for (int i = 0; i < 4; ++i) {
      const int n = (i + 1) % 4;
      for (double ratio = 0.0; ratio < 1.0 + 1e-10; ratio += ratio_step)
        points->push_back(Vec2d(cx[i] * (1 - ratio) + cx[n] * ratio, cy[i] * (1 - ratio) + cy[n] * ratio));
    }
  }

  bool LaneSide(const std::vector<Vec2d>& boundary, bool is_left, LaneConstraints* const out) const {
    out->clear();
    if (boundary.empty()) return false;
    std::vector<double> flat(boundary.size() * 2), rows(boundary.size() * 7 + 7);
    for (size_t i = 0; i < boundary.size(); ++i) {
      flat[2 * i] = boundary[i].x();
      flat[2 * i + 1] = boundary[i].y();
    }
    const int m = cilqr_lane_constraints(flat.data(), (int32_t)boundary.size(), config_.lane_segment_length,
                                         is_left ? 1 : 0, rows.data(), (int32_t)boundary.size() + 1);
    if (m < 1) return false;                                                            // cc:273-275
    for (int k = 0; k < m; ++k) {
      const double* r = &rows[(size_t)k * 7];
      out->push_back(std::make_pair(Vec3(r[0], r[1], r[2]), LineSegment2d(Vec2d(r[3], r[4]), Vec2d(r[5], r[6]))));
    }
    return true;
  }

  bool Acquire() {
    if (h_ != nullptr) return true;
    cilqr_config cfg;
    if (cilqr_default_config(&cfg, 10) != CILQR_OK) return false;
    return cilqr_create(&cfg, 0, 1, 16, 8, &h_) == CILQR_OK;   // a minimal handle: it only names the device
  }
  void Release() {
    if (h_ != nullptr) cilqr_destroy(h_);
    h_ = nullptr;
  }

  CorridorConfig config_;
  Env env_;
  std::vector<std::vector<Vec2d>> points_for_corridors_;
  cilqr_handle h_ = nullptr;
};

}  // namespace cilqr

#endif  // CILQR_CORRIDOR_HPP_
