// include/cilqr/dp_planner.hpp -- header-only C++14: the coarse-trajectory producer in front of the
// CILQR solve, host side.
//
// What it restates (behaviour, not code) from the reference:
//   DpPlanner               algorithm/planner/dp_planner.{h,cpp}: 5 x 7 x 10 (time, station, lateral) sampling DP in
//                           the Frenet frame of the centre line -- GetCost cpp:88-133, GetCollisionCost cpp:44-86,
//                           Plan cpp:135-281, InterpolateLinearly cpp:283-320
//   ComputePathProfile      algorithm/utils/discrete_points_math.cc:27-176 (finite-difference heading / s / v / a / kappa)
//   reference-line queries  algorithm/utils/discretized_trajectory.cpp: EvaluateStation :117-128, GetProjection
//                           :165-197, GetCartesian :199-203, LinearInterpolateTrajectory :66-89, math::slerp
//   Environment             algorithm/utils/environment.cpp: set_reference :20-43 (road barriers every 0.1 m),
//                           CheckStaticCollision :45-80, CheckDynamicCollision :113-130, CheckOptimizationCollision :92-111
//   geometry                Polygon2d::HasOverlap(Box2d) polygon2d.cpp:150-164, Polygon2d::IsPointIn :120-140,
//                           Box2d::IsPointIn box2d.cpp:123-129, VehicleParam disc positions vehicle_param.h:76-95
//
// Every number is produced by the reference's expressions in the reference's order (the DP compares costs with
// '<', so a different rounding could pick another cell); what differs is the bookkeeping around them: the point count
// of a layer's path segment is computed once, the last point of the parent's segment in closed form instead of
// rebuilding the whole segment, obstacle polygons are placed once per trajectory sample with their bounding boxes, and
// the lateral offsets of a cell are cached.  tests/test_dp_planner.py holds it to the line-by-line restatement in
// oracle/dp_oracle.cc, bit for bit.
//
// Quirks kept: the DP-local epsilon of 1e-3 next to the geometry epsilon of 1e-10 (cpp:25 vs vec2d.h:33); the loop
// that counts a segment's points accumulates t += delta_t in floating point (cpp:288-298); GetDiscPositions returns
// (front, rear) into variables named (rear, front) (environment.cpp:100-101) -- harmless, both discs are tested;
// a dynamic obstacle whose last sample time equals the query time dereferences end() in the reference
// (environment.cpp:119-125): here the last sample is used.
#ifndef CILQR_DP_PLANNER_HPP_
#define CILQR_DP_PLANNER_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <initializer_list>
#include <limits>
#include <utility>
#include <vector>

namespace cilqr {

constexpr int kDpNT = 5;    // dp_planner.h:27-29
constexpr int kDpNS = 7;
constexpr int kDpNL = 10;

// live fields of PlannerConfig (planner_config.h:88-133) and VehicleParam (vehicle_param.h:26-46) for this stage
struct DpConfig {
  double tf = 8.0;
  double delta_t = 0.1;
  double dp_nominal_velocity = 10.0;
  double dp_w_obstacle = 1000.0;
  double dp_w_lateral = 0.1;
  double dp_w_lateral_change = 0.5;
  double dp_w_lateral_velocity_change = 1.0;
  double dp_w_longitudinal_velocity_bias = 10.0;
  double dp_w_longitudinal_velocity_change = 1.0;
  double front_hang_length = 0.96;
  double wheel_base = 1.0;
  double rear_hang_length = 0.929;
  double width = 1.942;
  double max_velocity = 20.0;
};

struct DpPoint2 {
  double x, y;
};

// TrajectoryPoint fields the coarse trajectory carries (discretized_trajectory.h:26-43)
struct CoarsePoint {
  double time = 0.0, s = 0.0, x = 0.0, y = 0.0, theta = 0.0, kappa = 0.0, velocity = 0.0, a = 0.0, delta = 0.0;
};

namespace dp_detail {

constexpr double kGeomEps = 1e-10;   // math::kMathEpsilon, vec2d.h:33
constexpr double kDpEps = 1e-3;      // dp_planner.cpp:25

inline double NormalizeAngle(double angle) {   // math_utils.cpp:53-59
  double a = std::fmod(angle + M_PI, 2.0 * M_PI);
  if (a < 0.0) a += 2.0 * M_PI;
  return a - M_PI;
}

inline double Slerp(double a0, double t0, double a1, double t1, double t) {   // math_utils.h:208-225
  if (std::abs(t1 - t0) <= kGeomEps) return NormalizeAngle(a0);
  const double a0_n = NormalizeAngle(a0);
  const double a1_n = NormalizeAngle(a1);
  double d = a1_n - a0_n;
  if (d > M_PI) d = d - 2 * M_PI;
  else if (d < -M_PI) d = d + 2 * M_PI;
  const double r = (t - t0) / (t1 - t0);
  const double a = a0_n + d * r;
  return NormalizeAngle(a);
}

template <int N>
inline std::array<double, N> LinSpaced(double start, double end) {   // math_utils.h:245-254
  std::array<double, N> res;
  const double step = (end - start) / (N - 1);
  for (int i = 0; i < N; ++i) res[i] = start + step * i;
  return res;
}

}  // namespace dp_detail

// ---- the centre line (CenterLinePoint: s x y theta kappa left_bound right_bound) -------------------------------
struct RefPoint {
  double s = 0.0, x = 0.0, y = 0.0, theta = 0.0, kappa = 0.0, left_bound = 0.0, right_bound = 0.0;
};

class ReferenceLine {
 public:
  ReferenceLine() = default;
  explicit ReferenceLine(const std::vector<std::array<double, 7>>& center) {
    pts_.resize(center.size());
    for (size_t i = 0; i < center.size(); ++i) {
      const auto& c = center[i];
      pts_[i] = RefPoint{c[0], c[1], c[2], c[3], c[4], c[5], c[6]};
    }
  }
  const std::vector<RefPoint>& points() const { return pts_; }
  bool empty() const { return pts_.size() < 2; }

  // discretized_trajectory.cpp:117-128 (+ :33-47, :66-89)
  RefPoint EvaluateStation(double station) const {
    size_t it;
    if (station >= pts_.back().s) {
      it = pts_.size() - 1;
    } else if (station < pts_.front().s) {
      it = 0;
    } else {
      size_t lo = 0, hi = pts_.size();   // first point with s >= station
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (pts_[mid].s < station) lo = mid + 1;
        else hi = mid;
      }
      it = lo;
    }
    if (it == 0) it = 1;
    return Interpolate(pts_[it - 1], pts_[it], station);
  }

  DpPoint2 GetCartesian(double station, double lateral) const {   // cpp:199-203
    const RefPoint ref = EvaluateStation(station);
    return DpPoint2{ref.x - lateral * std::sin(ref.theta), ref.y + lateral * std::cos(ref.theta)};
  }

  // (station, lateral) of a point, cpp:165-197
  DpPoint2 GetProjection(double px, double py) const {
    size_t idx = 0;
    double nearest = std::numeric_limits<double>::max();
    for (size_t i = 0; i < pts_.size(); ++i) {
      const double dx = pts_[i].x - px, dy = pts_[i].y - py;
      const double d = dx * dx + dy * dy;
      if (d < nearest) {
        idx = i;
        nearest = d;
      }
    }
    RefPoint proj = pts_[idx];
    const size_t i0 = idx > 0 ? idx - 1 : 0;
    const size_t i1 = std::min(pts_.size() - 1, idx + 1);
    if (i0 < i1) {
      const double v0x = px - pts_[i0].x, v0y = py - pts_[i0].y;
      const double v1x = pts_[i1].x - pts_[i0].x, v1y = pts_[i1].y - pts_[i0].y;
      const double v1_norm = std::sqrt(v1x * v1x + v1y * v1y);
      const double dot = v0x * v1x + v0y * v1y;
      const double delta_s = dot / v1_norm;
      proj = Interpolate(pts_[i0], pts_[i1], pts_[i0].s + delta_s);
    }
    const double nr_x = px - proj.x, nr_y = py - proj.y;
    const double lateral = std::copysign(std::hypot(nr_x, nr_y), nr_y * std::cos(proj.theta) - nr_x * std::sin(proj.theta));
    return DpPoint2{proj.s, lateral};
  }

 private:
  static RefPoint Interpolate(const RefPoint& p0, const RefPoint& p1, double s) {   // cpp:66-89
    const double s0 = p0.s, s1 = p1.s;
    if (std::abs(s1 - s0) < dp_detail::kGeomEps) return p0;
    RefPoint pt;
    const double weight = (s - s0) / (s1 - s0);
    pt.s = s;
    pt.x = (1 - weight) * p0.x + weight * p1.x;
    pt.y = (1 - weight) * p0.y + weight * p1.y;
    pt.theta = dp_detail::Slerp(p0.theta, p0.s, p1.theta, p1.s, s);
    pt.kappa = (1 - weight) * p0.kappa + weight * p1.kappa;
    pt.left_bound = (1 - weight) * p0.left_bound + weight * p1.left_bound;
    pt.right_bound = (1 - weight) * p0.right_bound + weight * p1.right_bound;
    return pt;
  }
  std::vector<RefPoint> pts_;
};

// ---- the scene as the reference's Environment holds it ------------------------------------------------------------
class DpEnvironment {
 public:
  struct Poly {   // a placed polygon with its bounding box (Polygon2d::BuildFromPoints, polygon2d.cpp:240-257)
    std::vector<DpPoint2> pts;
    double min_x, max_x, min_y, max_y;
  };
  struct Dynamic {
    std::vector<double> time;
    std::vector<Poly> poly;   // one per trajectory sample (PlanningNode::DynamicObstaclesCallback, planning_node.cc:63-78)
  };

  DpEnvironment() = default;
  DpEnvironment(const DpConfig& cfg, const ReferenceLine& ref) : cfg_(cfg), ref_(ref) {
    const double length = cfg.wheel_base + cfg.rear_hang_length + cfg.front_hang_length;   // vehicle_param.h:80-85
    radius_ = std::hypot(0.25 * length, 0.5 * cfg.width);
    r2x_ = 0.25 * length - cfg.rear_hang_length;
    f2x_ = 0.75 * length - cfg.rear_hang_length;
    // set_reference, environment.cpp:20-43: both road barriers sampled every 0.1 m, sorted by x
    constexpr double kSampleStep = 0.1;
    const double start_s = ref_.points().front().s, back_s = ref_.points().back().s;
    const int sample_points = int((back_s - start_s) / kSampleStep);
    for (int i = 0; i <= sample_points; ++i) {
      const double s = start_s + i * kSampleStep;
      const RefPoint r = ref_.EvaluateStation(s);
      barrier_.push_back(ref_.GetCartesian(s, r.left_bound));
      barrier_.push_back(ref_.GetCartesian(s, -r.right_bound));
    }
    std::stable_sort(barrier_.begin(), barrier_.end(), [](const DpPoint2& a, const DpPoint2& b) { return a.x < b.x; });
  }

  const ReferenceLine& reference() const { return ref_; }

  void AddStatic(const std::vector<DpPoint2>& world_polygon) { statics_.push_back(MakePoly(world_polygon)); }

  // body-frame polygon + trajectory (time, x, y, theta): one placed polygon per sample (Pose::transform)
  void AddDynamic(const std::vector<DpPoint2>& body_polygon, const std::vector<std::array<double, 4>>& trajectory) {
    Dynamic d;
    for (const auto& tp : trajectory) {
      const double c = std::cos(tp[3]), s = std::sin(tp[3]);
      std::vector<DpPoint2> w;
      for (const auto& v : body_polygon)   // Pose::transform, pose.h:40-46: x + rx cos - ry sin, in that order
        w.push_back(DpPoint2{tp[1] + v.x * c - v.y * s, tp[2] + v.x * s + v.y * c});
      d.time.push_back(tp[0]);
      d.poly.push_back(MakePoly(w));
    }
    if (!d.time.empty()) dynamics_.push_back(std::move(d));
  }

  // the same obstacle given as the reference's Environment holds it: one world-frame polygon per trajectory sample
  void AddDynamicPlaced(const std::vector<double>& times, const std::vector<std::vector<DpPoint2>>& world_polygons) {
    Dynamic d;
    for (size_t i = 0; i < times.size() && i < world_polygons.size(); ++i) {
      d.time.push_back(times[i]);
      d.poly.push_back(MakePoly(world_polygons[i]));
    }
    if (!d.time.empty()) dynamics_.push_back(std::move(d));
  }

  // environment.cpp:92-111 (collision_buffer = 0)
  bool CheckOptimizationCollision(double time, double x, double y, double theta) const {
    const double ct = std::cos(theta), st = std::sin(theta);
    const double ax = x + f2x_ * ct, ay = y + f2x_ * st;   // vehicle_param.h:88-95
    const double bx = x + r2x_ * ct, by = y + r2x_ * st;
    return StaticCollision(bx, by) || StaticCollision(ax, ay) || DynamicCollision(time, bx, by) ||
           DynamicCollision(time, ax, ay);
  }

 private:
  static Poly MakePoly(const std::vector<DpPoint2>& p) {
    Poly q;
    q.pts = p;
    q.min_x = q.max_x = p[0].x;
    q.min_y = q.max_y = p[0].y;
    for (const auto& v : p) {
      q.min_x = std::min(q.min_x, v.x);
      q.max_x = std::max(q.max_x, v.x);
      q.min_y = std::min(q.min_y, v.y);
      q.max_y = std::max(q.max_y, v.y);
    }
    return q;
  }

  // axis-aligned square of half side radius_ centred at (cx, cy), as Box2d(AABox2d) (box2d.cpp:93-105)
  struct Square {
    double cx, cy, h, min_x, max_x, min_y, max_y;
  };
  Square MakeSquare(double cx, double cy) const {
    return Square{cx, cy, radius_, cx - radius_, cx + radius_, cy - radius_, cy + radius_};
  }
  static bool SquareHasPoint(const Square& b, const DpPoint2& p) {   // Box2d::IsPointIn, heading 0 (box2d.cpp:123-129)
    const double x0 = p.x - b.cx, y0 = p.y - b.cy;
    const double dx = std::abs(x0 * 1.0 + y0 * 0.0);
    const double dy = std::abs(-x0 * 0.0 + y0 * 1.0);
    return dx <= b.h + dp_detail::kGeomEps && dy <= b.h + dp_detail::kGeomEps;
  }
  static bool PolyHasPoint(const Poly& q, double px, double py) {   // Polygon2d::IsPointIn, polygon2d.cpp:120-140
    if (px < q.min_x || px > q.max_x || py < q.min_y || py > q.max_y) return false;
    const int n = (int)q.pts.size();
    int j = n - 1, c = 0;
    for (int i = 0; i < n; ++i) {
      if ((q.pts[i].y > py) != (q.pts[j].y > py)) {
        // CrossProd(point, points_[i], points_[j]) = (pi - p) x (pj - p)
        const double side = (q.pts[i].x - px) * (q.pts[j].y - py) - (q.pts[j].x - px) * (q.pts[i].y - py);
        if (q.pts[i].y < q.pts[j].y ? side > 0.0 : side < 0.0) ++c;
      }
      j = i;
    }
    return (c & 1) != 0;
  }
  static bool Overlap(const Poly& q, const Square& b) {   // Polygon2d::HasOverlap(Box2d), polygon2d.cpp:150-164
    if (b.max_x < q.min_x || b.min_x > q.max_x || b.max_y < q.min_y || b.min_y > q.max_y) return false;
    for (const auto& p : q.pts)
      if (SquareHasPoint(b, p)) return true;
    // AABox2d::GetAllCorners order (aabox2d.cpp:63-71)
    return PolyHasPoint(q, b.cx + b.h, b.cy - b.h) || PolyHasPoint(q, b.cx + b.h, b.cy + b.h) ||
           PolyHasPoint(q, b.cx - b.h, b.cy + b.h) || PolyHasPoint(q, b.cx - b.h, b.cy - b.h);
  }

  bool StaticCollision(double cx, double cy) const {   // environment.cpp:45-80
    const Square b = MakeSquare(cx, cy);
    for (const auto& q : statics_)
      if (Overlap(q, b)) return true;
    if (barrier_.empty()) return false;
    if (b.max_x < barrier_.front().x || b.min_x > barrier_.back().x) return false;
    auto upper = [&](double val) {   // first barrier point with val < point.x
      size_t lo = 0, hi = barrier_.size();
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (val < barrier_[mid].x) hi = mid;
        else lo = mid + 1;
      }
      return lo;
    };
    size_t first = upper(b.min_x);
    const size_t last = upper(b.max_x);
    if (first > 0) --first;
    for (size_t i = first; i < last; ++i)
      if (SquareHasPoint(b, barrier_[i])) return true;
    return false;
  }

  bool DynamicCollision(double time, double cx, double cy) const {   // environment.cpp:113-130
    const Square b = MakeSquare(cx, cy);
    for (const auto& d : dynamics_) {
      if (d.time.front() > time || d.time.back() < time) continue;
      size_t i = 0;   // first sample with time < sample time (std::upper_bound)
      while (i < d.time.size() && !(time < d.time[i])) ++i;
      if (i >= d.time.size()) i = d.time.size() - 1;   // the reference dereferences end() here
      if (Overlap(d.poly[i], b)) return true;
    }
    return false;
  }

  DpConfig cfg_;
  ReferenceLine ref_;
  double radius_ = 0.0, r2x_ = 0.0, f2x_ = 0.0;
  std::vector<DpPoint2> barrier_;
  std::vector<Poly> statics_;
  std::vector<Dynamic> dynamics_;
};

// ---- ComputePathProfile, discrete_points_math.cc:27-176 ---------------------------------------------------------
// Heading, arc length, speed, acceleration and curvature of a polyline sampled every `dt` seconds.  Every finite difference
// of the reference is one operator -- "value at the upper neighbour minus value at the lower neighbour", the neighbours clamped
// to the ends of the line -- applied to the points (heading, first derivative over s) and to its own output (second
// derivative); written as that operator here, the operands and the order of every operation as in the reference, so the
// results are the same bits (tests/test_reference_pins.py holds the checker against the reference's own function,
// tests/test_dp_planner.py this one against the checker).
namespace dp_detail {
struct Span {            // the neighbours a knot's differences are taken over
  std::size_t lo, hi;
  bool two_sided() const { return hi - lo == 2; }
};
inline Span SpanAt(std::size_t i, std::size_t n) { return Span{i > 0 ? i - 1 : 0, i + 1 < n ? i + 1 : n - 1}; }
}  // namespace dp_detail

inline bool ComputePathProfile(double dt, const std::vector<std::pair<double, double>>& xy, std::vector<double>* headings,
                               std::vector<double>* accumulated_s, std::vector<double>* speeds,
                               std::vector<double>* accelerations, std::vector<double>* kappas) {
  const std::size_t n = xy.size() >= 2 ? xy.size() : 0;
  for (std::vector<double>* out : {headings, accumulated_s, speeds, accelerations, kappas}) out->assign(n, 0.0);
  if (n == 0) return false;
  std::vector<double>& s = *accumulated_s;
  std::vector<double>& v = *speeds;
  std::vector<double>& a = *accelerations;

  // chord direction over the span (halved where the span is two-sided, before atan2 sees it) and the running chord length
  for (std::size_t i = 0; i < n; ++i) {
    const dp_detail::Span sp = dp_detail::SpanAt(i, n);
    const double scale = sp.two_sided() ? 0.5 : 1.0;
    const double cx = xy[sp.hi].first - xy[sp.lo].first, cy = xy[sp.hi].second - xy[sp.lo].second;
    (*headings)[i] = std::atan2(scale * cy, scale * cx);   // a factor of one changes no bit
    if (i > 0) {
      const double ex = xy[i - 1].first - xy[i].first, ey = xy[i - 1].second - xy[i].second;
      s[i] = std::sqrt(ex * ex + ey * ey) + s[i - 1];
    }
  }
  // forward differences in time; the last knot repeats its predecessor
  for (std::size_t i = 0; i + 1 < n; ++i) v[i] = (s[i + 1] - s[i]) / dt;
  v[n - 1] = v[n - 2];
  for (std::size_t i = 0; i + 1 < n; ++i) a[i] = (v[i + 1] - v[i]) / dt;
  a[n - 1] = a[n - 2];

  // derivatives with respect to arc length: d/ds of a sampled pair (p, q) = span difference over the span's arc length
  struct Pair { double p, q; };
  std::vector<Pair> d1(n), d2(n);
  auto over_s = [&](auto&& sample, std::size_t i) {
    const dp_detail::Span sp = dp_detail::SpanAt(i, n);
    const Pair hi = sample(sp.hi), lo = sample(sp.lo);
    return Pair{(hi.p - lo.p) / (s[sp.hi] - s[sp.lo]), (hi.q - lo.q) / (s[sp.hi] - s[sp.lo])};
  };
  for (std::size_t i = 0; i < n; ++i) d1[i] = over_s([&](std::size_t k) { return Pair{xy[k].first, xy[k].second}; }, i);
  for (std::size_t i = 0; i < n; ++i) d2[i] = over_s([&](std::size_t k) { return d1[k]; }, i);
  for (std::size_t i = 0; i < n; ++i) {
    const double g2 = d1[i].p * d1[i].p + d1[i].q * d1[i].q;
    (*kappas)[i] = (d1[i].p * d2[i].q - d1[i].q * d2[i].p) / (std::sqrt(g2) * g2 + 1e-6);
  }
  return true;
}

// ---- the DP ------------------------------------------------------------------------------------------------------
class DpPlanner {
 public:
  DpPlanner(const DpConfig& config, const DpEnvironment* env) : env_(env), cfg_(config), unit_time_(config.tf / kDpNT) {
    time_ = dp_detail::LinSpaced<kDpNT>(unit_time_, cfg_.tf);                          // cpp:29-34
    station_ = dp_detail::LinSpaced<kDpNS>(0, unit_time_ * cfg_.max_velocity);
    lateral_ = dp_detail::LinSpaced<kDpNL - 1>(0, 1);
    safe_margin_ = cfg_.width / 2 * 1.5;
    for (int t = 0; t < kDpNT; ++t) nseg_[t] = CountSegmentPoints(t);
  }

  // false = no collision-free path (min cost >= dp_w_obstacle); `result` is filled either way, as in the reference.
  // Three passes over the lattice: relax every layer from the one before (Relax), pick the cheapest leaf and walk the
  // parent links back (Backtrack), sample the chosen polyline at the knot times and profile it (Sample).
  bool Plan(double start_x, double start_y, double start_theta, std::vector<CoarsePoint>* result) {
    (void)start_theta;
    const DpPoint2 origin = env_->reference().GetProjection(start_x, start_y);
    start_s_ = origin.x;
    start_l_ = origin.y;
    Relax();
    Node chosen[kDpNT];
    const double best = Backtrack(chosen);
    Sample(chosen, result);
    return best < cfg_.dp_w_obstacle;
  }

  int segment_points(int layer) const { return nseg_[layer]; }

 private:
  struct Cell {
    double cost = std::numeric_limits<double>::max();
    double current_s = std::numeric_limits<double>::min();
    int parent_s = -1, parent_l = -1;
  };
  struct Node {   // a cell on the chosen path
    int s, l;
    Cell cell;
  };
  // What a transition needs to know about the cell it starts from -- the same for all kDpNS x kDpNL children, so it is
  // worked out once per parent instead of once per (parent, child) pair.  No parent (first layer): the start state.
  struct Origin {
    int layer = -1, l_index = -1;   // l_index: lateral sample of the parent, -1 = start state
    double s = 0, l = 0;            // where the new segment starts: the parent's station and lateral offset
    double before_s = 0, before_l = 0;   // where the parent's own segment started (its parent, or the start state)
    double tail_s = 0, tail_l = 0;  // last sampled point of the parent's own segment: the heading of the first new point hangs on it
    double time = 0;                // time of the parent's layer
    double cost = 0;                // accumulated cost of the parent
  };

  int CountSegmentPoints(int cur_t) const {   // the counting loop of InterpolateLinearly, cpp:287-299
    int nseg = 0;
    for (double t = 0.0; t < cfg_.tf + cfg_.delta_t - dp_detail::kGeomEps; t += cfg_.delta_t) {
      if (cur_t == 0) {
        if (t > 0.0 - dp_detail::kDpEps && t < unit_time_ + dp_detail::kDpEps) ++nseg;
      } else {
        if (t > time_[cur_t] - unit_time_ + dp_detail::kGeomEps && t < time_[cur_t] + dp_detail::kGeomEps) ++nseg;
      }
    }
    return nseg;
  }

  double LateralAt(double s, int l_ind) const {   // dp_planner.h:84-92
    if (l_ind == kDpNL - 1) return 0.0;
    const RefPoint ref = env_->reference().EvaluateStation(s);
    const double lb = -ref.right_bound + safe_margin_;
    const double ub = ref.left_bound - safe_margin_;
    return lb + (ub - lb) * lateral_[l_ind];
  }

  Origin StartOrigin() const {
    Origin o;
    o.s = o.before_s = o.tail_s = start_s_;
    o.l = o.before_l = o.tail_l = start_l_;
    return o;
  }
  // cell (t, si, li) as the origin of the next layer's transitions (cpp:45-61 and cpp:92-105 for one parent)
  Origin OriginOf(int t, int si, int li) const {
    const Cell& cell = cells_[t][si][li];
    Origin o;
    o.layer = t;
    o.l_index = li;
    o.cost = cell.cost;
    o.time = time_[t];
    o.s = cell.current_s;
    o.l = LateralAt(o.s, li);
    o.before_s = start_s_;
    o.before_l = start_l_;
    if (t >= 1) {
      o.before_s = cells_[t - 1][cell.parent_s][cell.parent_l].current_s;
      o.before_l = LateralAt(o.before_s, cell.parent_l);
    }
    // the parent's own segment ran from (before_s, before_l) to (before_s + station, its lateral there) in nseg_[t] samples;
    // its last sample is one step short of the end
    const int n_own = nseg_[t];
    const double end_s = o.before_s + station_[si];
    const double end_l = LateralAt(end_s, li);
    const double step_s = station_[si] / n_own;
    const double step_l = (end_l - o.before_l) / n_own;
    o.tail_s = o.before_s + (n_own - 1) * step_s;
    o.tail_l = o.before_l + (n_own - 1) * step_l;
    return o;
  }

  // Does the sampled segment from `from` to (end_s, end_l) leave the road or hit something (cpp:63-85)?
  bool SegmentBlocked(const Origin& from, int layer, int si, double end_l) const {
    const ReferenceLine& ref = env_->reference();
    const int n = nseg_[layer];
    const double step_s = station_[si] / n;
    const double step_l = (end_l - from.l) / n;
    double seen_s = from.tail_s, seen_l = from.tail_l;
    for (int i = 0; i < n; ++i) {
      const double at_s = from.s + i * step_s, at_l = from.l + i * step_l;
      const double rise = at_l - seen_l;
      const double run = std::max(at_s - seen_s, dp_detail::kDpEps);
      seen_l = at_l;
      seen_s = at_s;
      const RefPoint r = ref.EvaluateStation(at_s);
      const double lo = std::min(0.0, -r.right_bound + safe_margin_);
      const double hi = std::max(0.0, r.left_bound - safe_margin_);
      if (at_l < lo - dp_detail::kDpEps || at_l > hi + dp_detail::kDpEps) return true;
      // GetCartesian(at_s, at_l) evaluates the same station: same RefPoint
      const double cx = r.x - at_l * std::sin(r.theta), cy = r.y + at_l * std::cos(r.theta);
      const double heading = r.theta + std::atan((rise / run) / (1 - r.kappa * at_l));
      const double when = from.time + i * (unit_time_ / n);
      if (env_->CheckOptimizationCollision(when, cx, cy, heading)) return true;
    }
    return false;
  }

  // cost of going from `from` to sample (si, li) of `layer`, and the station reached (cpp:88-133).  The five smoothness
  // terms keep the reference's order of summation (the sum decides ties between parents).
  double Transition(const Origin& from, int layer, int si, int li, double* reached_s) const {
    const double to_s = from.s + station_[si];
    const double to_l = LateralAt(to_s, li);
    *reached_s = to_s;
    if (SegmentBlocked(from, layer, si, to_l)) return cfg_.dp_w_obstacle;
    const double advance = to_s - from.s, advance_before = from.s - from.before_s;
    const double shift = to_l - from.l, shift_before = from.l - from.before_l;
    const double off_centre = std::fabs(to_l);
    const double slope = std::fabs(from.l - to_l) / (station_[si] + dp_detail::kDpEps);
    const double lateral_rate_jump = std::fabs(shift - shift_before) / unit_time_;
    const double speed_error = std::fabs(advance / unit_time_ - cfg_.dp_nominal_velocity);
    const double speed_jump = std::fabs((advance - advance_before) / unit_time_);
    return (cfg_.dp_w_lateral * off_centre + cfg_.dp_w_lateral_change * slope + cfg_.dp_w_lateral_velocity_change * lateral_rate_jump +
            cfg_.dp_w_longitudinal_velocity_bias * speed_error + cfg_.dp_w_longitudinal_velocity_change * speed_jump);
  }

  // all children of one origin; a child keeps the first parent that reaches it most cheaply (strict <, cpp:176-178)
  void Expand(const Origin& from, int layer, int parent_s, int parent_l) {
    for (int si = 0; si < kDpNS; ++si)
      for (int li = 0; li < kDpNL; ++li) {
        double reached = 0.0;
        const double step_cost = Transition(from, layer, si, li, &reached);
        Cell& child = cells_[layer][si][li];
        if (from.layer < 0) {           // first layer: assigned, not compared (cpp:155-161)
          child.current_s = reached;
          child.cost = step_cost;
          continue;
        }
        const double total = from.cost + step_cost;
        if (total < child.cost) {
          child.cost = total;
          child.current_s = reached;
          child.parent_s = parent_s;
          child.parent_l = parent_l;
        }
      }
  }

  void Relax() {                         // cpp:144-184
    for (auto& layer : cells_)
      for (auto& row : layer)
        for (auto& c : row) c = Cell();
    Expand(StartOrigin(), 0, -1, -1);
    for (int t = 0; t + 1 < kDpNT; ++t)
      for (int si = 0; si < kDpNS; ++si)
        for (int li = 0; li < kDpNL; ++li) Expand(OriginOf(t, si, li), t + 1, si, li);
  }

  // cheapest cell of the last layer (first one on ties, cpp:187-198) and its ancestors; returns its cost
  double Backtrack(Node* chosen) const {
    double best = std::numeric_limits<double>::max();
    int at_s = 0, at_l = 0;
    for (int si = 0; si < kDpNS; ++si)
      for (int li = 0; li < kDpNL; ++li)
        if (cells_[kDpNT - 1][si][li].cost < best) {
          at_s = si;
          at_l = li;
          best = cells_[kDpNT - 1][si][li].cost;
        }
    for (int t = kDpNT - 1; t >= 0; --t) {   // every cell of a layer >= 1 has a parent (its first candidate always improves
      chosen[t] = Node{at_s, at_l, cells_[t][at_s][at_l]};   // on the initial cost), so the indices stay valid down to layer 0
      at_s = chosen[t].cell.parent_s;
      at_l = chosen[t].cell.parent_l;
    }
    return best;
  }

  // the samples of one segment of the chosen path (cpp:283-320)
  std::vector<DpPoint2> SegmentSamples(double from_s, int from_l_index, int layer, int si, int li) const {
    const int n = nseg_[layer];
    std::vector<DpPoint2> pts(n);
    double p_s = start_s_, p_l = start_l_;
    if (from_l_index >= 0) {
      p_s = from_s;
      p_l = LateralAt(p_s, from_l_index);
    }
    const double end_s = p_s + station_[si];
    const double end_l = LateralAt(end_s, li);
    const double step_s = station_[si] / n;
    const double step_l = (end_l - p_l) / n;
    for (int i = 0; i < n; ++i) pts[i] = DpPoint2{p_s + i * step_s, p_l + i * step_l};
    return pts;
  }

  // knots of the coarse trajectory along the chosen path, then speed / acceleration / curvature from the points (cpp:217-275)
  void Sample(const Node* chosen, std::vector<CoarsePoint>* result) const {
    const ReferenceLine& ref = env_->reference();
    const size_t n_knots = (size_t)(cfg_.tf / cfg_.delta_t + 1);
    std::vector<CoarsePoint> knots(n_knots);
    std::vector<std::pair<double, double>> xy;
    double seen_l = start_l_, seen_s = start_s_;
    size_t k = 0;
    for (int t = 0; t < kDpNT; ++t) {
      const double from_s = t > 0 ? chosen[t - 1].cell.current_s : start_s_;
      for (const DpPoint2& q : SegmentSamples(from_s, chosen[t].cell.parent_l, t, chosen[t].s, chosen[t].l)) {
        const double rise = q.y - seen_l;
        const double run = std::max(q.x - seen_s, dp_detail::kDpEps);
        seen_l = q.y;
        seen_s = q.x;
        const DpPoint2 p = ref.GetCartesian(q.x, q.y);
        const RefPoint r = ref.EvaluateStation(q.x);
        if (k < n_knots) {
          knots[k].time = cfg_.delta_t * k;
          knots[k].s = q.x;
          knots[k].x = p.x;
          knots[k].y = p.y;
          knots[k].theta = r.theta + std::atan((rise / run) / (1 - r.kappa * q.y));
        }
        xy.emplace_back(p.x, p.y);
        ++k;
      }
    }
    std::vector<double> headings, stations, speeds, accels, kappas;
    ComputePathProfile(cfg_.delta_t, xy, &headings, &stations, &speeds, &accels, &kappas);
    for (size_t i = 0; i < xy.size() && i < n_knots; ++i) {
      knots[i].kappa = kappas[i];
      knots[i].delta = std::atan(knots[i].kappa * cfg_.wheel_base);
      knots[i].velocity = speeds[i];
      knots[i].a = accels[i];
    }
    *result = std::move(knots);
  }

  const DpEnvironment* env_;
  DpConfig cfg_;
  double unit_time_;
  std::array<double, kDpNT> time_;
  std::array<double, kDpNS> station_;
  std::array<double, kDpNL - 1> lateral_;
  double safe_margin_ = 0.0;
  int nseg_[kDpNT];
  double start_s_ = 0.0, start_l_ = 0.0;
  Cell cells_[kDpNT][kDpNS][kDpNL];
};

}  // namespace cilqr

#endif  // CILQR_DP_PLANNER_HPP_
