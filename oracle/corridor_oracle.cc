// CPU ORACLE for the corridor producer -- TEST INFRASTRUCTURE ONLY.
//
// Scalar restatement of the reference's safe-corridor construction, the producer of the hot path's
// per-knot half-plane inputs:
//   Corridor::AddCorridorPoints     algorithm/ilqr/corridor.cc:89-120
//   Corridor::BuildCorridor         corridor.cc:122-263
//   Corridor::LaneBoundarySample    corridor.cc:298-311
//   Corridor::Cal{Left,Right}LaneConstraints, HalfPlaneConstraint   corridor.cc:265-296, 313-321
//
// PARITY UNPINNED.  BuildCorridor calls cv::convexHull (OpenCV imgproc, Sklansky's scan on
// float32 points) three times; OpenCV is not part of /root/reference nor of this image, and the
// reference holds no tests or golden vectors for this path.  The hull below is Andrew's monotone
// chain on float32 with collinear points dropped -- the same hull (strictly convex vertices,
// counter-clockwise with y up, or clockwise on request) whenever no three input points are
// collinear to float32 rounding; the START vertex of the returned cycle may differ from OpenCV's,
// which only permutes the order of the emitted half-planes.  Everything around the hulls follows
// the reference statement by statement, including its mixed float32 / float64 arithmetic and
// these quirks:
//   * safe_radius is the norm of the LAST filtered point closer than `radius`, not the minimum
//     (cc:169-171); it only matters when the origin is a hull vertex;
//   * last_index = (OriginIndex - 1) % size is evaluated in unsigned 64-bit arithmetic (cc:203), so
//     OriginIndex == 0 gives (2^64 - 1) % size, not size - 1;
//   * the flipped-point array has points.size() + 1 zero-initialised entries (cc:159); entries past
//     the filtered count are copies of the origin and can never be hull vertices -- they are not
//     materialised here.
//
// Only tests/, __graft_entry__.smoke() and bench.py may use this file; the product never links it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct P2f {
  float x, y;
};

constexpr double kMathEpsilon = 1e-10;  // algorithm/math/vec2d.h:33

// strictly convex hull of float32 points; returns indices into `p`, counter-clockwise (y up)
// starting at the lexicographically smallest point, or clockwise when asked
std::vector<int> hull_indices(const std::vector<P2f>& p, bool clockwise) {
  const int n = (int)p.size();
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return p[a].x < p[b].x || (p[a].x == p[b].x && p[a].y < p[b].y);
  });
  auto cross = [&](int o, int a, int b) {
    const float ax = p[a].x - p[o].x, ay = p[a].y - p[o].y;
    const float bx = p[b].x - p[o].x, by = p[b].y - p[o].y;
    const float t1 = ax * by, t2 = ay * bx;
    return t1 - t2;
  };
  std::vector<int> h(2 * n + 2);
  int k = 0;
  for (int i = 0; i < n; ++i) {  // lower chain
    while (k >= 2 && cross(h[k - 2], h[k - 1], order[i]) <= 0.0f) --k;
    h[k++] = order[i];
  }
  for (int i = n - 2, t = k + 1; i >= 0; --i) {  // upper chain
    while (k >= t && cross(h[k - 2], h[k - 1], order[i]) <= 0.0f) --k;
    h[k++] = order[i];
  }
  if (k > 1) --k;  // last point repeats the first
  h.resize(k);
  // duplicates of the start point can survive as a 2-cycle; collapse exact repeats
  if (h.size() == 2 && p[h[0]].x == p[h[1]].x && p[h[0]].y == p[h[1]].y) h.resize(1);
  if (clockwise) std::reverse(h.begin() + (h.empty() ? 0 : 1), h.end());
  return h;
}

}  // namespace

extern "C" {

// cfg: max_diff_x, max_diff_y, radius, max_axis_x, max_axis_y   (planner_config.h:75-86)
// pts: n obstacle points (x, y) valid at this knot's time (Environment::Query*ObstaclesPoints).
// Appends the box points (AddCorridorPoints: both ends of every edge, or six samples per edge when cfg[5] =
// is_multiple_sample is set) and builds the corridor.  cfg = max_diff_x, max_diff_y, radius, max_axis_x, max_axis_y,
// is_multiple_sample, trig override flag, cos(theta), sin(theta)  (9 entries; see the hook below).
// Outputs: cons[.][3] = (a, b, c) with a x + b y <= c; poly[.][2] the polygon vertices.
// Returns the number of half-planes, or -1 (no points), -2 (fewer than 4 flipped points), -3 (more
// than max_out half-planes), -4 (degenerate hull).
int oracle_build_corridor(double ox, double oy, double theta, const double* pts, int n, const double* cfg,
                          double* cons, double* poly, int max_out) {
  const double max_diff_x = cfg[0], max_diff_y = cfg[1], radius = cfg[2];
  const double max_axis_x = cfg[3], max_axis_y = cfg[4];
  std::vector<double> px, py;
  for (int i = 0; i < n; ++i) {
    px.push_back(pts[2 * i]);
    py.push_back(pts[2 * i + 1]);
  }
  {  // AddCorridorPoints cc:89-120
    // cfg[6] != 0 (test hook): cos / sin of theta handed in as cfg[7], cfg[8] -- the values of ANOTHER libm, to show that
    // a knot on which two implementations differ differs through the last bit of a box corner and nothing else
    const double ch = (cfg[6] != 0.0) ? cfg[7] : std::cos(theta), sh = (cfg[6] != 0.0) ? cfg[8] : std::sin(theta);
    const double dx1 = ch * max_axis_x, dy1 = sh * max_axis_x;
    const double dx2 = sh * max_axis_y, dy2 = -ch * max_axis_y;
    const double cx[4] = {ox + dx1 + dx2, ox + dx1 - dx2, ox - dx1 - dx2, ox - dx1 + dx2};
    const double cy[4] = {oy + dy1 + dy2, oy + dy1 - dy2, oy - dy1 - dy2, oy - dy1 + dy2};
    const double kSampleMultiple = (cfg[5] != 0.0) ? 5.0 : 1.0;   // cc:110, is_multiple_sample
    const double ratio_step = 1.0 / kSampleMultiple;
    for (int i = 0; i < 4; ++i) {
      const int nx = (i + 1) % 4;
      for (double ratio = 0.0; ratio < 1.0 + kMathEpsilon; ratio += ratio_step) {
        px.push_back(cx[i] * (1 - ratio) + cx[nx] * ratio);
        py.push_back(cy[i] * (1 - ratio) + cy[nx] * ratio);
      }
    }
  }
  if (px.empty()) return -1;
  // filter cc:136-149
  std::vector<double> fx, fy;
  for (size_t i = 0; i < px.size(); ++i) {
    const double dx = px[i] - ox, dy = py[i] - oy;
    if (std::fabs(dx) > max_diff_x || std::fabs(dy) > max_diff_y) continue;
    const double norm2 = std::sqrt(dx * dx + dy * dy);
    if (std::fabs(norm2) < kMathEpsilon) continue;
    fx.push_back(px[i]);
    fy.push_back(py[i]);
  }
  // sphere flip cc:154-177 (double expression, stored as float32)
  double safe_radius = radius;
  const int nf = (int)fx.size();
  std::vector<P2f> flip(nf + 1, P2f{0.0f, 0.0f});
  int sum = 0;
  for (int i = 0; i < nf; ++i) {
    const double dx = fx[i] - ox, dy = fy[i] - oy;
    const double norm2 = std::sqrt(dx * dx + dy * dy);
    if (norm2 < radius) safe_radius = norm2;
    flip[i].x = (float)(dx + 2 * (radius - norm2) * dx / norm2);
    flip[i].y = (float)(dy + 2 * (radius - norm2) * dy / norm2);
    ++sum;
  }
  if (sum < 4) return -2;
  const std::vector<int> v1 = hull_indices(flip, false);  // cc:184
  if (v1.size() < 3) return -4;
  // star-shaped polygon through the visible points cc:186-198
  bool origin_is_vertex = false;
  int origin_index = -1;
  std::vector<P2f> vd;
  for (size_t i = 0; i < v1.size(); ++i) {
    if (v1[i] == nf) {
      origin_is_vertex = true;
      origin_index = (int)i;
      vd.push_back(P2f{(float)ox, (float)oy});
    } else {
      vd.push_back(P2f{(float)fx[v1[i]], (float)fy[v1[i]]});
    }
  }
  double ix = ox, iy = oy;  // cc:200-216
  if (origin_is_vertex) {
    const uint64_t sz = v1.size();
    const int last = (int)(((uint64_t)(int64_t)(origin_index - 1)) % sz);
    const int next = (int)(((uint64_t)(int64_t)(origin_index + 1)) % sz);
    const int vl = v1[last], vn = v1[next];
    const double lx = (vl == nf) ? ox : fx[vl], ly = (vl == nf) ? oy : fy[vl];
    const double nx = (vn == nf) ? ox : fx[vn], ny = (vn == nf) ? oy : fy[vn];
    const double dx = (lx + ox + nx) / 3 - ox;
    const double dy = (ly + oy + ny) / 3 - oy;
    const double d = std::sqrt(dx * dx + dy * dy);
    ix = 0.99 * safe_radius * dx / d + ox;
    iy = 0.99 * safe_radius * dy / d + oy;
  }
  const std::vector<int> v2 = hull_indices(vd, false);  // cc:218
  if (v2.size() < 3) return -4;
  // one half-plane per star vertex, normal of the hull edge it hides behind  cc:220-233
  struct C3f { float a, b, c; };
  std::vector<C3f> temp;
  const int nv = (int)vd.size();
  for (size_t j = 0; j < v2.size(); ++j) {
    const size_t j1 = (j + 1) % v2.size();
    const float rx = vd[v2[j1]].x - vd[v2[j]].x, ry = vd[v2[j1]].y - vd[v2[j]].y;
    float n0 = ry, n1 = -rx;
    const float z = n0 * n0 + n1 * n1;  // Eigen normalize(): x / sqrt(squaredNorm) when > 0
    if (z > 0.0f) {
      const float s = std::sqrt(z);
      n0 = n0 / s;
      n1 = n1 / s;
    }
    int idx = v2[j];
    int guard = 0;
    while (idx != v2[j1] && guard++ <= nv) {
      const double c = (vd[idx].x - ix) * n0 + (vd[idx].y - iy) * n1;
      temp.push_back(C3f{n0, n1, (float)c});
      idx = (idx + 1) % nv;
    }
  }
  std::vector<P2f> dual(temp.size());  // cc:235-239
  for (size_t i = 0; i < temp.size(); ++i) {
    dual[i].x = temp[i].a / temp[i].c;
    dual[i].y = temp[i].b / temp[i].c;
  }
  const std::vector<int> v3 = hull_indices(dual, true);  // cc:241-242
  if (v3.size() < 3) return -4;
  const int m = (int)v3.size();
  if (m > max_out) return -3;
  std::vector<double> qx(m), qy(m);
  for (int i = 0; i < m; ++i) {  // cc:244-249
    const P2f a = dual[v3[i]], b = dual[v3[(i + 1) % m]];
    const float rx = b.x - a.x, ry = b.y - a.y;
    const float t1 = ry * a.x, t2 = rx * a.y;
    const double c = t1 - t2;
    qx[i] = ix + ry / c;
    qy[i] = iy - rx / c;
    if (poly) {
      poly[2 * i] = qx[i];
      poly[2 * i + 1] = qy[i];
    }
  }
  for (int i = 0; i < m; ++i) {  // cc:251-261
    const int i1 = (i + 1) % m;
    const double rx = qx[i1] - qx[i], ry = qy[i1] - qy[i];
    const double c = -ry * qx[i] + rx * qy[i];
    cons[3 * i] = -ry;
    cons[3 * i + 1] = rx;
    cons[3 * i + 2] = c;
  }
  return m;
}

// LaneBoundarySample cc:298-311 + CalLeft/RightLaneConstraints cc:265-296 + HalfPlaneConstraint
// cc:313-321.  boundary: n points (x, y) of one road barrier.  rows[.][7] = a b c sx sy ex ey in
// the layout of cilqr_problem_batch::left_lane / right_lane.  Returns the number of rows, -1 when
// fewer than two sampled points remain, -3 when max_rows is too small.
int oracle_lane_constraints(const double* boundary, int n, double segment_length, int is_left, double* rows,
                            int max_rows) {
  if (n < 1) return -1;
  std::vector<double> sx, sy;
  double lx = boundary[0], ly = boundary[1];
  sx.push_back(lx);
  sy.push_back(ly);
  for (int i = 0; i < n; ++i) {
    const double x = boundary[2 * i], y = boundary[2 * i + 1];
    if (std::hypot(x - lx, y - ly) >= segment_length - kMathEpsilon) {
      sx.push_back(x);
      sy.push_back(y);
      lx = x;
      ly = y;
    }
  }
  if (sx.size() < 2) return -1;
  const int m = (int)sx.size() - 1;
  if (m > max_rows) return -3;
  for (int i = 1; i <= m; ++i) {
    // left: segment and half-plane run from point i to point i-1; right: from i-1 to i
    const double ax = is_left ? sx[i] : sx[i - 1], ay = is_left ? sy[i] : sy[i - 1];
    const double bx = is_left ? sx[i - 1] : sx[i], by = is_left ? sy[i - 1] : sy[i];
    const double nx = bx - ax, ny = by - ay;
    const double a = ny, b = -nx;
    double* r = rows + 7 * (i - 1);
    r[0] = a; r[1] = b; r[2] = a * ax + b * ay;
    r[3] = ax; r[4] = ay; r[5] = bx; r[6] = by;
  }
  return m;
}

}  // extern "C"
