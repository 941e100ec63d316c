// TEST-ONLY arithmetic (compiled with -DCILQR_REF_ORDER into lib/libcilqr_hip_reforder.so, never into the product
// library): the cost and quadratisation kernels evaluated in the REFERENCE's operation order -- one barrier value
// (one library log) per constraint, running sums over knots / discs / planes exactly as the loops of
// ilqr_optimizer.cc:497-603 and :620-769 nest, the nearest lane segment by the reference's DistanceTo arithmetic
// (hypot / fabs, line_segment2d.cpp:61-75) over the full scan, IEEE divisions and library sin / cos / tan / log.
// What is left between this build and the CPU oracle is the difference between the device's libm and glibc.
// tests/parity_report.py runs both builds against the oracle: the share of problems that differ from the oracle
// in THIS build is what the reference's own ill-conditioning costs any implementation with another libm; the
// product build's extra share is what its re-associations (grouped logs, per-plane sums, squared distances) add.
#pragma once
#ifdef CILQR_REF_ORDER

#include "dev_model.hpp"

namespace cilqr {
namespace reforder {

CILQR_DEV double bar_value(const Params& p, double x) {   // barrier_function.h:104-113
  if (x < -p.bar_eps) return -p.bar_r * log(-x);
  const double q = (-x - 2.0 * p.bar_eps) / p.bar_eps;
  return 0.5 * p.bar_r * (q * q - 1) - p.bar_rlogeps;
}
CILQR_DEV double bar_jac(const Params& p, double x) {     // :115-125
  if (x < -p.bar_eps) return -p.bar_r / x;
  return p.bar_r * (x + 2.0 * p.bar_eps) / p.bar_eps / p.bar_eps;
}
// Hessian coefficients (:127-140): out(i,j) = (c1 d_i) d_j - c2 ddx(i,j); relaxed branch: c2 term absent
CILQR_DEV void bar_hess(const Params& p, double x, double& c1, double& c2, bool& lg) {
  lg = x < -p.bar_eps;
  if (lg) {
    c1 = p.bar_r / x / x;
    c2 = p.bar_r / x;
  } else {
    c1 = p.bar_r * (x + 2.0 * p.bar_eps) / p.bar_eps / p.bar_eps;
    c2 = 0.0;
  }
}

// LineSegment2d::DistanceTo, line_segment2d.cpp:61-75 (rows: a b c | sx sy | ux uy | len | ex ey)
CILQR_DEV double seg_distance(const double* __restrict__ r, double px, double py) {
  const double sx = r[3], sy = r[4], ux = r[5], uy = r[6], len = r[7], ex = r[8], ey = r[9];
  if (len <= kMathEps) return hypot(px - sx, py - sy);
  const double x0 = px - sx, y0 = py - sy;
  const double proj = x0 * ux + y0 * uy;
  if (proj <= 0.0) return hypot(x0, y0);
  if (proj >= len) return hypot(px - ex, py - ey);
  return fabs(x0 * uy - y0 * ux);
}
CILQR_DEV const double* nearest(const double* __restrict__ tab, int n, double px, double py) {   // cc:605-618
  double min_dis = DBL_MAX;
  int idx = 0;
  for (int i = 0; i < n; ++i) {
    const double d = seg_distance(tab + i * kLaneFields, px, py);
    if (d < min_dis) {
      min_dis = d;
      idx = i;
    }
  }
  return tab + idx * kLaneFields;
}

// TotalCost (cc:417-436) of one trajectory; LX(i, x[6]) / LU(i, u[2]) fetch knot i
template <class LX, class LU>
CILQR_DEV void total_cost(const DeviceState& s, int slot, LX load_x, LU load_u, double* c5) {
  const Params& p = s.p;
  const int K = p.K, N = p.N, Bc = s.Bcap;
  const double* lanes = s.lanes;
  double x[6], u[2];
  double j_cost = 0.0;                                              // JCost cc:497-516
  for (int i = 0; i < K; ++i) {
    load_x(i, x);
    const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
    const double2 g0 = gp[0];
    const double gth = gp[(size_t)Bc].x;
    const double dx = x[0] - g0.x, dy = x[1] - g0.y, dth = x[2] - gth;
    j_cost += p.w_x * (dx * dx) + p.w_y * (dy * dy) + p.w_theta * (dth * dth);
  }
  for (int i = 0; i < N; ++i) {
    load_u(i, u);
    j_cost += p.w_jerk * (u[0] * u[0]) + p.w_delta_rate * (u[1] * u[1]);
  }
  double x_cost = 0.0;                                              // DynamicsCost cc:518-551
  for (int i = 0; i < K; ++i) {
    load_x(i, x);
    x_cost += bar_value(p, -x[3]);
    x_cost += bar_value(p, x[3] - p.max_velocity);
    x_cost += bar_value(p, x[4] - p.max_acc);
    x_cost += bar_value(p, p.min_acc - x[4]);
    x_cost += bar_value(p, x[5] - p.delta_max);
    x_cost += bar_value(p, p.delta_min - x[5]);
  }
  double u_cost = 0.0;
  for (int i = 0; i < N; ++i) {
    load_u(i, u);
    u_cost += bar_value(p, u[0] - p.jerk_max);
    u_cost += bar_value(p, p.jerk_min - u[0]);
    u_cost += bar_value(p, u[1] - p.delta_rate_max);
    u_cost += bar_value(p, p.delta_rate_min - u[1]);
  }
  const double dyn_cost = x_cost + u_cost;
  double cor_cost = 0.0;                                            // CorridorCost cc:553-581
  for (int i = 0; i < K; ++i) {
    load_x(i, x);
    const int cnt = s.ccnt[(size_t)i * Bc + slot];
    const double* cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
    for (int j = 0; j < p.num_of_disc; ++j) {
      const double px = x[0] + p.disc_off[j] * cos(x[2]);
      const double py = x[1] + p.disc_off[j] * sin(x[2]);
      for (int c = 0; c < cnt; ++c) {
        const double* q = cor + (size_t)c * 3 * Bc;
        cor_cost += bar_value(p, q[0] * px + q[(size_t)Bc] * py - q[(size_t)2 * Bc]);
      }
    }
  }
  double lane_cost = 0.0;                                           // LaneBoundaryCost cc:583-603
  for (int i = 0; i < K; ++i) {
    load_x(i, x);
    for (int j = 0; j < p.num_of_disc; ++j) {
      const double px = x[0] + p.disc_off[j] * cos(x[2]);
      const double py = x[1] + p.disc_off[j] * sin(x[2]);
      const double* l = nearest(lanes, s.nl, px, py);
      lane_cost += bar_value(p, l[0] * px + l[1] * py - l[2]);
      const double* r = nearest(lanes + s.nl * kLaneFields, s.nr, px, py);
      lane_cost += bar_value(p, r[0] * px + r[1] * py - r[2]);
    }
  }
  c5[0] = j_cost + dyn_cost + cor_cost + lane_cost;
  c5[1] = j_cost; c5[2] = dyn_cost; c5[3] = cor_cost; c5[4] = lane_cost;
}

// CostJacbian cc:620-636 (+ :657-671, :690-706, :729-746) and CostHessian cc:638-655 (+ :673-688, :708-727, :748-769)
// of knot i; lx[6], lu[2], h[9] = lxx rows / cols 0..2, hd[3] = lxx(3,3), (4,4), (5,5), huu[2]
CILQR_DEV void knot_quadratize(const DeviceState& s, int i, int slot, const double* x, const double* u, double* lx,
                               double* lu, double* h, double* hd, double* huu) {
  const Params& p = s.p;
  const int Bc = s.Bcap;
  const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
  const double2 g0 = gp[0];
  const double gth = gp[(size_t)Bc].x;
  lx[0] = 2.0 * p.w_x * (x[0] - g0.x);
  lx[1] = 2.0 * p.w_y * (x[1] - g0.y);
  lx[2] = 2.0 * p.w_theta * (x[2] - gth);
  lx[3] = 0.0; lx[4] = 0.0; lx[5] = 0.0;
  lu[0] = 2.0 * p.w_jerk * u[0];
  lu[1] = 2.0 * p.w_delta_rate * u[1];
  for (int e = 0; e < 9; ++e) h[e] = 0.0;
  h[0] = 2.0 * p.w_x; h[4] = 2.0 * p.w_y; h[8] = 2.0 * p.w_theta;
  hd[0] = 2.0 * p.w_v; hd[1] = 2.0 * p.w_a; hd[2] = 2.0 * p.w_delta;
  huu[0] = 2.0 * p.w_jerk; huu[1] = 2.0 * p.w_delta_rate;
  {  // bounds (cc:657-688): the gradient vectors / Hessian matrices of the constraints are summed first, then added.
     // A constraint touches one component (d = +-1 there, 0 elsewhere): the other constraints add coefficient * 0.0
     // = +-0 to it, which changes no sum, so each component is the sum of its two terms in the reference's order
     // (lower bound, upper bound); (c1 d_i) d_j = c1 on the diagonal.
    const double g[6] = {0.0 - x[3], x[3] - p.max_velocity, p.min_acc - x[4], x[4] - p.max_acc, p.delta_min - x[5],
                         x[5] - p.delta_max};
    for (int e = 0; e < 3; ++e) {
      lx[3 + e] += bar_jac(p, g[2 * e]) * -1.0 + bar_jac(p, g[2 * e + 1]) * 1.0;
      double c1a, c1b, c2;
      bool lg;
      bar_hess(p, g[2 * e], c1a, c2, lg);
      bar_hess(p, g[2 * e + 1], c1b, c2, lg);
      hd[e] += (c1a * -1.0) * -1.0 + (c1b * 1.0) * 1.0;
    }
    const double gu[4] = {p.jerk_min - u[0], u[0] - p.jerk_max, p.delta_rate_min - u[1], u[1] - p.delta_rate_max};
    for (int e = 0; e < 2; ++e) {
      lu[e] += bar_jac(p, gu[2 * e]) * -1.0 + bar_jac(p, gu[2 * e + 1]) * 1.0;
      double c1a, c1b, c2;
      bool lg;
      bar_hess(p, gu[2 * e], c1a, c2, lg);
      bar_hess(p, gu[2 * e + 1], c1b, c2, lg);
      huu[e] += (c1a * -1.0) * -1.0 + (c1b * 1.0) * 1.0;
    }
  }
  const int cnt = s.ccnt[(size_t)i * Bc + slot];
  const double* cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
  auto add_plane = [&](double a, double b, double c, double px, double py, double length_cos, double length_sin, bool jac,
                       bool hes) {
    const double gval = a * px + b * py - c;
    const double d[3] = {a, b, -a * length_sin + b * length_cos};
    if (jac) {
      const double coef = bar_jac(p, gval);
      for (int e = 0; e < 3; ++e) lx[e] += coef * d[e];
    }
    if (hes) {
      double c1, c2;
      bool lg;
      bar_hess(p, gval, c1, c2, lg);
      const double dd22 = -a * length_cos - b * length_sin;
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
          const double ddx = (r == 2 && q == 2) ? dd22 : 0.0;
          h[r * 3 + q] += lg ? (c1 * d[r]) * d[q] - c2 * ddx : (c1 * d[r]) * d[q];
        }
    }
  };
  // the reference runs the Jacobian passes (corridor, lanes) and then the Hessian passes; the sums of lx and of h
  // are independent, so one pass per constraint family in the same (disc, plane) order gives the same numbers
  for (int j = 0; j < p.num_of_disc; ++j) {                           // cc:690-706 / :708-727
    const double length_cos = p.disc_off[j] * cos(x[2]), length_sin = p.disc_off[j] * sin(x[2]);
    const double px = x[0] + length_cos, py = x[1] + length_sin;
    for (int c = 0; c < cnt; ++c) {
      const double* q = cor + (size_t)c * 3 * Bc;
      add_plane(q[0], q[(size_t)Bc], q[(size_t)2 * Bc], px, py, length_cos, length_sin, true, true);
    }
  }
  for (int j = 0; j < p.num_of_disc; ++j) {                           // cc:729-746 / :748-769
    const double length_cos = p.disc_off[j] * cos(x[2]), length_sin = p.disc_off[j] * sin(x[2]);
    const double px = x[0] + length_cos, py = x[1] + length_sin;
    const double* l = nearest(s.lanes, s.nl, px, py);
    const double* r = nearest(s.lanes + s.nl * kLaneFields, s.nr, px, py);
    add_plane(l[0], l[1], l[2], px, py, length_cos, length_sin, true, true);
    add_plane(r[0], r[1], r[2], px, py, length_cos, length_sin, true, true);
  }
}

}  // namespace reforder
}  // namespace cilqr
#endif  // CILQR_REF_ORDER
