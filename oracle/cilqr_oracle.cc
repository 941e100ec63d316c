/*
 * oracle/cilqr_oracle.cc -- scalar-fp64 CPU restatement of the reference CILQR solve.
 *
 * TEST INFRASTRUCTURE ONLY (see cilqr_oracle.h).  PARITY UNPINNED for the solve as a whole: the reference
 * has no tests/golden vectors and ilqr_optimizer.cc / vehicle_model.cc / barrier_function.h cannot be compiled
 * here (Eigen 3.4 + ROS + OpenCV absent).  PINNED against the reference's own code, bit for bit, where that code
 * builds with g++ alone (oracle/_ref, oracle/ref_shim.cc; tests/test_reference_pins.py): NormalizeAngle
 * (math_utils.cpp:53-59), LineSegment2d::DistanceTo (line_segment2d.cpp:61-75) and the nearest-segment loop over it.
 *
 * Every function cites the reference file:line it follows ("cc" = algorithm/ilqr/
 * ilqr_optimizer.cc, "vm" = algorithm/ilqr/vehicle_model.cc, "bf" = algorithm/ilqr/
 * barrier_function.h).  Plain arrays replace Eigen; the Eigen semantics reproduced by hand:
 *  - matrix products associate left to right ((A^T Vxx) A), each nested product is a temporary;
 *  - the order in which the six terms of a dot product are added: SWITCH CILQR_DOT_ORDER below;
 *  - `dst = xpr + product` evaluates the whole right side before writing dst (assume-aliasing);
 *  - 2x2 inverse is the closed form invdet = 1/(m00 m11 - m10 m01);
 *  - the lazy `auto` expressions of Backward (cc:348-363) are re-evaluated at every use, so the
 *    delta_V_ updates at cc:383-384 see the ALREADY UPDATED Vx/Vxx: SWITCH CILQR_DV_EVAL below;
 *  - cc:381 symmetrises Vxx in place without a temporary (column-major traversal);
 *  - iqr's R (cc:811-813) has indeterminate off-diagonals; 0 is used.
 */
#include "cilqr_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

constexpr int NX = 6;  // vm.h:11
constexpr int NU = 2;  // vm.h:12
constexpr double kMathEpsilon = 1e-10;  // vec2d.h:33
constexpr int kTraceCols = 10;  // columns of the per-iteration trace, see cilqr_oracle.h

// math_utils.cpp:53-59
inline double NormalizeAngle(double angle) {
  double a = std::fmod(angle + M_PI, 2.0 * M_PI);
  if (a < 0.0) a += (2.0 * M_PI);
  return a - M_PI;
}

// line_segment2d.cpp:38-48 (ctor) and :61-75 (DistanceTo)
struct Segment {
  double sx, sy, ex, ey, len, ux, uy;
  void Set(double sx_, double sy_, double ex_, double ey_) {
    sx = sx_; sy = sy_; ex = ex_; ey = ey_;
    const double dx = ex - sx, dy = ey - sy;
    len = std::hypot(dx, dy);
    if (len <= kMathEpsilon) { ux = 0.0; uy = 0.0; }
    else { ux = dx / len; uy = dy / len; }
  }
  double DistanceTo(double px, double py) const {
    if (len <= kMathEpsilon) return std::hypot(px - sx, py - sy);
    const double x0 = px - sx, y0 = py - sy;
    const double proj = x0 * ux + y0 * uy;
    if (proj <= 0.0) return std::hypot(x0, y0);
    if (proj >= len) return std::hypot(px - ex, py - ey);
    return std::abs(x0 * uy - y0 * ux);
  }
};

struct Lane { double a, b, c; Segment seg; };

// ---------------------------------------------------------------------------------------------------------------
// Two assumptions about Eigen 3.4 that NOTHING in this image can check (no Eigen here), kept as switches.  Both have a
// compile-time default (-DCILQR_DOT_ORDER=n, -DCILQR_DV_EVAL_EAGER) and a run-time override (oracle_set_semantics) so
// that tests/semantics_report.py can run the variants side by side and count the solves each one moves.
//
// CILQR_DOT_ORDER -- how the six products of a 6-term dot product are added (2-term sums have one order):
//   0 "sequential"   ((((t0+t1)+t2)+t3)+t4)+t5 everywhere.  This oracle's default in rounds 1-4; now the variant the
//                    test-only product twin libcilqr_hip_dotseq.so is held against.
//   1 "eigen_redux"  (t0+(t1+t2)) + (t3+(t4+t5)) everywhere: a coefficient of a small fixed-size product is
//                    (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum() (Eigen/src/Core/ProductEvaluators.h,
//                    product_evaluator<..LazyProduct..>::coeff), and sum() of a fixed-size expression unrolls through
//                    redux_novec_unroller, which halves the range (Eigen/src/Core/Redux.h).  What an Eigen build
//                    WITHOUT vectorisation (EIGEN_DONT_VECTORIZE, or a target without SIMD) computes.
//   2 "eigen_sse2"   what the same headers do with 2-double packets (x86-64's baseline SSE2; also NEON), read off the
//                    evaluator flags:
//                      * lhs stored the other way round -- every `X.transpose() * Y` of Backward / iqr (A^T Vx, B^T Vx,
//                        A^T Vxx, B^T Vxx, B^T P, A^T P): the product has no PacketAccessBit, each coefficient goes
//                        through coeff(), and the redux itself is vectorised (both operands are contiguous columns):
//                        packets p_j = (t_2j, t_2j+1), P = p0 + (p1 + p2), result = P[0] + P[1]
//                        = (t0 + (t2 + t4)) + (t1 + (t3 + t5));
//                      * lhs a column-major temporary or matrix -- (B^T Vxx) A, (B^T Vxx) B, (A^T Vxx) A, K (x - x_ref),
//                        (A^T P)(A - B K) ...: CanVectorizeLhs, the assignment runs packet(row, col) =
//                        etor_product_packet_impl: res = pmul(l0, r0); res = pmadd(l_k, r_k, res), k = 1..5, and pmadd
//                        without FMA is padd(pmul(a, b), c): sequential.
//                    Best reading of the default x86-64 -O2 build of CMakeLists.txt:9; not executable here.
//                    THE DEFAULT since round 5, and what the product kernels implement (cilqr_amd/csrc/dev_model.hpp:
//                    sum6_xty; backward_core.hpp, kernels_load.hip).
//
// CILQR_DV_EVAL -- cc:348-352 declare Qu, Quu with `auto`: lazy expressions that hold references to Vx, Vxx.
//   lazy  (default) cc:383-384 re-evaluate them AFTER cc:379-381 overwrote Vx, Vxx: delta_V_ uses the new value function
//         (how Eigen's expression templates behave).
//   eager delta_V_ uses the Qu, Quu that K and k were computed from (what the author most likely meant; what a
//         compiler would do if `auto` were a Matrix).  -DCILQR_DV_EVAL_EAGER, also in backward_core.hpp.
#ifndef CILQR_DOT_ORDER
#define CILQR_DOT_ORDER 2
#endif
#ifdef CILQR_DV_EVAL_EAGER
constexpr int kDvEagerDefault = 1;
#else
constexpr int kDvEagerDefault = 0;
#endif
int g_dot_order = CILQR_DOT_ORDER;   // written only by oracle_set_semantics, between solves
int g_dv_eager = kDvEagerDefault;

// sum of six products in the order `order` (0 sequential, 1 halving tree, 2 even/odd packets)
inline double Sum6(const double* t, int order) {
  if (order == 1) return (t[0] + (t[1] + t[2])) + (t[3] + (t[4] + t[5]));
  if (order == 2) return (t[0] + (t[2] + t[4])) + (t[1] + (t[3] + t[5]));
  return ((((t[0] + t[1]) + t[2]) + t[3]) + t[4]) + t[5];
}
// Small dense helpers, row-major.  MatMul: lhs as stored (Eigen: column-major temporary / matrix on the left);
// MatTMul: lhs transposed (Eigen: X.transpose() * Y).  See CILQR_DOT_ORDER for what that decides.
template <int R, int I, int C>
inline void MatMul(const double* a, const double* b, double* out) {  // out[R][C] = a[R][I] b[I][C]
  const int order = (I == 6) ? (g_dot_order == 1 ? 1 : 0) : 0;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      if (I == 6 && order != 0) {
        double t[6];
        for (int k = 0; k < 6; ++k) t[k] = a[r * I + (k % I)] * b[(k % I) * C + c];
        out[r * C + c] = Sum6(t, order);
        continue;
      }
      double s = a[r * I + 0] * b[0 * C + c];
      for (int k = 1; k < I; ++k) s += a[r * I + k] * b[k * C + c];
      out[r * C + c] = s;
    }
}
template <int R, int I, int C>
inline void MatTMul(const double* a, const double* b, double* out) {  // out[R][C] = a[I][R]^T b[I][C]
  const int order = (I == 6) ? g_dot_order : 0;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      if (I == 6 && order != 0) {
        double t[6];
        for (int k = 0; k < 6; ++k) t[k] = a[(k % I) * R + r] * b[(k % I) * C + c];
        out[r * C + c] = Sum6(t, order);
        continue;
      }
      double s = a[0 * R + r] * b[0 * C + c];
      for (int k = 1; k < I; ++k) s += a[k * R + r] * b[k * C + c];
      out[r * C + c] = s;
    }
}

struct Oracle {
  oracle_config cfg;
  int N, K;
  double disc_radius;
  double start[4];
  std::vector<double> goals;                   // K*6
  int cmax = 0;
  std::vector<double> corridor;                // K*cmax*3, shrunk + normalised
  std::vector<int> ccount;                     // K
  std::vector<Lane> left, right;
  bool has_problem = false;

  // ---- bf:82-147 RelaxBarrierFunction ----
  double BarrierValue(double x) const {  // bf:104-113
    const double r = 1.0 / cfg.barrier_t, eps = cfg.barrier_eps;
    if (x < -eps) return -r * std::log(-x);
    const double q = (-x - 2.0 * eps) / eps;
    return 0.5 * r * (q * q - 1) - r * std::log(eps);
  }
  double BarrierJacCoef(double x) const {  // bf:115-125, coefficient multiplying dx
    const double r = 1.0 / cfg.barrier_t, eps = cfg.barrier_eps;
    if (x < -eps) return -r / x;
    return r * (x + 2.0 * eps) / eps / eps;
  }
  // bf:127-140.  out(i,j) = (coef*dx_i)*dx_j [- (r/x)*ddx(i,j) on the log branch]
  template <int Nn>
  void BarrierHessian(double x, const double* dx, const double* ddx, double* out) const {
    const double r = 1.0 / cfg.barrier_t, eps = cfg.barrier_eps;
    if (x < -eps) {
      const double c1 = r / x / x, c2 = r / x;
      for (int i = 0; i < Nn; ++i)
        for (int j = 0; j < Nn; ++j)
          out[i * Nn + j] = (c1 * dx[i]) * dx[j] - c2 * (ddx ? ddx[i * Nn + j] : 0.0);
    } else {
      const double c1 = r * (x + 2.0 * eps) / eps / eps;
      for (int i = 0; i < Nn; ++i)
        for (int j = 0; j < Nn; ++j) out[i * Nn + j] = (c1 * dx[i]) * dx[j];
    }
  }

  // ---- vm:123-138 ----
  void DynamicsContinuous(const double* s, const double* u, double* res) const {
    const double theta = NormalizeAngle(s[2]);
    const double v = s[3], a = s[4];
    const double delta = NormalizeAngle(s[5]);
    res[0] = v * std::cos(theta);
    res[1] = v * std::sin(theta);
    res[2] = v * std::tan(delta) / cfg.wheel_base;
    res[3] = a;
    res[4] = u[0];
    res[5] = u[1];
  }
  // ---- vm:88-121 (RK2 midpoint) ----
  void Dynamics(const double* s, const double* u, double* next) const {
    double k1[NX], mid[NX], k2[NX], out[NX];
    DynamicsContinuous(s, u, k1);
    const double h = 0.5 * cfg.dt;
    for (int i = 0; i < NX; ++i) mid[i] = s[i] + h * k1[i];
    DynamicsContinuous(mid, u, k2);
    for (int i = 0; i < NX; ++i) out[i] = s[i] + cfg.dt * k2[i];
    out[2] = NormalizeAngle(out[2]);
    out[5] = NormalizeAngle(out[5]);
    std::memcpy(next, out, sizeof(out));
  }
  // ---- vm:21-86 ----
  void DynamicsJacobian(const double* s, const double* u, double* A, double* B) const {
    const double L = cfg.wheel_base, dt = cfg.dt;
    const double v = s[3];
    const double theta = NormalizeAngle(s[2]);
    const double delta = NormalizeAngle(s[5]);
    const double a = s[4];
    const double delta_rate = u[1];
    const double theta_mid = theta + 0.5 * dt * v * std::tan(delta) / L;
    const double tan_delta = std::tan(delta);
    const double tan_delta_rate = std::tan(delta + 0.5 * dt * delta_rate);
    const double cos_theta_mid = std::cos(theta_mid);
    const double sin_theta_mid = std::sin(theta_mid);
    const double tan_delta_square = tan_delta * tan_delta;
    const double tan_delta_rate_square = tan_delta_rate * tan_delta_rate;
    const double v_tan_delta_rate = v * (tan_delta_rate_square + 1);
    for (int i = 0; i < 36; ++i) A[i] = 0.0;
    for (int i = 0; i < 12; ++i) B[i] = 0.0;
    A[0 * 6 + 0] = 1.0;
    A[0 * 6 + 2] = -dt * (0.5 * a * dt + v) * sin_theta_mid;
    A[0 * 6 + 3] = dt * cos_theta_mid - 0.5 * dt * dt * (0.5 * a * dt + v) * sin_theta_mid * tan_delta / L;
    A[0 * 6 + 4] = 0.5 * dt * dt * cos_theta_mid;
    A[0 * 6 + 5] = -0.5 * dt * dt * v * (0.5 * a * dt + v) * (tan_delta_square + 1) * sin_theta_mid / L;
    A[1 * 6 + 1] = 1.0;
    A[1 * 6 + 2] = dt * (0.5 * a * dt + v) * cos_theta_mid;
    A[1 * 6 + 3] = dt * sin_theta_mid + 0.5 * dt * dt * (0.5 * a * dt + v) * cos_theta_mid * tan_delta / L;
    A[1 * 6 + 4] = 0.5 * dt * dt * sin_theta_mid;
    A[1 * 6 + 5] = 0.5 * dt * dt * v * (0.5 * a * dt + v) * (tan_delta_square + 1) * cos_theta_mid / L;
    A[2 * 6 + 2] = 1.0;
    A[2 * 6 + 3] = dt * tan_delta_rate / L;
    A[2 * 6 + 4] = 0.5 * dt * dt * tan_delta_rate / L;
    A[2 * 6 + 5] = dt * v_tan_delta_rate / L;
    A[3 * 6 + 3] = 1.0;
    A[3 * 6 + 4] = dt;
    A[4 * 6 + 4] = 1.0;
    A[5 * 6 + 5] = 1.0;
    B[2 * 2 + 1] = 0.5 * dt * dt * v * (tan_delta_rate_square + 1) / L;
    B[3 * 2 + 0] = 0.5 * dt * dt;
    B[4 * 2 + 0] = dt;
    B[5 * 2 + 1] = dt;
  }

  // ---- cc:97-104 ----
  void CalculateDiscRadius() {
    const double length = cfg.front_hang + cfg.wheel_base + cfg.rear_hang;
    disc_radius = std::hypot(cfg.width / 2.0, length / 2.0 / cfg.num_of_disc);
  }
  double DiscL() const {  // cc:556
    return (cfg.rear_hang + cfg.wheel_base + cfg.front_hang) / cfg.num_of_disc;
  }

  // ---- cc:605-618 ----
  const Lane& FindNearestLaneSegment(double x, double y, const std::vector<Lane>& lanes) const {
    double min_dis = std::numeric_limits<double>::max();
    int min_index = -1;
    for (int i = 0; i < (int)lanes.size(); ++i) {
      const double dis = lanes[i].seg.DistanceTo(x, y);
      if (dis < min_dis) { min_dis = dis; min_index = i; }
    }
    if (min_index < 0) min_index = 0;  // reference indexes [-1] (UB) when every distance is NaN
    return lanes[min_index];
  }

  // ---- cc:497-516 ----
  double JCost(const double* X, const double* U) const {
    double cost = 0.0;
    for (int i = 0; i < K; ++i) {
      const double dx = X[i * 6 + 0] - goals[i * 6 + 0];
      const double dy = X[i * 6 + 1] - goals[i * 6 + 1];
      const double dth = X[i * 6 + 2] - goals[i * 6 + 2];
      cost += cfg.w_x * (dx * dx) + cfg.w_y * (dy * dy) + cfg.w_theta * (dth * dth);
    }
    for (int i = 0; i < N; ++i)
      cost += cfg.w_jerk * (U[i * 2] * U[i * 2]) + cfg.w_delta_rate * (U[i * 2 + 1] * U[i * 2 + 1]);
    return cost;
  }
  // ---- cc:518-551 ----
  double DynamicsCost(const double* X, const double* U) const {
    double x_cost = 0.0;
    for (int i = 0; i < K; ++i) {
      const double* s = X + i * 6;
      x_cost += BarrierValue(-s[3]);
      x_cost += BarrierValue(s[3] - cfg.max_velocity);
      x_cost += BarrierValue(s[4] - cfg.max_acceleration);
      x_cost += BarrierValue(cfg.min_acceleration - s[4]);
      x_cost += BarrierValue(s[5] - cfg.delta_max);
      x_cost += BarrierValue(cfg.delta_min - s[5]);
    }
    double u_cost = 0.0;
    for (int i = 0; i < N; ++i) {
      const double* u = U + i * 2;
      u_cost += BarrierValue(u[0] - cfg.jerk_max);
      u_cost += BarrierValue(cfg.jerk_min - u[0]);
      u_cost += BarrierValue(u[1] - cfg.delta_rate_max);
      u_cost += BarrierValue(cfg.delta_rate_min - u[1]);
    }
    return x_cost + u_cost;
  }
  // ---- cc:553-581 ----
  double CorridorCost(const double* X) const {
    double cost = 0.0;
    const double L = DiscL(), rf = cfg.rear_hang;
    for (int i = 0; i < K; ++i) {
      const double* s = X + i * 6;
      for (int j = 0; j < cfg.num_of_disc; ++j) {
        const double x = s[0] + (L * (j - 0.5) - rf) * std::cos(s[2]);
        const double y = s[1] + (L * (j - 0.5) - rf) * std::sin(s[2]);
        for (int c = 0; c < ccount[i]; ++c) {
          const double* p = &corridor[(i * cmax + c) * 3];
          cost += BarrierValue(p[0] * x + p[1] * y - p[2]);
        }
      }
    }
    return cost;
  }
  // ---- cc:583-603 ----
  double LaneBoundaryCost(const double* X) const {
    double cost = 0.0;
    const double L = DiscL(), rf = cfg.rear_hang;
    for (int i = 0; i < K; ++i) {
      const double* s = X + i * 6;
      for (int j = 0; j < cfg.num_of_disc; ++j) {
        const double x = s[0] + (L * (j - 0.5) - rf) * std::cos(s[2]);
        const double y = s[1] + (L * (j - 0.5) - rf) * std::sin(s[2]);
        const Lane& l = FindNearestLaneSegment(x, y, left);
        cost += BarrierValue(l.a * x + l.b * y - l.c);
        const Lane& r = FindNearestLaneSegment(x, y, right);
        cost += BarrierValue(r.a * x + r.b * y - r.c);
      }
    }
    return cost;
  }
  // ---- cc:417-436 ----
  double TotalCost(const double* X, const double* U, double* cost5) const {
    const double j_cost = JCost(X, U);
    const double dynamics_cost = DynamicsCost(X, U);
    const double corridor_cost = CorridorCost(X);
    const double lane_cost = LaneBoundaryCost(X);
    const double total = j_cost + dynamics_cost + corridor_cost + lane_cost;
    if (cost5) {
      cost5[0] = total; cost5[1] = j_cost; cost5[2] = dynamics_cost;
      cost5[3] = corridor_cost; cost5[4] = lane_cost;
    }
    return total;
  }

  // ---- cc:620-636 with cc:657-671, 690-706, 729-746 ----
  void CostJacobian(int index, const double* s, const double* u, double* Jx, double* Ju) const {
    Jx[0] = 2.0 * cfg.w_x * (s[0] - goals[index * 6 + 0]);
    Jx[1] = 2.0 * cfg.w_y * (s[1] - goals[index * 6 + 1]);
    Jx[2] = 2.0 * cfg.w_theta * (s[2] - goals[index * 6 + 2]);
    Jx[3] = 0.0; Jx[4] = 0.0; Jx[5] = 0.0;
    Ju[0] = 2.0 * cfg.w_jerk * u[0];
    Ju[1] = 2.0 * cfg.w_delta_rate * u[1];
    // DynamicsConsJacbian cc:657-671: the six (four) gradient vectors are summed first, then added.
    {
      const double g[6] = {0.0 - s[3], s[3] - cfg.max_velocity, cfg.min_acceleration - s[4],
                           s[4] - cfg.max_acceleration, cfg.delta_min - s[5], s[5] - cfg.delta_max};
      const int comp[6] = {3, 3, 4, 4, 5, 5};
      const double sign[6] = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0};
      double sum[6];
      for (int e = 0; e < 6; ++e) {
        double acc = 0.0;
        for (int t = 0; t < 6; ++t) {
          const double d = (comp[t] == e) ? sign[t] : 0.0;
          const double term = BarrierJacCoef(g[t]) * d;
          acc = (t == 0) ? term : acc + term;
        }
        sum[e] = acc;
      }
      for (int e = 0; e < 6; ++e) Jx[e] += sum[e];
      const double gu[4] = {cfg.jerk_min - u[0], u[0] - cfg.jerk_max, cfg.delta_rate_min - u[1],
                            u[1] - cfg.delta_rate_max};
      const int compu[4] = {0, 0, 1, 1};
      const double signu[4] = {-1.0, 1.0, -1.0, 1.0};
      for (int e = 0; e < 2; ++e) {
        double acc = 0.0;
        for (int t = 0; t < 4; ++t) {
          const double d = (compu[t] == e) ? signu[t] : 0.0;
          const double term = BarrierJacCoef(gu[t]) * d;
          acc = (t == 0) ? term : acc + term;
        }
        Ju[e] += acc;
      }
    }
    const double L = DiscL(), rf = cfg.rear_hang;
    // CorridorConsJacbian cc:690-706
    for (int j = 0; j < cfg.num_of_disc; ++j) {
      const double length_cos = (L * (j - 0.5) - rf) * std::cos(s[2]);
      const double length_sin = (L * (j - 0.5) - rf) * std::sin(s[2]);
      const double x = s[0] + length_cos, y = s[1] + length_sin;
      for (int c = 0; c < ccount[index]; ++c) {
        const double* p = &corridor[(index * cmax + c) * 3];
        const double coef = BarrierJacCoef(p[0] * x + p[1] * y - p[2]);
        const double d[3] = {p[0], p[1], -p[0] * length_sin + p[1] * length_cos};
        for (int e = 0; e < 3; ++e) Jx[e] += coef * d[e];
      }
    }
    // LaneBoundaryConsJacbian cc:729-746
    for (int j = 0; j < cfg.num_of_disc; ++j) {
      const double length_cos = (L * (j - 0.5) - rf) * std::cos(s[2]);
      const double length_sin = (L * (j - 0.5) - rf) * std::sin(s[2]);
      const double x = s[0] + length_cos, y = s[1] + length_sin;
      const Lane* ls[2] = {&FindNearestLaneSegment(x, y, left), &FindNearestLaneSegment(x, y, right)};
      for (int side = 0; side < 2; ++side) {
        const Lane& l = *ls[side];
        const double coef = BarrierJacCoef(l.a * x + l.b * y - l.c);
        const double d[3] = {l.a, l.b, -l.a * length_sin + l.b * length_cos};
        for (int e = 0; e < 3; ++e) Jx[e] += coef * d[e];
      }
    }
  }

  // ---- cc:638-655 with cc:673-688, 708-727, 748-769 ----
  void CostHessian(int index, const double* s, const double* u, double* Hx, double* Hu) const {
    for (int i = 0; i < 36; ++i) Hx[i] = 0.0;
    Hx[0 * 6 + 0] = 2.0 * cfg.w_x;
    Hx[1 * 6 + 1] = 2.0 * cfg.w_y;
    Hx[2 * 6 + 2] = 2.0 * cfg.w_theta;
    Hx[3 * 6 + 3] = 2.0 * cfg.w_v;
    Hx[4 * 6 + 4] = 2.0 * cfg.w_a;
    Hx[5 * 6 + 5] = 2.0 * cfg.w_delta;
    Hu[0] = 2.0 * cfg.w_jerk; Hu[1] = 0.0; Hu[2] = 0.0; Hu[3] = 2.0 * cfg.w_delta_rate;
    // DynamicsConsHessian cc:673-688: six (four) matrices summed, then added.
    {
      const double g[6] = {0.0 - s[3], s[3] - cfg.max_velocity, cfg.min_acceleration - s[4],
                           s[4] - cfg.max_acceleration, cfg.delta_min - s[5], s[5] - cfg.delta_max};
      const int comp[6] = {3, 3, 4, 4, 5, 5};
      const double sign[6] = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0};
      double sum[36], h[36];
      for (int t = 0; t < 6; ++t) {
        double d[6] = {0, 0, 0, 0, 0, 0};
        d[comp[t]] = sign[t];
        BarrierHessian<6>(g[t], d, nullptr, h);
        for (int e = 0; e < 36; ++e) sum[e] = (t == 0) ? h[e] : sum[e] + h[e];
      }
      for (int e = 0; e < 36; ++e) Hx[e] += sum[e];
      const double gu[4] = {cfg.jerk_min - u[0], u[0] - cfg.jerk_max, cfg.delta_rate_min - u[1],
                            u[1] - cfg.delta_rate_max};
      const int compu[4] = {0, 0, 1, 1};
      const double signu[4] = {-1.0, 1.0, -1.0, 1.0};
      double sumu[4], hu[4];
      for (int t = 0; t < 4; ++t) {
        double d[2] = {0, 0};
        d[compu[t]] = signu[t];
        BarrierHessian<2>(gu[t], d, nullptr, hu);
        for (int e = 0; e < 4; ++e) sumu[e] = (t == 0) ? hu[e] : sumu[e] + hu[e];
      }
      for (int e = 0; e < 4; ++e) Hu[e] += sumu[e];
    }
    const double L = DiscL(), rf = cfg.rear_hang;
    double ddx[36], h[36];
    for (int i = 0; i < 36; ++i) ddx[i] = 0.0;
    // CorridorConsHessian cc:708-727
    for (int j = 0; j < cfg.num_of_disc; ++j) {
      const double length_cos = (L * (j - 0.5) - rf) * std::cos(s[2]);
      const double length_sin = (L * (j - 0.5) - rf) * std::sin(s[2]);
      const double x = s[0] + length_cos, y = s[1] + length_sin;
      for (int c = 0; c < ccount[index]; ++c) {
        const double* p = &corridor[(index * cmax + c) * 3];
        ddx[2 * 6 + 2] = -p[0] * length_cos - p[1] * length_sin;
        const double d[6] = {p[0], p[1], -p[0] * length_sin + p[1] * length_cos, 0.0, 0.0, 0.0};
        BarrierHessian<6>(p[0] * x + p[1] * y - p[2], d, ddx, h);
        for (int e = 0; e < 36; ++e) Hx[e] += h[e];
      }
    }
    // LaneBoundaryConsHessian cc:748-769
    for (int j = 0; j < cfg.num_of_disc; ++j) {
      const double length_cos = (L * (j - 0.5) - rf) * std::cos(s[2]);
      const double length_sin = (L * (j - 0.5) - rf) * std::sin(s[2]);
      const double x = s[0] + length_cos, y = s[1] + length_sin;
      const Lane* ls[2] = {&FindNearestLaneSegment(x, y, left), &FindNearestLaneSegment(x, y, right)};
      for (int side = 0; side < 2; ++side) {
        const Lane& l = *ls[side];
        ddx[2 * 6 + 2] = -l.a * length_cos - l.b * length_sin;
        const double d[6] = {l.a, l.b, -l.a * length_sin + l.b * length_cos, 0.0, 0.0, 0.0};
        BarrierHessian<6>(l.a * x + l.b * y - l.c, d, ddx, h);
        for (int e = 0; e < 36; ++e) Hx[e] += h[e];
      }
    }
  }

  // cc:203-213: linearise around (X, U); terminal knot with u = 0, its lu/luu discarded.
  void Quadratize(const double* X, const double* U, double* A, double* B, double* lx, double* lu,
                  double* lxx, double* luu) const {
    for (int i = 0; i < N; ++i) {
      DynamicsJacobian(X + i * 6, U + i * 2, A + i * 36, B + i * 12);
      CostJacobian(i, X + i * 6, U + i * 2, lx + i * 6, lu + i * 2);
      CostHessian(i, X + i * 6, U + i * 2, lxx + i * 36, luu + i * 4);
    }
    const double zero_u[2] = {0.0, 0.0};
    double tmp_ju[2], tmp_hu[4];
    CostJacobian(N, X + N * 6, zero_u, lx + N * 6, tmp_ju);
    CostHessian(N, X + N * 6, zero_u, lxx + N * 36, tmp_hu);
  }

  // ---- cc:334-390 ----
  void Backward(double lambda, const double* A, const double* B, const double* lx, const double* lu,
                const double* lxx, const double* luu, double* Ks, double* ks, double* dV) const {
    dV[0] = 0.0; dV[1] = 0.0;
    double Vx[6], Vxx[36];
    std::memcpy(Vx, lx + N * 6, sizeof(Vx));
    std::memcpy(Vxx, lxx + N * 36, sizeof(Vxx));
    for (int i = N - 1; i >= 0; --i) {
      const double* Ai = A + i * 36;
      const double* Bi = B + i * 12;
      double* Kc = Ks + i * 12;  // 2x6 row-major
      double* kc = ks + i * 2;
      double BtV[12], Qux[12], BtVB[4], Quu[4], BtVx[2], Qu[2];
      MatTMul<2, 6, 6>(Bi, Vxx, BtV);      // B^T Vxx
      MatMul<2, 6, 6>(BtV, Ai, Qux);       // (B^T Vxx) A             cc:353
      MatMul<2, 6, 2>(BtV, Bi, BtVB);
      for (int e = 0; e < 4; ++e) Quu[e] = luu[i * 4 + e] + BtVB[e];          // cc:352
      MatTMul<2, 6, 1>(Bi, Vx, BtVx);
      for (int e = 0; e < 2; ++e) Qu[e] = lu[i * 2 + e] + BtVx[e];            // cc:349
      // cc:361: Quu + lambda * Identity (coefficient-wise)
      double Qt[4] = {Quu[0] + lambda * 1.0, Quu[1] + lambda * 0.0, Quu[2] + lambda * 0.0,
                      Quu[3] + lambda * 1.0};
      // cc:363: Eigen fixed 2x2 inverse
      const double invdet = 1.0 / (Qt[0] * Qt[3] - Qt[2] * Qt[1]);
      const double inv[4] = {Qt[3] * invdet, -Qt[1] * invdet, -Qt[2] * invdet, Qt[0] * invdet};
      const double ninv[4] = {-inv[0], -inv[1], -inv[2], -inv[3]};
      MatMul<2, 2, 6>(ninv, Qux, Kc);      // cc:365
      MatMul<2, 2, 1>(ninv, Qu, kc);       // cc:366
      // cc:379-380, everything on the right side sees the OLD Vx/Vxx
      double AtVx[6], AtV[36], AtVA[36], KtQuu[12], t1[6], t2[6], t3[6];
      MatTMul<6, 6, 1>(Ai, Vx, AtVx);
      MatTMul<6, 6, 6>(Ai, Vxx, AtV);
      MatMul<6, 6, 6>(AtV, Ai, AtVA);
      MatTMul<6, 2, 2>(Kc, Quu, KtQuu);    // K^T Quu (6x2)
      MatMul<6, 2, 1>(KtQuu, kc, t1);      // (K^T Quu) k
      MatTMul<6, 2, 1>(Kc, Qu, t2);        // K^T Qu
      MatTMul<6, 2, 1>(Qux, kc, t3);       // Qux^T k
      double nVx[6], nVxx[36], m1[36], m2[36], m3[36];
      for (int e = 0; e < 6; ++e) nVx[e] = (((lx[i * 6 + e] + AtVx[e]) + t1[e]) + t2[e]) + t3[e];
      MatMul<6, 2, 6>(KtQuu, Kc, m1);      // (K^T Quu) K
      MatTMul<6, 2, 6>(Kc, Qux, m2);       // K^T Qux
      MatTMul<6, 2, 6>(Qux, Kc, m3);       // Qux^T K
      for (int e = 0; e < 36; ++e) nVxx[e] = (((lxx[i * 36 + e] + AtVA[e]) + m1[e]) + m2[e]) + m3[e];
      std::memcpy(Vx, nVx, sizeof(Vx));
      std::memcpy(Vxx, nVxx, sizeof(Vxx));
      // cc:381: in-place, column-major traversal, no temporary
      for (int c = 0; c < 6; ++c)
        for (int r = 0; r < 6; ++r) Vxx[r * 6 + c] = 0.5 * (Vxx[r * 6 + c] + Vxx[c * 6 + r]);
      // cc:383-384: lazy Qu / Quu re-evaluated with the NEW Vx / Vxx (CILQR_DV_EVAL lazy; eager: the ones K, k came from)
      double BtV2[12], BtVB2[4], Quu2[4], BtVx2[2], Qu2[2];
      if (g_dv_eager) {
        for (int e = 0; e < 2; ++e) Qu2[e] = Qu[e];
        for (int e = 0; e < 4; ++e) Quu2[e] = Quu[e];
      } else {
        MatTMul<2, 6, 1>(Bi, Vx, BtVx2);
        for (int e = 0; e < 2; ++e) Qu2[e] = lu[i * 2 + e] + BtVx2[e];
        MatTMul<2, 6, 6>(Bi, Vxx, BtV2);
        MatMul<2, 6, 2>(BtV2, Bi, BtVB2);
        for (int e = 0; e < 4; ++e) Quu2[e] = luu[i * 4 + e] + BtVB2[e];
      }
      dV[0] += kc[0] * Qu2[0] + kc[1] * Qu2[1];
      const double hk[2] = {0.5 * kc[0], 0.5 * kc[1]};
      const double hkQ[2] = {hk[0] * Quu2[0] + hk[1] * Quu2[2], hk[0] * Quu2[1] + hk[1] * Quu2[3]};
      dV[1] += hkQ[0] * kc[0] + hkQ[1] * kc[1];
    }
  }

  // ---- cc:322-332 ----
  double GradNorm(const double* ks, const double* U) const {
    double acc = 0.0;
    for (int i = 0; i < N; ++i) {
      const double v0 = std::abs(ks[i * 2]) / (std::abs(U[i * 2]) + 1);
      const double v1 = std::abs(ks[i * 2 + 1]) / (std::abs(U[i * 2 + 1]) + 1);
      acc += (v0 > v1 ? v0 : v1);  // maxCoeff: strict '<' update, first wins on ties
    }
    return acc / N;
  }

  // ---- cc:392-415 ----
  void Forward(double alpha, const double* X, const double* U, const double* Ks, const double* ks,
               double* Xn, double* Un) const {
    double x[6];
    std::memcpy(x, &goals[0], sizeof(x));
    std::memcpy(Xn, x, sizeof(x));
    for (int i = 0; i < N; ++i) {
      double dx[6], Kdx[2];
      for (int e = 0; e < 6; ++e) dx[e] = x[e] - X[i * 6 + e];
      MatMul<2, 6, 1>(Ks + i * 12, dx, Kdx);
      double u[2];
      for (int e = 0; e < 2; ++e) u[e] = (U[i * 2 + e] + Kdx[e]) + alpha * ks[i * 2 + e];
      u[1] = NormalizeAngle(u[1]);
      Un[i * 2] = u[0]; Un[i * 2 + 1] = u[1];
      Dynamics(x, u, x);
      std::memcpy(Xn + (i + 1) * 6, x, sizeof(x));
    }
  }

  // ---- cc:793-842 ----
  void InitGuess(double* X, double* U) const {
    std::vector<double> Ks(N * 12);
    double Q[36];
    for (int i = 0; i < 36; ++i) Q[i] = 0.0;
    Q[0] = 0.001; Q[7] = 0.001; Q[14] = 0.001; Q[21] = 0.001; Q[28] = 0.01; Q[35] = 0.005;
    const double R[4] = {0.2, 0.0, 0.0, 0.05};
    double P[36];
    std::memcpy(P, Q, sizeof(P));
    const double zero_u[2] = {0.0, 0.0};
    double A[36], B[12];
    for (int i = N - 1; i >= 0; --i) {
      DynamicsJacobian(&goals[i * 6], zero_u, A, B);
      double BtP[12], BtPB[4], BtPA[12], M[4];
      MatTMul<2, 6, 6>(B, P, BtP);
      MatMul<2, 6, 2>(BtP, B, BtPB);
      MatMul<2, 6, 6>(BtP, A, BtPA);
      for (int e = 0; e < 4; ++e) M[e] = R[e] + BtPB[e];
      const double invdet = 1.0 / (M[0] * M[3] - M[2] * M[1]);
      const double inv[4] = {M[3] * invdet, -M[1] * invdet, -M[2] * invdet, M[0] * invdet};
      MatMul<2, 2, 6>(inv, BtPA, &Ks[i * 12]);                              // cc:822
      double BK[36], AmBK[36], AtP[36], AtPA[36];
      MatMul<6, 2, 6>(B, &Ks[i * 12], BK);
      for (int e = 0; e < 36; ++e) AmBK[e] = A[e] - BK[e];
      MatTMul<6, 6, 6>(A, P, AtP);
      MatMul<6, 6, 6>(AtP, AmBK, AtPA);
      for (int e = 0; e < 36; ++e) P[e] = Q[e] + AtPA[e];                   // cc:823
    }
    double x[6];
    std::memcpy(x, &goals[0], sizeof(x));
    std::memcpy(X, x, sizeof(x));
    for (int i = 0; i < N; ++i) {
      double dx[6], nK[12], u[2];
      for (int e = 0; e < 6; ++e) dx[e] = x[e] - goals[i * 6 + e];
      for (int e = 0; e < 12; ++e) nK[e] = -Ks[i * 12 + e];
      MatMul<2, 6, 1>(nK, dx, u);                                            // cc:834
      u[0] = std::fmin(cfg.jerk_max, std::fmax(u[0], cfg.jerk_min));
      u[1] = std::fmin(cfg.delta_rate_max, std::fmax(u[1], cfg.delta_rate_min));
      U[i * 2] = u[0]; U[i * 2 + 1] = u[1];
      Dynamics(x, u, x);
      std::memcpy(X + (i + 1) * 6, x, sizeof(x));
    }
  }

  // ---- cc:771-791 ----
  void ToTrajectory(const double* X, const double* U, double* traj) const {
    for (int i = 0; i < K; ++i) {
      double* t = traj + i * 10;
      t[0] = i * cfg.dt;
      t[1] = X[i * 6 + 0]; t[2] = X[i * 6 + 1]; t[3] = X[i * 6 + 2];
      t[4] = X[i * 6 + 3]; t[5] = X[i * 6 + 4]; t[6] = X[i * 6 + 5];
      t[7] = std::tan(X[i * 6 + 5]) / cfg.wheel_base;
      t[8] = 0.0; t[9] = 0.0;
      if (i < K - 1) { t[8] = U[i * 2]; t[9] = U[i * 2 + 1]; }
    }
  }

  // ---- cc:141-152, 438-495 ----
  int SetProblem(const double* start4, const double* coarse, int n_coarse, const double* cor,
                 const int* cc, int cmax_, const double* l, int nl, const double* r, int nr) {
    has_problem = false;
    bool empty_corridor = (cor == nullptr || cc == nullptr);
    if (empty_corridor || nl == 0 || nr == 0) return -1;                      // cc:68-73
    if (K != n_coarse) return -1;                                             // cc:75-78
    std::memcpy(start, start4, sizeof(start));
    goals.assign(coarse, coarse + K * 6);                                     // cc:147-150
    goals[0] = start[0]; goals[1] = start[1]; goals[2] = start[2]; goals[3] = start[3];
    goals[4] = 0.0; goals[5] = 0.0;                                           // cc:151
    cmax = cmax_;
    ccount.assign(cc, cc + K);
    corridor.assign(cor, cor + (size_t)K * cmax * 3);
    const double shrink_c = disc_radius + cfg.safe_margin;
    for (int i = 0; i < K; ++i)
      for (int c = 0; c < ccount[i]; ++c) {
        double* e = &corridor[(i * cmax + c) * 3];
        e[2] = e[2] - shrink_c * (e[0] * e[0] + e[1] * e[1]) / std::hypot(e[0], e[1]);   // cc:448
        const double norm = std::hypot(std::hypot(e[0], e[1]), e[2]);                    // cc:479
        e[0] = e[0] / norm; e[1] = e[1] / norm; e[2] = e[2] / norm;
      }
    auto load_lane = [&](const double* src, int n, std::vector<Lane>& dst) {
      dst.resize(n);
      for (int i = 0; i < n; ++i) {
        const double* p = src + i * 7;
        double a = p[0], b = p[1], c = p[2];
        c = c - disc_radius * (a * a + b * b) / std::hypot(a, b);                        // cc:463,471
        const double norm = std::hypot(std::hypot(a, b), c);                             // cc:486,492
        dst[i].a = a / norm; dst[i].b = b / norm; dst[i].c = c / norm;
        dst[i].seg.Set(p[3], p[4], p[5], p[6]);
      }
    };
    load_lane(l, nl, left);
    load_lane(r, nr, right);
    has_problem = true;
    return 0;
  }

  static double RelMargin(double a, double b) {
    const double s = std::max(std::abs(a), std::abs(b));
    if (!(s > 0.0)) return 0.0;
    return std::abs(a - b) / s;
  }

  // Re-entry into the loop of Optimize() for the step-by-step parity tests: instead of the init guess
  // the solve starts from a given iterate with the given regularisation state, at iteration `iter`,
  // and (stop_after_accept) returns as soon as one iteration was accepted.
  struct Warm {
    const double* X;
    const double* U;
    double lambda, dlambda;
    int iter;
    bool stop_after_accept;
  };

  // ---- cc:154-320 ----
  int Plan(double* traj, double* cost_hist, int* n_cost_out, int* status_out, int* n_iter_out,
           double* iter_trajs, int max_iter_trajs, int* n_iter_trajs_out, double* trace,
           double* min_margin_out, const Warm* warm = nullptr, double* U_out = nullptr,
           double* lambda_out = nullptr) const {
    std::vector<double> X(K * 6), U(N * 2), Xo(K * 6), Uo(N * 2);
    std::vector<double> A(N * 36), B(N * 12), lx(K * 6), lu(N * 2), lxx(K * 36), luu(N * 4);
    std::vector<double> Ks(N * 12), ks(N * 2);
    int n_cost = 0, n_it = 0, status = ORACLE_ST_RUNNING;
    double margin = std::numeric_limits<double>::infinity();
    auto push_iter_traj = [&](const double* Xs, const double* Us) {
      if (iter_trajs && n_it < max_iter_trajs) ToTrajectory(Xs, Us, iter_trajs + (size_t)n_it * K * 10);
      ++n_it;
    };
    auto push_cost = [&](const double* c5) {
      std::memcpy(cost_hist + n_cost * 5, c5, 5 * sizeof(double));
      ++n_cost;
    };
    if (warm) {
      std::memcpy(X.data(), warm->X, X.size() * sizeof(double));
      std::memcpy(U.data(), warm->U, U.size() * sizeof(double));
    } else {
      InitGuess(X.data(), U.data());                       // cc:169
    }
    push_iter_traj(X.data(), U.data());                    // cc:170
    double cost_data[5];
    double cost_old = TotalCost(X.data(), U.data(), cost_data);  // cc:172
    push_cost(cost_data);
    bool updated = true;
    double dcost = 0.0, lambda = 1.0, dlambda = 1.0, z = 0.0, cost_new = 0.0;
    const double ratio = 1.6, reg_min = 1e-8, reg_max = 1e11, gnorm_min = 1e-6;
    const double beta_min = 1e-4, beta_max = 10.0;
    static const double alpha_list[11] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316,
                                          0.0158, 0.0079, 0.0040, 0.0020, 0.0010};   // cc:197
    double dV[2];
    int iter = 0;
    if (warm) { lambda = warm->lambda; dlambda = warm->dlambda; iter = warm->iter; }
    const int iter_first = iter;
    bool stopped = false;
    for (; iter < cfg.max_iter; ++iter) {
      if (updated) {                                                               // cc:203-214
        Quadratize(X.data(), U.data(), A.data(), B.data(), lx.data(), lu.data(), lxx.data(), luu.data());
        updated = false;
      }
      Backward(lambda, A.data(), B.data(), lx.data(), lu.data(), lxx.data(), luu.data(), Ks.data(),
               ks.data(), dV);                                                     // cc:218 (never diverges)
      const double gnorm = GradNorm(ks.data(), U.data());                          // cc:235
      if (trace) {
        double* t = trace + (iter - iter_first) * kTraceCols;
        t[0] = -1; t[1] = lambda; t[2] = dV[0]; t[3] = dV[1]; t[4] = 0; t[5] = 0; t[6] = 0; t[7] = gnorm;
        t[8] = 0; t[9] = 0;
      }
      // smallest relative distance of this iteration's decisions to their thresholds
      double it_margin = (lambda < 1e-5) ? RelMargin(gnorm, gnorm_min) : std::numeric_limits<double>::infinity();
      if (gnorm < gnorm_min && lambda < 1e-5) {                                    // cc:236-241
        margin = std::min(margin, it_margin);
        if (trace) { trace[(iter - iter_first) * kTraceCols + 0] = -2; trace[(iter - iter_first) * kTraceCols + 8] = it_margin; }
        status = ORACLE_ST_GNORM; ++iter; break;
      }
      bool done = false;
      int acc_idx = -1, n_trials = 0;
      for (int ai = 0; ai < 11; ++ai) {                                            // cc:246-265
        ++n_trials;
        const double alpha = alpha_list[ai];
        Forward(alpha, X.data(), U.data(), Ks.data(), ks.data(), Xo.data(), Uo.data());
        cost_new = TotalCost(Xo.data(), Uo.data(), cost_data);
        dcost = cost_old - cost_new;
        const double expected = -alpha * (dV[0] + alpha * dV[1]);
        z = dcost / expected;
        it_margin = std::min(it_margin, RelMargin(z, beta_min));
        it_margin = std::min(it_margin, RelMargin(z, beta_max));
        it_margin = std::min(it_margin, std::abs(dcost) / std::max(std::abs(cost_old), 1e-300));
        if ((z > beta_min && z < beta_max) && dcost > 0.0) { done = true; acc_idx = ai; break; }
      }
      if (done) {
        it_margin = std::min(it_margin, RelMargin(dcost, cfg.abs_cost_tol));
        it_margin = std::min(it_margin, RelMargin(dcost / cost_old, cfg.rel_cost_tol));
      }
      margin = std::min(margin, it_margin);
      if (trace) {
        double* t = trace + (iter - iter_first) * kTraceCols;
        t[0] = acc_idx; t[4] = cost_new; t[5] = dcost; t[6] = z; t[8] = it_margin; t[9] = n_trials;
      }
      if (done) {
        X.swap(Xo); U.swap(Uo);
        dlambda = std::fmin(dlambda / ratio, 1.0 / ratio);                         // cc:273
        lambda = lambda * dlambda * (lambda > reg_min);                            // cc:275
        updated = true;
        if (dcost < cfg.abs_cost_tol || dcost / cost_old < cfg.rel_cost_tol) {     // cc:281-293
          push_cost(cost_data);
          status = (dcost < cfg.abs_cost_tol) ? ORACLE_ST_CONVERGED_ABS : ORACLE_ST_CONVERGED_REL;
          ++iter;
          break;
        }
        push_iter_traj(X.data(), U.data());                                        // cc:294
        cost_old = cost_new;
        push_cost(cost_data);
        if (warm && warm->stop_after_accept) { ++iter; stopped = true; break; }
      } else {
        dlambda = std::fmax(dlambda * ratio, ratio);                               // cc:298
        lambda = std::fmax(lambda * dlambda, reg_min);                             // cc:299
        if (lambda > reg_max) { status = ORACLE_ST_UNSOLVED; ++iter; break; }      // cc:302-307
      }
    }
    if (status == ORACLE_ST_RUNNING && (!stopped || iter >= cfg.max_iter)) status = ORACLE_ST_MAX_ITER;  // cc:312-319
    ToTrajectory(X.data(), U.data(), traj);
    if (U_out) std::memcpy(U_out, U.data(), U.size() * sizeof(double));
    if (lambda_out) { lambda_out[0] = lambda; lambda_out[1] = dlambda; }
    *n_cost_out = n_cost;
    *status_out = status;
    if (n_iter_out) *n_iter_out = iter;
    if (n_iter_trajs_out) *n_iter_trajs_out = n_it;
    if (min_margin_out) *min_margin_out = margin;
    return 0;
  }
};

}  // namespace

extern "C" {

/* dv_eval: 0 lazy, 1 eager; dot_order: 0 sequential, 1 eigen_redux, 2 eigen_sse2 (see the top of this file); a negative
 * value leaves that switch alone.  Process-wide; call between solves only.  Returns dv_eval | dot_order << 8 as now set. */
int oracle_set_semantics(int dv_eval, int dot_order) {
  if (dv_eval == 0 || dv_eval == 1) g_dv_eager = dv_eval;
  if (dot_order >= 0 && dot_order <= 2) g_dot_order = dot_order;
  return g_dv_eager | (g_dot_order << 8);
}

/* test hook: one 6-term dot product through the helpers every matrix product of this file uses.  transposed_lhs = 1:
 * the path of X.transpose() * Y (MatTMul), 0: plain lhs (MatMul); with the dot order currently set. */
double oracle_dot6(const double* a, const double* b, int transposed_lhs) {
  double out = 0.0;
  if (transposed_lhs) MatTMul<1, 6, 1>(a, b, &out);
  else MatMul<1, 6, 1>(a, b, &out);
  return out;
}

void oracle_default_config(oracle_config* c, int n_steps) {
  c->n_steps = n_steps;
  c->dt = 0.1;                       // planner_config.h:94
  c->num_of_disc = 5;                // :58
  c->safe_margin = 0.2;              // :59
  c->w_jerk = 1; c->w_delta_rate = 1;            // :46-47
  c->w_x = 0.5; c->w_y = 0.5; c->w_theta = 1e-3; // :49-51
  c->w_v = 0.0; c->w_a = 0.0; c->w_delta = 0.0;  // :52-54
  c->max_iter = 200;                 // :63
  c->abs_cost_tol = 1e-2; c->rel_cost_tol = 1e-2;  // :65-66
  c->front_hang = 0.96; c->wheel_base = 1.0; c->rear_hang = 0.929; c->width = 1.942;  // vehicle_param.h:26-41
  c->max_velocity = 20.0;            // :46
  c->min_acceleration = -5.0; c->max_acceleration = 5.0;  // :51-52
  c->jerk_min = -10.0; c->jerk_max = 10.0;       // :57-58
  c->delta_min = -40.0 / 180 * M_PI; c->delta_max = 40.0 / 180 * M_PI;  // :60-61
  c->delta_rate_min = c->delta_min / 3.0; c->delta_rate_max = c->delta_max / 3.0;  // :63-64
  c->barrier_t = 5.0; c->barrier_eps = 0.01;     // barrier_function.h:144-145
}

void* oracle_create(const oracle_config* c) {
  Oracle* o = new Oracle();
  o->cfg = *c;
  o->N = c->n_steps;
  o->K = c->n_steps + 1;
  o->CalculateDiscRadius();
  return o;
}
void oracle_destroy(void* h) { delete static_cast<Oracle*>(h); }

int oracle_set_problem(void* h, const double* start4, const double* coarse, int n_coarse,
                       const double* corridor, const int* ccount, int cmax, const double* left,
                       int n_left, const double* right, int n_right) {
  return static_cast<Oracle*>(h)->SetProblem(start4, coarse, n_coarse, corridor, ccount, cmax, left,
                                             n_left, right, n_right);
}

int oracle_plan(void* h, double* traj, double* cost_hist, int* n_cost, int* status, int* n_iter,
                double* iter_trajs, int max_iter_trajs, int* n_iter_trajs, double* trace,
                double* min_margin) {
  Oracle* o = static_cast<Oracle*>(h);
  if (traj == nullptr || cost_hist == nullptr || n_cost == nullptr || status == nullptr) return -1;  // cc:64
  if (!o->has_problem) return -1;
  return o->Plan(traj, cost_hist, n_cost, status, n_iter, iter_trajs, max_iter_trajs, n_iter_trajs,
                 trace, min_margin);
}

/* Step-by-step replay: Optimize() re-entered at iteration `iter` from the iterate (X, U) with the
 * regularisation state (lambda, dlambda); runs until one iteration is accepted (or the solve ends).
 * cost_hist row 0 = TotalCost(X, U), row 1 = the accepted trial (if any); trace rows count from the
 * first replayed iteration; lambda_out = (lambda, dlambda) afterwards. */
int oracle_replay(void* h, const double* X, const double* U, double lambda, double dlambda, int iter,
                  double* traj, double* cost_hist, int* n_cost, int* status, int* n_iter, double* trace,
                  double* lambda_out) {
  Oracle* o = static_cast<Oracle*>(h);
  if (!o->has_problem) return -1;
  Oracle::Warm w{X, U, lambda, dlambda, iter, true};
  return o->Plan(traj, cost_hist, n_cost, status, n_iter, nullptr, 0, nullptr, trace, nullptr, &w, nullptr,
                 lambda_out);
}

void oracle_get_constraints(void* h, double* goals, double* corridor, double* left_abc,
                            double* right_abc, double* disc_radius) {
  Oracle* o = static_cast<Oracle*>(h);
  if (goals) std::memcpy(goals, o->goals.data(), o->goals.size() * sizeof(double));
  if (corridor) std::memcpy(corridor, o->corridor.data(), o->corridor.size() * sizeof(double));
  if (left_abc)
    for (size_t i = 0; i < o->left.size(); ++i) {
      left_abc[i * 3] = o->left[i].a; left_abc[i * 3 + 1] = o->left[i].b; left_abc[i * 3 + 2] = o->left[i].c;
    }
  if (right_abc)
    for (size_t i = 0; i < o->right.size(); ++i) {
      right_abc[i * 3] = o->right[i].a; right_abc[i * 3 + 1] = o->right[i].b; right_abc[i * 3 + 2] = o->right[i].c;
    }
  if (disc_radius) *disc_radius = o->disc_radius;
}

void oracle_init_guess(void* h, double* X, double* U) { static_cast<Oracle*>(h)->InitGuess(X, U); }

void oracle_open_loop_rollout(void* h, const double* x0, const double* U, double* X) {
  Oracle* o = static_cast<Oracle*>(h);
  double x[6];
  std::memcpy(x, x0, sizeof(x));
  std::memcpy(X, x, sizeof(x));
  for (int i = 0; i < o->N; ++i) {
    o->Dynamics(x, U + i * 2, x);
    std::memcpy(X + (i + 1) * 6, x, sizeof(x));
  }
}

double oracle_total_cost(void* h, const double* X, const double* U, double* cost5) {
  return static_cast<Oracle*>(h)->TotalCost(X, U, cost5);
}
void oracle_quadratize(void* h, const double* X, const double* U, double* A, double* B, double* lx,
                       double* lu, double* lxx, double* luu) {
  static_cast<Oracle*>(h)->Quadratize(X, U, A, B, lx, lu, lxx, luu);
}
void oracle_backward(void* h, double lambda, const double* A, const double* B, const double* lx,
                     const double* lu, const double* lxx, const double* luu, double* Kfb, double* kff,
                     double* dV2) {
  static_cast<Oracle*>(h)->Backward(lambda, A, B, lx, lu, lxx, luu, Kfb, kff, dV2);
}
double oracle_grad_norm(void* h, const double* kff, const double* U) {
  return static_cast<Oracle*>(h)->GradNorm(kff, U);
}
void oracle_forward(void* h, double alpha, const double* X, const double* U, const double* Kfb,
                    const double* kff, double* Xn, double* Un) {
  static_cast<Oracle*>(h)->Forward(alpha, X, U, Kfb, kff, Xn, Un);
}
void oracle_dynamics(void* h, const double* x, const double* u, double* xn) {
  static_cast<Oracle*>(h)->Dynamics(x, u, xn);
}
void oracle_dynamics_jacobian(void* h, const double* x, const double* u, double* A, double* B) {
  static_cast<Oracle*>(h)->DynamicsJacobian(x, u, A, B);
}
double oracle_normalize_angle(double a) { return NormalizeAngle(a); }
double oracle_segment_distance(const double* seg4, double px, double py) {
  Segment s;
  s.Set(seg4[0], seg4[1], seg4[2], seg4[3]);
  return s.DistanceTo(px, py);
}
// index FindNearestLaneSegment picks among n segments (start x, y, end x, y each); held against the reference's own
// LineSegment2d by tests/test_reference_pins.py
int oracle_nearest_segment(void* h, const double* segs, int n, double px, double py) {
  std::vector<Lane> lanes(n);
  for (int i = 0; i < n; ++i) {
    lanes[i].a = lanes[i].b = lanes[i].c = 0.0;
    lanes[i].seg.Set(segs[4 * i], segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3]);
  }
  const Lane& l = static_cast<Oracle*>(h)->FindNearestLaneSegment(px, py, lanes);
  return static_cast<int>(&l - lanes.data());
}
double oracle_barrier_value(void* h, double g) { return static_cast<Oracle*>(h)->BarrierValue(g); }
void oracle_barrier_jacobian(void* h, double g, const double* dg, int n, double* out) {
  const double c = static_cast<Oracle*>(h)->BarrierJacCoef(g);
  for (int i = 0; i < n; ++i) out[i] = c * dg[i];
}
void oracle_barrier_hessian(void* h, double g, const double* dg, const double* ddg, int n, double* out) {
  Oracle* o = static_cast<Oracle*>(h);
  if (n == 6) o->BarrierHessian<6>(g, dg, ddg, out);
  else if (n == 2) o->BarrierHessian<2>(g, dg, ddg, out);
}

/* Batch driver: B independent Plan() calls, single thread.  seconds (nullable) receives the
 * steady_clock time of the set_problem+plan span (what the reference times at cc:82-93).
 * alpha_trace / iter_margin (nullable, [B][max_iter]): per iteration the accepted alpha index (-1 all
 * rejected, -2 gradient-norm exit, -3 iteration not run) and the trace's decision margin. */
int oracle_solve_batch_trace(const oracle_config* c, int B, const double* start, const double* coarse,
                             const double* corridor, const int* ccount, int cmax, const double* left,
                             int n_left, const double* right, int n_right, double* traj, double* cost_hist,
                             int* n_cost, int* status, int* n_iter, double* min_margin, double* seconds,
                             signed char* alpha_trace, double* iter_margin, double* problem_seconds) {
  Oracle* o = static_cast<Oracle*>(oracle_create(c));
  const int K = o->K, M = c->max_iter;
  const bool want_trace = alpha_trace != nullptr || iter_margin != nullptr;
  std::vector<double> trace(want_trace ? (size_t)M * kTraceCols : 0);
  const auto t0 = std::chrono::steady_clock::now();
  int rc = 0;
  for (int b = 0; b < B; ++b) {
    const auto tb = std::chrono::steady_clock::now();
    rc = o->SetProblem(start + (size_t)b * 4, coarse + (size_t)b * K * 6, K,
                       corridor + (size_t)b * K * cmax * 3, ccount + (size_t)b * K, cmax, left, n_left,
                       right, n_right);
    if (rc != 0) break;
    int iters = 0;
    rc = o->Plan(traj + (size_t)b * K * 10, cost_hist + (size_t)b * (M + 1) * 5, n_cost + b,
                 status + b, &iters, nullptr, 0, nullptr, want_trace ? trace.data() : nullptr,
                 min_margin ? min_margin + b : nullptr);
    if (rc != 0) break;
    if (problem_seconds)   // the span the reference times around Plan (ilqr_optimizer.cc:82-93)
      problem_seconds[b] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
    if (n_iter) n_iter[b] = iters;
    for (int i = 0; i < M && want_trace; ++i) {
      if (alpha_trace) alpha_trace[(size_t)b * M + i] = (i < iters) ? (signed char)trace[(size_t)i * kTraceCols] : -3;
      if (iter_margin) iter_margin[(size_t)b * M + i] = (i < iters) ? trace[(size_t)i * kTraceCols + 8] : 0.0;
    }
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  delete o;
  return rc;
}

int oracle_solve_batch(const oracle_config* c, int B, const double* start, const double* coarse,
                       const double* corridor, const int* ccount, int cmax, const double* left,
                       int n_left, const double* right, int n_right, double* traj, double* cost_hist,
                       int* n_cost, int* status, int* n_iter, double* min_margin, double* seconds) {
  return oracle_solve_batch_trace(c, B, start, coarse, corridor, ccount, cmax, left, n_left, right, n_right,
                                  traj, cost_hist, n_cost, status, n_iter, min_margin, seconds, nullptr, nullptr,
                                  nullptr);
}

// The same loop over problems on `n_threads` host threads (contiguous slices, one Oracle each, no shared state): what the
// box's cores give the reference's algorithm when a caller has many problems at once.  bench.py's cpu_baseline.all_cores.
int oracle_solve_batch_threads(const oracle_config* c, int B, const double* start, const double* coarse,
                               const double* corridor, const int* ccount, int cmax, const double* left, int n_left,
                               const double* right, int n_right, double* traj, double* cost_hist, int* n_cost,
                               int* status, int* n_iter, int n_threads, double* seconds) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > B) n_threads = B > 0 ? B : 1;
  const int K = c->n_steps + 1, M = c->max_iter;
  std::vector<int> rcs((size_t)n_threads, 0);
  std::vector<std::thread> pool;
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < n_threads; ++t) {
    const int b0 = (int)((long long)B * t / n_threads), b1 = (int)((long long)B * (t + 1) / n_threads);
    pool.emplace_back([=, &rcs]() {
      rcs[(size_t)t] = oracle_solve_batch(c, b1 - b0, start + (size_t)b0 * 4, coarse + (size_t)b0 * K * 6,
                                          corridor + (size_t)b0 * K * cmax * 3, ccount + (size_t)b0 * K, cmax, left, n_left,
                                          right, n_right, traj + (size_t)b0 * K * 10, cost_hist + (size_t)b0 * (M + 1) * 5,
                                          n_cost + b0, status + b0, n_iter ? n_iter + b0 : nullptr, nullptr, nullptr);
    });
  }
  for (std::thread& th : pool) th.join();
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int rc : rcs)
    if (rc != 0) return rc;
  return 0;
}

}  // extern "C"
