// Cost evaluation and quadratisation kernels: one thread per (problem, knot).
//
// Reference behaviour (algorithm/ilqr/ilqr_optimizer.cc):
//   TotalCost cc:417-436 = JCost cc:497 + DynamicsCost cc:518 + CorridorCost cc:553 +
//   LaneBoundaryCost cc:583;  CostJacbian cc:620-636 (+cc:657-671, 690-706, 729-746);
//   CostHessian cc:638-655 (+cc:673-688, 708-727, 748-769); DynamicsJacbian vehicle_model.cc:21-86.
//
// Work decomposition: grid = (ceil(n/256), K); blockIdx.y is the knot, threads walk a list of
// slots, so a wave reads 64 consecutive slots of one knot (coalesced) and the lane tables through
// scalar loads.  Per-knot cost partials go to `part`; a per-problem pass sums them over knots in
// index order (kernels_search.hip).
#include "dev_model.hpp"

namespace cilqr {

// ---------------------------------------------------------------------------------------------
// cost partials of knot i of buffer `buf`
// ---------------------------------------------------------------------------------------------
CILQR_DEV void knot_cost(const DeviceState& s, int buf, int i, int slot) {
  const Params& p = s.p;
  const int Bc = s.Bcap;
  double x[6];
  load_x(s, buf, i, slot, x);
  const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
  const double2 g0 = gp[0];
  const double gth = gp[(size_t)Bc].x;
  // JCost cc:501-513
  const double ex = x[0] - g0.x, ey = x[1] - g0.y, eth = x[2] - gth;
  const double jx = p.w_x * (ex * ex) + p.w_y * (ey * ey) + p.w_theta * (eth * eth);
  double ju = 0.0, du = 0.0;
  if (i < p.N) {
    double u[2];
    load_u(s, buf, i, slot, u);
    ju = p.w_jerk * (u[0] * u[0]) + p.w_delta_rate * (u[1] * u[1]);
    du += bar_value(p, u[0] - p.jerk_max);          // cc:543-546
    du += bar_value(p, p.jerk_min - u[0]);
    du += bar_value(p, u[1] - p.delta_rate_max);
    du += bar_value(p, p.delta_rate_min - u[1]);
  }
  double dx = 0.0;                                  // cc:523-528
  dx += bar_value(p, -x[3]);
  dx += bar_value(p, x[3] - p.max_velocity);
  dx += bar_value(p, x[4] - p.max_acc);
  dx += bar_value(p, p.min_acc - x[4]);
  dx += bar_value(p, x[5] - p.delta_max);
  dx += bar_value(p, p.delta_min - x[5]);
  double sn, cs;
  sincos(x[2], &sn, &cs);
  const int cnt = s.ccnt[(size_t)i * Bc + slot];
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
  double ccost = 0.0, lcost = 0.0;
  for (int j = 0; j < p.num_of_disc; ++j) {
    const double px = x[0] + p.disc_off[j] * cs;
    const double py = x[1] + p.disc_off[j] * sn;
    for (int c = 0; c < cnt; ++c) {                 // cc:566-574
      const double* q = cor + (size_t)c * 3 * Bc;
      const double a = q[0], b = q[(size_t)Bc], cc = q[(size_t)2 * Bc];
      ccost += bar_value(p, a * px + b * py - cc);
    }
    {                                               // cc:594-598
      const double* L = s.lanes + nearest_segment(s.lanes, s.nl, px, py) * kLaneFields;
      lcost += bar_value(p, L[0] * px + L[1] * py - L[2]);
      const double* Rr = s.lanes + (s.nl + nearest_segment(s.lanes + s.nl * kLaneFields, s.nr, px, py)) * kLaneFields;
      lcost += bar_value(p, Rr[0] * px + Rr[1] * py - Rr[2]);
    }
  }
  double2* o = s.part + (size_t)i * kPartPairs * Bc + slot;
  o[0] = make_double2(jx, ju);
  o[(size_t)Bc] = make_double2(dx, du);
  o[(size_t)2 * Bc] = make_double2(ccost, lcost);
}

// list == nullptr: slots 0..n-1.  skip_done: ignore slots that already left the iteration.
__global__ __launch_bounds__(256) void k_cost_knots(DeviceState s, const int* __restrict__ list,
                                                    const int* __restrict__ n_ptr, int n_max, int cand,
                                                    int skip_done) {
  const int n = n_ptr ? min(*n_ptr, n_max) : n_max;
  const int i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list ? list[j] : j;
    if (skip_done && s.acc_idx[slot] != -1) continue;
    knot_cost(s, s.cur[slot] ^ cand, i, slot);
  }
}

// sum of the knot partials in index order -> trial[5] (total, J, dynamics, corridor, lane)
CILQR_DEV void reduce_cost(const DeviceState& s, int slot, double* c5) {
  const int Bc = s.Bcap, K = s.p.K, N = s.p.N;
  double j = 0.0, dx = 0.0, du = 0.0, cc = 0.0, lc = 0.0;
  for (int i = 0; i < K; ++i) {
    const double2* o = s.part + (size_t)i * kPartPairs * Bc + slot;
    const double2 a = o[0], b = o[(size_t)Bc], c = o[(size_t)2 * Bc];
    j += a.x;
    dx += b.x;
    cc += c.x;
    lc += c.y;
  }
  for (int i = 0; i < N; ++i) {   // control terms follow the state terms (cc:510-513)
    const double2* o = s.part + (size_t)i * kPartPairs * Bc + slot;
    j += o[0].y;
    du += o[(size_t)Bc].y;
  }
  const double dyn = dx + du;                      // cc:550
  c5[0] = j + dyn + cc + lc;                       // cc:429
  c5[1] = j; c5[2] = dyn; c5[3] = cc; c5[4] = lc;
}

__global__ __launch_bounds__(64) void k_reduce_only(DeviceState s, const int* __restrict__ list, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int slot = list ? list[j] : j;
  double c5[5];
  reduce_cost(s, slot, c5);
#pragma unroll
  for (int c = 0; c < 5; ++c) s.trial[(size_t)c * s.Bcap + slot] = c5[c];
}

// n_max bounds the list length; n_grid sizes the grid (threads stride over the rest)
void launch_cost_knots(const DeviceState& s, const int* list, const int* n_ptr, int n_max, int n_grid,
                       int cand, int skip_done, hipStream_t st) {
  if (n_grid <= 0) return;
  dim3 g((n_grid + 255) / 256, s.p.K);
  hipLaunchKernelGGL(k_cost_knots, g, dim3(256), 0, st, s, list, n_ptr, n_max, cand, skip_done);
}

void launch_cost_only(const DeviceState& s, const int* list, int n, int cand, hipStream_t st) {
  if (n == 0) return;
  dim3 g((n + 255) / 256, s.p.K);
  hipLaunchKernelGGL(k_cost_knots, g, dim3(256), 0, st, s, list, (const int*)nullptr, n, cand, 0);
  hipLaunchKernelGGL(k_reduce_only, dim3((n + 63) / 64), dim3(64), 0, st, s, list, n);
}

// cost of the init guess -> cost_old, cost_[0]; iter_trajs gets the init guess (cc:170-173)
__global__ void k_init_cost_commit(DeviceState s, int n) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
#pragma unroll
  for (int c = 0; c < 5; ++c) s.hist[(size_t)c * s.Bcap + slot] = s.trial[(size_t)c * s.Bcap + slot];
  s.cost_old[slot] = s.trial[slot];
  s.n_cost[slot] = 1;
  s.n_iter_trajs[slot] = 1;
  s.emit[slot] = 1;
}
void launch_init_cost_commit(const DeviceState& s, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_init_cost_commit, dim3((n + 255) / 256), dim3(256), 0, st, s, n);
}

// ---------------------------------------------------------------------------------------------
// quadratisation of knot i: A, B, lx, lu, lxx, luu (terminal knot: lx, lxx with u = 0)
// ---------------------------------------------------------------------------------------------
struct Quad {
  double lx[6];
  double lu[2];
  double h[9];     // lxx rows/cols 0..2 (full 3x3; the reference block is not bitwise symmetric)
  double hd[3];    // lxx(3,3), (4,4), (5,5)
  double huu[2];   // luu(0,0), (1,1)
};

CILQR_DEV void add_plane(const Params& p, Quad& q, double a, double b, double c, double px, double py,
                         double lc, double ls) {
  const double g = a * px + b * py - c;
  const double d[3] = {a, b, -a * ls + b * lc};
  const double jc = bar_jcoef(p, g);
#pragma unroll
  for (int e = 0; e < 3; ++e) q.lx[e] += jc * d[e];
  double c1, c2;
  bool lg;
  bar_hcoef(p, g, c1, c2, lg);
  const double dd22 = -a * lc - b * ls;            // cc:723
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const double ce = c1 * d[e];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      double v = ce * d[f];
      if (lg) v = v - c2 * ((e == 2 && f == 2) ? dd22 : 0.0);
      q.h[e * 3 + f] += v;
    }
  }
}

CILQR_DEV void knot_quadratize(const DeviceState& s, int buf, int i, int slot) {
  const Params& p = s.p;
  const int Bc = s.Bcap;
  const bool term = (i == p.N);
  double x[6], u[2] = {0.0, 0.0};
  load_x(s, buf, i, slot, x);
  if (!term) load_u(s, buf, i, slot, u);
  const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
  const double2 g0 = gp[0];
  const double gth = gp[(size_t)Bc].x;
  Quad q;
  q.lx[0] = 2.0 * p.w_x * (x[0] - g0.x);           // cc:623-628
  q.lx[1] = 2.0 * p.w_y * (x[1] - g0.y);
  q.lx[2] = 2.0 * p.w_theta * (x[2] - gth);
  q.lx[3] = 0.0; q.lx[4] = 0.0; q.lx[5] = 0.0;
  q.lu[0] = 2.0 * p.w_jerk * u[0];                 // cc:630-631
  q.lu[1] = 2.0 * p.w_delta_rate * u[1];
#pragma unroll
  for (int e = 0; e < 9; ++e) q.h[e] = 0.0;
  q.h[0] = 2.0 * p.w_x; q.h[4] = 2.0 * p.w_y; q.h[8] = 2.0 * p.w_theta;   // cc:642-647
  q.hd[0] = 2.0 * p.w_v; q.hd[1] = 2.0 * p.w_a; q.hd[2] = 2.0 * p.w_delta;
  q.huu[0] = 2.0 * p.w_jerk; q.huu[1] = 2.0 * p.w_delta_rate;             // cc:649-650
  // state / control bounds (cc:657-688): lower bound first, the pair is summed, then added
  {
    const double gl[3] = {0.0 - x[3], p.min_acc - x[4], p.delta_min - x[5]};
    const double gh[3] = {x[3] - p.max_velocity, x[4] - p.max_acc, x[5] - p.delta_max};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      q.lx[3 + e] += bar_jcoef(p, gl[e]) * -1.0 + bar_jcoef(p, gh[e]) * 1.0;
      double c1l, c1h, c2;
      bool lg;
      bar_hcoef(p, gl[e], c1l, c2, lg);
      bar_hcoef(p, gh[e], c1h, c2, lg);
      q.hd[e] += c1l + c1h;
    }
    const double ul[2] = {p.jerk_min - u[0], p.delta_rate_min - u[1]};
    const double uh[2] = {u[0] - p.jerk_max, u[1] - p.delta_rate_max};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      q.lu[e] += bar_jcoef(p, ul[e]) * -1.0 + bar_jcoef(p, uh[e]) * 1.0;
      double c1l, c1h, c2;
      bool lg;
      bar_hcoef(p, ul[e], c1l, c2, lg);
      bar_hcoef(p, uh[e], c1h, c2, lg);
      q.huu[e] += c1l + c1h;
    }
  }
  double sn, cs;
  sincos(x[2], &sn, &cs);
  const int cnt = s.ccnt[(size_t)i * Bc + slot];
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
  // corridor planes, all discs (cc:690-727)
  for (int j = 0; j < p.num_of_disc; ++j) {
    const double lc = p.disc_off[j] * cs, ls = p.disc_off[j] * sn;
    const double px = x[0] + lc, py = x[1] + ls;
    for (int c = 0; c < cnt; ++c) {
      const double* r = cor + (size_t)c * 3 * Bc;
      add_plane(p, q, r[0], r[(size_t)Bc], r[(size_t)2 * Bc], px, py, lc, ls);
    }
  }
  // nearest left / right lane plane, all discs (cc:729-769)
  for (int j = 0; j < p.num_of_disc; ++j) {
    const double lc = p.disc_off[j] * cs, ls = p.disc_off[j] * sn;
    const double px = x[0] + lc, py = x[1] + ls;
    const double* L = s.lanes + nearest_segment(s.lanes, s.nl, px, py) * kLaneFields;
    add_plane(p, q, L[0], L[1], L[2], px, py, lc, ls);
    const double* Rr = s.lanes + (s.nl + nearest_segment(s.lanes + s.nl * kLaneFields, s.nr, px, py)) * kLaneFields;
    add_plane(p, q, Rr[0], Rr[1], Rr[2], px, py, lc, ls);
  }
  if (term) {
    double2* o = s.term + slot;
    o[0] = make_double2(q.lx[0], q.lx[1]);
    o[(size_t)Bc] = make_double2(q.lx[2], q.lx[3]);
    o[(size_t)2 * Bc] = make_double2(q.lx[4], q.lx[5]);
    o[(size_t)3 * Bc] = make_double2(q.h[0], q.h[1]);
    o[(size_t)4 * Bc] = make_double2(q.h[2], q.h[3]);
    o[(size_t)5 * Bc] = make_double2(q.h[4], q.h[5]);
    o[(size_t)6 * Bc] = make_double2(q.h[6], q.h[7]);
    o[(size_t)7 * Bc] = make_double2(q.h[8], q.hd[0]);
    o[(size_t)8 * Bc] = make_double2(q.hd[1], q.hd[2]);
    return;
  }
  DynJac J;
  dynamics_jacobian(p, x, u, J);
  double2* o = s.lin + (size_t)i * kLinPairs * Bc + slot;
  o[(size_t)0 * Bc] = make_double2(J.a02, J.a03);
  o[(size_t)1 * Bc] = make_double2(J.a04, J.a05);
  o[(size_t)2 * Bc] = make_double2(J.a12, J.a13);
  o[(size_t)3 * Bc] = make_double2(J.a14, J.a15);
  o[(size_t)4 * Bc] = make_double2(J.a23, J.a24);
  o[(size_t)5 * Bc] = make_double2(J.a25, J.b21);
  o[(size_t)6 * Bc] = make_double2(q.lx[0], q.lx[1]);
  o[(size_t)7 * Bc] = make_double2(q.lx[2], q.lx[3]);
  o[(size_t)8 * Bc] = make_double2(q.lx[4], q.lx[5]);
  o[(size_t)9 * Bc] = make_double2(q.lu[0], q.lu[1]);
  o[(size_t)10 * Bc] = make_double2(q.h[0], q.h[1]);
  o[(size_t)11 * Bc] = make_double2(q.h[2], q.h[3]);
  o[(size_t)12 * Bc] = make_double2(q.h[4], q.h[5]);
  o[(size_t)13 * Bc] = make_double2(q.h[6], q.h[7]);
  o[(size_t)14 * Bc] = make_double2(q.h[8], q.hd[0]);
  o[(size_t)15 * Bc] = make_double2(q.hd[1], q.hd[2]);
  o[(size_t)16 * Bc] = make_double2(q.huu[0], q.huu[1]);
}

__global__ __launch_bounds__(256) void k_quadratize(DeviceState s, const int* __restrict__ list, int n,
                                                    int only_upd) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int slot = list ? list[j] : j;
  if (only_upd && !s.upd[slot]) return;
  knot_quadratize(s, s.cur[slot], blockIdx.y, slot);
}

void launch_quadratize(const DeviceState& s, const int* list, int n, int only_upd, hipStream_t st) {
  if (n == 0) return;
  dim3 g((n + 255) / 256, s.p.K);
  hipLaunchKernelGGL(k_quadratize, g, dim3(256), 0, st, s, list, n, only_upd);
}

}  // namespace cilqr
