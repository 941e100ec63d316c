// Per-knot cost and quadratisation (device functions shared by the lockstep kernels of kernels_quad.hip and the
// per-problem tail kernel of kernels_tail.hip).
//
// Reference behaviour (algorithm/ilqr/ilqr_optimizer.cc):
//   TotalCost cc:417-436 = JCost cc:497 + DynamicsCost cc:518 + CorridorCost cc:553 + LaneBoundaryCost cc:583;
//   CostJacbian cc:620-636 (+cc:657-671, 690-706, 729-746); CostHessian cc:638-655 (+cc:673-688, 708-727, 748-769);
//   DynamicsJacbian vehicle_model.cc:21-86.
#pragma once
#include "cost_reduce.hpp"

namespace cilqr {

// corridor planes are read in chunks (all loads of a chunk in flight at once).  Both kernels keep their chunks at two
// planes: with that, the lane-grid cells fetched disc by disc, the disc centres recomputed in the lane loop and
// finished outputs stored as soon as they are complete, the cost function needs 114-120 VGPRs and the quadratisation
// 127 -- four waves per SIMD, no spills (kernels_quad.hip, Makefile: QUADFLAGS).  Three-plane chunks spill at that budget.
#ifndef CILQR_COST_CHUNK
#define CILQR_COST_CHUNK 2
#endif
#ifndef CILQR_QUAD_CHUNK
#define CILQR_QUAD_CHUNK 2
#endif
constexpr int kCostChunk = CILQR_COST_CHUNK;
constexpr int kQuadChunk = CILQR_QUAD_CHUNK;

// lane tables -> LDS (call from every thread of the block, before any early exit)
CILQR_DEV const double* stage_lanes(const DeviceState& s, double* lds) {
  const int n = (s.nl + s.nr) * kLaneFields;
  for (int e = threadIdx.x; e < n; e += blockDim.x) lds[e] = s.lanes[e];
  __syncthreads();
  return lds;
}
static inline size_t lane_lds_bytes(const DeviceState& s) {
  return (size_t)(s.nl + s.nr) * kLaneFields * sizeof(double);
}

// one chunk of up to C corridor planes; missing planes are (0, 0, 1): g = -1, which multiplies
// the barrier product by exactly 1 and adds exact zeros to every gradient / Hessian entry
template <int C>
struct PlaneChunk {
  double a[C], b[C], c[C];
};
template <int C>
CILQR_DEV void load_chunk(const double* __restrict__ cor, int Bc, int c0, int cnt, PlaneChunk<C>& pc) {
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const bool live = (c0 + k) < cnt;
    const double* q = cor + (size_t)(live ? (c0 + k) : 0) * 3 * Bc;
    const double a = q[0], b = q[(size_t)Bc], c = q[(size_t)2 * Bc];
    pc.a[k] = live ? a : 0.0;
    pc.b[k] = live ? b : 0.0;
    pc.c[k] = live ? c : 1.0;
  }
}

// first chunk of a knot: requested before the plane count is known (its addresses do not depend on
// it), masked once the count has arrived -- one dependent memory round trip less per knot
template <int C>
CILQR_DEV void load_first_chunk(const double* __restrict__ cor, int Bc, int cmax, PlaneChunk<C>& pc) {
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const double* q = cor + (size_t)min(k, cmax - 1) * 3 * Bc;
    pc.a[k] = q[0];
    pc.b[k] = q[(size_t)Bc];
    pc.c[k] = q[(size_t)2 * Bc];
  }
}
template <int C>
CILQR_DEV void mask_first_chunk(int cnt, PlaneChunk<C>& pc) {
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const bool live = k < cnt;
    pc.a[k] = live ? pc.a[k] : 0.0;
    pc.b[k] = live ? pc.b[k] : 0.0;
    pc.c[k] = live ? pc.c[k] : 1.0;
  }
}

// ---------------------------------------------------------------------------------------------
// cost partials of one knot.  x, u: the knot's state / control.  Three pairs: out[0] = (J, bound barriers) of the STATE,
// out[stride] = the same of the CONTROL, out[2*stride] = (corridor, lane).  The totals add all state terms before the first
// control term (cc:510-513, 550): a reader sums pairs 0 and 2 over the knots, then pair 1 over the steps -- every pair is
// read once (until round 5 the pairs were (J state, J control), (bounds state, bounds control): two of them read twice)
// ---------------------------------------------------------------------------------------------
// bound barriers of one knot (DynamicsCost cc:518-551); returns {state part, control part}
CILQR_DEV double2 knot_bound_cost(const Params& p, int i, const double* x, const double* u) {
  double du = 0.0;
  if (i < p.N) {
    BarGroup g;
    const double gu[4] = {u[0] - p.jerk_max, p.jerk_min - u[0],                    // cc:543-546
                          u[1] - p.delta_rate_max, p.delta_rate_min - u[1]};
    bar_accumulate(p, gu, g);
    du = bar_group_value(p, g);
  }
  BarGroup g;
  const double gx[6] = {-x[3], x[3] - p.max_velocity, x[4] - p.max_acc,            // cc:523-528
                        p.min_acc - x[4], x[5] - p.delta_max, p.delta_min - x[5]};
  bar_accumulate(p, gx, g);
  return make_double2(bar_group_value(p, g), du);
}

#ifdef CILQR_COST_PROFILE
// Tuning build only (make OBJDIR=build/costprof OUT=../lib/variants/libcilqr_hip_costprof.so EXTRA=-DCILQR_COST_PROFILE;
// tools/cost_phase_profile.py): wall-clock stamps (100 MHz) of the phases of a knot cost, one record per wave.
constexpr int kCostProfWaves = 1 << 16;
__device__ unsigned long long g_cost_prof[kCostProfWaves * 8];
#define CP_WAVE ((int)(((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kCostProfWaves - 1)))
#define CP_STAMP(k) do { const unsigned long long n_ = wall_clock64(); if ((threadIdx.x & 63) == 0) g_cost_prof[CP_WAVE * 8 + (k)] = n_ - cp_t; cp_t = n_; } while (0)
#else
#define CP_STAMP(k)
#endif
// InLds: goals, cor, ccnt of the view and `out` are in LDS (the tail kernel; dev_model.hpp: assume_lds)
template <int D, bool EX, bool InLds = false>
CILQR_DEV void knot_cost_core(const DeviceState& s, const double* __restrict__ lanes, int i, int slot,
                              const double* x, const double* u, double2* __restrict__ out, size_t stride) {
  constexpr int C = kCostChunk;
#ifdef CILQR_COST_PROFILE
  unsigned long long cp_t = wall_clock64();
#endif
  const Params& p = s.p;
  const int Bc = s.Bcap;
  const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
  assume_lds<InLds>(gp);
  const double2 g0 = gp[0];
  const double gth = gp[(size_t)Bc].x;
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
  assume_lds<InLds>(cor);
  assume_lds<InLds>(out);
  PlaneChunk<C> pc;
  load_first_chunk(cor, Bc, s.cmax, pc);
  const int* cntp = s.ccnt + (size_t)i * Bc + slot;
  assume_lds<InLds>(cntp);
  const int cnt = *cntp;
  mask_first_chunk(cnt, pc);
  // JCost cc:501-513
  const double ex = x[0] - g0.x, ey = x[1] - g0.y, eth = x[2] - gth;
  const double jx = p.w_x * (ex * ex) + p.w_y * (ey * ey) + p.w_theta * (eth * eth);
  const double ju = (i < p.N) ? p.w_jerk * (u[0] * u[0]) + p.w_delta_rate * (u[1] * u[1]) : 0.0;
  const double2 dyn = knot_bound_cost(p, i, x, u);
  double sn, cs;
  lean_sincos(x[2], &sn, &cs);
#ifdef CILQR_COST_PROFILE
  asm volatile("" :: "v"(jx), "v"(dyn.x), "v"(sn), "v"(pc.a[0]));
  CP_STAMP(2);   // operands arrived + J, bounds, sincos
#endif
  static_assert(D == 5, "the lane loop below selects among five disc offsets");
  const double x0 = x[0], x1 = x[1];
  double px[D], py[D];
  BarGroup grp[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    px[j] = x0 + p.disc_off[j] * cs;                       // cc:564-565
    py[j] = x1 + p.disc_off[j] * sn;
  }
  // the partial sums that are complete leave now (eight registers less through the two loops below)
  out[0] = make_double2(jx, dyn.x);
  out[stride] = make_double2(ju, dyn.y);
  // CorridorCost cc:553-581: planes outer (each read once), discs inner; one log for the knot
  for (int c0 = 0; c0 < cnt; c0 += C) {
    PlaneChunk<C> nx;
    if (c0 + C < cnt) load_chunk(cor, Bc, c0 + C, cnt, nx);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double g[C];
#pragma unroll
      for (int k = 0; k < C; ++k) g[k] = pc.a[k] * px[j] + pc.b[k] * py[j] - pc.c[k];
      bar_accumulate(p, g, grp[j]);
    }
    if ((c0 & 63) == 64 - C) {   // every 64 planes: keep the products in range
#pragma unroll
      for (int j = 0; j < D; ++j) bar_renormalize(grp[j]);
    }
    pc = nx;
  }
  BarGroup call;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    bar_renormalize(grp[j]);
    bar_merge(call, grp[j]);
  }
  const double ccost = bar_group_value(p, call);
#ifdef CILQR_COST_PROFILE
  asm volatile("" :: "v"(ccost));
  CP_STAMP(3);   // corridor
#endif
  // LaneBoundaryCost cc:583-603, disc by disc (a rolled loop: fetching the ten candidate lists up front hides
  // their latency but holds 40 registers through the searches, which costs the third wave per SIMD)
  BarGroup lall;
#ifdef CILQR_COST_PROFILE
  unsigned long long cp_fetch = 0, cp_search = 0;
#endif
#pragma unroll 1
  for (int j = 0; j < D; ++j) {
#ifdef CILQR_COST_PROFILE
    const unsigned long long cp_a = wall_clock64();
#endif
    // the disc's centre again from the state (the same expression as above, so the same bits) instead of px[j], py[j]:
    // indexing the register arrays with the loop counter costs a chain of selects per access and keeps all ten
    // values alive through the searches
    const double doff = (j == 0) ? p.disc_off[0] : (j == 1) ? p.disc_off[1] : (j == 2) ? p.disc_off[2] : (j == 3) ? p.disc_off[3] : p.disc_off[4];
    const double pxj = x0 + doff * cs, pyj = x1 + doff * sn;
    const uint4 cl = lane_cell_fetch(s, 0, pxj, pyj);
    const uint4 cr = lane_cell_fetch(s, 1, pxj, pyj);
#ifdef CILQR_COST_PROFILE
    asm volatile("" :: "v"(cl.x), "v"(cr.x));
    const unsigned long long cp_b = wall_clock64();
#endif
    const double* L = lanes + nearest_from_cell<EX>(s, lanes, 0, cl, pxj, pyj) * kLaneFields;
    const double* Rr = lanes + (s.nl + nearest_from_cell<EX>(s, lanes, 1, cr, pxj, pyj)) * kLaneFields;
    const double g[2] = {L[0] * pxj + L[1] * pyj - L[2], Rr[0] * pxj + Rr[1] * pyj - Rr[2]};
    bar_accumulate(p, g, lall);
#ifdef CILQR_COST_PROFILE
    asm volatile("" :: "v"(lall.prod));
    cp_fetch += cp_b - cp_a;
    cp_search += wall_clock64() - cp_b;
#endif
  }
  const double lcost = bar_group_value(p, lall);
#ifdef CILQR_COST_PROFILE
  asm volatile("" :: "v"(lcost));
  CP_STAMP(4);   // lanes
  if ((threadIdx.x & 63) == 0) g_cost_prof[CP_WAVE * 8 + 7] = (cp_fetch << 32) | (cp_search & 0xffffffffull);   // cell words / searches
#endif
  out[2 * stride] = make_double2(ccost, lcost);
}

// discs as a run-time count (any num_of_disc != 5): same arithmetic, disc-major loops
template <bool EX>
CILQR_DEV void knot_cost_generic(const DeviceState& s, const double* __restrict__ lanes, int i, int slot,
                                 const double* x, const double* u, double2* __restrict__ out, size_t stride) {
  const Params& p = s.p;
  const int Bc = s.Bcap;
  const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
  const double2 g0 = gp[0];
  const double gth = gp[(size_t)Bc].x;
  const double ex = x[0] - g0.x, ey = x[1] - g0.y, eth = x[2] - gth;
  const double jx = p.w_x * (ex * ex) + p.w_y * (ey * ey) + p.w_theta * (eth * eth);
  const double ju = (i < p.N) ? p.w_jerk * (u[0] * u[0]) + p.w_delta_rate * (u[1] * u[1]) : 0.0;
  const double2 dyn = knot_bound_cost(p, i, x, u);
  double sn, cs;
  lean_sincos(x[2], &sn, &cs);
  const int cnt = s.ccnt[(size_t)i * Bc + slot];
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
  BarGroup call, lall;
  for (int j = 0; j < p.num_of_disc; ++j) {
    const double px = x[0] + p.disc_off[j] * cs, py = x[1] + p.disc_off[j] * sn;
    for (int c = 0; c < cnt; ++c) {
      const double* q = cor + (size_t)c * 3 * Bc;
      const double g[1] = {q[0] * px + q[(size_t)Bc] * py - q[(size_t)2 * Bc]};
      bar_accumulate(p, g, call);
      if ((c & 63) == 63) bar_renormalize(call);
    }
    bar_renormalize(call);
    const double* L = lanes + nearest_segment<EX>(s, lanes, 0, px, py) * kLaneFields;
    const double* Rr = lanes + (s.nl + nearest_segment<EX>(s, lanes, 1, px, py)) * kLaneFields;
    const double g[2] = {L[0] * px + L[1] * py - L[2], Rr[0] * px + Rr[1] * py - Rr[2]};
    bar_accumulate(p, g, lall);
    bar_renormalize(lall);
  }
  out[0] = make_double2(jx, dyn.x);
  out[stride] = make_double2(ju, dyn.y);
  out[2 * stride] = make_double2(bar_group_value(p, call), bar_group_value(p, lall));
}

// D = 5: the reference's disc count, unrolled (knot_cost_core); D = 0: any other count
template <int D, bool EX = false, bool InLds = false>
CILQR_DEV void knot_cost(const DeviceState& s, const double* __restrict__ lanes, int i, int slot,
                         const double* x, const double* u, double2* __restrict__ out, size_t stride) {
  if constexpr (D == 5) knot_cost_core<5, EX, InLds>(s, lanes, i, slot, x, u, out, stride);
  else knot_cost_generic<EX>(s, lanes, i, slot, x, u, out, stride);
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

// One half-plane g = a x + b y - c seen from the discs of a knot, disc point (px, py) =
// (x, y) + (lc, ls).  dg = (a, b, d2) with d2 = -a ls + b lc (cc:703), ddg(2,2) = -a lc - b ls
// (cc:723-724).  a and b are the same for every disc, so the sums over the discs of
//   jc dg            = (a T0, b T0, T1)
//   c1 dg dg^T       = [a a S0, a b S0, a S1; . , b b S0, b S1; . , . , S2]
//   c2 ddg           = W   (entry (2,2) only; c2 = 0 on the relaxed branch)
// need five running sums per plane instead of a 3-vector and a 3x3 update per (plane, disc).
// This re-associates the reference's accumulation (and makes the 3x3 block exactly symmetric,
// which the reference's is only to rounding).
struct PlaneSums {
  double T0 = 0.0, T1 = 0.0, S0 = 0.0, S1 = 0.0, S2 = 0.0, W = 0.0;
};
CILQR_DEV void plane_disc(const Params& p, double a, double b, double c, double px, double py, double lc,
                          double ls, PlaneSums& m) {
  const double g = a * px + b * py - c;
  const double d2 = -a * ls + b * lc;
  const double dd22 = -a * lc - b * ls;
  double jc, c1, c2;
  bool lg;
  bar_coefs(p, g, jc, c1, c2, lg);
  m.T0 += jc;
  m.T1 += jc * d2;
  m.S0 += c1;
  const double t = c1 * d2;
  m.S1 += t;
  m.S2 += t * d2;
  m.W += c2 * dd22;
}
CILQR_DEV void plane_commit(Quad& q, double a, double b, const PlaneSums& m) {
  q.lx[0] += a * m.T0;
  q.lx[1] += b * m.T0;
  q.lx[2] += m.T1;
  const double aS = a * m.S0, bS = b * m.S0;
  const double h01 = aS * b, h02 = a * m.S1, h12 = b * m.S1;
  // upper triangle only: the mirrored entries receive the same addends in the same order, so they are copies
  // (quad_mirror, once per knot) -- three running sums less in registers
  q.h[0] += aS * a; q.h[1] += h01; q.h[2] += h02;
  q.h[4] += bS * b; q.h[5] += h12;
  q.h[8] += m.S2 - m.W;
}
CILQR_DEV void quad_mirror(Quad& q) {
  q.h[3] = q.h[1];
  q.h[6] = q.h[2];
  q.h[7] = q.h[5];
}

#ifdef CILQR_QUAD_PROFILE
// Tuning build only (make OBJDIR=build/quadprof OUT=../lib/variants/libcilqr_hip_quadprof.so EXTRA=-DCILQR_QUAD_PROFILE;
// tools/quad_phase_profile.py): wall-clock stamps (100 MHz) of the phases of a knot's quadratisation, one record per wave.
constexpr int kQuadProfWaves = 1 << 16;
__device__ unsigned long long g_quad_prof[kQuadProfWaves * 8];
#define QP_WAVE ((int)(((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kQuadProfWaves - 1)))
#define QP_STAMP(k) do { const unsigned long long n_ = wall_clock64(); if ((threadIdx.x & 63) == 0) g_quad_prof[QP_WAVE * 8 + (k)] = n_ - qp_t; qp_t = n_; } while (0)
#else
#define QP_STAMP(k)
#endif
// InLds: X, U, goals, cor, ccnt, lin, term of the view are in LDS (the tail kernel; dev_model.hpp: assume_lds)
// OnlyFirstPart: stop after the part that is a function of the knot's state and control alone -- the bounds' barriers,
// the dynamics' Jacobian and the entries of lx / lu / lxx / luu no corridor or lane plane touches (the tail kernel's
// split quadratisation evaluates the planes on other lanes: knot_plane_items / knot_commit_items below).
template <int D, bool EX = false, bool InLds = false, bool OnlyFirstPart = false>
CILQR_DEV void knot_quadratize(const DeviceState& s, const double* __restrict__ lanes, int buf, int i, int slot) {
#ifdef CILQR_QUAD_PROFILE
  unsigned long long qp_t = wall_clock64();
  const unsigned long long qp_t0 = qp_t;
#endif
  const Params& p = s.p;
  const int Bc = s.Bcap;
  const bool term = (i == p.N);
  const int nd = (D > 0) ? D : p.num_of_disc;
  double x[6], u[2] = {0.0, 0.0};
  {
    const double2* xb = s.X + ((size_t)buf * p.K + i) * 3 * Bc + slot;
    assume_lds<InLds>(xb);
    const double2 p0 = xb[0], p1 = xb[(size_t)Bc], p2 = xb[(size_t)2 * Bc];
    x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y;
    if (!term) {
      const double2* ub = s.U + ((size_t)buf * p.N + i) * Bc + slot;
      assume_lds<InLds>(ub);
      const double2 uq = *ub;
      u[0] = uq.x; u[1] = uq.y;
    }
  }
  const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
  assume_lds<InLds>(gp);
  const double2 g0 = gp[0];
  const double gth = gp[(size_t)Bc].x;
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3 * Bc + slot;
  assume_lds<InLds>(cor);
  constexpr int C = kQuadChunk;
  PlaneChunk<C> pc;
  load_first_chunk(cor, Bc, s.cmax, pc);
  const int* cntp = s.ccnt + (size_t)i * Bc + slot;
  assume_lds<InLds>(cntp);
  const int cnt = *cntp;
  mask_first_chunk(cnt, pc);
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
      double jl, jh, c1l, c1h, c2;
      bool lg;
      bar_coefs(p, gl[e], jl, c1l, c2, lg);
      bar_coefs(p, gh[e], jh, c1h, c2, lg);
      q.lx[3 + e] += jl * -1.0 + jh * 1.0;
      q.hd[e] += c1l + c1h;
    }
    const double ul[2] = {p.jerk_min - u[0], p.delta_rate_min - u[1]};
    const double uh[2] = {u[0] - p.jerk_max, u[1] - p.delta_rate_max};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      double jl, jh, c1l, c1h, c2;
      bool lg;
      bar_coefs(p, ul[e], jl, c1l, c2, lg);
      bar_coefs(p, uh[e], jh, c1h, c2, lg);
      q.lu[e] += jl * -1.0 + jh * 1.0;
      q.huu[e] += c1l + c1h;
    }
  }
  // What is complete at this point leaves now -- the dynamics' Jacobian (a function of x, u alone) and the entries
  // the constraints below do not touch -- so that v, a, delta, the control and eight sums are dead through the two
  // loops that follow (26 registers less where the pressure is highest; same values, same places as before).
  if (term) {
    double2* o = s.term + scratch_index(s, slot);
    assume_lds<InLds>(o);
    o[(size_t)Bc].y = q.lx[3];
    o[(size_t)2 * Bc] = make_double2(q.lx[4], q.lx[5]);
    o[(size_t)7 * Bc].y = q.hd[0];
    o[(size_t)8 * Bc] = make_double2(q.hd[1], q.hd[2]);
  } else {
    DynJac J;
    dynamics_jacobian(p, x, u, J);
    double2* o = s.lin + (size_t)i * kLinPairs * Bc + scratch_index(s, slot);
    assume_lds<InLds>(o);
    o[(size_t)0 * Bc] = make_double2(J.a02, J.a03);
    o[(size_t)1 * Bc] = make_double2(J.a04, J.a05);
    o[(size_t)2 * Bc] = make_double2(J.a12, J.a13);
    o[(size_t)3 * Bc] = make_double2(J.a14, J.a15);
    o[(size_t)4 * Bc] = make_double2(J.a23, J.a24);
    o[(size_t)5 * Bc] = make_double2(J.a25, J.b21);
    o[(size_t)7 * Bc].y = q.lx[3];
    o[(size_t)8 * Bc] = make_double2(q.lx[4], q.lx[5]);
    o[(size_t)9 * Bc] = make_double2(q.lu[0], q.lu[1]);
    o[(size_t)14 * Bc].y = q.hd[0];
    o[(size_t)15 * Bc] = make_double2(q.hd[1], q.hd[2]);
    o[(size_t)16 * Bc] = make_double2(q.huu[0], q.huu[1]);
  }
  if constexpr (OnlyFirstPart) return;
  double sn, cs;
  lean_sincos(x[2], &sn, &cs);
#ifdef CILQR_QUAD_PROFILE
  asm volatile("" :: "v"(sn), "v"(cs), "v"(pc.a[0]), "v"(q.lx[0]));
  QP_STAMP(0);   // state, goals, first planes arrived; bounds, Jacobian, early stores, sincos
#endif
  // corridor planes x discs (cc:690-727); planes outer (each read once), discs inner
  for (int c0 = 0; c0 < cnt; c0 += C) {
    PlaneChunk<C> nx;
    if (c0 + C < cnt) load_chunk(cor, Bc, c0 + C, cnt, nx);
#pragma unroll
    for (int k = 0; k < C; ++k) {
      PlaneSums m;
      if constexpr (D > 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const double lc = p.disc_off[j] * cs, ls = p.disc_off[j] * sn;
          plane_disc(p, pc.a[k], pc.b[k], pc.c[k], x[0] + lc, x[1] + ls, lc, ls, m);
        }
      } else {
#pragma unroll 1
        for (int j = 0; j < nd; ++j) {
          const double lc = p.disc_off[j] * cs, ls = p.disc_off[j] * sn;
          plane_disc(p, pc.a[k], pc.b[k], pc.c[k], x[0] + lc, x[1] + ls, lc, ls, m);
        }
      }
      plane_commit(q, pc.a[k], pc.b[k], m);
    }
    pc = nx;
  }
#ifdef CILQR_QUAD_PROFILE
  asm volatile("" :: "v"(q.h[0]), "v"(q.h[8]), "v"(q.lx[2]));
  QP_STAMP(1);   // corridor
#endif
  // nearest left / right lane plane, all discs (cc:729-769)
#pragma unroll 1
  for (int j = 0; j < nd; ++j) {
    // the offset of disc j without indexing the parameter block by a run-time value: where the state is a local
    // copy (kernels_tail.hip) a dynamically indexed member would put the whole structure into scratch memory
    double doff;
    if constexpr (D == 5) doff = (j == 0) ? p.disc_off[0] : (j == 1) ? p.disc_off[1] : (j == 2) ? p.disc_off[2] : (j == 3) ? p.disc_off[3] : p.disc_off[4];
    else doff = p.disc_off[j];
    const double lcj = doff * cs, lsj = doff * sn;
    const double px = x[0] + lcj, py = x[1] + lsj;
    const double* L = lanes + nearest_segment<EX>(s, lanes, 0, px, py) * kLaneFields;
    PlaneSums ml, mr;
    plane_disc(p, L[0], L[1], L[2], px, py, lcj, lsj, ml);
    plane_commit(q, L[0], L[1], ml);
    const double* Rr = lanes + (s.nl + nearest_segment<EX>(s, lanes, 1, px, py)) * kLaneFields;
    plane_disc(p, Rr[0], Rr[1], Rr[2], px, py, lcj, lsj, mr);
    plane_commit(q, Rr[0], Rr[1], mr);
  }
#ifdef CILQR_QUAD_PROFILE
  asm volatile("" :: "v"(q.h[0]), "v"(q.h[8]), "v"(q.lx[2]));
  QP_STAMP(2);   // lanes
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) g_quad_prof[QP_WAVE * 8 + 5] = wall_clock64() - qp_t0;
#endif
  quad_mirror(q);
  if (term) {
    double2* o = s.term + scratch_index(s, slot);
    assume_lds<InLds>(o);
    o[0] = make_double2(q.lx[0], q.lx[1]);
    o[(size_t)Bc].x = q.lx[2];
    o[(size_t)3 * Bc] = make_double2(q.h[0], q.h[1]);
    o[(size_t)4 * Bc] = make_double2(q.h[2], q.h[3]);
    o[(size_t)5 * Bc] = make_double2(q.h[4], q.h[5]);
    o[(size_t)6 * Bc] = make_double2(q.h[6], q.h[7]);
    o[(size_t)7 * Bc].x = q.h[8];
    return;
  }
  double2* o = s.lin + (size_t)i * kLinPairs * Bc + scratch_index(s, slot);
  assume_lds<InLds>(o);
  o[(size_t)6 * Bc] = make_double2(q.lx[0], q.lx[1]);
  o[(size_t)7 * Bc].x = q.lx[2];
  o[(size_t)10 * Bc] = make_double2(q.h[0], q.h[1]);
  o[(size_t)11 * Bc] = make_double2(q.h[2], q.h[3]);
  o[(size_t)12 * Bc] = make_double2(q.h[4], q.h[5]);
  o[(size_t)13 * Bc] = make_double2(q.h[6], q.h[7]);
  o[(size_t)14 * Bc].x = q.h[8];
}

// ---------------------------------------------------------------------------------------------
// The same quadratisation with a knot's planes spread over lanes (the tail kernel: one problem, a workgroup, and a
// dependent chain of ~5300 instructions per knot if one lane does it all).  What a plane contributes is its five sums
// over the discs (plane_disc) -- independent of every other plane -- followed by a handful of additions into the knot's
// lx / lxx in plane order (plane_commit).  knot_plane_items evaluates the sums of the planes (and of the nearest lane
// planes of the discs) that lane `part` of `parts` owns and leaves them in `scr`; knot_commit_items, one lane per knot,
// adds them up in the order knot_quadratize does: corridor planes 0, 1, ..., then disc 0 left, disc 0 right, disc 1
// left, ...  Same operations on the same operands in the same order: bit-identical to knot_quadratize (tested).
// Problem alone in its arena (capacity 1, slot 0), D = 5.  scr: quad_split_doubles(cmax) doubles per knot, cmax = the batch's.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int quad_split_planes(int cmax) { return (cmax + kQuadChunk - 1) / kQuadChunk * kQuadChunk; }
// (an odd count: the committing lanes, one per knot, then read their rows from different LDS banks)
__host__ __device__ constexpr int quad_split_stride(int cap) { return (cap * 5 + 2 * 5 * 7) | 1; }   // cap = quad_split_planes(cmax)
__host__ __device__ constexpr int quad_split_doubles(int cmax) { return quad_split_stride(quad_split_planes(cmax)); }
// cap: quad_split_planes of the batch's plane capacity (rows of a knot's corridor block in scr); stride: quad_split_doubles of it
template <bool EX, bool InLds>
CILQR_DEV void knot_plane_items(const DeviceState& s, const double* __restrict__ lanes, int buf, int i, int part, int parts,
                                double* __restrict__ scr, int cap, int stride) {
  const Params& p = s.p;
  const double2* xb = s.X + ((size_t)buf * p.K + i) * 3;
  assume_lds<InLds>(xb);
  const double2 p0 = xb[0];
  const double x0 = p0.x, x1 = p0.y, th = xb[1].x;
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3;
  assume_lds<InLds>(cor);
  const int* cntp = s.ccnt + i;
  assume_lds<InLds>(cntp);
  const int cnt = *cntp;
  // knot_quadratize walks the planes in chunks of kQuadChunk and fills the last chunk up with (0, 0, 1) planes
  const int padded = (cnt + kQuadChunk - 1) / kQuadChunk * kQuadChunk;
  double sn, cs;
  lean_sincos(th, &sn, &cs);
  double* __restrict__ o = scr + (size_t)i * stride;
  for (int k = part; k < padded; k += parts) {
    const bool live = k < cnt;
    const double* q = cor + (size_t)(live ? k : 0) * 3;
    const double qa = q[0], qb = q[1], qc = q[2];
    const double a = live ? qa : 0.0, b = live ? qb : 0.0, c = live ? qc : 1.0;
    PlaneSums m;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double lc = p.disc_off[j] * cs, ls = p.disc_off[j] * sn;
      plane_disc(p, a, b, c, x0 + lc, x1 + ls, lc, ls, m);
    }
    double* w = o + k * 5;
    w[0] = m.T0; w[1] = m.T1; w[2] = m.S0; w[3] = m.S1; w[4] = m.S2 - m.W;
  }
  double* __restrict__ ol = o + cap * 5;
#pragma unroll 1
  for (int e = part; e < 10; e += parts) {
    const int j = e >> 1, side = e & 1;
    const double doff = (j == 0) ? p.disc_off[0] : (j == 1) ? p.disc_off[1] : (j == 2) ? p.disc_off[2] : (j == 3) ? p.disc_off[3] : p.disc_off[4];
    const double lcj = doff * cs, lsj = doff * sn;
    const double px = x0 + lcj, py = x1 + lsj;
    const double* L = lanes + ((side ? s.nl : 0) + nearest_segment<EX>(s, lanes, side, px, py)) * kLaneFields;
    const double la = L[0], lb = L[1];
    PlaneSums m;
    plane_disc(p, la, lb, L[2], px, py, lcj, lsj, m);
    double* w = ol + e * 7;
    w[0] = m.T0; w[1] = m.T1; w[2] = m.S0; w[3] = m.S1; w[4] = m.S2 - m.W; w[5] = la; w[6] = lb;
  }
}
// plane_commit from the stored sums (d = S2 - W)
CILQR_DEV void plane_commit_sums(Quad& q, double a, double b, double T0, double T1, double S0, double S1, double d) {
  q.lx[0] += a * T0;
  q.lx[1] += b * T0;
  q.lx[2] += T1;
  const double aS = a * S0, bS = b * S0;
  const double h01 = aS * b, h02 = a * S1, h12 = b * S1;
  q.h[0] += aS * a; q.h[1] += h01; q.h[2] += h02;
  q.h[4] += bS * b; q.h[5] += h12;
  q.h[8] += d;
}
template <bool InLds>
CILQR_DEV void knot_commit_items(const DeviceState& s, int buf, int i, const double* __restrict__ scr, int cap, int stride) {
  const Params& p = s.p;
  const bool term = (i == p.N);
  const double2* xb = s.X + ((size_t)buf * p.K + i) * 3;
  assume_lds<InLds>(xb);
  const double2 p0 = xb[0];
  const double th = xb[1].x;
  const double2* gp = s.goals + (size_t)i * 3;
  assume_lds<InLds>(gp);
  const double2 g0 = gp[0];
  const double gth = gp[1].x;
  const double* __restrict__ cor = s.cor + (size_t)i * s.cmax * 3;
  assume_lds<InLds>(cor);
  const int* cntp = s.ccnt + i;
  assume_lds<InLds>(cntp);
  const int cnt = *cntp;
  const int padded = (cnt + kQuadChunk - 1) / kQuadChunk * kQuadChunk;
  Quad q;
  q.lx[0] = 2.0 * p.w_x * (p0.x - g0.x);           // cc:623-628
  q.lx[1] = 2.0 * p.w_y * (p0.y - g0.y);
  q.lx[2] = 2.0 * p.w_theta * (th - gth);
#pragma unroll
  for (int e = 0; e < 9; ++e) q.h[e] = 0.0;
  q.h[0] = 2.0 * p.w_x; q.h[4] = 2.0 * p.w_y; q.h[8] = 2.0 * p.w_theta;   // cc:642-647
  const double* __restrict__ o = scr + (size_t)i * stride;
  for (int k = 0; k < padded; ++k) {
    const bool live = k < cnt;
    const double* qq = cor + (size_t)(live ? k : 0) * 3;
    const double qa = qq[0], qb = qq[1];
    const double* w = o + k * 5;
    plane_commit_sums(q, live ? qa : 0.0, live ? qb : 0.0, w[0], w[1], w[2], w[3], w[4]);
  }
  const double* __restrict__ ol = o + cap * 5;
#pragma unroll 2
  for (int e = 0; e < 10; ++e) {
    const double* w = ol + e * 7;
    plane_commit_sums(q, w[5], w[6], w[0], w[1], w[2], w[3], w[4]);
  }
  quad_mirror(q);
  if (term) {
    double2* t = s.term;
    assume_lds<InLds>(t);
    t[0] = make_double2(q.lx[0], q.lx[1]);
    t[1].x = q.lx[2];
    t[3] = make_double2(q.h[0], q.h[1]);
    t[4] = make_double2(q.h[2], q.h[3]);
    t[5] = make_double2(q.h[4], q.h[5]);
    t[6] = make_double2(q.h[6], q.h[7]);
    t[7].x = q.h[8];
    return;
  }
  double2* t = s.lin + (size_t)i * kLinPairs;
  assume_lds<InLds>(t);
  t[6] = make_double2(q.lx[0], q.lx[1]);
  t[7].x = q.lx[2];
  t[10] = make_double2(q.h[0], q.h[1]);
  t[11] = make_double2(q.h[2], q.h[3]);
  t[12] = make_double2(q.h[4], q.h[5]);
  t[13] = make_double2(q.h[6], q.h[7]);
  t[14].x = q.h[8];
}

#ifdef CILQR_REF_ORDER
// test-only: the knot's quadratisation in the reference's operation order (ref_order.hpp), same storage layout
CILQR_DEV void knot_quadratize_ref(const DeviceState& s, int buf, int i, int slot) {
  const Params& p = s.p;
  const int Bc = s.Bcap;
  const bool term = (i == p.N);
  double x[6], u[2] = {0.0, 0.0};
  load_x(s, buf, i, slot, x);
  if (!term) load_u(s, buf, i, slot, u);
  Quad q;
  reforder::knot_quadratize(s, i, slot, x, u, q.lx, q.lu, q.h, q.hd, q.huu);
  if (term) {
    double2* o = s.term + scratch_index(s, slot);
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
  double2* o = s.lin + (size_t)i * kLinPairs * Bc + scratch_index(s, slot);
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
#endif

}  // namespace cilqr
