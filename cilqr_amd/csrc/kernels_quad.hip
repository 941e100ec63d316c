// Cost evaluation and quadratisation kernels: one thread per (problem, knot).
//
// Reference behaviour (algorithm/ilqr/ilqr_optimizer.cc):
//   TotalCost cc:417-436 = JCost cc:497 + DynamicsCost cc:518 + CorridorCost cc:553 +
//   LaneBoundaryCost cc:583;  CostJacbian cc:620-636 (+cc:657-671, 690-706, 729-746);
//   CostHessian cc:638-655 (+cc:673-688, 708-727, 748-769); DynamicsJacbian vehicle_model.cc:21-86.
//
// Work decomposition: grid = (ceil(n/256), K[, alpha]); blockIdx.y is the knot, threads walk a
// list of slots, so a wave reads 64 consecutive slots of one knot (coalesced).  Inside a thread:
//   * the corridor planes of the knot are read ONCE, four at a time (12 loads in flight), and
//     every chunk is applied to all discs from registers -- the reference loops disc-major and
//     would re-read each plane per disc;
//   * the lane tables (a few KiB, shared by the batch) are staged in LDS per block; the nearest
//     segment comes from the uniform-grid candidate lists (dev_model.hpp);
//   * barrier values are grouped so that one log() serves all constraints of a disc.
// Per-knot cost partials go to `part`; a per-problem pass sums them over knots in index order.
// Sums over (disc, plane) are therefore re-associated with respect to the reference's running
// sums; every individual term is computed with the reference's expression.
#include "quad_core.hpp"

namespace cilqr {

// The cost kernels exist per disc count: D = 5 (the reference's; unrolled) and D = 0 (any other count), so that the
// generic path does not set the register budget of the common one.  Four waves per SIMD (128 VGPRs).  What made that
// possible is a compiler switch as much as the code: this file is built with -mllvm -disable-machine-licm (Makefile,
// QUADFLAGS).  With machine LICM on, the ~20 polynomial coefficients of sincos / log and the derived barrier constants
// were materialised ONCE in VGPR pairs ahead of the block's problem loop and kept there through it -- 40-odd vector
// registers holding uniform constants (the scalar file was full already), 168 VGPRs, three waves, and spills as soon as
// a fourth was asked for.  Rematerialised at their uses they cost a v_mov pair each and no live range.
#ifndef CILQR_COST_OCC
#define CILQR_COST_OCC 4
#endif
#define CILQR_COST_ATTR __attribute__((amdgpu_waves_per_eu(CILQR_COST_OCC, CILQR_COST_OCC)))
// ... and per tie rule (EX: CILQR_OPT_EXACT_LANE_TIES, see nearest_from_cell)
#define CILQR_LAUNCH_BY_DISCS(kernel, grid, block, lds, st, ...)                                            \
  do {                                                                                                     \
    if (s.p.num_of_disc == 5) {                                                                            \
      if (s.exact_ties) hipLaunchKernelGGL((kernel<5, true>), grid, block, lds, st, __VA_ARGS__);          \
      else hipLaunchKernelGGL((kernel<5, false>), grid, block, lds, st, __VA_ARGS__);                      \
    } else {                                                                                               \
      if (s.exact_ties) hipLaunchKernelGGL((kernel<0, true>), grid, block, lds, st, __VA_ARGS__);          \
      else hipLaunchKernelGGL((kernel<0, false>), grid, block, lds, st, __VA_ARGS__);                      \
    }                                                                                                      \
  } while (0)

// list == nullptr: slots 0..n-1.  skip_done: ignore slots that already left the iteration.
template <int D, bool EX>
__global__ __launch_bounds__(256) CILQR_COST_ATTR void k_cost_knots(DeviceState s, const int* __restrict__ list,
                                                    const int* __restrict__ n_ptr, int n_max, int cand,
                                                    int skip_done) {
  extern __shared__ double lds[];
#ifdef CILQR_COST_PROFILE
  unsigned long long cp_t = wall_clock64();
  const unsigned long long cp_t0 = cp_t;
#endif
  const int n = n_ptr ? min(*n_ptr, n_max) : active_count(s, n_max);
  if ((int)(blockIdx.x * blockDim.x) >= n) return;   // whole block idle: skip the LDS staging too
  const double* lanes = stage_lanes(s, lds);
  CP_STAMP(0);   // kernel arguments + lane tables staged
  const int i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list ? list[j] : j;
    if (skip_done && s.acc_idx[slot] != -1) continue;
    const int buf = s.cur[slot] ^ cand;
    double x[6], u[2] = {0.0, 0.0};
    load_x(s, buf, i, slot, x);
    if (i < s.p.N) load_u(s, buf, i, slot, u);
#ifdef CILQR_COST_PROFILE
    asm volatile("" :: "v"(x[0]), "v"(x[5]), "v"(u[0]));
    CP_STAMP(1);   // cur[slot] -> state loaded
#endif
    knot_cost<D, EX>(s, lanes, i, slot, x, u, s.part + (size_t)i * kPartPairs * s.Bcap + slot, (size_t)s.Bcap);
  }
#ifdef CILQR_COST_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) {
    g_cost_prof[CP_WAVE * 8 + 5] = wall_clock64() - cp_t0;   // whole life of the wave
    g_cost_prof[CP_WAVE * 8 + 6] = cp_t0;                    // its start
  }
#endif
}
#ifdef CILQR_COST_PROFILE
extern "C" void cilqr_debug_cost_profile(unsigned long long* out, int reset) {   // tuning build only, not part of the C-ABI
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cost_prof), sizeof(g_cost_prof));
  if (reset) { void* p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_cost_prof)); (void)hipMemset(p, 0, sizeof(g_cost_prof)); }
}
#endif

// speculative line search: knot i of candidate alpha_{r0 + blockIdx.z} of list entry j
template <int D, bool EX>
__global__ __launch_bounds__(256) CILQR_COST_ATTR void k_spec_cost(DeviceState s, const int* __restrict__ list,
                                                   const int* __restrict__ n_ptr, int off, int n_max, int r0) {
  extern __shared__ double lds[];
  const int n = list_count(s, n_ptr, off, n_max);   // `list` points at entry `off` already
  if ((int)(blockIdx.x * blockDim.x) >= n) return;
  const double* lanes = stage_lanes(s, lds);
  const int i = blockIdx.y, r = r0 + blockIdx.z;
  const size_t cap = (size_t)s.spec_cap;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list[j];
    if (s.acc_idx[slot] != -1) continue;   // left at the gradient-norm exit
    const double2* xb = s.Xs + ((size_t)r * s.p.K + i) * 3 * cap + j;
    const double2 p0 = xb[0], p1 = xb[cap], p2 = xb[2 * cap];
    const double x[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
    double u[2] = {0.0, 0.0};
    if (i < s.p.N) {
      const double2 q = s.Us[((size_t)r * s.p.N + i) * cap + j];
      u[0] = q.x; u[1] = q.y;
    }
    knot_cost<D, EX>(s, lanes, i, slot, x, u, s.parts + ((size_t)r * s.p.K + i) * kPartPairs * cap + j, cap);
  }
}
// The same with P consecutive lanes per problem: knot i of candidates alpha_{r0} .. alpha_{r_end-1} of list entry j; the lanes
// hold P step sizes of ONE problem.  What depends on the problem and the knot only (corridor planes, goal, plane
// count) is then read once per P lanes (same address: one cache line), and only the candidate itself (3 + 1 pairs)
// per lane.  For the SPARSE list of the hybrid schedule (the problems that rejected every sequential round: a few
// percent of the slots, in no particular order) one lane per problem made every such load touch 64 different lines;
// packed, the same knot costs are 4.6x faster (measured: 897 -> 155 us in the first iteration of the bench batch).
template <int D, int P, bool EX>
__global__ __launch_bounds__(256) CILQR_COST_ATTR void k_spec_cost_packed(DeviceState s, const int* __restrict__ list,
                                                                          const int* __restrict__ n_ptr, int off, int n_max,
                                                                          int r0, int r_end) {
  extern __shared__ double lds[];
  const int n = list_count(s, n_ptr, off, n_max);
  constexpr int per_block = 256 / P;
  if ((int)(blockIdx.x * per_block) >= n) return;
  const double* lanes = stage_lanes(s, lds);
  const int i = blockIdx.y, r = r0 + (int)(threadIdx.x % P);
  const size_t cap = (size_t)s.spec_cap;
  if (r >= r_end) return;
  for (int j = blockIdx.x * per_block + threadIdx.x / P; j < n; j += gridDim.x * per_block) {
    const int slot = list[j];
    if (s.acc_idx[slot] != -1) continue;   // left at the gradient-norm exit
    const double2* xb = s.Xs + ((size_t)r * s.p.K + i) * 3 * cap + j;
    const double2 p0 = xb[0], p1 = xb[cap], p2 = xb[2 * cap];
    const double x[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
    double u[2] = {0.0, 0.0};
    if (i < s.p.N) {
      const double2 q = s.Us[((size_t)r * s.p.N + i) * cap + j];
      u[0] = q.x; u[1] = q.y;
    }
    knot_cost<D, EX>(s, lanes, i, slot, x, u, s.parts + ((size_t)r * s.p.K + i) * kPartPairs * cap + j, cap);
  }
}

// Candidates alpha_{r0} .. alpha_10 of every listed problem.  The step sizes of a problem sit in neighbouring lanes
// (eight, then four for what is left), so what depends on the problem and the knot only -- corridor planes, goal --
// travels once per group instead of once per candidate; for the sparse pending list of the hybrid schedule this is
// also what keeps a load instruction from touching 64 different cache lines.
void launch_spec_cost(const DeviceState& s, const int* list, const int* n_ptr, int off, int n_max, int n_grid, int r0,
                      int sparse, hipStream_t st) {
  const size_t lds = lane_lds_bytes(s);
  const bool five = s.p.num_of_disc == 5;
  if (!sparse) {   // a dense list: one lane per problem is already coalesced, one grid layer per candidate (measured: 2 % faster)
    dim3 g((n_grid + 255) / 256, s.p.K, kNumAlpha - r0);
    CILQR_LAUNCH_BY_DISCS(k_spec_cost, g, dim3(256), lds, st, s, list, n_ptr, off, n_max, r0);
    return;
  }
  for (int a = r0; a < kNumAlpha;) {
    const int left = kNumAlpha - a;
    const int P = left > 4 ? 8 : (left > 2 ? 4 : (left > 1 ? 2 : 1));
    const int e = a + (P < left ? P : left);
    const int per_block = 256 / P;
    dim3 g((n_grid + per_block - 1) / per_block, s.p.K);
#define CILQR_SC1(DD, PP, EE) hipLaunchKernelGGL((k_spec_cost_packed<DD, PP, EE>), g, dim3(256), lds, st, s, list, n_ptr, off, n_max, a, e)
#define CILQR_SC(DD, PP) do { if (s.exact_ties) CILQR_SC1(DD, PP, true); else CILQR_SC1(DD, PP, false); } while (0)
    if (P == 8) { if (five) CILQR_SC(5, 8); else CILQR_SC(0, 8); }
    else if (P == 4) { if (five) CILQR_SC(5, 4); else CILQR_SC(0, 4); }
    else if (P == 2) { if (five) CILQR_SC(5, 2); else CILQR_SC(0, 2); }
    else { if (five) CILQR_SC(5, 1); else CILQR_SC(0, 1); }
#undef CILQR_SC1
#undef CILQR_SC
    a = e;
  }
}

// Sequential rounds, G step sizes per round: knot i of candidates alpha_{r0} .. alpha_{r0+G-1} of the problems that
// rejected alpha_0 .. alpha_{r0-1} (pending list r0).  G consecutive lanes hold the G candidates of one problem, so
// its corridor planes and goals travel once per G knot costs (see k_spec_cost_packed); with G = 2 a round costs
// a candidate that the sequential loop might not have reached (alpha_1 for the 15 % that accept alpha_0), and saves
// half of the plane traffic and half of the launches.  The first passing index still wins (k_round_pick).
template <int D, int G, bool EX>
__global__ __launch_bounds__(256) CILQR_COST_ATTR void k_round_cost(DeviceState s, int r0, int n_max) {
  extern __shared__ double lds[];
  const int* __restrict__ list = s.pend + (size_t)r0 * s.Bcap;
  const int n = (r0 == 0) ? active_count(s, n_max) : min(s.counters[r0], n_max);
  constexpr int per_block = 256 / G;
  if ((int)(blockIdx.x * per_block) >= n) return;
  const double* lanes = stage_lanes(s, lds);
  const int i = blockIdx.y, r = r0 + (int)(threadIdx.x % G);
  const size_t cap = (size_t)s.spec_cap;
  for (int e = blockIdx.x * per_block + threadIdx.x / G; e < n; e += gridDim.x * per_block) {
    const int j = (r0 == 0) ? e : list[e];
    const int slot = s.act[j];
    if (s.acc_idx[slot] != -1) continue;
    const double2* xb = s.Xs + ((size_t)r * s.p.K + i) * 3 * cap + j;
    const double2 p0 = xb[0], p1 = xb[cap], p2 = xb[2 * cap];
    const double x[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
    double u[2] = {0.0, 0.0};
    if (i < s.p.N) {
      const double2 q = s.Us[((size_t)r * s.p.N + i) * cap + j];
      u[0] = q.x; u[1] = q.y;
    }
    knot_cost<D, EX>(s, lanes, i, slot, x, u, s.parts + ((size_t)r * s.p.K + i) * kPartPairs * cap + j, cap);
  }
}

// n_grid: problems the grid is sized for (threads stride over the rest); group: 1, 2 or 4 candidates per round
void launch_round_cost(const DeviceState& s, int r0, int group, int n_max, int n_grid, hipStream_t st) {
  const int per_block = 256 / group;
  dim3 g((n_grid + per_block - 1) / per_block, s.p.K);
  const size_t lds = lane_lds_bytes(s);
  const bool five = s.p.num_of_disc == 5;
#define CILQR_RC1(DD, GG, EE) hipLaunchKernelGGL((k_round_cost<DD, GG, EE>), g, dim3(256), lds, st, s, r0, n_max)
#define CILQR_RC(DD, GG) do { if (s.exact_ties) CILQR_RC1(DD, GG, true); else CILQR_RC1(DD, GG, false); } while (0)
  if (group == 1) { if (five) CILQR_RC(5, 1); else CILQR_RC(0, 1); }
  else if (group == 2) { if (five) CILQR_RC(5, 2); else CILQR_RC(0, 2); }
  else { if (five) CILQR_RC(5, 4); else CILQR_RC(0, 4); }
#undef CILQR_RC
#undef CILQR_RC1
}

__global__ __launch_bounds__(64) void k_reduce_only(DeviceState s, const int* __restrict__ list, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int slot = list ? list[j] : j;
  double c5[5];
  reduce_cost(s, slot, 0, c5);
#pragma unroll
  for (int c = 0; c < 5; ++c) s.trial[(size_t)c * s.Bcap + slot] = c5[c];
}

// n_max bounds the list length; n_grid sizes the grid (threads stride over the rest)
void launch_cost_knots(const DeviceState& s, const int* list, const int* n_ptr, int n_max, int n_grid,
                       int cand, int skip_done, hipStream_t st) {
  if (n_grid <= 0) return;
  dim3 g((n_grid + 255) / 256, s.p.K);
  CILQR_LAUNCH_BY_DISCS(k_cost_knots, g, dim3(256), lane_lds_bytes(s), st, s, list, n_ptr, n_max, cand, skip_done);
}

void launch_cost_only(const DeviceState& s, const int* list, int n, int cand, hipStream_t st) {
  if (n == 0) return;
  launch_cost_knots(s, list, nullptr, n, n, cand, 0, st);
  hipLaunchKernelGGL(k_reduce_only, dim3((n + 63) / 64), dim3(64), 0, st, s, list, n);
}

// cost of the init guess -> cost_old, cost_[0]; iter_trajs gets the init guess (cc:170-173)
__global__ void k_init_cost_commit(DeviceState s, int n) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const int pb = s.pid[slot];
#pragma unroll
  for (int c = 0; c < 5; ++c) s.hist[(size_t)c * s.Pcap + pb] = s.trial[(size_t)c * s.Bcap + slot];
  s.cost_old[slot] = s.trial[slot];
  s.n_cost[pb] = 1;
  s.n_iter_trajs[pb] = 1;
  s.emit[slot] = 1;
}
void launch_init_cost_commit(const DeviceState& s, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_init_cost_commit, dim3((n + 255) / 256), dim3(256), 0, st, s, n);
}

#ifndef CILQR_QUAD_OCC
#define CILQR_QUAD_OCC 4
#endif
#define CILQR_QUAD_ATTR __attribute__((amdgpu_waves_per_eu(CILQR_QUAD_OCC, CILQR_QUAD_OCC)))
template <int D, bool EX>
__global__ __launch_bounds__(256) CILQR_QUAD_ATTR void k_quadratize(DeviceState s, const int* __restrict__ list, int n,
                                                    int only_upd) {
  extern __shared__ double lds[];
  n = active_count(s, n);
  if ((int)(blockIdx.x * blockDim.x) >= n) return;
  const double* lanes = stage_lanes(s, lds);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int slot = list ? list[j] : j;
  if (only_upd && !s.upd[slot]) return;
#ifdef CILQR_REF_ORDER
  (void)lanes;
  knot_quadratize_ref(s, s.cur[slot], blockIdx.y, slot);
#else
  knot_quadratize<D, EX>(s, lanes, s.cur[slot], blockIdx.y, slot);
#endif
}

#ifdef CILQR_QUAD_PROFILE
extern "C" void cilqr_debug_quad_profile(unsigned long long* out, int reset) {   // tuning build only, not part of the C-ABI
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_prof), sizeof(g_quad_prof));
  if (reset) { void* p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_quad_prof)); (void)hipMemset(p, 0, sizeof(g_quad_prof)); }
}
#endif

void launch_quadratize(const DeviceState& s, const int* list, int n, int only_upd, hipStream_t st) {
  if (n == 0) return;
  dim3 g((n + 255) / 256, s.p.K);
  CILQR_LAUNCH_BY_DISCS(k_quadratize, g, dim3(256), lane_lds_bytes(s), st, s, list, n, only_upd);
}

}  // namespace cilqr
