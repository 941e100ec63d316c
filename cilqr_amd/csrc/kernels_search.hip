// Forward pass, 11-step line search and the per-problem state machine of Optimize().
//
// Reference behaviour (algorithm/ilqr/ilqr_optimizer.cc):
//   Forward cc:392-415; line search + acceptance test cc:243-270 (alpha list cc:197,
//   beta in (1e-4, 10), dcost > 0); regularisation schedule and exits cc:272-308, 312-319;
//   gradient-norm exit cc:235-241.
//
// Lockstep design: every trial of the line search depends only on the pre-trial iterate, so
// round r evaluates alpha_r for exactly the problems that rejected alpha_0..alpha_{r-1}:
//   forward(alpha_0) on the active list -> cost (problem x knot threads) ->
//   [reduce + accept test, or roll out alpha_{r+1} and join pending list r+1] -> cost -> ...
// The first passing list index wins, as in the sequential loop.  Pending lists are compacted
// with wave-aggregated atomics; their order only affects coalescing, never results.
#include <cstdio>
#include <cstdlib>

#include "search_core.hpp"

namespace cilqr {

// One lane per rollout: given a wave per SIMD the compiler spends all 256 registers on unrolling and prefetching, and such a
// wave then keeps two or three cost-kernel waves of another solve off its SIMD.  CILQR_ROLL_OCC > 0 caps the registers instead.
#ifndef CILQR_ROLL_OCC
#define CILQR_ROLL_OCC 0
#endif
#if CILQR_ROLL_OCC > 0
#define CILQR_ROLL_ATTR __attribute__((amdgpu_waves_per_eu(CILQR_ROLL_OCC, CILQR_ROLL_OCC)))
#else
#define CILQR_ROLL_ATTR
#endif
// stage API: plain rollout of the listed slots with one alpha
__global__ __launch_bounds__(64) CILQR_ROLL_ATTR void k_forward(DeviceState s, const int* __restrict__ list, int n,
                                                double alpha, int skip_done) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int slot = list ? list[j] : j;
  if (skip_done && s.acc_idx[slot] != -1) return;
  forward_problem(s, slot, alpha);
}
void launch_forward(const DeviceState& s, const int* list, int n, double alpha, int skip_done,
                    hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_forward, dim3((n + 63) / 64), dim3(64), 0, st, s, list, n, alpha, skip_done);
}

// round 0 opener: gradient-norm exit (cc:235-241), else roll out alpha_0
__global__ __launch_bounds__(64) CILQR_ROLL_ATTR void k_search_open(DeviceState s, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= active_count(s, n)) return;
  const int slot = s.act[j];
  if (leaves_before_search(s, slot, true)) {
    s.acc_idx[slot] = -2;
    return;
  }
  s.acc_idx[slot] = -1;
  forward_problem(s, slot, kAlpha[0]);
}

// round r: total cost of the alpha_r candidate, acceptance test (cc:252-261); on rejection roll
// out alpha_{r+1} and queue the slot for the next round.
__global__ __launch_bounds__(64) CILQR_ROLL_ATTR void k_search_round(DeviceState s, int r, int n_max, int roll_next) {
  const int* __restrict__ list = (r == 0) ? s.act : s.pend + (size_t)r * s.Bcap;
  const int n = (r == 0) ? active_count(s, n_max) : min(s.counters[r], n_max);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list[j];
    if (s.acc_idx[slot] != -1) continue;   // left at the gradient-norm exit
    double c5[5];
    reduce_cost(s, slot, 1, c5);
    const double alpha = kAlpha[r];
    const double dcost = s.cost_old[slot] - c5[0];                                  // cc:254
    const double expected = -alpha * (s.dV[slot] + alpha * s.dV[(size_t)s.Bcap + slot]);  // cc:255
    const double z = dcost / expected;                                              // cc:257
#pragma unroll
    for (int c = 0; c < 5; ++c) s.trial[(size_t)c * s.Bcap + slot] = c5[c];
    if ((z > 1e-4 && z < 10.0) && dcost > 0.0) {                                    // cc:258
      s.acc_idx[slot] = r;
      s.dcost[slot] = dcost;
      s.cur[slot] ^= 1;   // the candidate becomes the iterate
    } else if (r + 1 < kNumAlpha) {
      if (roll_next) forward_problem(s, slot, kAlpha[r + 1]);
      const int pos = atomicAdd(&s.counters[r + 1], 1);
      s.pend[(size_t)(r + 1) * s.Bcap + pos] = slot;
    }
  }
}

// ---- speculative mode: step sizes r0..10 of every listed problem at once ----
// Used for the whole line search of small active sets (r0 = 0, list = active list, `open` = also
// take the gradient-norm exit) and for the tail of the round-by-round search (r0 = number of
// sequential rounds done, list = the problems that rejected all of them).
// rollouts of a SPARSE list: eight consecutive lanes hold the remaining step sizes of one problem, so the nominal
// trajectory and the gains of a step (11 pairs, the same for every step size) are read once per eight lanes
// (see k_spec_cost_packed in kernels_quad.hip)
__global__ __launch_bounds__(64) CILQR_ROLL_ATTR void k_spec_forward_packed(DeviceState s, const int* __restrict__ list,
                                                            const int* __restrict__ n_ptr, int off, int n_max, int r0) {
  const int n = list_count(s, n_ptr, off, n_max);   // `list` points at entry `off` of the device-side list already
  constexpr int per_block = 64 / 8;
  const int r = r0 + (threadIdx.x & 7);
  if (r >= kNumAlpha) return;
  for (int j = blockIdx.x * per_block + threadIdx.x / 8; j < n; j += gridDim.x * per_block) {
    const int slot = list[j];
    forward_core(s, slot, kAlpha[r], OutSpec{s, r, j});
  }
}

__global__ __launch_bounds__(64) CILQR_ROLL_ATTR void k_spec_forward(DeviceState s, const int* __restrict__ list,
                                                     const int* __restrict__ n_ptr, int off, int n_max, int r0, int open) {
  const int n = list_count(s, n_ptr, off, n_max);
  const int r = r0 + blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list[j];
    if (open) {
      // status is only written by the r == 0 lanes, and only from 0 to 3: the other step sizes' lanes
      // read 0 or 3 and come to the same verdict through the gradient norm
      if (leaves_before_search(s, slot, r == 0)) {
        if (r == 0) s.acc_idx[slot] = -2;
        continue;
      }
      if (r == 0) s.acc_idx[slot] = -1;
    }
    forward_core(s, slot, kAlpha[r], OutSpec{s, r, j});
  }
}

// total cost of candidate alpha_r of list entry j: knot partials summed in index order
__global__ __launch_bounds__(64) void k_spec_reduce(DeviceState s, const int* __restrict__ list,
                                                    const int* __restrict__ n_ptr, int off, int n_max, int r0, int open) {
  const int n = list_count(s, n_ptr, off, n_max);
  const int r = r0 + blockIdx.y;
  const size_t cap = (size_t)s.spec_cap;
  const int K = s.p.K, N = s.p.N;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list[j];
    if (open) {   // acc_idx is written by the r == 0 lanes of k_spec_forward, already complete here
      if (s.acc_idx[slot] != -1) continue;
    }
    double* t = s.spec_tot + (size_t)r * 5 * cap + j;
#ifdef CILQR_REF_ORDER
    double c5[5];
    spec_total_cost(s, slot, r, j, c5);
    t[0] = c5[0];
    t[cap] = c5[1]; t[2 * cap] = c5[2]; t[3 * cap] = c5[3]; t[4 * cap] = c5[4];
    (void)K; (void)N;
#else
    double jj = 0.0, dx = 0.0, du = 0.0, cc = 0.0, lc = 0.0;
    const double2* pb = s.parts + (size_t)r * K * kPartPairs * cap + j;
    // loads of several knots in flight; the sums stay in knot order
#pragma unroll 8
    for (int i = 0; i < K; ++i) {
      const double2* o = pb + (size_t)i * kPartPairs * cap;
      const double2 a = o[0], c = o[2 * cap];     // (J, bounds) of the state; (corridor, lane): quad_core.hpp, knot_cost
      jj += a.x;
      dx += a.y;
      cc += c.x;
      lc += c.y;
    }
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
      const double2 b = pb[((size_t)i * kPartPairs + 1) * cap];   // (J, bounds) of the control
      jj += b.x;
      du += b.y;
    }
    const double dyn = dx + du;
    t[0] = jj + dyn + cc + lc;
    t[cap] = jj; t[2 * cap] = dyn; t[3 * cap] = cc; t[4 * cap] = lc;
#endif
  }
}

// first passing alpha wins (cc:246-261)
__global__ __launch_bounds__(64) void k_spec_pick(DeviceState s, const int* __restrict__ list,
                                                  const int* __restrict__ n_ptr, int off, int n_max, int r0) {
  const int n = list_count(s, n_ptr, off, n_max);
  const size_t cap = (size_t)s.spec_cap;
  const int Bc = s.Bcap;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list[j];
    if (s.acc_idx[slot] != -1) continue;
    const double cost_old = s.cost_old[slot], dV0 = s.dV[slot], dV1 = s.dV[(size_t)Bc + slot];
    int acc = -1, last = kNumAlpha - 1;
    double dcost = 0.0;
    double tot[kNumAlpha];   // all totals requested at once, then tested in order
#pragma unroll
    for (int r = 0; r < kNumAlpha; ++r) tot[r] = (r >= r0) ? s.spec_tot[(size_t)r * 5 * cap + j] : 0.0;
#pragma unroll
    for (int r = 0; r < kNumAlpha; ++r) {
      if (r < r0) continue;
      const double alpha = kAlpha[r];
      dcost = cost_old - tot[r];
      const double expected = -alpha * (dV0 + alpha * dV1);
      const double z = dcost / expected;
      if ((z > 1e-4 && z < 10.0) && dcost > 0.0) {
        acc = r;
        last = r;
        break;
      }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c)
      s.trial[(size_t)c * Bc + slot] = s.spec_tot[((size_t)last * 5 + c) * cap + j];
    if (acc < 0) continue;
    s.acc_idx[slot] = acc;
    s.dcost[slot] = dcost;
    s.cur[slot] ^= 1;   // k_spec_copy fills the new current buffer
  }
}

// the accepted candidate becomes the iterate: one thread per (list entry, knot)
__global__ __launch_bounds__(256) void k_spec_copy(DeviceState s, const int* __restrict__ list,
                                                   const int* __restrict__ n_ptr, int off, int n_max, int r0) {
  const int n = list_count(s, n_ptr, off, n_max);
  const int i = blockIdx.y;
  const size_t cap = (size_t)s.spec_cap;
  const int K = s.p.K, N = s.p.N, Bc = s.Bcap;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = list[j];
    const int acc = s.acc_idx[slot];
    if (acc < r0) continue;   // not accepted (or accepted in an earlier sequential round)
    const int nb = s.cur[slot];
    const double2* xb = s.Xs + ((size_t)acc * K + i) * 3 * cap + j;
    double2* o = s.X + ((size_t)nb * K + i) * 3 * Bc + slot;
    o[0] = xb[0];
    o[(size_t)Bc] = xb[cap];
    o[(size_t)2 * Bc] = xb[2 * cap];
    if (i < N) s.U[((size_t)nb * N + i) * Bc + slot] = s.Us[((size_t)acc * N + i) * cap + j];
  }
}

// ---- pre-rolled rounds: alpha_0 .. alpha_{G-1} of every active problem are rolled out in one pass
// (forward_multi), then evaluated round by round: round r costs and tests only the problems that
// rejected alpha_0 .. alpha_{r-1}.  The arena position of a problem is its position j in the active
// list; pending lists 1 .. G-1 hold such positions, the last one (problems that rejected all G)
// holds slots, which is what the speculative pass over the remaining step sizes takes.
constexpr int kMaxPreRolled = 6;
// The same rollouts with one (problem, step size) per lane: P neighbouring lanes share a problem's nominal
// trajectory and gains (same addresses), four times the waves of k_multi_forward at a third of its registers, so
// that the waves of a SIMD fill each other's latency -- a rollout is a chain of ~300 dependent fp64 instructions
// per step, which one wave per SIMD cannot hide however many independent chains its lanes carry.
#ifndef CILQR_FWD_OCC
#define CILQR_FWD_OCC 3
#endif
#ifndef CILQR_FWD_AHEAD
#define CILQR_FWD_AHEAD 1
#endif
template <int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CILQR_FWD_OCC, CILQR_FWD_OCC)))
void k_multi_forward_packed(DeviceState s, int n, int G) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t / P, r = t % P;
  if (j >= active_count(s, n) || r >= G) return;
  const int slot = s.act[j];
  if (leaves_before_search(s, slot, r == 0)) {   // see k_spec_forward: every lane of a problem comes to the same verdict
    if (r == 0) s.acc_idx[slot] = -2;
    return;
  }
  if (r == 0) s.acc_idx[slot] = -1;
  forward_core<OutSpec, CILQR_FWD_AHEAD>(s, slot, kAlpha[r], OutSpec{s, r, j});
}
   // more sequential rounds than this: the round-by-round rollouts below
template <int G>
__global__ __launch_bounds__(64) void k_multi_forward(DeviceState s, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= active_count(s, n)) return;
  const int slot = s.act[j];
  if (leaves_before_search(s, slot, true)) {
    s.acc_idx[slot] = -2;
    return;
  }
  s.acc_idx[slot] = -1;
  forward_multi<G>(s, slot, j);
}

// round of G step sizes alpha_{r0} .. alpha_{r0+G-1}: total cost of each candidate (knot partials summed in index
// order) and acceptance test (cc:252-261), IN LIST ORDER and stopping at the first that passes -- what the
// sequential loop does; rejected all G -> pending list r0 + G (positions, or slots when `last`)
__global__ __launch_bounds__(64) void k_round_pick(DeviceState s, int r0, int G, int n_max, int last) {
  const int* __restrict__ list = s.pend + (size_t)r0 * s.Bcap;
  const int n = (r0 == 0) ? active_count(s, n_max) : min(s.counters[r0], n_max);
  const size_t cap = (size_t)s.spec_cap;
  const int K = s.p.K, N = s.p.N;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int j = (r0 == 0) ? e : list[e];
    const int slot = s.act[j];
    if (s.acc_idx[slot] != -1) continue;   // left at the gradient-norm exit
    const double cost_old = s.cost_old[slot], dV0 = s.dV[slot], dV1 = s.dV[(size_t)s.Bcap + slot];
    bool accepted = false;
    for (int r = r0; r < r0 + G && !accepted; ++r) {
#ifdef CILQR_REF_ORDER
      double c5[5];
      spec_total_cost(s, slot, r, j, c5);
      (void)K; (void)N; (void)cap;
#else
      double jj = 0.0, dx = 0.0, du = 0.0, cc = 0.0, lc = 0.0;
      const double2* pb = s.parts + (size_t)r * K * kPartPairs * cap + j;
      // loads of several knots in flight; the sums stay in knot order
#pragma unroll 8
      for (int i = 0; i < K; ++i) {
        const double2* o = pb + (size_t)i * kPartPairs * cap;
        const double2 a = o[0], c = o[2 * cap];     // (J, bounds) of the state; (corridor, lane): quad_core.hpp, knot_cost
        jj += a.x;
        dx += a.y;
        cc += c.x;
        lc += c.y;
      }
#pragma unroll 8
      for (int i = 0; i < N; ++i) {
        const double2 b = pb[((size_t)i * kPartPairs + 1) * cap];   // (J, bounds) of the control
        jj += b.x;
        du += b.y;
      }
      const double dyn = dx + du;
      const double c5[5] = {jj + dyn + cc + lc, jj, dyn, cc, lc};
#endif
      const double alpha = kAlpha[r];
      const double dcost = cost_old - c5[0];                                          // cc:254
      const double expected = -alpha * (dV0 + alpha * dV1);                           // cc:255
      const double z = dcost / expected;                                              // cc:257
#pragma unroll
      for (int c = 0; c < 5; ++c) s.trial[(size_t)c * s.Bcap + slot] = c5[c];
      if ((z > 1e-4 && z < 10.0) && dcost > 0.0) {                                    // cc:258
        s.acc_idx[slot] = r;
        s.dcost[slot] = dcost;
        s.cur[slot] ^= 1;   // k_multi_copy fills the new current buffer
        accepted = true;
      }
    }
    if (!accepted && r0 + G < kNumAlpha) {
      const int pos = atomicAdd(&s.counters[r0 + G], 1);
      s.pend[(size_t)(r0 + G) * s.Bcap + pos] = last ? slot : j;
    }
  }
}

// the candidate accepted in a pre-rolled round becomes the iterate: one thread per (position, kCopyKnots knots) -- the three
// dependent look-ups (position -> slot -> accepted index, buffer) are paid once per kCopyKnots x 4 pairs instead of once per
// four (until round 5: one thread per knot), and a thread has that many independent copies in flight
constexpr int kCopyKnots = 4;
__global__ __launch_bounds__(256) void k_multi_copy(DeviceState s, int n_max, int G) {
  const int n = active_count(s, n_max);
  const int i0 = blockIdx.y * kCopyKnots;
  const size_t cap = (size_t)s.spec_cap;
  const int K = s.p.K, N = s.p.N, Bc = s.Bcap;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int slot = s.act[j];
    const int acc = s.acc_idx[slot];
    if (acc < 0 || acc >= G) continue;
    const int nb = s.cur[slot];
#pragma unroll
    for (int d = 0; d < kCopyKnots; ++d) {
      const int i = i0 + d;
      if (i >= K) break;
      const double2* xb = s.Xs + ((size_t)acc * K + i) * 3 * cap + j;
      double2* o = s.X + ((size_t)nb * K + i) * 3 * Bc + slot;
      o[0] = xb[0];
      o[(size_t)Bc] = xb[cap];
      o[(size_t)2 * Bc] = xb[2 * cap];
      if (i < N) s.U[((size_t)nb * N + i) * Bc + slot] = s.Us[((size_t)acc * N + i) * cap + j];
    }
  }
}

// ---- views of the candidate arena (DeviceState::spec_rows) ----
// The allocation holds spec_rows x Bcap cells (a cell = one candidate: K x 3 + N pairs of Xs / Us, K x 3 pairs of parts).
// Eleven rows: every view is the plain [step size][...][Bcap] layout.  Four rows (arenas of >= 32768 slots, where the
// eleven-row arena was 4.1 GB of a handle's 11.7): the pre-rolled rounds use rows 0..3 at stride Bcap; once the last
// k_round_pick has run and k_multi_copy has moved the accepted candidates out, ALL of those cells are dead, and the step
// sizes that are left (7 after four rounds) of the problems that rejected every round are laid over them with a shorter
// stride E = rows x Bcap / 7: entry j of the pending list, step size r -> row r - R of a [7][...][E] layout.  E entries fit
// at once; a pending list longer than E (possible only while more than E problems are active: early iterations, adversarial
// scenes) takes further passes over the same cells.  The base pointers are moved back by R rows of the NEW stride so that the
// kernels' `row r` arithmetic lands on row r - R (address arithmetic only; nothing below the allocation is touched).
static int spec_entries(const DeviceState& s, int n_steps_sizes) {
  static const int forced = [] {   // test hook: a tiny E drives the multi-pass path with a handful of problems
    const char* e = std::getenv("CILQR_SPEC_PASS_ENTRIES");
    return e ? std::atoi(e) : 0;
  }();
  if (s.spec_rows >= kNumAlpha) return s.Bcap;
  const long long cells = (long long)s.spec_rows * s.Bcap;
  int E = (int)(cells / n_steps_sizes);
  if (E >= 64) E = E / 64 * 64;
  if (E > s.Bcap) E = s.Bcap;
  if (forced > 0 && forced < E) E = forced;
  return E;
}
static DeviceState spec_view(const DeviceState& s, int row0, int stride) {
  DeviceState v = s;
  if (s.spec_rows >= kNumAlpha) return v;          // the plain layout: rows are step sizes, stride is the capacity
  const size_t K = (size_t)s.p.K, N = (size_t)s.p.N;
  // The base pointers move back by row0 rows of the new stride: rows below row0 are never addressed (every kernel of a pass
  // starts at r0 = row0), and rows row0..10 at that stride must lie inside the cells the arena owns.  Checked here, on the
  // host, for every launch (nothing else holds these three facts together).
  if (row0 < 0 || stride <= 0 || (size_t)(kNumAlpha - row0) * (size_t)stride > (size_t)s.spec_rows * (size_t)s.Bcap) {
    std::fprintf(stderr, "cilqr: candidate view [%d..10] x %d does not fit %d x %d cells\n", row0, stride, s.spec_rows, s.Bcap);
    std::abort();
  }
  v.spec_cap = stride;
  v.Xs = reinterpret_cast<double2*>(reinterpret_cast<uintptr_t>(s.Xs) - (uintptr_t)row0 * K * 3 * (size_t)stride * sizeof(double2));
  v.Us = reinterpret_cast<double2*>(reinterpret_cast<uintptr_t>(s.Us) - (uintptr_t)row0 * N * (size_t)stride * sizeof(double2));
  v.parts = reinterpret_cast<double2*>(reinterpret_cast<uintptr_t>(s.parts) - (uintptr_t)row0 * K * kPartPairs * (size_t)stride * sizeof(double2));
  return v;
}
// most problems an all-eleven pass over the ACTIVE list can take (launch_linesearch falls back to rounds above it)
int spec_open_capacity(const DeviceState& s) { return spec_entries(s, kNumAlpha); }

// one pass: step sizes r0..10 of entries [off, off + n_max) of the list (n_ptr: its device-side length; nullptr = the active list)
static void launch_spec(const DeviceState& s, const int* list, const int* n_ptr, int off, int n_max, int n_grid, int r0,
                        int open, hipStream_t st) {
  const int na = kNumAlpha - r0;
  const int* lp = list + off;
  // the pending list of the hybrid schedule is sparse and unordered: packed kernels (eight lanes per problem)
  const int sparse = (!open && na <= 8) ? 1 : 0;
  if (sparse)
    hipLaunchKernelGGL(k_spec_forward_packed, dim3((n_grid + 7) / 8), dim3(64), 0, st, s, lp, n_ptr, off, n_max, r0);
  else
    hipLaunchKernelGGL(k_spec_forward, dim3((n_grid + 63) / 64, na), dim3(64), 0, st, s, lp, n_ptr, off, n_max, r0, open);
  launch_spec_cost(s, lp, n_ptr, off, n_max, n_grid, r0, sparse, st);
  hipLaunchKernelGGL(k_spec_reduce, dim3((n_grid + 63) / 64, na), dim3(64), 0, st, s, lp, n_ptr, off, n_max, r0, open);
  hipLaunchKernelGGL(k_spec_pick, dim3((n_grid + 63) / 64), dim3(64), 0, st, s, lp, n_ptr, off, n_max, r0);
  hipLaunchKernelGGL(k_spec_copy, dim3((n_grid + 255) / 256, s.p.K), dim3(256), 0, st, s, lp, n_ptr, off, n_max, r0);
}
// the step sizes R..10 of the problems on pending list R (all n_act of them at worst), in as many passes as the arena asks for
static void launch_spec_remainder(const DeviceState& s, int R, int n_act, hipStream_t st) {
  const int E = spec_entries(s, kNumAlpha - R);
  const DeviceState v = spec_view(s, R, E);
  for (int off = 0; off < n_act; off += E) {
    const int n_max = (n_act - off < E) ? n_act - off : E;
    // the first pass is sized for the usual few per cent of the active problems (threads stride over the rest); a further
    // pass finds its share of the list empty unless the scenes are adversarial, and is sized for that
    const int want = (off == 0) ? (n_act + 3) / 4 : (n_max + 15) / 16;
    const int n_grid = want < n_max ? (want > 0 ? want : 1) : n_max;
    launch_spec(v, s.pend + (size_t)R * s.Bcap, s.counters + R, off, n_max, n_grid, R, 0, st);
  }
}

// seq_rounds: how many step sizes are tried round by round before the rest is evaluated at once
void launch_linesearch(const DeviceState& s, int n_act, int spec_threshold, int seq_rounds, int round_group,
                       hipStream_t st) {
  if (n_act == 0) return;
  if (n_act <= spec_threshold && n_act <= spec_open_capacity(s)) {
    launch_spec(spec_view(s, 0, spec_open_capacity(s)), s.act, nullptr, 0, n_act, n_act, 0, 1, st);
    return;
  }
  const int R = seq_rounds < 1 ? 1 : (seq_rounds > kNumAlpha ? kNumAlpha : seq_rounds);
  if (R <= kMaxPreRolled && R <= s.spec_rows) {
    // pre-rolled rounds: one pass rolls out alpha_0 .. alpha_{R-1} of every active problem
    const dim3 gf((n_act + 63) / 64), bf(64);
    if (R <= 4) {   // one rollout per lane (measured: 3.9 -> 2.9 ms per solve against four rollouts per lane)
      if (R <= 2) hipLaunchKernelGGL(k_multi_forward_packed<2>, dim3((n_act * 2 + 63) / 64), bf, 0, st, s, n_act, R);
      else hipLaunchKernelGGL(k_multi_forward_packed<4>, dim3((n_act * 4 + 63) / 64), bf, 0, st, s, n_act, R);
    }
    else if (R == 5) hipLaunchKernelGGL(k_multi_forward<5>, gf, bf, 0, st, s, n_act);
    else hipLaunchKernelGGL(k_multi_forward<6>, gf, bf, 0, st, s, n_act);
    // G step sizes per round (round_group; the last round takes what is left of R)
    for (int r0 = 0; r0 < R;) {
      const int G = (round_group >= 4 && R - r0 >= 4) ? 4 : ((round_group >= 2 && R - r0 >= 2) ? 2 : 1);
      // later rounds carry a fraction of the batch: shrink the grids, stride inside
      const int shrink = (r0 == 0) ? 1 : ((r0 == 1 || G > 1) ? 2 : 8);
      const int n_grid = (n_act + shrink - 1) / shrink;
      launch_round_cost(s, r0, G, n_act, n_grid, st);
      // one lane per pending problem (never strided: a lane sums whole cost columns)
      hipLaunchKernelGGL(k_round_pick, dim3((n_act + 63) / 64), dim3(64), 0, st, s, r0, G, n_act, (r0 + G == R) ? 1 : 0);
      r0 += G;
    }
    hipLaunchKernelGGL(k_multi_copy, dim3((n_act + 255) / 256, (s.p.K + kCopyKnots - 1) / kCopyKnots), dim3(256), 0, st, s, n_act, R);
    if (R < kNumAlpha) launch_spec_remainder(s, R, n_act, st);
    return;
  }
  hipLaunchKernelGGL(k_search_open, dim3((n_act + 63) / 64), dim3(64), 0, st, s, n_act);
  for (int r = 0; r < R; ++r) {
    // later rounds carry a fraction of the batch: shrink the cost grid, stride inside
    const int shrink = (r == 0) ? 1 : (r == 1 ? 2 : 8);
    const int n_grid = (n_act + shrink - 1) / shrink;
    launch_cost_knots(s, (r == 0) ? (const int*)s.act : (const int*)(s.pend + (size_t)r * s.Bcap),
                      (r == 0) ? (const int*)nullptr : (const int*)(s.counters + r), n_act, n_grid, 1, 1, st);
    // one lane per pending problem (never strided: a lane's work is a whole 50-step rollout)
    hipLaunchKernelGGL(k_search_round, dim3((n_act + 63) / 64), dim3(64), 0, st, s, r, n_act,
                       (r + 1 < R || R == kNumAlpha) ? 1 : 0);
  }
  // the problems that rejected alpha_0..alpha_{R-1}: all remaining step sizes at once (the round-by-round rollouts went
  // into the iterates' own other buffers, so the whole candidate arena is free for them)
  if (R < kNumAlpha) launch_spec_remainder(s, R, n_act, st);
}

// per-problem bookkeeping after the line search (cc:272-308, 312-319) + next active list
// once per solve: the ring of active counts starts with the batch size, everything else at zero
// (afterwards k_update's epilogue keeps the counters)
__global__ void k_init_counters(DeviceState s, int first_n) {
  if (threadIdx.x < 64) s.counters[threadIdx.x] = (threadIdx.x == kCntActive) ? first_n : 0;
}
void launch_init_counters(const DeviceState& s, int first_n, hipStream_t st) {
  hipLaunchKernelGGL(k_init_counters, dim3(1), dim3(64), 0, st, s, first_n);
}

// The last block to finish publishes the survivor count to the host (pinned memory: no copy
// kernel), clears the pending-list counters and the ring entry of the iteration after next: the
// bookkeeping between two lockstep iterations needs no kernel of its own.
CILQR_DEV void update_epilogue(const DeviceState& s) {
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = (atomicAdd(&s.counters[kCntTicket], 1) == (int)gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    const int total = atomicAdd(s.n_next, 0);
    *s.h_count_dev = total;
    *s.n_clear = 0;
    s.counters[kCntTicket] = 0;
#pragma unroll
    for (int r = 0; r <= kNumAlpha; ++r) s.counters[r] = 0;
    __threadfence_system();
  }
}

CILQR_DEV void update_problem(const DeviceState& s, int j) {
  const int slot = s.act[j];
  if (!update_state(s, s, slot)) {
    const int pos = atomicAdd(s.n_next, 1);
    s.act_next[pos] = slot;
    if (s.posn) s.posn[slot] = pos;       // where the next iteration's lin / term / gains of this slot will live
  }
}
__global__ __launch_bounds__(256) void k_update(DeviceState s, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < active_count(s, n)) update_problem(s, j);
  update_epilogue(s);
}

void launch_update(const DeviceState& s, int n_act, hipStream_t st) {
  if (n_act == 0) return;
  hipLaunchKernelGGL(k_update, dim3((n_act + 255) / 256), dim3(256), 0, st, s, n_act);
}

}  // namespace cilqr
