// Tail of a batch: one workgroup per problem runs ALL remaining iterations of Optimize() in one launch.
//
// Reference behaviour: the body of the loop of IlqrOptimizer::Optimize (algorithm/ilqr/ilqr_optimizer.cc:201-319):
// quadratisation cc:203-214, Backward cc:218, gradient-norm exit cc:235-241, the 11-step line search cc:243-270,
// regularisation schedule and exits cc:272-308, iteration cap cc:312-319.
//
// Why.  After ~30 lockstep iterations under 2 % of a batch is still iterating and stays so for up to 70 more
// iterations (the stragglers are the problems whose iteration is chaotic).  A lockstep iteration over a few hundred
// problems is nine launches that each sit on a chain of dependent work (50 backward steps, 50 rollout steps) with
// the GPU otherwise idle: ~255 us per iteration, a quarter of a solve.  Here a problem's iterations follow each
// other inside one kernel: no launches, no bookkeeping kernels between the phases, and a problem that converges
// leaves at once instead of waiting for the slowest one of its iteration.
//
// Same arithmetic.  Every phase calls the device function the lockstep kernels call (knot_quadratize,
// backward_wave_problem, forward_core, knot_cost, update_state) on a view of the state in which the problem is
// the only one: the block copies its problem's working set into a private, contiguous arena and runs the functions
// with Bcap = 1, slot = 0.  Results are bit-identical to the lockstep path (tested with the tail switched off).
//
// Phases of one iteration (256 threads; `|` = __syncthreads):
//   quadratize (four lanes share a knot's planes | one lane per knot adds their sums up in plane order while another wave
//   evaluates the state-only part; one knot per thread where the split form's scratch rows do not fit)
//   | backward, wave 0 (one output element per lane) | exit test | rollouts of all 11 step sizes,
//   11 lanes | knot costs of alpha_0..4 (5 K items over the block) | totals, 5 lanes | first passing index;
//   only if none: alpha_5..9, then alpha_10 | the winner becomes the iterate | update_state | exports.
// The first passing list index wins whatever the evaluation order (cc:246-265), so evaluating the candidates in
// chunks of five and stopping at the first chunk with a winner gives the sequential loop's answer.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "backward_core.hpp"
#include "quad_core.hpp"
#include "search_core.hpp"

namespace cilqr {

#ifndef CILQR_TAIL_THREADS
#define CILQR_TAIL_THREADS 256
#endif
constexpr int kTailThreads = CILQR_TAIL_THREADS;
constexpr int kTailChunk = 5;   // step sizes costed together: 5 x 51 knots = 255 items for 256 threads

// byte offsets of the tensors inside one block's private arena
struct TailLayout {
  size_t X, U, goals, cor, ccnt, lin, term, gains, Xs, Us, parts, spec_tot, dbl, ints, stride;
};
constexpr size_t kNotInLds = ~(size_t)0;
struct TailArgs {
  char* ws;
  TailLayout L;
  TailLayout S;     // byte offsets of the tensors that live in LDS instead (kNotInLds: in the private arena of `ws`)
  int lds_base;     // byte offset of that block inside the kernel's dynamic shared memory
  double* traj;
  double* iter_trajs;
  int it_cap;
  int* max_iter;   // host-visible: largest iteration count a problem of the tail reached
  int bwd_in_lds;  // lin, term, gains, U and the scalars are in LDS: the backward pass runs on ds_* instructions
  int fwd_in_lds;  // X, U, gains, goals, Xs, Us are: so do the rollouts
  int fwd_src_in_lds;   // X, U, gains, goals only (the rollouts' loads)
  int quad_in_lds; // X, U, goals, cor, ccnt, lin, term
  int quad_scr;    // >= 0: the quadratisation runs in its split form, with its per-plane sums at this byte offset of the
                   // block's LDS (the candidates' rows -- parts, Xs, Us -- which are dead until the line search); -1: one lane per knot
  int cost_in_lds; // Xs, Us, parts, goals, cor, ccnt
};

// Plane capacity of the view's corridor tensor.  A knot's planes are read by the lane that owns the knot (quadratisation,
// knot costs): with the batch's capacity as the stride -- 16 planes x 3 doubles = 96 dwords -- the 64 lanes of a wave fall
// into TWO bank pairs of LDS (96 i mod 64), every plane load a 32-way conflict; an odd capacity spreads them over all banks.
__host__ __device__ constexpr int tail_cmax(int cmax) { return cmax | 1; }

static TailLayout tail_layout(const DeviceState& s) {
  const size_t K = s.p.K, N = s.p.N;
  TailLayout L;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 15) / 16 * 16;
    return at;
  };
  L.X = take(2 * K * 3 * sizeof(double2));
  L.U = take(2 * N * sizeof(double2));
  L.goals = take(K * 3 * sizeof(double2));
  L.cor = take(K * tail_cmax(s.cmax) * 3 * sizeof(double));
  L.ccnt = take(K * sizeof(int));
  L.lin = take(N * kLinPairs * sizeof(double2));
  L.term = take(kTermPairs * sizeof(double2));
  L.gains = take(N * kGainPairs * sizeof(double2));
  L.Xs = take((size_t)kNumAlpha * K * 3 * sizeof(double2));
  L.Us = take((size_t)kNumAlpha * N * sizeof(double2));
  L.parts = take((size_t)kNumAlpha * K * kPartPairs * sizeof(double2));
  L.spec_tot = take((size_t)kNumAlpha * 5 * sizeof(double));
  L.dbl = take(16 * sizeof(double));   // dV[2] gnorm trial[5] lambda dlambda cost_old dcost
  L.ints = take(8 * sizeof(int));      // cur pid done_now upd acc_idx emit
  L.stride = (o + 255) / 256 * 256;
  return L;
}
size_t tail_workspace_bytes(const DeviceState& s) { return tail_layout(s).stride; }

// the state as the problem of block `blk` sees it: itself in slot 0 of an arena of capacity 1
CILQR_DEV DeviceState tail_view(const DeviceState& g, const TailArgs& a, int blk, char* lds_block) {
  DeviceState t = g;
  char* p = a.ws + (size_t)blk * a.L.stride;
  // a tensor of the problem lives in LDS when the launch found room for it (tail_lds_layout), else in the arena
  auto at = [&](size_t in_arena, size_t in_lds) -> char* { return in_lds != kNotInLds ? lds_block + in_lds : p + in_arena; };
  t.Bcap = 1;
  t.spec_cap = 1;
  t.cmax = tail_cmax(g.cmax);
  t.X = reinterpret_cast<double2*>(at(a.L.X, a.S.X));
  t.U = reinterpret_cast<double2*>(at(a.L.U, a.S.U));
  t.goals = reinterpret_cast<double2*>(at(a.L.goals, a.S.goals));
  t.cor = reinterpret_cast<double*>(at(a.L.cor, a.S.cor));
  t.ccnt = reinterpret_cast<int*>(at(a.L.ccnt, a.S.ccnt));
  t.lin = reinterpret_cast<double2*>(at(a.L.lin, a.S.lin));
  t.term = reinterpret_cast<double2*>(at(a.L.term, a.S.term));
  t.gains = reinterpret_cast<double2*>(at(a.L.gains, a.S.gains));
  t.Xs = reinterpret_cast<double2*>(at(a.L.Xs, a.S.Xs));
  t.Us = reinterpret_cast<double2*>(at(a.L.Us, a.S.Us));
  t.parts = reinterpret_cast<double2*>(at(a.L.parts, a.S.parts));
  t.spec_tot = reinterpret_cast<double*>(at(a.L.spec_tot, a.S.spec_tot));
  double* d = reinterpret_cast<double*>(at(a.L.dbl, a.S.dbl));
  t.dV = d;            // [2]
  t.gnorm = d + 2;
  t.trial = d + 3;     // [5]
  t.lambda = d + 8;
  t.dlambda = d + 9;
  t.cost_old = d + 10;
  t.dcost = d + 11;
  int* q = reinterpret_cast<int*>(at(a.L.ints, a.S.ints));
  t.cur = q;
  t.pid = q + 1;
  t.done_now = q + 2;
  t.upd = q + 3;
  t.acc_idx = q + 4;
  t.emit = q + 5;
  t.part = nullptr;    // the tail costs candidates only (parts)
  t.n_dev = nullptr;
  t.posn = nullptr;    // one problem, slot 0: its scratch is at position 0
  return t;
}

// The backward phase as a function of its own: inlined, its loop inherits the scalar-register pressure of the whole
// kernel (exec masks of its lane roles spilled to vector lanes: 57 v_readlane per step); called, it is allocated
// on its own.  LDS is reached through the kernel's dynamic shared array, so the accesses stay ds_* instructions.
// InLds: every tensor the phase touches lives in LDS (TailArgs::bwd_in_lds / fwd_in_lds; the usual case: horizons up to ~100)
template <bool InLds>
__device__ __attribute__((noinline)) void tail_backward(int off_T, int off_view) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  backward_wave_problem<WaveSync, true, InLds>(t, 0, t.lambda[0], (int)threadIdx.x, lds + off_T, WaveSync{});
}

// ... and so are the other three heavy phases (the lane tables sit at the start of the dynamic shared array)
#ifndef CILQR_TAIL_AHEAD
#define CILQR_TAIL_AHEAD 4
#endif
// SrcLds: what a rollout READS (X, U, gains, goals) is in LDS; OutLds: so are the candidates' rows it writes (Xs, Us).  Long
// horizons keep the first but not the second (eleven candidates of N = 100 no longer fit): the loads are what a step waits for.
template <bool SrcLds, bool OutLds>
__device__ __attribute__((noinline)) void tail_forward(int off_view) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  const int tid = (int)threadIdx.x;
  // operands in LDS are a short round trip away: one step ahead is enough (and a quarter of the registers)
  forward_core<OutSpecSolo<OutLds>, SrcLds ? 1 : CILQR_TAIL_AHEAD, true, SrcLds>(t, 0, kAlpha[tid], OutSpecSolo<OutLds>(t, tid));
}
template <int D, bool EX, bool InLds>
__device__ __attribute__((noinline)) void tail_quadratize(int off_view, int i) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  knot_quadratize<D, EX, InLds>(t, lds, t.cur[0], i, 0);
}
// The split form (quad_core.hpp: knot_plane_items / knot_commit_items): kQuadParts lanes share a knot's planes, then one lane
// adds their sums up in order while another evaluates the part that depends on the knot's state and control alone.
constexpr int kQuadParts = 4;
template <bool EX>
__device__ __attribute__((noinline)) void tail_quad_items(int off_view, int off_scr, int cap, int i, int part) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  knot_plane_items<EX, true>(t, lds, t.cur[0], i, part, kQuadParts, reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + off_scr), cap,
                             quad_split_stride(cap));
}
__device__ __attribute__((noinline)) void tail_quad_commit(int off_view, int off_scr, int cap, int i) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  knot_commit_items<true>(t, t.cur[0], i, reinterpret_cast<const double*>(reinterpret_cast<const char*>(lds) + off_scr), cap,
                          quad_split_stride(cap));
}
template <bool EX>
__device__ __attribute__((noinline)) void tail_quad_first_part(int off_view, int i) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  knot_quadratize<5, EX, true, true>(t, lds, t.cur[0], i, 0);
}
template <int D, bool EX, bool InLds>
__device__ __attribute__((noinline)) void tail_knot_cost(int off_view, int r, int i) {
  extern __shared__ double lds[];
  const DeviceState& t = *reinterpret_cast<const DeviceState*>(reinterpret_cast<const char*>(lds) + off_view);
  const int K = t.p.K, N = t.p.N;
  const double2* xb = t.Xs + ((size_t)r * K + i) * 3;
  assume_lds<InLds>(xb);
  const double2 p0 = xb[0], p1 = xb[1], p2 = xb[2];
  const double x[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
  double u[2] = {0.0, 0.0};
  if (i < N) {
    const double2* ub = t.Us + (size_t)r * N + i;
    assume_lds<InLds>(ub);
    const double2 q = *ub;
    u[0] = q.x; u[1] = q.y;
  }
  knot_cost<D, EX, InLds>(t, lds, i, 0, x, u, t.parts + ((size_t)r * K + i) * kPartPairs, 1);
}

#ifdef CILQR_TAIL_PROFILE
#define TP_DECL long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tp_t = __builtin_readcyclecounter(); int tp_it = 0;
#define TP(k) do { const long long n_ = __builtin_readcyclecounter(); tp_[k] += n_ - tp_t; tp_t = n_; } while (0)
#else
#define TP_DECL
#define TP(k)
#endif
#ifdef CILQR_TAIL_OCC
#define CILQR_TAIL_ATTR __attribute__((amdgpu_waves_per_eu(CILQR_TAIL_OCC, CILQR_TAIL_OCC)))
#else
#define CILQR_TAIL_ATTR
#endif
template <int D, bool EX>
__global__ __launch_bounds__(kTailThreads) CILQR_TAIL_ATTR void k_tail(DeviceState g, TailArgs a, int n_max) {
  extern __shared__ double lds[];
  const int n = active_count(g, n_max);
  // list position (and private arena) of this workgroup: eight consecutive positions per XCD, because the copy-in below
  // gathers 8 or 16 bytes per row of the batch arena and the rest of each 128-byte line belongs to the neighbours
  // (dev_model.hpp: xcd_local_position; needs a grid of a multiple of 64 workgroups -- a handful of problems, the drop-in call
  // among them, is launched as it is)
  const int blk = (gridDim.x & 63u) == 0 ? xcd_local_position((int)blockIdx.x) : (int)blockIdx.x;
  if (blk >= n) return;
  const int tid = threadIdx.x;
  (void)stage_lanes(g, lds);   // lane tables -> the start of the dynamic shared array (read by the phase functions)
  double* T = lds + ((g.nl + g.nr) * kLaneFields + 1) / 2 * 2;   // operands of the backward pass
  double* tot = T + wave::lds_doubles(g.p.N);                    // [11][5] candidate totals
  int* flag = reinterpret_cast<int*>(tot + kNumAlpha * 5);       // [0] leaves before the search [1] accepted index [2] done
  // The view sits in LDS, not in registers: its ~70 uniform fields on top of the kernel's own would spill the scalar
  // registers into vector lanes (385 spills, 159 v_readlane per backward step measured: +60 % on that phase).
  DeviceState* tv = reinterpret_cast<DeviceState*>(flag + 16);
  // The problem's working set lives in LDS as far as it fits (all of it at N = 50, Cmax = 16: 111 KB; a workgroup has
  // a CU to itself anyway): every phase of an iteration is a chain of dependent loads and stores -- 51 L2 round trips
  // for a candidate's total, a round trip per step of the backward pass and of a rollout -- and LDS answers in a tenth
  // of the time of L2.  What does not fit (the corridor planes of long horizons first) stays in the private arena.
  if (tid == 0) *tv = tail_view(g, a, blk, reinterpret_cast<char*>(lds) + a.lds_base);
  __syncthreads();
  const DeviceState& t = *tv;
  const int off_view = (int)(reinterpret_cast<const char*>(tv) - reinterpret_cast<const char*>(lds));
  const int K = g.p.K, N = g.p.N;
  const size_t Bc = (size_t)g.Bcap;

  {  // working set of the problem -> private arena (what k_compact moves, same rules: iterate into buffer 0, upd = 1)
    const int src = g.act[blk];
    const int buf = g.cur[src];
    for (int i = tid; i < K; i += kTailThreads) {
      const double2* x = g.X + ((size_t)buf * K + i) * 3 * Bc + src;
      t.X[i * 3 + 0] = x[0];
      t.X[i * 3 + 1] = x[Bc];
      t.X[i * 3 + 2] = x[2 * Bc];
      if (i < N) t.U[i] = g.U[((size_t)buf * N + i) * Bc + src];
      const double2* gg = g.goals + (size_t)i * 3 * Bc + src;
      t.goals[i * 3 + 0] = gg[0];
      t.goals[i * 3 + 1] = gg[Bc];
      t.goals[i * 3 + 2] = gg[2 * Bc];
      t.ccnt[i] = g.ccnt[(size_t)i * Bc + src];
    }
    const int rows = g.cmax * 3, rows_t = tail_cmax(g.cmax) * 3;
    for (int e = tid; e < K * rows; e += kTailThreads) {
      const int i = e / rows, r = e - i * rows;
      if (r < g.ccnt[(size_t)i * Bc + src] * 3) t.cor[i * rows_t + r] = g.cor[((size_t)i * rows + r) * Bc + src];
    }
    if (tid == 0) {
      t.cur[0] = 0;
      t.pid[0] = g.pid[src];
      t.lambda[0] = g.lambda[src];
      t.dlambda[0] = g.dlambda[src];
      t.cost_old[0] = g.cost_old[src];
      t.dcost[0] = g.dcost[src];
      t.upd[0] = 1;
      t.acc_idx[0] = -1;
      t.emit[0] = 0;
      t.done_now[0] = 0;
    }
  }
  __syncthreads();
  const int pb = t.pid[0];
  TP_DECL
  for (;;) {
    TP(7);
    if (t.upd[0]) {                                                        // cc:203-214
      if (D == 5 && a.quad_scr >= 0) {
        const int off_scr = a.lds_base + a.quad_scr, cap = quad_split_planes(g.cmax);   // the batch's plane capacity, not the view's padded one
        for (int w = tid; w < K * kQuadParts; w += kTailThreads) tail_quad_items<EX>(off_view, off_scr, cap, w / kQuadParts, w % kQuadParts);
        __syncthreads();
        // even waves add the planes' sums up, odd waves evaluate the state-only part: different SIMDs, disjoint outputs
        const int wv = tid >> 6;
        for (int i = (wv >> 1) * 64 + (tid & 63); i < K; i += kTailThreads / 2) {
          if (wv & 1) tail_quad_first_part<EX>(off_view, i);
          else tail_quad_commit(off_view, off_scr, cap, i);
        }
      } else if (a.quad_in_lds) { for (int i = tid; i < K; i += kTailThreads) tail_quadratize<D, EX, true>(off_view, i); }
      else { for (int i = tid; i < K; i += kTailThreads) tail_quadratize<D, EX, false>(off_view, i); }
    }
    __syncthreads();
    TP(0);
    if (tid < 64) {                                                        // cc:218 (wave 0)
      if (a.bwd_in_lds) tail_backward<true>((int)(T - lds), off_view);
      else tail_backward<false>((int)(T - lds), off_view);
    }
    __syncthreads();
    TP(1);
    if (tid == 0) {                                                        // cc:235-241
      const bool leave = leaves_before_search(t, 0, true);
      t.acc_idx[0] = leave ? -2 : -1;
      flag[0] = leave ? 1 : 0;
      flag[1] = -1;
    }
    __syncthreads();
    if (!flag[0]) {
      if (tid < kNumAlpha) {                                               // cc:246-250, all step sizes
        if (a.fwd_in_lds) tail_forward<true, true>(off_view);
        else if (a.fwd_src_in_lds) tail_forward<true, false>(off_view);
        else tail_forward<false, false>(off_view);
      }
      __syncthreads();
      TP(2);
      int acc = -1;
      for (int r0 = 0; r0 < kNumAlpha && acc < 0; r0 += kTailChunk) {
        const int nr = min(kTailChunk, kNumAlpha - r0);
        for (int e = tid; e < nr * K; e += kTailThreads) {
          const int rr = e / K, i = e - rr * K, r = r0 + rr;
          if (a.cost_in_lds) tail_knot_cost<D, EX, true>(off_view, r, i);
          else tail_knot_cost<D, EX, false>(off_view, r, i);
        }
        __syncthreads();
        TP(3);
        if (tid < nr) {   // total of candidate r: knot partials in index order (k_spec_reduce)
          const int r = r0 + tid;
          double jj = 0.0, dx = 0.0, du = 0.0, cc = 0.0, lc = 0.0;
          const double2* pp = t.parts + (size_t)r * K * kPartPairs;
#pragma unroll 8
          for (int i = 0; i < K; ++i) {
            const double2* o = pp + (size_t)i * kPartPairs;
            const double2 aa = o[0], c2 = o[2];     // (J, bounds) of the state; (corridor, lane)
            jj += aa.x;
            dx += aa.y;
            cc += c2.x;
            lc += c2.y;
          }
#pragma unroll 8
          for (int i = 0; i < N; ++i) {
            const double2 bb = pp[(size_t)i * kPartPairs + 1];   // (J, bounds) of the control
            jj += bb.x;
            du += bb.y;
          }
          const double dyn = dx + du;
          double* tr = tot + r * 5;
          tr[0] = jj + dyn + cc + lc;
          tr[1] = jj; tr[2] = dyn; tr[3] = cc; tr[4] = lc;
        }
        __syncthreads();
        if (tid == 0) {   // first passing step size of the chunk (cc:252-261, k_spec_pick)
          const double cost_old = t.cost_old[0], dV0 = t.dV[0], dV1 = t.dV[1];
          int won = -1;
          double dcost = 0.0;
          for (int r = r0; r < r0 + nr; ++r) {
            const double alpha = kAlpha[r];
            dcost = cost_old - tot[r * 5];
            const double expected = -alpha * (dV0 + alpha * dV1);
            const double z = dcost / expected;
            if ((z > 1e-4 && z < 10.0) && dcost > 0.0) {
              won = r;
              break;
            }
          }
          const bool all_tried = (r0 + nr == kNumAlpha);
          if (won >= 0 || all_tried) {
            const int last = (won >= 0) ? won : kNumAlpha - 1;
#pragma unroll
            for (int c = 0; c < 5; ++c) t.trial[c] = tot[last * 5 + c];
          }
          if (won >= 0) {
            t.acc_idx[0] = won;
            t.dcost[0] = dcost;
            t.cur[0] ^= 1;
          }
          flag[1] = won;
        }
        __syncthreads();
        acc = flag[1];
        TP(4);
      }
      if (acc >= 0) {   // the accepted candidate becomes the iterate
        const int nb = t.cur[0];
        for (int i = tid; i < K; i += kTailThreads) {
          const double2* xb = t.Xs + ((size_t)acc * K + i) * 3;
          double2* o = t.X + ((size_t)nb * K + i) * 3;
          o[0] = xb[0];
          o[1] = xb[1];
          o[2] = xb[2];
          if (i < N) t.U[(size_t)nb * N + i] = t.Us[(size_t)acc * N + i];
        }
      }
    }
    __syncthreads();
    if (tid == 0) flag[2] = update_state(t, g, 0) ? 1 : 0;                // cc:272-319
    __syncthreads();
    const bool done = flag[2] != 0;
    TP(5);
#ifdef CILQR_TAIL_PROFILE
    ++tp_it;
    if (done && tid == 0 && (blk & 31) == 0)
      printf("tail blk %d iters %d cycles: quad %lld bwd %lld exit+fwd %lld cost %lld reduce+pick %lld copy+update %lld export %lld\n", blk, tp_it,
             tp_[0] / tp_it, tp_[1] / tp_it, tp_[2] / tp_it, tp_[3] / tp_it, tp_[4] / tp_it, tp_[5] / tp_it, tp_[7] / tp_it);
#endif
    if (a.iter_trajs && t.emit[0]) {
      const int idx = g.n_iter_trajs[pb] - 1;
      if (idx < a.it_cap)
        for (int i = tid; i < K; i += kTailThreads)
          write_traj_point(t, t.cur[0], i, 0, a.iter_trajs + (((size_t)pb * a.it_cap + idx) * K + i) * 10);
    }
    if (done) {
      for (int i = tid; i < K; i += kTailThreads)
        write_traj_point(t, t.cur[0], i, 0, a.traj + ((size_t)pb * K + i) * 10);
      if (tid == 0) atomicMax(a.max_iter, g.iter[pb]);
      break;
    }
  }
}

static size_t tail_fixed_lds(const DeviceState& g) {
  const size_t lane_d = (size_t)((g.nl + g.nr) * kLaneFields + 1) / 2 * 2;
  return ((lane_d + wave::lds_doubles(g.p.N) + kNumAlpha * 5) * sizeof(double) + 16 * sizeof(int) + sizeof(DeviceState) + 31) / 16 * 16;
}
// lane tables + the backward pass's operands and per-step rows + the view: within the 64 KiB no attribute has to grant
bool tail_supported(const DeviceState& g) { return tail_fixed_lds(g) <= 62 * 1024; }

// g.act / g.n_dev: the active list the tail takes over (n_max bounds its length and sizes the grid)
void launch_tail(const DeviceState& g, void* workspace, int n_max, double* traj, double* iter_trajs,
                 int max_iter_trajs, int* max_iter_dev, hipStream_t st) {
  if (n_max <= 0) return;
  TailArgs a;
  a.ws = static_cast<char*>(workspace);
  a.L = tail_layout(g);
  a.traj = traj;
  a.iter_trajs = iter_trajs;
  a.it_cap = max_iter_trajs;
  a.max_iter = max_iter_dev;
  const size_t fixed = tail_fixed_lds(g);
  // Dynamic shared memory beyond the default has to be asked for, once per kernel and device; what the device grants
  // (160 KiB per workgroup on gfx950, 64 KiB on older CDNA) sizes the budget below.  One-time, guarded: several handles'
  // worker threads come through here.
  struct DeviceLds { std::once_flag once; size_t bytes = 0; };
  static DeviceLds lds_of[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  DeviceLds& dl = lds_of[dev & 63];
  std::call_once(dl.once, [&] {
    int max_lds = 0;
    if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || max_lds <= 0) max_lds = 64 * 1024;
    const void* fns[] = {reinterpret_cast<const void*>(&k_tail<5, false>), reinterpret_cast<const void*>(&k_tail<5, true>),
                         reinterpret_cast<const void*>(&k_tail<0, false>), reinterpret_cast<const void*>(&k_tail<0, true>)};
    bool ok = true;
    for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) == hipSuccess;
    if (!ok) {            // keep to what needs no permission; the tensors that do not fit stay in the global-memory arena
      (void)hipGetLastError();
      max_lds = std::min(max_lds, 64 * 1024);
    }
    dl.bytes = (size_t)max_lds;
  });
  // tensors into LDS by priority until the workgroup's share (one workgroup per CU: all of its LDS less a margin) is used up
  const size_t budget = dl.bytes > fixed + 2048 ? dl.bytes - 2048 - fixed : 0;
  const size_t K = g.p.K, N = g.p.N;
  struct Item { size_t TailLayout::*field; size_t bytes; };
  const Item items[] = {
      {&TailLayout::dbl, 16 * sizeof(double)}, {&TailLayout::ints, 8 * sizeof(int)}, {&TailLayout::spec_tot, (size_t)kNumAlpha * 5 * sizeof(double)},
      {&TailLayout::term, kTermPairs * sizeof(double2)}, {&TailLayout::gains, N * kGainPairs * sizeof(double2)},
      {&TailLayout::X, 2 * K * 3 * sizeof(double2)}, {&TailLayout::U, 2 * N * sizeof(double2)},
      {&TailLayout::lin, N * kLinPairs * sizeof(double2)}, {&TailLayout::parts, (size_t)kNumAlpha * K * kPartPairs * sizeof(double2)},
      {&TailLayout::Xs, (size_t)kNumAlpha * K * 3 * sizeof(double2)}, {&TailLayout::Us, (size_t)kNumAlpha * N * sizeof(double2)},
      {&TailLayout::goals, K * 3 * sizeof(double2)}, {&TailLayout::ccnt, K * sizeof(int)}, {&TailLayout::cor, K * tail_cmax(g.cmax) * 3 * sizeof(double)}};
  size_t used = 0;
  a.S = a.L;
  for (const Item& it : items) {
    const size_t b = (it.bytes + 15) / 16 * 16;
    if (used + b <= budget) { a.S.*(it.field) = used; used += b; }
    else a.S.*(it.field) = kNotInLds;
  }
  a.lds_base = (int)fixed;
  auto in_lds = [&](size_t TailLayout::*f) { return a.S.*f != kNotInLds; };
  a.bwd_in_lds = in_lds(&TailLayout::lin) && in_lds(&TailLayout::term) && in_lds(&TailLayout::gains) && in_lds(&TailLayout::U) &&
                 in_lds(&TailLayout::dbl);
  a.fwd_src_in_lds = in_lds(&TailLayout::X) && in_lds(&TailLayout::U) && in_lds(&TailLayout::gains) && in_lds(&TailLayout::goals);
  a.fwd_in_lds = a.fwd_src_in_lds && in_lds(&TailLayout::Xs) && in_lds(&TailLayout::Us);
  a.quad_in_lds = in_lds(&TailLayout::X) && in_lds(&TailLayout::U) && in_lds(&TailLayout::goals) && in_lds(&TailLayout::cor) &&
                  in_lds(&TailLayout::ccnt) && in_lds(&TailLayout::lin) && in_lds(&TailLayout::term);
  a.cost_in_lds = in_lds(&TailLayout::Xs) && in_lds(&TailLayout::Us) && in_lds(&TailLayout::parts) && in_lds(&TailLayout::goals) &&
                  in_lds(&TailLayout::cor) && in_lds(&TailLayout::ccnt);
  {  // split quadratisation: needs its tensors and the (then idle) candidate rows in LDS, the latter in one piece
    auto al = [](size_t b) { return (b + 15) / 16 * 16; };
    const size_t b_parts = al((size_t)kNumAlpha * K * kPartPairs * sizeof(double2)), b_xs = al((size_t)kNumAlpha * K * 3 * sizeof(double2)),
                 b_us = al((size_t)kNumAlpha * N * sizeof(double2));
    const bool one_piece = in_lds(&TailLayout::parts) && in_lds(&TailLayout::Xs) && in_lds(&TailLayout::Us) &&
                           a.S.Xs == a.S.parts + b_parts && a.S.Us == a.S.Xs + b_xs;
    const size_t need = K * (size_t)quad_split_doubles(g.cmax) * sizeof(double);
    const char* env = std::getenv("CILQR_TAIL_QUAD_SPLIT");   // "0": one lane per knot (A/B measurements; both forms are tested)
    const bool off = env && env[0] == '0';
    a.quad_scr = (!off && g.p.num_of_disc == 5 && a.quad_in_lds && one_piece && need <= b_parts + b_xs + b_us && kTailThreads == 256)
                     ? (int)a.S.parts : -1;
  }
  const size_t lds = fixed + used;
  if (g.p.num_of_disc == 5) {
    if (g.exact_ties) hipLaunchKernelGGL((k_tail<5, true>), dim3(n_max >= 64 ? (n_max + 63) / 64 * 64 : n_max), dim3(kTailThreads), lds, st, g, a, n_max);
    else hipLaunchKernelGGL((k_tail<5, false>), dim3(n_max >= 64 ? (n_max + 63) / 64 * 64 : n_max), dim3(kTailThreads), lds, st, g, a, n_max);
  } else {
    if (g.exact_ties) hipLaunchKernelGGL((k_tail<0, true>), dim3(n_max >= 64 ? (n_max + 63) / 64 * 64 : n_max), dim3(kTailThreads), lds, st, g, a, n_max);
    else hipLaunchKernelGGL((k_tail<0, false>), dim3(n_max >= 64 ? (n_max + 63) / 64 * 64 : n_max), dim3(kTailThreads), lds, st, g, a, n_max);
  }
}

}  // namespace cilqr
