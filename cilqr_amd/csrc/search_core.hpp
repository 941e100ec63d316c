// Forward pass and per-problem state machine (device functions shared by kernels_search.hip and kernels_tail.hip).
//
// Reference behaviour (algorithm/ilqr/ilqr_optimizer.cc): Forward cc:392-415; alpha list cc:197; gradient-norm
// exit cc:235-241; regularisation schedule and exits cc:272-308, 312-319.
#pragma once
#include "cost_reduce.hpp"

namespace cilqr {

static __device__ const double kAlpha[kNumAlpha] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316,
                                             0.0158, 0.0079, 0.0040, 0.0020, 0.0010};


// where a rollout is written: the slot's candidate buffer, or the speculative arena
struct OutSlot {
  const DeviceState& s;
  int nb, slot;
  CILQR_DEV void x(int i, const double* v) const { store_x(s, nb, i, slot, v); }
  CILQR_DEV void u(int i, const double* v) const { store_u(s, nb, i, slot, v); }
};
struct OutSpec {
  const DeviceState& s;
  int r, j;
  CILQR_DEV void x(int i, const double* v) const {
    const size_t cap = (size_t)s.spec_cap;
    double2* b = s.Xs + ((size_t)r * s.p.K + i) * 3 * cap + j;
    b[0] = make_double2(v[0], v[1]);
    b[cap] = make_double2(v[2], v[3]);
    b[2 * cap] = make_double2(v[4], v[5]);
  }
  CILQR_DEV void u(int i, const double* v) const {
    s.Us[((size_t)r * s.p.N + i) * (size_t)s.spec_cap + j] = make_double2(v[0], v[1]);
  }
};

// the same for a problem that is alone in its arena (the tail kernel's private view: capacity 1, list position 0): the
// candidate's rows are contiguous, so a store is a pointer plus a compile-time offset
template <bool InLds>
struct OutSpecSolo {
  double2* xs;   // Xs + r K 3
  double2* us;   // Us + r N
  CILQR_DEV OutSpecSolo(const DeviceState& s, int r) : xs(s.Xs + (size_t)r * s.p.K * 3), us(s.Us + (size_t)r * s.p.N) {
    assume_lds<InLds>(xs);
    assume_lds<InLds>(us);
  }
  CILQR_DEV void x(int i, const double* v) const {
    double2* b = xs + (size_t)i * 3;
    b[0] = make_double2(v[0], v[1]);
    b[1] = make_double2(v[2], v[3]);
    b[2] = make_double2(v[4], v[5]);
  }
  CILQR_DEV void u(int i, const double* v) const { us[i] = make_double2(v[0], v[1]); }
};

// what step i of a rollout reads: nominal state / control and the gains K_i, k_i
#ifndef CILQR_ROLL_AHEAD
#define CILQR_ROLL_AHEAD 4
#endif
constexpr int kFwdAhead = CILQR_ROLL_AHEAD;
struct FwdStep {
  double2 x0, x1, x2, u, kk[kGainPairs];
};
// the tensors a rollout reads, as locals (a view in LDS -- kernels_tail.hip -- would be read again after every store)
struct FwdSrc {
  const double2* X;      // nominal states of the current iterate: X + buf K 3 Bc + slot
  const double2* U;      // U + buf N Bc + slot
  const double2* gains;  // gains + sp (sp: scratch_index(s, slot), where the slot's gains live this iteration)
  size_t Bc;
};
CILQR_DEV FwdSrc fwd_source(const DeviceState& s, int buf, int slot, int sp, size_t Bc) {
  return FwdSrc{s.X + (size_t)buf * s.p.K * 3 * Bc + slot, s.U + (size_t)buf * s.p.N * Bc + slot, s.gains + sp, Bc};
}
CILQR_DEV void load_fwd_step(const FwdSrc& q, int i, FwdStep& f) {
  const double2* b = q.X + (size_t)i * 3 * q.Bc;
  f.x0 = b[0]; f.x1 = b[q.Bc]; f.x2 = b[2 * q.Bc];
  f.u = q.U[(size_t)i * q.Bc];
  const double2* g = q.gains + (size_t)i * kGainPairs * q.Bc;
#pragma unroll
  for (int r = 0; r < kGainPairs; ++r) f.kk[r] = g[(size_t)r * q.Bc];
}

// roll the closed-loop policy out from goals_[0] (cc:392-415).  kAhead: steps whose operands are requested ahead of
// the arithmetic -- 4 where a lane is alone on its SIMD (176 VGPRs of loads in flight), 1 where several waves per
// SIMD hide each other's latency instead.  Solo: the problem is alone in its arena (capacity 1, slot 0, position 0:
// kernels_tail.hip) -- every stride is a compile-time constant and a step's operands are a pointer bump away.
// InLds (with Solo): X, U, gains, goals of the view are in LDS (dev_model.hpp: assume_lds; Out says the same of its rows).
template <class Out, int kAhead = kFwdAhead, bool Solo = false, bool InLds = false>
CILQR_DEV void forward_core(const DeviceState& s, int slot, double alpha, const Out& out) {
  const DynP p = dyn_params(s.p);
  const size_t Bc = Solo ? (size_t)1 : (size_t)s.Bcap;
  const int N = s.p.N;
  const int buf = s.cur[slot];
  const int sp = Solo ? 0 : scratch_index(s, slot);
  double x[6];
  {
    const double2* gp = s.goals + slot;
    const double2 g0 = gp[0], g1 = gp[Bc], g2 = gp[2 * Bc];
    x[0] = g0.x; x[1] = g0.y; x[2] = g1.x; x[3] = g1.y; x[4] = g2.x; x[5] = g2.y;
  }
  const FwdSrc src = fwd_source(s, buf, slot, sp, Bc);
  assume_lds<InLds>(src.X);
  assume_lds<InLds>(src.U);
  assume_lds<InLds>(src.gains);
  out.x(0, x);
  // The nominal (xs, us) and the gains of a step do not depend on the rollout, and one lane's step
  // is short (~0.5 us of arithmetic) against the latency of a load that misses L2 (the gains were
  // just written by another kernel): keep the loads of the next kFwdAhead steps in flight.
  FwdStep pf[kAhead];
#pragma unroll
  for (int d = 0; d < kAhead; ++d)
    if (d < N) load_fwd_step(src, d, pf[d]);
  for (int i0 = 0; i0 < N; i0 += kAhead) {
#pragma unroll
    for (int d = 0; d < kAhead; ++d) {
      const int i = i0 + d;
      if (i < N) {
        const FwdStep c = pf[d];
        if (i + kAhead < N) load_fwd_step(src, i + kAhead, pf[d]);
        const double xs[6] = {c.x0.x, c.x0.y, c.x1.x, c.x1.y, c.x2.x, c.x2.y};
        const double us[2] = {c.u.x, c.u.y};
        double dx[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) dx[e] = x[e] - xs[e];
        double u[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          double acc = c.kk[r * 3].x * dx[0];
          acc += c.kk[r * 3].y * dx[1];
          acc += c.kk[r * 3 + 1].x * dx[2];
          acc += c.kk[r * 3 + 1].y * dx[3];
          acc += c.kk[r * 3 + 2].x * dx[4];
          acc += c.kk[r * 3 + 2].y * dx[5];
          const double kff = (r == 0) ? c.kk[6].x : c.kk[6].y;
          u[r] = (us[r] + acc) + alpha * kff;                           // cc:407
        }
        // the straight-line step where a wavefront is alone with its chain (the tail kernel), the branchy one where several
        // waves per SIMD fill each other's gaps (the lockstep rollout kernels: 139 against 161 us per bulk launch, r04 log 11)
        closed_loop_step<Solo>(p, x, u, x);                               // cc:408-410
        out.u(i, u);
        out.x(i + 1, x);
      }
    }
  }
}
CILQR_DEV void forward_problem(const DeviceState& s, int slot, double alpha) {
  forward_core(s, slot, alpha, OutSlot{s, s.cur[slot] ^ 1, slot});
}

// G rollouts of one problem at once (alpha_0 .. alpha_{G-1}) into the speculative arena at list
// position j: the nominal trajectory and the gains of a step are loaded once for all of them, and
// the G dynamics chains are independent, which fills the latency of each other.  Same arithmetic
// per rollout as forward_core.
template <int G>
CILQR_DEV void forward_multi(const DeviceState& s, int slot, int j) {
  const DynP p = dyn_params(s.p);
  const int Bc = s.Bcap, N = s.p.N;
  const int buf = s.cur[slot];
  const int sp = scratch_index(s, slot);
  const FwdSrc src = fwd_source(s, buf, slot, sp, (size_t)Bc);
  double x[G][6];
  {
    const double2* gp = s.goals + slot;
    const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      x[r][0] = g0.x; x[r][1] = g0.y; x[r][2] = g1.x; x[r][3] = g1.y; x[r][4] = g2.x; x[r][5] = g2.y;
      OutSpec{s, r, j}.x(0, x[r]);
    }
  }
  FwdStep pf[kFwdAhead];
#pragma unroll
  for (int d = 0; d < kFwdAhead; ++d)
    if (d < N) load_fwd_step(src, d, pf[d]);
  for (int i0 = 0; i0 < N; i0 += kFwdAhead) {
#pragma unroll
    for (int d = 0; d < kFwdAhead; ++d) {
      const int i = i0 + d;
      if (i < N) {
        const FwdStep c = pf[d];
        if (i + kFwdAhead < N) load_fwd_step(src, i + kFwdAhead, pf[d]);
        const double xs[6] = {c.x0.x, c.x0.y, c.x1.x, c.x1.y, c.x2.x, c.x2.y};
        const double us[2] = {c.u.x, c.u.y};
#pragma unroll
        for (int r = 0; r < G; ++r) {
          double dx[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) dx[e] = x[r][e] - xs[e];
          double u[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            double acc = c.kk[q * 3].x * dx[0];
            acc += c.kk[q * 3].y * dx[1];
            acc += c.kk[q * 3 + 1].x * dx[2];
            acc += c.kk[q * 3 + 1].y * dx[3];
            acc += c.kk[q * 3 + 2].x * dx[4];
            acc += c.kk[q * 3 + 2].y * dx[5];
            const double kff = (q == 0) ? c.kk[6].x : c.kk[6].y;
            u[q] = (us[q] + acc) + kAlpha[r] * kff;                         // cc:407
          }
          closed_loop_step<false>(p, x[r], u, x[r]);                        // cc:408-410
          const OutSpec out{s, r, j};
          out.u(i, u);
          out.x(i + 1, x[r]);
        }
      }
    }
  }
}

// Leaves the iteration before the line search (acc_idx = -2 must then be set by ONE lane of the
// problem): gradient-norm exit (cc:235-241), or a problem that was never admissible (a knot without
// corridor, status 6 -- the reference aborts such a Plan before Optimize, corridor.cc:78-81).
CILQR_DEV bool leaves_before_search(const DeviceState& s, int slot, bool write) {
  const int pb = s.pid[slot];
  if (s.status[pb] == 6) return true;
  if (s.gnorm[slot] < 1e-6 && s.lambda[slot] < 1e-5) {   // cc:235-241
    if (write) s.status[pb] = 3;   // CILQR_ST_GNORM
    return true;
  }
  return false;
}

// Per-problem bookkeeping after the line search (cc:272-308, 312-319).  `s` holds the slot-indexed state, `g` the
// problem-indexed one (cost history, alpha trace: rows of g.Bcap problems) -- the same object in the lockstep
// kernels; kernels_tail.hip passes its block-private view as `s`.  Returns whether the problem has terminated.
CILQR_DEV bool update_state(const DeviceState& s, const DeviceState& g, int slot) {
  const int pb = s.pid[slot];
  const Params& p = s.p;
  bool done = false;
  int st = g.status[pb];
  s.emit[slot] = 0;
  const int it0 = g.iter[pb];
  if (st == 3 || st == 6) {
    done = true;
    g.atrace[(size_t)it0 * g.Pcap + pb] = (signed char)-2;
  } else {
    const int a = s.acc_idx[slot];
    g.atrace[(size_t)it0 * g.Pcap + pb] = (signed char)a;
    const double lam = s.lambda[slot], dl = s.dlambda[slot];
    if (a >= 0) {
      const double ndl = fmin(dl / 1.6, 1.0 / 1.6);                                // cc:273
      s.dlambda[slot] = ndl;
      s.lambda[slot] = lam * ndl * ((lam > 1e-8) ? 1.0 : 0.0);                     // cc:275
      s.upd[slot] = 1;
      const double dc = s.dcost[slot], co = s.cost_old[slot];
      const int nc = g.n_cost[pb];
#pragma unroll
      for (int c = 0; c < 5; ++c)
        g.hist[((size_t)nc * 5 + c) * g.Pcap + pb] = s.trial[(size_t)c * s.Bcap + slot];
      g.n_cost[pb] = nc + 1;
      if (dc < p.abs_tol || dc / co < p.rel_tol) {                                 // cc:281-293
        st = (dc < p.abs_tol) ? 1 : 2;
        done = true;
      } else {
        g.n_iter_trajs[pb] += 1;                                                   // cc:294
        s.emit[slot] = 1;
        s.cost_old[slot] = s.trial[slot];                                          // cc:295
      }
    } else {
      const double ndl = fmax(dl * 1.6, 1.6);                                      // cc:298
      const double nl = fmax(lam * ndl, 1e-8);                                     // cc:299
      s.dlambda[slot] = ndl;
      s.lambda[slot] = nl;
      s.upd[slot] = 0;
      if (nl > 1e11) {                                                             // cc:302
        st = 4;
        done = true;
      }
    }
  }
  const int it = it0 + 1;
  g.iter[pb] = it;
  if (!done && it >= p.max_iter) {                                                 // cc:312
    st = 5;
    done = true;
  }
  g.status[pb] = st;
  s.acc_idx[slot] = -1;
  s.done_now[slot] = done ? 1 : 0;
  return done;
}

}  // namespace cilqr
