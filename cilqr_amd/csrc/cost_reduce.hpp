// Total cost of a trajectory from its per-knot partials (dev_model.hpp declares reduce_cost).
#pragma once
#include "dev_model.hpp"
#include "ref_order.hpp"

namespace cilqr {

CILQR_DEV void reduce_cost(const DeviceState& s, int slot, int cand, double* c5) {
#ifdef CILQR_REF_ORDER
  const int buf = s.cur[slot] ^ cand;
  reforder::total_cost(
      s, slot, [&](int i, double* x) { load_x(s, buf, i, slot, x); }, [&](int i, double* u) { load_u(s, buf, i, slot, u); }, c5);
#else
  (void)cand;
  const int Bc = s.Bcap, K = s.p.K, N = s.p.N;
  double j = 0.0, dx = 0.0, du = 0.0, cc = 0.0, lc = 0.0;
  // loads of several knots in flight; the sums stay in knot order
#pragma unroll 8
  for (int i = 0; i < K; ++i) {
    const double2* o = s.part + (size_t)i * kPartPairs * Bc + slot;
    const double2 a = o[0], c = o[(size_t)2 * Bc];     // (J, bounds) of the state; (corridor, lane)
    j += a.x;
    dx += a.y;
    cc += c.x;
    lc += c.y;
  }
#pragma unroll 8
  for (int i = 0; i < N; ++i) {   // control terms follow the state terms (cc:510-513)
    const double2 b = s.part[((size_t)i * kPartPairs + 1) * Bc + slot];   // (J, bounds) of the control
    j += b.x;
    du += b.y;
  }
  const double dyn = dx + du;                      // cc:550
  c5[0] = j + dyn + cc + lc;                       // cc:429
  c5[1] = j; c5[2] = dyn; c5[3] = cc; c5[4] = lc;
#endif
}

#ifdef CILQR_REF_ORDER
// candidate alpha_r of list position j in the speculative arena
CILQR_DEV void spec_total_cost(const DeviceState& s, int slot, int r, int j, double* c5) {
  const size_t cap = (size_t)s.spec_cap;
  reforder::total_cost(
      s, slot,
      [&](int i, double* x) {
        const double2* xb = s.Xs + ((size_t)r * s.p.K + i) * 3 * cap + j;
        const double2 p0 = xb[0], p1 = xb[cap], p2 = xb[2 * cap];
        x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y;
      },
      [&](int i, double* u) {
        const double2 q = s.Us[((size_t)r * s.p.N + i) * cap + j];
        u[0] = q.x; u[1] = q.y;
      },
      c5);
}
#endif

}  // namespace cilqr
