// Backward pass of one problem (device functions shared by kernels_backward.hip and kernels_tail.hip).
//
// Reference behaviour: IlqrOptimizer::Backward (algorithm/ilqr/ilqr_optimizer.cc:334-390) and
// CalGradientNorm (cc:322-332); see kernels_backward.hip for the quirks that are kept.
//
// Compile-time switch CILQR_DV_EVAL ("hard parts"; the CPU checker the tests use carries the same switch):
//   default ("lazy")        delta_V_ (cc:383-384) from Qu / Quu RE-EVALUATED on the Vx / Vxx that cc:379-381 have just
//                           overwritten -- how Eigen's lazy `auto` expressions of cc:348-352 behave (read off Eigen's
//                           expression semantics; not executable in this image, which has no Eigen);
//   -DCILQR_DV_EVAL_EAGER   delta_V_ from the Qu / Quu the gains of the step were computed from.  TEST-ONLY build
//                           (`make dveager` -> lib/libcilqr_hip_dveager.so, tests/test_gpu_parity.py): it is held
//                           against the checker's eager variant, so both readings stay checked on the device.
// All three mappings below (one lane, eight lanes, one wavefront per problem) honour it and stay bit-identical to each other.
#pragma once
#include "dev_model.hpp"

namespace cilqr {

// structure of A (6x6) and B (6x2): 0 = exact zero, 1 = exact one, 2 = value
__host__ __device__ constexpr int a_kind(int r, int c) {
  return (r == c) ? 1
         : ((r == 0 || r == 1) && c >= 2) ? 2
         : (r == 2 && c >= 3) ? 2
         : (r == 3 && c == 4) ? 2
         : 0;
}
__host__ __device__ constexpr int b_kind(int r, int c) {
  return ((r == 2 && c == 1) || (r == 3 && c == 0) || (r == 4 && c == 0) || (r == 5 && c == 1)) ? 2 : 0;
}
// structure of lxx: dense 3x3 block + diagonal
__host__ __device__ constexpr int h_kind(int r, int c) {
  return (r < 3 && c < 3) ? 2 : (r == c) ? 2 : 0;
}

// sequential accumulation over terms some of which are structural zeros (x_m below; Acc: dev_model.hpp)
#define ACC_TERM(acc, t)            \
  do {                              \
    if ((acc).any) (acc).v += (t);  \
    else { (acc).v = (t); (acc).any = true; } \
  } while (0)

// out[R][6] = M^T X with M = A (6x6) or B (6x2): out(r,c) = sum_k M(k,r) X(k,c) -- an `X.transpose() * Y` product of the
// reference: the six terms are added as sum6_xty says (dev_model.hpp), structural zeros dropping out of the tree
template <int R, int xcols, bool IsA>
CILQR_DEV void mt_x(const double* __restrict__ M, const double* __restrict__ X, double* out) {
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < xcols; ++c) {
      Acc t[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int kind = IsA ? a_kind(k, r) : b_kind(k, r);
        if (kind == 0) { t[k] = Acc{0.0, false}; continue; }
        const double x = X[k * xcols + c];
        t[k] = Acc{(kind == 1) ? x : M[k * R + r] * x, true};
      }
      const Acc a = acc_sum6_xty(t);
      out[r * xcols + c] = a.any ? a.v : 0.0;
    }
}
// out[rows][C] = X M with M = A (6x6) or B (6x2): out(r,c) = sum_k X(r,k) M(k,c)
template <int C, int rows, bool IsA>
CILQR_DEV void x_m(const double* __restrict__ X, const double* __restrict__ M, double* out) {
#pragma unroll
  for (int r = 0; r < rows; ++r)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      Acc a{0.0, false};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int kind = IsA ? a_kind(k, c) : b_kind(k, c);
        if (kind == 0) continue;
        const double x = X[r * 6 + k];
        const double t = (kind == 1) ? x : x * M[k * C + c];
        ACC_TERM(a, t);
      }
      out[r * C + c] = a.any ? a.v : 0.0;
    }
}

// `stage`: two buffers of 18 rows x 64 lanes x 16 B in LDS.  The operands of the NEXT step wait there, not in registers:
// gfx950 loads 16 B per lane straight into LDS (global_load_lds_dwordx4, lane l of a row at offset 16 l), so the software
// pipeline costs no registers and no copies.  Round 6: 352 -> 276 VGPRs (no accumulation registers as spill space), 137 -> 34
// register moves per step; a launch over 16384 / 32768 / 65536 problems 163 / 191 / 266 -> 147 / 174 / 254 us, the same bits.
template <bool kStore>
CILQR_DEV void backward_problem(const DeviceState& s, int slot, double lambda, double2* stage) {
  const Params& p = s.p;
  const int Bc = s.Bcap, N = p.N;
  const double dt = p.dt;
  const int sp = scratch_index(s, slot);   // where this slot's lin / term / gains live this iteration
  double Vx[6], Vxx[36];
  {
    const double2* t = s.term + sp;
    const double2 t0 = t[0], t1 = t[(size_t)Bc], t2 = t[(size_t)2 * Bc], t3 = t[(size_t)3 * Bc],
                  t4 = t[(size_t)4 * Bc], t5 = t[(size_t)5 * Bc], t6 = t[(size_t)6 * Bc],
                  t7 = t[(size_t)7 * Bc], t8 = t[(size_t)8 * Bc];
    Vx[0] = t0.x; Vx[1] = t0.y; Vx[2] = t1.x; Vx[3] = t1.y; Vx[4] = t2.x; Vx[5] = t2.y;
#pragma unroll
    for (int e = 0; e < 36; ++e) Vxx[e] = 0.0;
    Vxx[0] = t3.x; Vxx[1] = t3.y; Vxx[2] = t4.x;
    Vxx[6] = t4.y; Vxx[7] = t5.x; Vxx[8] = t5.y;
    Vxx[12] = t6.x; Vxx[13] = t6.y; Vxx[14] = t7.x;
    Vxx[21] = t7.y; Vxx[28] = t8.x; Vxx[35] = t8.y;
  }
  const int buf = s.cur[slot];
  double dV0 = 0.0, dV1 = 0.0, gsum = 0.0;
  // software pipeline: the operands of step i-1 are requested before step i is computed, so a
  // wave (one per SIMD at B = 65536) always has 18 KiB of loads in flight behind its arithmetic
  double2 w[kLinPairs];
  double2 uu;
  constexpr int kRows = kLinPairs + 1;
  auto fetch_to_lds = [&](int step) {
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
    const double2* q = s.lin + (size_t)step * kLinPairs * Bc + sp;
    double2* dst = stage + (size_t)(step & 1) * kRows * 64;
#pragma unroll
    for (int r = 0; r < kLinPairs; ++r)
      __builtin_amdgcn_global_load_lds((gptr)(q + (size_t)r * Bc), (lptr)(dst + r * 64), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr)(s.U + ((size_t)buf * N + step) * Bc + slot), (lptr)(dst + kLinPairs * 64), 16, 0, 0);
  };
  fetch_to_lds(N - 1);
  for (int i = N - 1; i >= 0; --i) {
    {
      // this step's operands have landed (requested a step ago; the gains stored since go by the same counter and are long
      // acknowledged), the next step's go out before the arithmetic
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const double2* src = stage + (size_t)(i & 1) * kRows * 64 + (threadIdx.x & 63);
#pragma unroll
      for (int r = 0; r < kLinPairs; ++r) w[r] = src[r * 64];
      uu = src[kLinPairs * 64];
      if (i > 0) fetch_to_lds(i - 1);
    }
    // A and B as dense register arrays; entries of kind 0/1 are never read
    double A[36], B[12];
    A[2] = w[0].x; A[3] = w[0].y; A[4] = w[1].x; A[5] = w[1].y;
    A[8] = w[2].x; A[9] = w[2].y; A[10] = w[3].x; A[11] = w[3].y;
    A[15] = w[4].x; A[16] = w[4].y; A[17] = w[5].x;
    A[22] = dt;
    B[5] = w[5].y; B[6] = 0.5 * dt * dt; B[8] = dt; B[11] = dt;
    const double lx[6] = {w[6].x, w[6].y, w[7].x, w[7].y, w[8].x, w[8].y};
    const double lu[2] = {w[9].x, w[9].y};
    // lxx (kind 2 entries only)
    double H[36];
    H[0] = w[10].x; H[1] = w[10].y; H[2] = w[11].x;
    H[6] = w[11].y; H[7] = w[12].x; H[8] = w[12].y;
    H[12] = w[13].x; H[13] = w[13].y; H[14] = w[14].x;
    H[21] = w[14].y; H[28] = w[15].x; H[35] = w[15].y;
    const double luu0 = w[16].x, luu1 = w[16].y;

    // ---- quantities from the OLD Vx / Vxx ----
    double BtV[12], Qux[12], BtVB[4], BtVx[2];
    mt_x<2, 6, false>(B, Vxx, BtV);          // B^T Vxx
    x_m<6, 2, true>(BtV, A, Qux);            // (B^T Vxx) A                    cc:353
    x_m<2, 2, false>(BtV, B, BtVB);          // (B^T Vxx) B
    mt_x<2, 1, false>(B, Vx, BtVx);
    const double Quu[4] = {luu0 + BtVB[0], BtVB[1], BtVB[2], luu1 + BtVB[3]};      // cc:352
    const double Qu[2] = {lu[0] + BtVx[0], lu[1] + BtVx[1]};                       // cc:349
    // (Quu + lambda I)^-1, closed form                                             cc:361-363
    const double m00 = Quu[0] + lambda, m01 = Quu[1], m10 = Quu[2], m11 = Quu[3] + lambda;
    const double invdet = 1.0 / (m00 * m11 - m10 * m01);
    const double n00 = -(m11 * invdet), n01 = -(-m01 * invdet), n10 = -(-m10 * invdet),
                 n11 = -(m00 * invdet);
    double Kc[12], kc[2];
#pragma unroll
    for (int c = 0; c < 6; ++c) {                                                   // cc:365
      Kc[c] = n00 * Qux[c] + n01 * Qux[6 + c];
      Kc[6 + c] = n10 * Qux[c] + n11 * Qux[6 + c];
    }
    kc[0] = n00 * Qu[0] + n01 * Qu[1];                                              // cc:366
    kc[1] = n10 * Qu[0] + n11 * Qu[1];
    if (kStore) {
      double2* g = s.gains + (size_t)i * kGainPairs * Bc + sp;
#pragma unroll
      for (int r = 0; r < 6; ++r) g[(size_t)r * Bc] = make_double2(Kc[2 * r], Kc[2 * r + 1]);
      g[(size_t)6 * Bc] = make_double2(kc[0], kc[1]);
    }
    {  // CalGradientNorm term, cc:328-329
      const double v0 = fabs(kc[0]) / (fabs(uu.x) + 1), v1 = fabs(kc[1]) / (fabs(uu.y) + 1);
      gsum += (v0 > v1 ? v0 : v1);
    }
    double AtVx[6], AtV[36], AtVA[36];
    mt_x<6, 1, true>(A, Vx, AtVx);
    mt_x<6, 6, true>(A, Vxx, AtV);
    x_m<6, 6, true>(AtV, A, AtVA);
    // K^T Quu (6x2), then the three correction terms of each update              cc:379-380
    double KtQ[12];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      KtQ[r * 2 + 0] = Kc[r] * Quu[0] + Kc[6 + r] * Quu[2];
      KtQ[r * 2 + 1] = Kc[r] * Quu[1] + Kc[6 + r] * Quu[3];
    }
    double nVx[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double t1 = KtQ[r * 2] * kc[0] + KtQ[r * 2 + 1] * kc[1];
      const double t2 = Kc[r] * Qu[0] + Kc[6 + r] * Qu[1];
      const double t3 = Qux[r] * kc[0] + Qux[6 + r] * kc[1];
      nVx[r] = (((lx[r] + AtVx[r]) + t1) + t2) + t3;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double m1 = KtQ[r * 2] * Kc[c] + KtQ[r * 2 + 1] * Kc[6 + c];
        const double m2 = Kc[r] * Qux[c] + Kc[6 + r] * Qux[6 + c];
        const double m3 = Qux[r] * Kc[c] + Qux[6 + r] * Kc[6 + c];
        const double qxx = (h_kind(r, c) == 2) ? (H[r * 6 + c] + AtVA[r * 6 + c]) : AtVA[r * 6 + c];
        Vxx[r * 6 + c] = ((qxx + m1) + m2) + m3;
      }
#pragma unroll
    for (int r = 0; r < 6; ++r) Vx[r] = nVx[r];
    // in-place symmetrisation, column-major order                                 cc:381
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int r = 0; r < 6; ++r) Vxx[r * 6 + c] = 0.5 * (Vxx[r * 6 + c] + Vxx[c * 6 + r]);
    // ---- delta_V_ with Qu / Quu re-evaluated on the NEW Vx / Vxx ----            cc:383-384
#ifdef CILQR_DV_EVAL_EAGER
    const double Qu0 = Qu[0], Qu1 = Qu[1];
    dV0 += kc[0] * Qu0 + kc[1] * Qu1;
    const double q00 = Quu[0], q01 = Quu[1], q10 = Quu[2], q11 = Quu[3];
#else
    double BtV2[12], BtVB2[4], BtVx2[2];
    mt_x<2, 1, false>(B, Vx, BtVx2);
    const double Qu0 = lu[0] + BtVx2[0], Qu1 = lu[1] + BtVx2[1];
    dV0 += kc[0] * Qu0 + kc[1] * Qu1;
    mt_x<2, 6, false>(B, Vxx, BtV2);
    x_m<2, 2, false>(BtV2, B, BtVB2);
    const double q00 = luu0 + BtVB2[0], q01 = BtVB2[1], q10 = BtVB2[2], q11 = luu1 + BtVB2[3];
#endif
    const double hk0 = 0.5 * kc[0], hk1 = 0.5 * kc[1];
    const double r0 = hk0 * q00 + hk1 * q10, r1 = hk0 * q01 + hk1 * q11;
    dV1 += r0 * kc[0] + r1 * kc[1];
  }
  s.dV[slot] = dV0;
  s.dV[(size_t)Bc + slot] = dV1;
  s.gnorm[slot] = gsum / N;
}

// ---------------------------------------------------------------------------------------------
// Team variant for small active sets.  With a few hundred problems left, one lane per problem
// leaves the chip idle and every launch takes N dependent steps of ~1100 dependent fp64
// instructions each.  Here eight lanes share a problem: lane c (c < 6) owns column c of Vxx and of
// every 6-column intermediate (B^T Vxx, Qux, K, A^T Vxx, A^T Vxx A, the new Vxx) and entry c of
// Vx; the 2x2 / 2-vector quantities (Quu, Qu, the inverse, k, delta_V, the gradient norm) are
// evaluated redundantly by every lane.  Three exchanges per step go through LDS (the block is one
// wave, so a barrier is a wait on the LDS counter).
//
// Every number is produced by the same operations on the same operands in the same order as in
// backward_problem, so the results are bit-identical (tested): columns are independent in the
// reference's products, and where the sparsity of a product depends on the column (A's column c in
// X A, A^T x) the lanes use the dense form with A's exact zeros and ones -- x * 1.0 = x and
// s + x * 0.0 = s are exact (up to the sign of a zero).
// ---------------------------------------------------------------------------------------------
namespace team {
constexpr int kLanes = 8;          // lanes per problem
constexpr int kStride = 132;       // doubles per team in LDS; 132 * 8 B = 8 banks (mod 64): no conflicts between teams
constexpr int oAtV = 0;            // [6][4]  columns 0..3 of A^T Vxx
constexpr int oKc = 24;            // [2][6]
constexpr int oQux = 36;           // [2][6]
constexpr int oVn = 48;            // [6 columns][6 rows] new Vxx before symmetrisation
constexpr int oB0 = 84;            // (B^T Vxx)(0, .) of the updated Vxx
constexpr int oB1 = 90;            // (B^T Vxx)(1, .) with this step's B(2,1)
constexpr int oBn = 96;            // (B^T Vxx)(1, .) with the next step's B(2,1)
constexpr int oVx = 102;           // Vx
}  // namespace team

// Sync: how the lanes of a team wait for each other's LDS writes.  BlockSync for a block that is one wave of
// teams (k_backward_team); WaveSync when the team is part of a larger block whose other waves do not take part
// (kernels_tail.hip): LDS operations of one wave execute in order, so a wave-scope fence is enough.
struct BlockSync {
  CILQR_DEV void operator()() const { __syncthreads(); }
};
// WaveSync: dev_model.hpp

// cl: lane within the team (0..7); T: the team's kStride doubles of LDS; live: stores enabled
template <class Sync>
CILQR_DEV void backward_team_problem(const DeviceState& s, int slot, double lambda, bool live, int cl,
                                     double* __restrict__ T, const Sync& sync) {
  using namespace team;
  const Params& p = s.p;
  const int Bc = s.Bcap, N = p.N;
  const double dt = p.dt;
  const int c = cl < 6 ? cl : 5;               // column this lane computes (lanes 6, 7 shadow column 5)
  const bool owner = cl < 6;
  // column selectors
  const bool c0 = c == 0, c1 = c == 1, c2 = c == 2, c3 = c == 3, c4 = c == 4;
  auto sel6 = [&](double v0, double v1, double v2, double v3, double v4, double v5) {
    return c0 ? v0 : c1 ? v1 : c2 ? v2 : c3 ? v3 : c4 ? v4 : v5;
  };
  const int sp = scratch_index(s, slot);   // where this slot's lin / term / gains live this iteration
  // terminal value function: column c of Vxx, entry c of Vx
  double V[6], vx;
  {
    const double2* t = s.term + sp;
    const double2 t0 = t[0], t1 = t[(size_t)Bc], t2 = t[(size_t)2 * Bc], t3 = t[(size_t)3 * Bc],
                  t4 = t[(size_t)4 * Bc], t5 = t[(size_t)5 * Bc], t6 = t[(size_t)6 * Bc],
                  t7 = t[(size_t)7 * Bc], t8 = t[(size_t)8 * Bc];
    vx = sel6(t0.x, t0.y, t1.x, t1.y, t2.x, t2.y);
    V[0] = sel6(t3.x, t3.y, t4.x, 0.0, 0.0, 0.0);
    V[1] = sel6(t4.y, t5.x, t5.y, 0.0, 0.0, 0.0);
    V[2] = sel6(t6.x, t6.y, t7.x, 0.0, 0.0, 0.0);
    V[3] = sel6(0.0, 0.0, 0.0, t7.y, 0.0, 0.0);
    V[4] = sel6(0.0, 0.0, 0.0, 0.0, t8.x, 0.0);
    V[5] = sel6(0.0, 0.0, 0.0, 0.0, 0.0, t8.y);
  }
  const int buf = s.cur[slot];
  const double B30 = 0.5 * dt * dt, B40 = dt, B51 = dt;
  double dV0 = 0.0, dV1 = 0.0, gsum = 0.0;
  // What a lane needs from the linearisation of a step is a handful of scalars that depend on its
  // column: rows 0..3 of A's column c, lx(c), column c of lxx -- plus the pairs every lane needs
  // (B(2,1), lu, luu).  Each is fetched straight from its place in the [17 pairs][Bcap] record
  // (offsets in doubles from the step's first pair, fixed per lane), instead of loading all 17
  // pairs and selecting by column.  Structural zeros / ones never touch memory: `k` = kind.
  const size_t row = (size_t)Bc * 2;                       // doubles per pair row
  auto at = [&](int pair, int half) { return (size_t)pair * row + (size_t)half; };
  // A(0,c): pairs 0,1 = (A02,A03),(A04,A05); A(1,c): pairs 2,3; A(2,c): pair 4 = (A23,A24), 5.x = A25
  const size_t o_a0 = c >= 2 ? at((c - 2) >> 1, (c - 2) & 1) : 0;
  const size_t o_a1 = c >= 2 ? at(2 + ((c - 2) >> 1), (c - 2) & 1) : 0;
  const size_t o_a2 = c >= 3 ? at(4 + ((c - 3) >> 1), (c - 3) & 1) : 0;
  const size_t o_lx = at(6 + (c >> 1), c & 1);
  // lxx: pairs 10..15 = (H00,H01),(H02,H10),(H11,H12),(H20,H21),(H22,H33),(H44,H55)
  //   column c < 3: rows 0..2 -> H(r,c) is scalar number 3 r + c of the first nine
  //   column c >= 3: the diagonal entry, scalar number 6 + c
  auto hs = [&](int sidx) { return at(10 + (sidx >> 1), sidx & 1); };
  const size_t o_h0 = hs(c < 3 ? c : 6 + c), o_h1 = hs(c < 3 ? 3 + c : 6 + c), o_h2 = hs(c < 3 ? 6 + c : 6 + c);
  const double ka0 = c0 ? 1.0 : 0.0, ka1 = c1 ? 1.0 : 0.0, ka2 = c2 ? 1.0 : 0.0;   // value when not loaded
  struct StepIn {
    double a0, a1, a2, lx, h0, h1, h2;
    double2 wa[6], lu, luu, u;     // wa: the pairs that hold A (and B(2,1)): A^T x needs all of A
  };
  auto load_step = [&](int i, StepIn& o) {
    const double* q = reinterpret_cast<const double*>(s.lin + (size_t)i * kLinPairs * Bc + sp);
    o.a0 = q[o_a0]; o.a1 = q[o_a1]; o.a2 = q[o_a2]; o.lx = q[o_lx];
    o.h0 = q[o_h0]; o.h1 = q[o_h1]; o.h2 = q[o_h2];
    const double2* q2 = s.lin + (size_t)i * kLinPairs * Bc + sp;
#pragma unroll
    for (int r = 0; r < 6; ++r) o.wa[r] = q2[(size_t)r * Bc];
    o.lu = q2[(size_t)9 * Bc]; o.luu = q2[(size_t)16 * Bc];
    o.u = s.U[((size_t)buf * N + i) * Bc + slot];
  };
  StepIn w, wn;
  load_step(N - 1, w);
  // B^T Vxx and Vx of the terminal value function, for the first step
  if (owner) {
    T[oB0 + c] = B30 * V[3] + B40 * V[4];
    T[oBn + c] = w.wa[5].y * V[2] + B51 * V[5];
    T[oVx + c] = vx;
  }
  sync();
  double BtV0[6], BtV1[6], Vxa[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    BtV0[k] = T[oB0 + k];
    BtV1[k] = T[oBn + k];
    Vxa[k] = T[oVx + k];
  }
  for (int i = N - 1; i >= 0; --i) {
    load_step((i > 0) ? i - 1 : 0, wn);
    // A (the entries the column-independent product A^T x reads) and B
    double A[36], B[12];
    A[2] = w.wa[0].x; A[3] = w.wa[0].y; A[4] = w.wa[1].x; A[5] = w.wa[1].y;
    A[8] = w.wa[2].x; A[9] = w.wa[2].y; A[10] = w.wa[3].x; A[11] = w.wa[3].y;
    A[15] = w.wa[4].x; A[16] = w.wa[4].y; A[17] = w.wa[5].x;
    A[22] = dt;
    B[5] = w.wa[5].y; B[6] = B30; B[8] = B40; B[11] = B51;
    // rows 0..3 of A's column c, exact zeros and ones included; rows 4, 5 of a column hold only
    // its diagonal one
    const double a0 = c >= 2 ? w.a0 : ka0;
    const double a1 = c >= 2 ? w.a1 : ka1;
    const double a2 = c >= 3 ? w.a2 : ka2;
    const double a3 = c3 ? 1.0 : c4 ? dt : 0.0;
    const bool late = c >= 4;   // diagonal entry below row 3: one more term, the own column
    const double lxc = w.lx;
    const double lu[2] = {w.lu.x, w.lu.y};
    // column c of lxx (zeros where the reference's lxx has none)
    const bool blk = c < 3;
    const double hc[6] = {blk ? w.h0 : 0.0, blk ? w.h1 : 0.0, blk ? w.h2 : 0.0,
                          c3 ? w.h0 : 0.0, c4 ? w.h0 : 0.0, (c == 5) ? w.h0 : 0.0};
    const double luu0 = w.luu.x, luu1 = w.luu.y;
    const double2 uu = w.u;

    // ---- quantities from the OLD Vx / Vxx ----
    // column c of Qux = (B^T Vxx) A                                               cc:353
    double Qux0, Qux1;
    {
      double q0 = BtV0[0] * a0; q0 += BtV0[1] * a1; q0 += BtV0[2] * a2; q0 += BtV0[3] * a3;
      double q1 = BtV1[0] * a0; q1 += BtV1[1] * a1; q1 += BtV1[2] * a2; q1 += BtV1[3] * a3;
      const double own0 = c4 ? BtV0[4] : BtV0[5], own1 = c4 ? BtV1[4] : BtV1[5];
      Qux0 = late ? q0 + own0 : q0;
      Qux1 = late ? q1 + own1 : q1;
    }
    const double BtVB[4] = {BtV0[3] * B[6] + BtV0[4] * B[8], BtV0[2] * B[5] + BtV0[5] * B[11],
                            BtV1[3] * B[6] + BtV1[4] * B[8], BtV1[2] * B[5] + BtV1[5] * B[11]};
    const double BtVx[2] = {B[6] * Vxa[3] + B[8] * Vxa[4], B[5] * Vxa[2] + B[11] * Vxa[5]};
    const double Quu[4] = {luu0 + BtVB[0], BtVB[1], BtVB[2], luu1 + BtVB[3]};      // cc:352
    const double Qu[2] = {lu[0] + BtVx[0], lu[1] + BtVx[1]};                       // cc:349
    const double m00 = Quu[0] + lambda, m01 = Quu[1], m10 = Quu[2], m11 = Quu[3] + lambda;
    const double invdet = 1.0 / (m00 * m11 - m10 * m01);                            // cc:361-363
    const double n00 = -(m11 * invdet), n01 = -(-m01 * invdet), n10 = -(-m10 * invdet),
                 n11 = -(m00 * invdet);
    const double Kc0 = n00 * Qux0 + n01 * Qux1;                                     // cc:365
    const double Kc1 = n10 * Qux0 + n11 * Qux1;
    const double kc[2] = {n00 * Qu[0] + n01 * Qu[1], n10 * Qu[0] + n11 * Qu[1]};    // cc:366
    {  // CalGradientNorm term, cc:328-329
      const double v0 = fabs(kc[0]) / (fabs(uu.x) + 1), v1 = fabs(kc[1]) / (fabs(uu.y) + 1);
      gsum += (v0 > v1 ? v0 : v1);
    }
    // entry c of A^T Vx, column c of A^T Vxx
    double AtVxc;
    {
      // A^T Vx is an X^T Y product (sum6_xty): dense over this column's rows, with the exact zeros / ones of rows 4, 5
      AtVxc = sum6_xty(a0 * Vxa[0], a1 * Vxa[1], a2 * Vxa[2], a3 * Vxa[3], c4 ? Vxa[4] : 0.0, (c == 5) ? Vxa[5] : 0.0);
    }
    double AtVc[6];
    mt_x<6, 1, true>(A, V, AtVc);
    if (owner) {
      if (c < 4) {
#pragma unroll
        for (int r = 0; r < 6; ++r) T[oAtV + r * 4 + c] = AtVc[r];
      }
      T[oKc + c] = Kc0;
      T[oKc + 6 + c] = Kc1;
      T[oQux + c] = Qux0;
      T[oQux + 6 + c] = Qux1;
    }
    sync();
    double Ka[12], Qa[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      Ka[e] = T[oKc + e];
      Qa[e] = T[oQux + e];
    }
    if (live) {   // gains: lane r < 6 stores pair r of (K row 0 | K row 1), lane 6 stores k
      double2* g = s.gains + (size_t)i * kGainPairs * Bc + sp;
      if (cl < 6) g[(size_t)cl * Bc] = make_double2(Ka[2 * cl], Ka[2 * cl + 1]);
      else if (cl == 6) g[(size_t)6 * Bc] = make_double2(kc[0], kc[1]);
    }
    // column c of (A^T Vxx) A
    double AtVAc[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double q = T[oAtV + r * 4 + 0] * a0;
      q += T[oAtV + r * 4 + 1] * a1;
      q += T[oAtV + r * 4 + 2] * a2;
      q += T[oAtV + r * 4 + 3] * a3;
      AtVAc[r] = late ? q + AtVc[r] : q;
    }
    // K^T Quu, all rows                                                          cc:379-380
    double KtQ[12];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      KtQ[r * 2 + 0] = Ka[r] * Quu[0] + Ka[6 + r] * Quu[2];
      KtQ[r * 2 + 1] = Ka[r] * Quu[1] + Ka[6 + r] * Quu[3];
    }
    double nvx;
    {
      const double kq0 = sel6(KtQ[0], KtQ[2], KtQ[4], KtQ[6], KtQ[8], KtQ[10]);
      const double kq1 = sel6(KtQ[1], KtQ[3], KtQ[5], KtQ[7], KtQ[9], KtQ[11]);
      const double t1 = kq0 * kc[0] + kq1 * kc[1];
      const double t2 = Kc0 * Qu[0] + Kc1 * Qu[1];
      const double t3 = Qux0 * kc[0] + Qux1 * kc[1];
      nvx = (((lxc + AtVxc) + t1) + t2) + t3;
    }
    double Vn[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double m1 = KtQ[r * 2] * Kc0 + KtQ[r * 2 + 1] * Kc1;
      const double m2 = Ka[r] * Qux0 + Ka[6 + r] * Qux1;
      const double m3 = Qa[r] * Kc0 + Qa[6 + r] * Kc1;
      const double qxx = hc[r] + AtVAc[r];
      Vn[r] = ((qxx + m1) + m2) + m3;
    }
    if (owner) {
#pragma unroll
      for (int r = 0; r < 6; ++r) T[oVn + c * 6 + r] = Vn[r];
    }
    sync();
    // in-place symmetrisation, column-major order (cc:381): entries below the diagonal average
    // the two old values; entries above it average the old value with the already averaged one
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double o_cr = T[oVn + r * 6 + c];   // old (c, r): row c of column r
      const double lower = 0.5 * (Vn[r] + o_cr);
      const double upper = 0.5 * (Vn[r] + 0.5 * (o_cr + Vn[r]));
      V[r] = (r < c) ? upper : lower;
    }
    vx = nvx;
    // B^T Vxx of the updated Vxx: with this step's B for delta_V, with the next step's for the
    // next step's gains
    if (owner) {
      T[oB0 + c] = B30 * V[3] + B40 * V[4];
      T[oB1 + c] = B[5] * V[2] + B51 * V[5];
      T[oBn + c] = wn.wa[5].y * V[2] + B51 * V[5];
      T[oVx + c] = vx;
    }
    sync();
    double Bt2_1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      BtV0[k] = T[oB0 + k];
      Bt2_1[k] = T[oB1 + k];
      BtV1[k] = T[oBn + k];
      Vxa[k] = T[oVx + k];
    }
    // ---- delta_V_ with Qu / Quu re-evaluated on the NEW Vx / Vxx ----            cc:383-384
    {
#ifdef CILQR_DV_EVAL_EAGER
      const double Qu0 = Qu[0], Qu1 = Qu[1];
      dV0 += kc[0] * Qu0 + kc[1] * Qu1;
      const double q00 = Quu[0], q01 = Quu[1], q10 = Quu[2], q11 = Quu[3];
#else
      const double BtVx2[2] = {B[6] * Vxa[3] + B[8] * Vxa[4], B[5] * Vxa[2] + B[11] * Vxa[5]};
      const double Qu0 = lu[0] + BtVx2[0], Qu1 = lu[1] + BtVx2[1];
      dV0 += kc[0] * Qu0 + kc[1] * Qu1;
      const double BtVB2[4] = {BtV0[3] * B[6] + BtV0[4] * B[8], BtV0[2] * B[5] + BtV0[5] * B[11],
                               Bt2_1[3] * B[6] + Bt2_1[4] * B[8], Bt2_1[2] * B[5] + Bt2_1[5] * B[11]};
      const double q00 = luu0 + BtVB2[0], q01 = BtVB2[1], q10 = BtVB2[2], q11 = luu1 + BtVB2[3];
#endif
      const double hk0 = 0.5 * kc[0], hk1 = 0.5 * kc[1];
      const double r0 = hk0 * q00 + hk1 * q10, r1 = hk0 * q01 + hk1 * q11;
      dV1 += r0 * kc[0] + r1 * kc[1];
    }
    w = wn;
  }
  if (live && cl == 0) {
    s.dV[slot] = dV0;
    s.dV[(size_t)Bc + slot] = dV1;
    s.gnorm[slot] = gsum / N;
  }
}


// ---------------------------------------------------------------------------------------------
// Wave variant: one wavefront per problem, operands staged in LDS.
//
// The chain of dependent work of a backward step is short (the 2x2 inverse and a few 6-term dot products); what
// makes a step slow with one or eight lanes per problem is the NUMBER of instructions every lane issues.  Here the
// 64 lanes of a wave each own one output element of every stage, all stages are written as one generic 6-term dot
// product whose operand addresses are per-lane constants, and every lane issues ~200 instructions per step instead
// of ~500 (team) or ~1100 (lane).  Lane roles:
//   0..35   (r, c)  element of A^T Vxx, (A^T Vxx) A, the new Vxx
//   36..47  (q, c)  element of B^T Vxx, Qux, K
//   48..53  r       entry of A^T Vx, of the new Vx
//   54..57          B^T Vx with this step's B (gains) / with the previous step's B (its delta_V, see below)
//   58..63  c       (B^T Vxx)(1, c) with the previous step's B
// The 2x2 / 2-vector quantities (Quu, Qu, the inverse, k) are evaluated by every lane.  delta_V of a step needs
// Qu / Quu on the UPDATED Vx / Vxx (cc:383-384), which are the next step's inputs: the spare lanes evaluate their
// B^T Vx / B^T Vxx B parts during the next step's first two stages and leave them in a row of the step they belong to;
// the per-step terms of delta_V and of the gradient norm are evaluated after the recursion, a step per lane, and summed by
// one lane in the recursion's order (end of backward_wave_problem).
// Dense products with A's and B's exact zeros and ones stand for the sparse ones of backward_problem (x * 1 = x,
// s + x * 0 = s); stage 1 (the X^T Y products) adds its terms as sum6_xty says, stage 2 in index order, like the other two
// kernels: bit-identical results (tested against both).
// ---------------------------------------------------------------------------------------------
namespace wave {
constexpr int oA = 0, oB = 36, oBo = 48, oV = 60, oVx = 96, oAtV = 102, oBtV = 138, oBtV2 = 150, oAtVx = 162,
              oBtVx = 168, oBtVx2 = 170, oQux = 172, oBtVB = 188, oBtVB2 = 192, oK = 196, oVn = 212, oLx = 248,
              oLu = 254, oLuu = 256, oU = 258, oH = 260, oDummy = 296, kDoubles = 304;
// Qux and K are [2][8]: columns 0..5, column 6 = (Qu | k) so that the new Vx is the "seventh column" of the Vxx update
// Behind the stage operands: one row per step (and one spare) of what delta_V and the gradient norm need from it --
// (B^T Vx)(2) and (B^T Vxx B)(4) on the UPDATED value function, k (2).  Their sums are taken after the recursion (below).
constexpr int kDefer = 8, oD = kDoubles;
__host__ __device__ constexpr size_t lds_doubles(int N) { return (size_t)kDoubles + (size_t)(N + 1) * kDefer; }
}  // namespace wave

// Solo: the problem is alone in its arena (capacity 1, slot 0, position 0 -- the tail kernel's private view): every
// stride is a compile-time constant.  L: wave::lds_doubles(N) doubles of LDS.
// InLds (with Solo): lin, term, gains, U, dV and gnorm of the view are in LDS (dev_model.hpp: assume_lds).
#ifdef CILQR_BWD_STAGE_PROFILE   // tuning build: cycles per stage of the wave form, printed by lane 0 of block 0
#define BSP_DECL long long bsp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long bsp_t = __builtin_readcyclecounter();
#define BSP(k) do { const long long n_ = __builtin_readcyclecounter(); bsp_[k] += n_ - bsp_t; bsp_t = n_; } while (0)
#else
#define BSP_DECL
#define BSP(k)
#endif
template <class Sync, bool Solo = false, bool InLds = false>
CILQR_DEV void backward_wave_problem(const DeviceState& s, int slot, double lambda, int lane,
                                     double* __restrict__ L, const Sync& sync) {
  using namespace wave;
  const Params& p = s.p;
  const size_t Bc = Solo ? (size_t)1 : (size_t)s.Bcap;
  const int N = p.N;
  const double dt = p.dt;
  const int buf = s.cur[slot];
  const int sp = Solo ? 0 : scratch_index(s, slot);   // where this slot's lin / term / gains live this iteration
  // the pointers as locals: when the state is a large by-value copy (kernels_tail.hip) its fields live in scratch
  const double2* __restrict__ lin_p = s.lin + sp;
  const double2* __restrict__ u_p = s.U + (size_t)buf * N * Bc + slot;
  double2* __restrict__ gains_p = s.gains + sp;
  const double2* __restrict__ term_p = s.term + sp;
  double* __restrict__ dV_p = s.dV;
  double* __restrict__ gnorm_p = s.gnorm;
  assume_lds<InLds>(lin_p);
  assume_lds<InLds>(u_p);
  assume_lds<InLds>(gains_p);
  assume_lds<InLds>(term_p);
  assume_lds<InLds>(dV_p);
  assume_lds<InLds>(gnorm_p);
  // ---- per-lane roles ----
  const bool mat = lane < 36, row2 = lane >= 36 && lane < 48, vec = lane >= 48 && lane < 54;
  const int r = mat ? lane / 6 : (vec ? lane - 48 : 0);
  const int c = mat ? lane % 6 : (row2 ? (lane - 36) % 6 : (lane >= 58 ? lane - 58 : 0));
  const int q = row2 ? (lane - 36) / 6 : ((lane >= 54 && lane < 58) ? (lane & 1) : 0);
  // stage 1: sum_k M[pm1 + k sm1] * X[px1 + k sx1] -> L[po1]; the lanes that evaluate a DEFERRED quantity (B^T Vx on the
  // updated value function: lanes 56, 57) write into the row of the step it belongs to, d1 doubles further every step
  int pm1, sm1, px1, sx1, po1, d1 = 0;
  if (mat)            { pm1 = oA + r;  sm1 = 6; px1 = oV + c; sx1 = 6; po1 = oAtV + r * 6 + c; }
  else if (row2)      { pm1 = oB + q;  sm1 = 2; px1 = oV + c; sx1 = 6; po1 = oBtV + q * 6 + c; }
  else if (vec)       { pm1 = oA + r;  sm1 = 6; px1 = oVx;    sx1 = 1; po1 = oAtVx + r; }
  else if (lane < 56) { pm1 = oB + q;  sm1 = 2; px1 = oVx;    sx1 = 1; po1 = oBtVx + q; }
#ifdef CILQR_DV_EVAL_EAGER   // the rows keep the step's OWN Qu / Quu (lane 0, stage 3); these lanes' results go unused
  else if (lane < 58) { pm1 = oBo + q; sm1 = 2; px1 = oVx;    sx1 = 1; po1 = oBtVx2 + q; }
#else
  else if (lane < 58) { pm1 = oBo + q; sm1 = 2; px1 = oVx;    sx1 = 1; po1 = oD + q; d1 = kDefer; }
#endif
  else                { pm1 = oBo + 1; sm1 = 2; px1 = oV + c; sx1 = 6; po1 = oBtV2 + 6 + c; }
  // stage 2: sum_k X[px2 + k] * M[pm2 + k sm2] -> L[po2] (matrix lanes keep the result); lanes 52..55: deferred (B^T Vxx B)
  int px2, pm2, sm2, po2, d2 = 0;
  if (mat)            { px2 = oAtV + r * 6; pm2 = oA + c; sm2 = 6; po2 = oDummy; }
  else if (row2)      { px2 = oBtV + q * 6; pm2 = oA + c; sm2 = 6; po2 = oQux + q * 8 + c; }
  else if (lane < 52) { const int a = (lane - 48) >> 1, b = (lane - 48) & 1;
                        px2 = oBtV + a * 6; pm2 = oB + b; sm2 = 2; po2 = oBtVB + a * 2 + b; }
#ifdef CILQR_DV_EVAL_EAGER
  else if (lane < 56) { const int a = (lane - 52) >> 1, b = (lane - 52) & 1;
                        px2 = (a == 0) ? oBtV : oBtV2 + 6; pm2 = oBo + b; sm2 = 2; po2 = oBtVB2 + a * 2 + b; }
#else
  else if (lane < 56) { const int a = (lane - 52) >> 1, b = (lane - 52) & 1;
                        px2 = (a == 0) ? oBtV : oBtV2 + 6; pm2 = oBo + b; sm2 = 2; po2 = oD + 2 + a * 2 + b; d2 = kDefer; }
#endif
  else                { px2 = oBtV; pm2 = oB; sm2 = 2; po2 = oDummy + 1 + (lane & 3); }
  // the quantities evaluated while step i runs belong to step i + 1 (its B is in Bo): row i + 1, starting at row N (spare)
  po1 += d1 * N;
  po2 += d2 * N;
  // stage 3: K(q, c) (lanes 36..47), the others write to the dummy cell
  const int pk3 = row2 ? oK + q * 8 + c : oDummy + 5;
  const int pq3 = row2 ? oQux + c : oQux;          // Qux(0, c), Qux(1, c) = +8
  // stage 4: column quantities (K(., c), Qux(., c)), c = 6 for the Vx lanes
  const int c4 = mat ? c : 6;
  const int pbase = mat ? oH + r * 6 + c : oLx + r;
  const int po4 = mat ? oVn + r * 6 + c : (vec ? oVx + r : oDummy + 6);
  // stage 5: symmetrisation partner
  const int psym = mat ? oVn + c * 6 + r : oVn;
  // ---- constants of A, B, H; terminal value function ----
  if (lane < 36) {
    L[oA + lane] = (lane / 6 == lane % 6) ? 1.0 : (lane == 3 * 6 + 4 ? dt : 0.0);
    L[oH + lane] = 0.0;
  }
  if (lane < 12) {
    const double b = (lane == 3 * 2 + 0) ? 0.5 * dt * dt : ((lane == 4 * 2 + 0 || lane == 5 * 2 + 1) ? dt : 0.0);
    L[oB + lane] = b;
    L[oBo + lane] = b;
  }
  if (lane < 36) L[oV + lane] = 0.0;
  if (lane < 16) { L[oQux + lane] = 0.0; L[oK + lane] = 0.0; }
  sync();
  if (lane < 9) {
    const double2 t = term_p[(size_t)lane * Bc];
    // pairs 0..2: Vx; 3..8: (h00,h01) (h02,h10) (h11,h12) (h20,h21) (h22,h33) (h44,h55) of Vxx
    constexpr int px[9] = {oVx + 0, oVx + 2, oVx + 4, oV + 0, oV + 2, oV + 7, oV + 12, oV + 14, oV + 28};
    constexpr int py[9] = {oVx + 1, oVx + 3, oVx + 5, oV + 1, oV + 6, oV + 8, oV + 13, oV + 21, oV + 35};
    L[px[lane]] = t.x;
    L[py[lane]] = t.y;
  }
  // where the pair a lane fetches (17 pairs of lin, then u) goes: two cells of A / B / lx / lu / H / luu / u
  int in0 = oDummy + 2, in1 = oDummy + 3;
  {
    constexpr int t0[18] = {oA + 2, oA + 4, oA + 8, oA + 10, oA + 15, oA + 17, oLx + 0, oLx + 2, oLx + 4, oLu,
                            oH + 0, oH + 2, oH + 7, oH + 12, oH + 14, oH + 28, oLuu, oU};
    constexpr int t1[18] = {oA + 3, oA + 5, oA + 9, oA + 11, oA + 16, oB + 5, oLx + 1, oLx + 3, oLx + 5, oLu + 1,
                            oH + 1, oH + 6, oH + 8, oH + 13, oH + 21, oH + 35, oLuu + 1, oU + 1};
    if (lane < 18) { in0 = t0[lane]; in1 = t1[lane]; }
  }
  // one pair per lane and step: pair `lane` of the step's 17, lane 17 the control
  const double2* __restrict__ fp = (lane < kLinPairs) ? lin_p + (size_t)lane * Bc : u_p;
  const size_t fstride = (lane < kLinPairs) ? (size_t)kLinPairs * Bc : Bc;
  auto fetch = [&](int i) -> double2 {
    if (lane <= kLinPairs) return fp[(size_t)i * fstride];
    return make_double2(0.0, 0.0);
  };
  // operands of the next kWaveAhead steps in flight (one pair per lane and step): a step is ~0.9 us of arithmetic,
  // a load that misses L2 takes longer
  constexpr int kWaveAhead = 3;
  double2 pre[kWaveAhead];
#pragma unroll
  for (int d = 0; d < kWaveAhead; ++d) pre[d] = fetch(N - 1 - d >= 0 ? N - 1 - d : 0);
  BSP_DECL
  double b21_prev = dt;   // lane 5: B(2,1) of the step before (moves to Bo when the next one arrives); first: unused
  for (int i = N - 1; i >= -1; --i) {
    // inputs of step i; step -1 only finishes the deferred quantities of step 0.  The previous B(2,1) moves to Bo first --
    // by the lane that overwrites B(2,1) (lane 5, from its own register: no round trip through LDS), so the move needs no
    // exchange of its own (every reader of the old Bo is at least one exchange behind)
    if (lane == 5 && i < N - 1) L[oBo + 5] = b21_prev;
    if (i >= 0 && lane <= kLinPairs) { L[in0] = pre[0].x; L[in1] = pre[0].y; }
    b21_prev = pre[0].y;
#pragma unroll
    for (int d = 0; d + 1 < kWaveAhead; ++d) pre[d] = pre[d + 1];
    if (i - kWaveAhead >= 0) pre[kWaveAhead - 1] = fetch(i - kWaveAhead);
    sync();
    BSP(0);
    // ---- stage 1 ----
    double res1;
    {
      // every product of this stage is an X^T Y product of the reference: its six terms in sum6_xty's order
      const double a = sum6_xty(L[pm1] * L[px1], L[pm1 + sm1] * L[px1 + sx1], L[pm1 + 2 * sm1] * L[px1 + 2 * sx1],
                                L[pm1 + 3 * sm1] * L[px1 + 3 * sx1], L[pm1 + 4 * sm1] * L[px1 + 4 * sx1],
                                L[pm1 + 5 * sm1] * L[px1 + 5 * sx1]);
      res1 = a;
      L[po1] = a;
    }
    sync();
    BSP(1);
    // ---- stage 2 ----
    double res2;
    {
      double a = L[px2] * L[pm2];
#pragma unroll
      for (int k = 1; k < 6; ++k) a += L[px2 + k] * L[pm2 + k * sm2];
      res2 = a;
      if (!mat && lane < 56) L[po2] = a;
    }
    po1 -= d1;
    po2 -= d2;
    sync();
    BSP(2);
    if (i < 0) break;
    // ---- stage 3: Quu, Qu, inverse, k (every lane), K (lanes 36..47) ----
    const double lu0 = L[oLu], lu1 = L[oLu + 1], luu0 = L[oLuu], luu1 = L[oLuu + 1];
    const double Quu[4] = {luu0 + L[oBtVB], L[oBtVB + 1], L[oBtVB + 2], luu1 + L[oBtVB + 3]};     // cc:352
    const double Qu[2] = {lu0 + L[oBtVx], lu1 + L[oBtVx + 1]};                                     // cc:349
    const double m00 = Quu[0] + lambda, m01 = Quu[1], m10 = Quu[2], m11 = Quu[3] + lambda;
    const double invdet = 1.0 / (m00 * m11 - m10 * m01);                                           // cc:361-363
    const double n00 = -(m11 * invdet), n01 = -(-m01 * invdet), n10 = -(-m10 * invdet), n11 = -(m00 * invdet);
    const double kc0 = n00 * Qu[0] + n01 * Qu[1], kc1 = n10 * Qu[0] + n11 * Qu[1];                 // cc:366
    {
      const double x0 = L[pq3], x1 = L[pq3 + 8];
      const double n0 = (q == 0) ? n00 : n10, n1 = (q == 0) ? n01 : n11;
      const double kq = n0 * x0 + n1 * x1;                                                         // cc:365
      if (row2) L[pk3] = kq;
    }
    if (lane == 0) {
      L[oK + 6] = kc0; L[oK + 14] = kc1; L[oQux + 6] = Qu[0]; L[oQux + 14] = Qu[1];
      double* D = L + oD + i * kDefer;
      D[6] = kc0; D[7] = kc1;
#ifdef CILQR_DV_EVAL_EAGER
      // delta_V_ from the Qu / Quu the gains were computed from (see the switch at the top of this file)
      D[0] = Qu[0]; D[1] = Qu[1]; D[2] = Quu[0]; D[3] = Quu[1]; D[4] = Quu[2]; D[5] = Quu[3];
#endif
    }
    sync();
    BSP(3);
    // gains: lanes 0..5 store pair r of (K row 0 | K row 1), lane 6 stores k
    if (lane < 7) {
      double2 g2;
      if (lane < 6) {
        const int e0 = 2 * lane, e1 = 2 * lane + 1;
        g2 = make_double2(L[oK + (e0 / 6) * 8 + e0 % 6], L[oK + (e1 / 6) * 8 + e1 % 6]);
      } else {
        g2 = make_double2(kc0, kc1);
      }
      gains_p[((size_t)i * kGainPairs + lane) * Bc] = g2;
    }
    // ---- stage 4: new Vxx (unsymmetrised) and new Vx ----
    double own;
    {
      const double K0r = L[oK + r], K1r = L[oK + 8 + r], Q0r = L[oQux + r], Q1r = L[oQux + 8 + r];
      const double a0 = L[oK + c4], a1 = L[oK + 8 + c4], b0 = L[oQux + c4], b1 = L[oQux + 8 + c4];
      const double KtQ0 = K0r * Quu[0] + K1r * Quu[2], KtQ1 = K0r * Quu[1] + K1r * Quu[3];        // cc:379-380
      const double m1 = KtQ0 * a0 + KtQ1 * a1;
      const double m2 = K0r * b0 + K1r * b1;
      const double m3 = Q0r * a0 + Q1r * a1;
      const double base = L[pbase] + (mat ? res2 : res1);
      own = ((base + m1) + m2) + m3;
      if (mat || vec) L[po4] = own;
    }
    sync();
    BSP(4);
    // ---- stage 5: in-place symmetrisation, column-major order (cc:381) ----
    if (mat) {
      const double o_cr = L[psym];
      const double lower = 0.5 * (own + o_cr);
      const double upper = 0.5 * (own + 0.5 * (o_cr + own));
      L[oV + r * 6 + c] = (r < c) ? upper : lower;
    }
    BSP(5);
  }
  // ---- delta_V_ (cc:383-384) and the gradient norm (cc:328-329), after the recursion ----
  // Their per-step terms do not feed the recursion; evaluated inside it they were ~50 instructions of every lane's step
  // (two IEEE divisions among them).  Here the steps' terms are evaluated side by side, a step per lane, from the rows kept
  // above -- Qu / Quu of a step on the UPDATED Vx / Vxx (lazy reading; the eager build keeps the step's own) -- and then
  // summed by one lane in the recursion's order, i = N-1 .. 0: the same operations on the same operands in the same order.
  for (int j = lane; j < N; j += 64) {
    double* D = L + oD + j * kDefer;
    const double k0 = D[6], k1 = D[7];
    const double2 luj = lin_p[((size_t)j * kLinPairs + kRowLu) * Bc], luuj = lin_p[((size_t)j * kLinPairs + kRowLuu) * Bc];
    const double2 uj = u_p[(size_t)j * Bc];
#ifdef CILQR_DV_EVAL_EAGER
    const double Qu0 = D[0], Qu1 = D[1];
    const double q00 = D[2], q01 = D[3], q10 = D[4], q11 = D[5];
    (void)luj; (void)luuj;
#else
    const double Qu0 = luj.x + D[0], Qu1 = luj.y + D[1];
    const double q00 = luuj.x + D[2], q01 = D[3], q10 = D[4], q11 = luuj.y + D[5];
#endif
    const double t0 = k0 * Qu0 + k1 * Qu1;
    const double hk0 = 0.5 * k0, hk1 = 0.5 * k1;
    const double r0 = hk0 * q00 + hk1 * q10, r1 = hk0 * q01 + hk1 * q11;
    const double t1 = r0 * k0 + r1 * k1;
    const double v0 = fabs(k0) / (fabs(uj.x) + 1), v1 = fabs(k1) / (fabs(uj.y) + 1);
    D[0] = t0;
    D[1] = t1;
    D[2] = (v0 > v1 ? v0 : v1);
  }
  sync();
  if (lane == 0) {
    double dV0 = 0.0, dV1 = 0.0, gsum = 0.0;
#pragma unroll 5
    for (int i = N - 1; i >= 0; --i) {
      const double* D = L + oD + i * kDefer;
      dV0 += D[0];
      dV1 += D[1];
      gsum += D[2];
    }
    dV_p[slot] = dV0;
    dV_p[Bc + slot] = dV1;
    gnorm_p[slot] = gsum / N;
  }
#ifdef CILQR_BWD_STAGE_PROFILE
  BSP(6);
  if (lane == 0 && blockIdx.x == 0)
    printf("bwd wave cycles/step: inputs %lld st1 %lld st2 %lld st3 %lld st4 %lld st5 %lld | after the loop %lld\n", bsp_[0] / N, bsp_[1] / N,
           bsp_[2] / N, bsp_[3] / N, bsp_[4] / N, bsp_[5] / N, bsp_[6]);
#endif
}

}  // namespace cilqr
