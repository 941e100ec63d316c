// Backward pass: Riccati-like gain recursion, one lane per problem (the roofline kernel).
//
// Reference behaviour: IlqrOptimizer::Backward (algorithm/ilqr/ilqr_optimizer.cc:334-390) and
// CalGradientNorm (cc:322-332).  Quirks kept:
//   * K, k and the Vx/Vxx updates use the un-regularised Quu except inside the inverse (cc:361-380);
//   * delta_V_ is accumulated from Qu / Quu re-evaluated with the UPDATED Vx / Vxx, because the
//     reference holds them as lazy Eigen `auto` expressions (cc:348-352 vs cc:383-384);
//   * Vxx is symmetrised in place, column-major, without a temporary (cc:381);
//   * products associate left to right and each dot product accumulates k = 0..5 in order.
//
// Memory: per step a lane reads 17 double2 (the non-constant entries of A, B, lx, lu, lxx, luu;
// see state.hpp) + 1 double2 of U, and writes 7 double2 of gains: 18 KiB in + 7 KiB out per
// wave-step, every access 16 B/lane and 1 KiB contiguous per wave.  Structural zeros/ones of A and
// B are skipped at compile time; adding an exact zero never changes an IEEE sum, so the results
// are those of the dense recursion.
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

struct Acc {
  double v;
  bool any;
};
#define ACC_TERM(acc, t)            \
  do {                              \
    if ((acc).any) (acc).v += (t);  \
    else { (acc).v = (t); (acc).any = true; } \
  } while (0)

// out[R][6] = M^T X with M = A (6x6) or B (6x2): out(r,c) = sum_k M(k,r) X(k,c)
template <int R, int xcols, bool IsA>
CILQR_DEV void mt_x(const double* __restrict__ M, const double* __restrict__ X, double* out) {
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < xcols; ++c) {
      Acc a{0.0, false};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int kind = IsA ? a_kind(k, r) : b_kind(k, r);
        if (kind == 0) continue;
        const double x = X[k * xcols + c];
        const double t = (kind == 1) ? x : M[k * R + r] * x;
        ACC_TERM(a, t);
      }
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

template <bool kStore>
CILQR_DEV void backward_problem(const DeviceState& s, int slot, double lambda) {
  const Params& p = s.p;
  const int Bc = s.Bcap, N = p.N;
  const double dt = p.dt;
  double Vx[6], Vxx[36];
  {
    const double2* t = s.term + slot;
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
  double2 w[kLinPairs], wn[kLinPairs];
  double2 uu, uun;
  {
    const double2* q = s.lin + (size_t)(N - 1) * kLinPairs * Bc + slot;
#pragma unroll
    for (int r = 0; r < kLinPairs; ++r) w[r] = q[(size_t)r * Bc];
    uu = s.U[((size_t)buf * N + (N - 1)) * Bc + slot];
  }
  for (int i = N - 1; i >= 0; --i) {
    {
      const int ip = (i > 0) ? i - 1 : 0;
      const double2* q = s.lin + (size_t)ip * kLinPairs * Bc + slot;
#pragma unroll
      for (int r = 0; r < kLinPairs; ++r) wn[r] = q[(size_t)r * Bc];
      uun = s.U[((size_t)buf * N + ip) * Bc + slot];
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
      double2* g = s.gains + (size_t)i * kGainPairs * Bc + slot;
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
    double BtV2[12], BtVB2[4], BtVx2[2];
    mt_x<2, 1, false>(B, Vx, BtVx2);
    const double Qu0 = lu[0] + BtVx2[0], Qu1 = lu[1] + BtVx2[1];
    dV0 += kc[0] * Qu0 + kc[1] * Qu1;
    mt_x<2, 6, false>(B, Vxx, BtV2);
    x_m<2, 2, false>(BtV2, B, BtVB2);
    const double q00 = luu0 + BtVB2[0], q01 = BtVB2[1], q10 = BtVB2[2], q11 = luu1 + BtVB2[3];
    const double hk0 = 0.5 * kc[0], hk1 = 0.5 * kc[1];
    const double r0 = hk0 * q00 + hk1 * q10, r1 = hk0 * q01 + hk1 * q11;
    dV1 += r0 * kc[0] + r1 * kc[1];
#pragma unroll
    for (int r = 0; r < kLinPairs; ++r) w[r] = wn[r];
    uu = uun;
  }
  s.dV[slot] = dV0;
  s.dV[(size_t)Bc + slot] = dV1;
  s.gnorm[slot] = gsum / N;
}

__global__ __launch_bounds__(64, 1) void k_backward(DeviceState s, const int* __restrict__ list, int n,
                                                     const double* __restrict__ lambda_override) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= active_count(s, n)) return;
  const int slot = list ? list[j] : j;
  const double lambda = lambda_override ? lambda_override[slot] : s.lambda[slot];
  backward_problem<true>(s, slot, lambda);
}

void launch_backward(const DeviceState& s, const int* list, int n, const double* lambda_override,
                     hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_backward, dim3((n + 63) / 64), dim3(64), 0, st, s, list, n, lambda_override);
}

}  // namespace cilqr
