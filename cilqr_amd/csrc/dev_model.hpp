// Device-side model functions: angle wrap, relaxed log barrier, kinematic-bicycle dynamics and
// its Jacobian, nearest lane segment.  All IEEE fp64; built with -ffp-contract=off so every
// product and sum is rounded once, as on the reference's CPU build.
//
// Behaviour follows (not copied from) the reference:
//   NormalizeAngle            algorithm/math/math_utils.cpp:53-59
//   RelaxBarrierFunction      algorithm/ilqr/barrier_function.h:82-147
//   VehicleModel::Dynamics    algorithm/ilqr/vehicle_model.cc:88-138
//   VehicleModel::DynamicsJacbian  vehicle_model.cc:21-86
//   LineSegment2d::DistanceTo algorithm/math/line_segment2d.cpp:61-75
//   FindNeastLaneSegment      algorithm/ilqr/ilqr_optimizer.cc:605-618
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#include "state.hpp"

namespace cilqr {

#define CILQR_DEV __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kMathEps = 1e-10;  // algorithm/math/vec2d.h:33

// ---- how the six products of a coefficient of `X.transpose() * Y` are added ----
// Every such product of the path -- A^T Vx, B^T Vx, A^T Vxx, B^T Vxx (Backward, cc:348-353) and B^T P, A^T P (iqr,
// cc:822-823) -- has a row-major left operand, so Eigen 3.4 evaluates each coefficient as
// (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum() (ProductEvaluators.h, product_evaluator<..LazyProduct..>::coeff),
// and the reference's build (x86-64, -O2, no -march: CMakeLists.txt:9, i.e. SSE2, 2-double packets, no FMA) runs that sum
// through the vectorised, completely unrolled redux (Redux.h: redux_vec_unroller<.., 0, 3> = p0 + (p1 + p2) on the packets
// p_j = (t_2j, t_2j+1), then predux = low + high):
//     (t0 + (t2 + t4)) + (t1 + (t3 + t5)).
// Products whose left operand is a column-major matrix or temporary ((B^T Vxx) A, (A^T Vxx) A, K dx, ...) run
// etor_product_packet_impl: pmul, then pmadd = padd(pmul) without FMA, k = 1..5 -- in index order; those stay sequential.
// The CPU checker the tests use carries the same rule as its default (its CILQR_DOT_ORDER switch).
// -DCILQR_DOT_ORDER_SEQUENTIAL: the other reading, index order everywhere (what rounds 1-4 shipped).  TEST-ONLY build
// (`make dotseq` -> lib/libcilqr_hip_dotseq.so), held against the checker's sequential variant.
CILQR_DEV double sum6_xty(double t0, double t1, double t2, double t3, double t4, double t5) {
#ifdef CILQR_DOT_ORDER_SEQUENTIAL
  return ((((t0 + t1) + t2) + t3) + t4) + t5;
#else
  return (t0 + (t2 + t4)) + (t1 + (t3 + t5));
#endif
}
// The same sum over terms some of which are structural zeros (any == false): a missing term drops out of the tree
// (s + 0 = s exactly), so sparse and dense evaluations of a product agree bit for bit (up to the sign of a zero).
struct Acc {
  double v;
  bool any;
};
CILQR_DEV Acc acc_add(const Acc a, const Acc b) {
  if (a.any && b.any) return Acc{a.v + b.v, true};
  return a.any ? a : b;
}
CILQR_DEV Acc acc_sum6_xty(const Acc (&t)[6]) {
#ifdef CILQR_DOT_ORDER_SEQUENTIAL
  return acc_add(acc_add(acc_add(acc_add(acc_add(t[0], t[1]), t[2]), t[3]), t[4]), t[5]);
#else
  return acc_add(acc_add(t[0], acc_add(t[2], t[4])), acc_add(t[1], acc_add(t[3], t[5])));
#endif
}

// fmod(a + pi, 2 pi) with the IEEE-exact fast paths (fmod is always exact, so the short
// branches return bit-identical values to the library call).
CILQR_DEV double normalize_angle(double angle) {
  const double t = angle + kPi;
  double r;
  if (t >= 0.0 && t < kTwoPi) {
    r = t;
  } else if (t >= kTwoPi && t < 2.0 * kTwoPi) {
    r = t - kTwoPi;  // exact (Sterbenz)
  } else if (t < 0.0 && t > -kTwoPi) {
    r = t;
  } else {
    r = fmod(t, kTwoPi);
  }
  if (r < 0.0) r += kTwoPi;
  return r - kPi;
}

// ---- lean fp64 math for the barrier terms ----
// The barrier arguments are finite, normal and of moderate size (distances in metres), so the
// special-case handling of the library routines (NaN / inf / denormal / sign tests, IEEE-exact
// division) is dead weight in kernels that are bound by the fp64 pipe.  Both routines below are
// accurate to about 1 ulp; the reference's results are reproduced to ~1e-15 relative, far inside
// the 1e-4 parity tolerance (the grouped logs already re-associate at that level).

// 1 / g for normal finite g != 0: hardware estimate + two Newton steps (6 instructions; the
// IEEE-exact division sequence is 12)
#ifdef CILQR_REF_ORDER   // test-only build (ref_order.hpp): IEEE division and the library's sin / cos / tan
CILQR_DEV double fast_rcp(double g) { return 1.0 / g; }
#else
CILQR_DEV double fast_rcp(double g) {
  double r = __builtin_amdgcn_rcp(g);
  r = fma(fma(-g, r, 1.0), r, r);
  r = fma(fma(-g, r, 1.0), r, r);
  return r;
}
#endif

// log(x * 2^e2) for normal finite x > 0.  Classic argument reduction x = 2^k (1 + f),
// sqrt(1/2) <= 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with
// the degree-7 minimax polynomial R of Sun's fdlibm e_log.c (error < 1 ulp); ~40 instructions
// against ~100 for the library log().  e2 lets a caller carry the exponent of a long product
// separately (see bar_group_value).
CILQR_DEV double log_pos(double x, int e2) {
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                   Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                   Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                   Lg7 = 1.479819860511658591e-01;
  double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x) + e2;
  const bool low = m < 0.70710678118654752440;
  m = low ? 2.0 * m : m;                              // [sqrt(1/2), sqrt(2))
  k -= low ? 1 : 0;
  const double f = m - 1.0;
  const double s = f * fast_rcp(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// sin and cos of x for |x| <= 1e5 (the states keep every angle wrapped to [-pi, pi)): three-step
// Cody-Waite reduction by pi/2 with fma, then the kernel polynomials of fdlibm k_sin.c / k_cos.c
// on |r| <= pi/4.  ~35 instructions against ~160 for the library sincos(), error about 1 ulp.
#ifdef CILQR_REF_ORDER
CILQR_DEV void lean_sincos(double x, double* sn, double* cs) {
  *sn = sin(x);
  *cs = cos(x);
}
CILQR_DEV double lean_tan(double x) { return tan(x); }
#else
CILQR_DEV void lean_sincos(double x, double* sn, double* cs) {
  constexpr double two_over_pi = 6.36619772367581382433e-01;
  constexpr double pio2_hi = 1.5707963267948966, pio2_lo = 6.123233995736766e-17, pio2_lo2 = -1.4973849048591698e-33;
  constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                   S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                   S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                   C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                   C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double k = rint(x * two_over_pi);
  double r = fma(-k, pio2_hi, x);
  r = fma(-k, pio2_lo, r);
  r = fma(-k, pio2_lo2, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
  const double s = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k;
  const bool odd = (q & 1) != 0;
  const double ss = odd ? c : s, cc = odd ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}
CILQR_DEV double lean_tan(double x) {
  double s, c;
  lean_sincos(x, &s, &c);
  return s * fast_rcp(c);
}
#endif
// x / wheel_base as the reference writes it (vehicle_model.cc:59-82, 131); the product build multiplies by the
// reciprocal, which is the same number for the reference's wheel_base = 1
#ifdef CILQR_REF_ORDER
#define CILQR_OVER_L(p, x) ((x) / (p).wheel_base)
#else
#define CILQR_OVER_L(p, x) ((x) * (p).inv_wheel_base)
#endif

// ---- relaxed log barrier, barrier_function.h:104-140 ----
// value(g) = g < -eps ? -r log(-g) : r/2 (((-g - 2 eps)/eps)^2 - 1) - r log(eps)
// Sum of barrier values with one log per group instead of one per constraint:
//   sum_c -r log(-g_c) = -r log(prod_c -g_c)   over the constraints on the log branch.
// A BarGroup collects the product (mantissa and exponent apart, so any number of factors is
// safe) and the relaxed-branch part.  This only re-associates the reference's sum.
// Branch-free form.  With w = max(g, -eps) + 2 eps the relaxed value is
//   r/2 ((w/eps)^2 - 1) - r log(eps)  =  r/(2 eps^2) (w^2 - eps^2)  -  r log(max(-g, eps)),
// and on the log branch the same expression gives exactly 0 - r log(-g) (w = eps there, bit for
// bit).  So every constraint multiplies max(-g, eps) into the product and adds w^2 to a sum of
// squares: two max, two multiplies and an fma per constraint, no test, no divergence.
struct BarGroup {
  double prod = 1.0;   // running product of max(-g, eps) over the current run of factors
  double sq = 0.0;     // sum of w^2
  int n = 0;           // constraints so far
  int e2 = 0;          // exponent carried over from closed runs
};
template <int N>
CILQR_DEV void bar_accumulate(const Params& p, const double (&g)[N], BarGroup& grp) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    grp.prod *= fmax(-g[k], p.bar_eps);
    const double w = fmax(g[k], -p.bar_eps) + 2.0 * p.bar_eps;
    grp.sq = fma(w, w, grp.sq);
  }
  grp.n += N;
}
// closes a run of factors: the product's exponent moves to e2, the mantissa stays.  Call at
// least every ~100 factors (a factor is a distance in metres: 1e-2 .. 1e3).
CILQR_DEV void bar_renormalize(BarGroup& grp) {
  grp.e2 += __builtin_amdgcn_frexp_exp(grp.prod);
  grp.prod = __builtin_amdgcn_frexp_mant(grp.prod);
}
CILQR_DEV void bar_merge(BarGroup& into, const BarGroup& g) {
  into.prod *= g.prod;
  into.sq += g.sq;
  into.n += g.n;
  into.e2 += g.e2;
}
CILQR_DEV double bar_group_value(const Params& p, const BarGroup& grp) {
  const double quad = p.bar_half_r_inv_eps2 * (grp.sq - (double)grp.n * (p.bar_eps * p.bar_eps));
  return quad - p.bar_r * log_pos(grp.prod, grp.e2);
}
// Gradient and Hessian coefficients of one constraint:
//   Jacbian() = jc * dg;  Hessian() = (c1 dg_i) dg_j - c2 ddg_ij (c2 term only on the log branch;
//   the relaxed branch reuses the gradient coefficient and drops ddg -- reference quirk, kept).
// Log branch: c2 = r/g, c1 = r/g/g, jc = -r/g from one reciprocal; relaxed branch:
// r (g + 2 eps) / eps^2.  Branch-free: both sides are a handful of instructions.
CILQR_DEV void bar_coefs(const Params& p, double g, double& jc, double& c1, double& c2, bool& log_branch) {
  log_branch = g < -p.bar_eps;
  const double inv = fast_rcp(log_branch ? g : 1.0);
  const double c2l = p.bar_r * inv;
  const double jr = p.bar_r * (g + 2.0 * p.bar_eps) * p.bar_inv_eps * p.bar_inv_eps;
  c2 = log_branch ? c2l : 0.0;
  c1 = log_branch ? c2l * inv : jr;
  jc = log_branch ? -c2l : jr;
}

// ---- continuous dynamics f(x,u), vehicle_model.cc:123-138 ----
// The three scalars of the parameter block a dynamics step reads, as a value: a rollout keeps them in registers for its N
// steps (where the state is a view in LDS -- kernels_tail.hip -- every store through a generic pointer may alias the
// block, and its fields would be fetched again after each one).
struct DynP {
  double dt, wheel_base, inv_wheel_base;
};
CILQR_DEV DynP dyn_params(const Params& p) { return DynP{p.dt, p.wheel_base, p.inv_wheel_base}; }

// NormalizeAngle without its rare branch.  Every angle a rollout wraps lies within (-3 pi, 3 pi) -- the states are wrapped
// after every step -- and takes one of the three short, exact paths of normalize_angle; written as selects those are a
// straight line of five instructions.  The fmod branch (and with it three levels of exec-mask bookkeeping per call, seven
// calls per rollout step) moves out of the way: `rare` is raised instead, and the caller repeats the step with the complete
// function if any lane of the wave raised it.  Same operations on the same operands wherever `rare` stays false.
CILQR_DEV double normalize_angle_common(double angle, bool& rare) {
  const double t = angle + kPi;
  // |angle| < 9.4 puts t inside (-2 pi, 4 pi) with room to spare: one compare instead of the two ends (NaN: rare)
  rare |= !(fabs(angle) < 9.4);
  // t - k 2 pi with k = 1 for t >= 2 pi, -1 for t < 0, else 0: k 2 pi is exact, so the fused form rounds once, exactly as the
  // subtraction / addition of the corresponding short path of normalize_angle does (and leaves t alone for k = 0)
  unsigned hi = (t < 0.0) ? 0xBFF00000u : 0u;
  hi = (t >= kTwoPi) ? 0x3FF00000u : hi;
  const double k = __hiloint2double((int)hi, 0);
  const double r = fma(-k, kTwoPi, t);
  return r - kPi;
}
template <bool Common>
CILQR_DEV double wrap_angle(double a, bool& rare) {
  if constexpr (Common) return normalize_angle_common(a, rare);
  else return normalize_angle(a);
}

template <bool Common>
CILQR_DEV void dyn_continuous(const DynP& p, const double* s, const double* u, double* r, bool& rare) {
  const double theta = wrap_angle<Common>(s[2], rare);
  const double v = s[3];
  const double delta = wrap_angle<Common>(s[5], rare);
  double sn, cs;
  lean_sincos(theta, &sn, &cs);
  r[0] = v * cs;
  r[1] = v * sn;
  r[2] = CILQR_OVER_L(p, v * lean_tan(delta));
  r[3] = s[4];
  r[4] = u[0];
  r[5] = u[1];
}

// ---- RK2 midpoint step, vehicle_model.cc:88-121 ----
template <bool Common>
CILQR_DEV void dynamics_body(const DynP& p, const double* s, const double* u, double* o, bool& rare) {
  double k1[6], mid[6], k2[6];
  dyn_continuous<Common>(p, s, u, k1, rare);
  const double h = 0.5 * p.dt;
#pragma unroll
  for (int i = 0; i < 6; ++i) mid[i] = s[i] + h * k1[i];
  dyn_continuous<Common>(p, mid, u, k2, rare);
#pragma unroll
  for (int i = 0; i < 6; ++i) o[i] = s[i] + p.dt * k2[i];
  o[2] = wrap_angle<Common>(o[2], rare);
  o[5] = wrap_angle<Common>(o[5], rare);
}
// The complete step, out of line: only a wave in which some lane left the short paths ever comes here (never, on any
// scene of the test and bench sets), so its code -- the fmod loops of six angle wraps -- stays out of the callers' loops.
struct StepResult {
  double x[6];
  double u1;
};
__device__ __attribute__((noinline)) StepResult dynamics_exact(DynP p, double x0, double x1, double x2, double x3, double x4, double x5,
                                                              double u0, double u1_raw, int wrap_u1) {
  const double s[6] = {x0, x1, x2, x3, x4, x5};
  double u[2] = {u0, wrap_u1 ? normalize_angle(u1_raw) : u1_raw};
  bool unused = false;
  StepResult r;
  dynamics_body<false>(p, s, u, r.x, unused);
  r.u1 = u[1];
  return r;
}
// One step: the straight-line form first, the complete one again if a lane of the wave left the short paths (the lanes
// that did not would compute the same bits again, so only those that did are rewritten).  `out` may be `s`.
CILQR_DEV void dynamics(const DynP& p, const double* s, const double* u, double* out) {
  bool rare = false;
  double o[6];
  dynamics_body<true>(p, s, u, o, rare);
  if (__builtin_amdgcn_ballot_w64(rare) != 0) {
    if (rare) {
      const StepResult r = dynamics_exact(p, s[0], s[1], s[2], s[3], s[4], s[5], u[0], u[1], 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) o[i] = r.x[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) out[i] = o[i];
}
CILQR_DEV void dynamics(const Params& p, const double* s, const double* u, double* out) {
  dynamics(dyn_params(p), s, u, out);
}
// A closed-loop rollout step (Forward, cc:407-410): delta_rate wrapped (cc:408), then the dynamics; u[1] is replaced by its
// wrapped value, `xn` may be `x`.  One test for the rare path covers both.
// Straight = false: the step as a chain of complete NormalizeAngle calls (branches).  The rollout kernels of the lockstep
// loop use that form: their waves share a SIMD with others, the untaken branches are nearly free there, and the out-of-line
// call of the straight form costs them scalar-register spills (measured, r04 log 11: bulk rollout launch 139 us branchy,
// 148 us straight with the exact step inlined, 161 us straight with the call).  Same bits either way.
template <bool Straight = true>
CILQR_DEV void closed_loop_step(const DynP& p, const double* x, double* u, double* xn) {
  if constexpr (!Straight) {
    bool unused = false;
    u[1] = normalize_angle(u[1]);
    double o[6];
    dynamics_body<false>(p, x, u, o, unused);
#pragma unroll
    for (int i = 0; i < 6; ++i) xn[i] = o[i];
    return;
  }
  bool rare = false;
  const double u1_raw = u[1];
  u[1] = normalize_angle_common(u1_raw, rare);
  double o[6];
  dynamics_body<true>(p, x, u, o, rare);
  if (__builtin_amdgcn_ballot_w64(rare) != 0) {
    if (rare) {
      const StepResult r = dynamics_exact(p, x[0], x[1], x[2], x[3], x[4], x[5], u[0], u1_raw, 1);
#pragma unroll
      for (int i = 0; i < 6; ++i) o[i] = r.x[i];
      u[1] = r.u1;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) xn[i] = o[i];
}

// ---- analytic Jacobian of the midpoint map, vehicle_model.cc:21-86 ----
// Only the state-dependent entries are produced; the rest of A/B is constant:
//   A = I, A(3,4) = dt;  B(3,0) = dt^2/2, B(4,0) = dt, B(5,1) = dt.
struct DynJac {
  double a02, a03, a04, a05;
  double a12, a13, a14, a15;
  double a23, a24, a25;
  double b21;
};
CILQR_DEV void dynamics_jacobian(const Params& p, const double* s, const double* u, DynJac& J) {
  const double dt = p.dt;
  const double v = s[3];
  const double theta = normalize_angle(s[2]);
  const double delta = normalize_angle(s[5]);
  const double a = s[4];
  const double delta_rate = u[1];
  const double tan_delta = lean_tan(delta);
  const double theta_mid = theta + CILQR_OVER_L(p, 0.5 * dt * v * tan_delta);
  const double tan_dr = lean_tan(delta + 0.5 * dt * delta_rate);
  double sin_m, cos_m;
  lean_sincos(theta_mid, &sin_m, &cos_m);
  const double td2 = tan_delta * tan_delta;
  const double tdr2 = tan_dr * tan_dr;
  const double v_tdr = v * (tdr2 + 1);
  const double vm = 0.5 * a * dt + v;
  J.a02 = -dt * vm * sin_m;
  J.a03 = dt * cos_m - CILQR_OVER_L(p, 0.5 * dt * dt * vm * sin_m * tan_delta);
  J.a04 = 0.5 * dt * dt * cos_m;
  J.a05 = CILQR_OVER_L(p, -0.5 * dt * dt * v * vm * (td2 + 1) * sin_m);
  J.a12 = dt * vm * cos_m;
  J.a13 = dt * sin_m + CILQR_OVER_L(p, 0.5 * dt * dt * vm * cos_m * tan_delta);
  J.a14 = 0.5 * dt * dt * sin_m;
  J.a15 = CILQR_OVER_L(p, 0.5 * dt * dt * v * vm * (td2 + 1) * cos_m);
  J.a23 = CILQR_OVER_L(p, dt * tan_dr);
  J.a24 = CILQR_OVER_L(p, 0.5 * dt * dt * tan_dr);
  J.a25 = CILQR_OVER_L(p, dt * v_tdr);
  J.b21 = CILQR_OVER_L(p, 0.5 * dt * dt * v * (tdr2 + 1));
}

// ---- nearest lane segment (first minimum wins), cc:605-618 + line_segment2d.cpp:61-75 ----
// `tab` rows: a b c | sx sy | ux uy | len | ex ey.  Squared distances are compared (sqrt is
// monotone; exact ties -- the shared end point of two consecutive segments -- stay ties, so the
// strict '<' keeps the reference's "first index wins").
CILQR_DEV double segment_dist2(const double* __restrict__ r, double px, double py) {
  double sx = r[3], sy = r[4], ux = r[5], uy = r[6], len = r[7], ex = r[8], ey = r[9];
  // All seven values are requested before anything is computed (one LDS round trip per candidate):
  // left alone, the compiler turns the case selection below into branches and sinks the loads of
  // the end point into one of them, which puts a second dependent round trip and two exec-mask
  // switches on the critical path of every candidate test.
  asm volatile("" : "+v"(sx), "+v"(sy), "+v"(ux), "+v"(uy), "+v"(len), "+v"(ex), "+v"(ey));
  const double x0 = px - sx, y0 = py - sy;
  double d_start = x0 * x0 + y0 * y0;
  const double proj = x0 * ux + y0 * uy;
  const double x1 = px - ex, y1 = py - ey;
  double d_end = x1 * x1 + y1 * y1;
  const double c = x0 * uy - y0 * ux;
  double d_perp = c * c;
  asm volatile("" : "+v"(d_start), "+v"(d_end), "+v"(d_perp));   // three values, then selects
  // same case order as DistanceTo: degenerate, before the start, past the end, foot inside
  return (len <= kMathEps || proj <= 0.0) ? d_start : (proj >= len ? d_end : d_perp);
}

// segment_dist2 that also reports WHERE the distance was taken, as a code: the identity of the end point used (row field
// 10 carries two int32: ids of the segment's start and end point, equal for points with bitwise equal coordinates --
// k_load_lanes), or a code of its own for the foot of the perpendicular.  Two candidates with the same code measure
// to the same point: the same distance in any arithmetic (the shared end point of consecutive segments: the wedge
// outside every joint) -- no tie to re-examine.
constexpr int kPerpCode = 0x10000;   // + segment index; end-point ids are < 2 * CILQR_MAX_LANE_SEGMENTS
CILQR_DEV double segment_dist2_code(const double* __restrict__ r, int seg, double px, double py, int* code) {
  double sx = r[3], sy = r[4], ux = r[5], uy = r[6], len = r[7], ex = r[8], ey = r[9], ids = r[10];
  asm volatile("" : "+v"(sx), "+v"(sy), "+v"(ux), "+v"(uy), "+v"(len), "+v"(ex), "+v"(ey), "+v"(ids));
  const double x0 = px - sx, y0 = py - sy;
  double d_start = x0 * x0 + y0 * y0;
  const double proj = x0 * ux + y0 * uy;
  const double x1 = px - ex, y1 = py - ey;
  double d_end = x1 * x1 + y1 * y1;
  const double c = x0 * uy - y0 * ux;
  double d_perp = c * c;
  asm volatile("" : "+v"(d_start), "+v"(d_end), "+v"(d_perp));
  const bool at_start = (len <= kMathEps || proj <= 0.0), at_end = proj >= len;
  const long long both = __double_as_longlong(ids);
  *code = at_start ? (int)(both & 0xffffffffll) : (at_end ? (int)(both >> 32) : kPerpCode + seg);
  return at_start ? d_start : (at_end ? d_end : d_perp);
}

// Squared distances order like distances, with one exception that is left as it is.  The reference compares
// DISTANCES (hypot / |cross|, line_segment2d.cpp:61-75): two squared distances one or two ulp apart have the same
// square root, which its strict '<' treats as a tie (first index wins).  That happens on a strip ~1e-7 m wide along
// the normal through a segment end point (perpendicular distance to one segment against the end-point distance to its
// neighbour); there this search returns the strictly nearer segment and the reference the earlier one.  Measured
// alternatives (k_cost_knots at full batch, 481 us as is): comparing the roots inside a 4-ulp window, inline +34 %,
// out of line +18 %; a plain 2-ulp margin costs nothing but only moves the disagreement to the fuzzy edge of the
// strip.  The parity tests recognise the case (an exact tie of the oracle's own distances) and say so.
// hypot as the reference's libm evaluates it.  glibc 2.35 (sysdeps/ieee754/dbl-64/e_hypot.c, the build without FMA
// that x86-64 -O2 binaries get): h = sqrt(ax^2 + ay^2) with ax >= ay, corrected by one Newton step whose residual is
// evaluated in two exact-ish parts.  The device library's hypot differs from it in the last bit on ~0.6 % of
// arguments (measured against numpy on 200 000 pairs; this routine on none) -- enough to turn the reference's exact
// ties into non-ties.  Arguments here are lengths in metres: the scaling branches for huge / tiny values are not needed.
CILQR_DEV double hypot_ref(double x, double y) {
  x = fabs(x);
  y = fabs(y);
  const double ax = x < y ? y : x, ay = x < y ? x : y;
  if (ax >= ay * 0x1p54) return ax + ay;
  double h = sqrt(ax * ax + ay * ay);
  double t1, t2;
  if (h <= 2.0 * ay) {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}

// LineSegment2d::DistanceTo as the reference evaluates it (line_segment2d.cpp:61-75): hypot to an end point, |cross| to
// the foot.  Only the exact-tie rule below calls it.
CILQR_DEV double segment_dist_ref(const double* __restrict__ r, double px, double py) {
  const double sx = r[3], sy = r[4], ux = r[5], uy = r[6], len = r[7], ex = r[8], ey = r[9];
  const double x0 = px - sx, y0 = py - sy;
  if (len <= kMathEps) return hypot_ref(x0, y0);
  const double proj = x0 * ux + y0 * uy;
  if (proj <= 0.0) return hypot_ref(x0, y0);
  if (proj >= len) return hypot_ref(px - ex, py - ey);
  return fabs(x0 * uy - y0 * ux);
}
// CILQR_OPT_EXACT_LANE_TIES.  Squared distances order like the reference's distances except when the two smallest are
// within rounding of each other.  The search notes when a candidate comes within kTieWindow (relative; far wider
// than any rounding difference between the two forms) of the best so far; only then the candidate list is searched
// again with the reference's own distance values and its strict '<' (first index wins) -- once, after the loop, out
// of line.  Whichever of the two smallest comes first in the list, the other meets it as "best so far", so no near-tie
// escapes.  A rare branch: the iterates that come to rest on a tie strip.
constexpr double kTieWindow = 1e-13;
CILQR_DEV bool near_tie(double d2, double best) { return fabs(d2 - best) <= kTieWindow * best; }   // best = DBL_MAX at first: false
// segments first, first + step, ... (n of them; list == nullptr) or the n bytes of a grid cell's candidate list
__device__ __attribute__((noinline)) int nearest_by_reference_distance(const double* __restrict__ tab, int n, uint4 raw, int from_cell,
                                                                       double px, double py) {
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
  double best = DBL_MAX;
  int bi = 0;
#pragma unroll 1
  for (int k = 0; k < n; ++k) {
    const int kk = k + 1;
    const unsigned word = (kk < 4) ? w[0] : (kk < 8) ? w[1] : (kk < 12) ? w[2] : w[3];
    const int seg = from_cell ? (int)((word >> ((kk & 3) * 8)) & 0xffu) : k;
    const double d = segment_dist_ref(tab + seg * kLaneFields, px, py);
    if (d < best) {                                             // cc:611-614
      best = d;
      bi = seg;
    }
  }
  return bi;
}
CILQR_DEV int nearest_segment_scan(const double* __restrict__ tab, int n, double px, double py, bool exact) {
  double best = DBL_MAX;
  int bi = 0;
  bool suspect = false;
  int bcode = -1;
#pragma unroll 1
  for (int s = 0; s < n; ++s) {
    int code = -2;
    const double d2 = exact ? segment_dist2_code(tab + s * kLaneFields, s, px, py, &code) : segment_dist2(tab + s * kLaneFields, px, py);
    if (exact) suspect |= near_tie(d2, best) & (code != bcode);
    if (d2 < best) {
      best = d2;
      bi = s;
      bcode = code;
    }
  }
  if (exact && suspect) bi = nearest_by_reference_distance(tab, n, make_uint4(0u, 0u, 0u, 0u), 0, px, py);
  return bi;
}

// Grid-accelerated version: only the candidate segments of the point's cell are tested, in
// ascending index order.  The candidate sets are conservative (triangle inequality, see
// k_build_lane_grid), so the result is the one of the full scan.
// candidate list of the grid cell that contains (px, py); off-grid points get the "scan
// everything" marker.  Split from the search so a kernel can issue the loads of all its discs
// first and hide their latency behind each other.
CILQR_DEV uint4 lane_cell_fetch(const DeviceState& s, int side, double px, double py) {
  const double fx = (px - s.gx0) * s.ginv_h, fy = (py - s.gy0) * s.ginv_h;
  if (!(fx >= 0.0 && fy >= 0.0 && fx < (double)s.gnx && fy < (double)s.gny))
    return make_uint4((unsigned)kGridFullScan, 0u, 0u, 0u);
  const int cell = (int)fy * s.gnx + (int)fx;
  return *reinterpret_cast<const uint4*>(s.lgrid + ((size_t)side * s.gnx * s.gny + cell) * kGridCellBytes);
}
// EX: CILQR_OPT_EXACT_LANE_TIES as a compile-time choice -- the kernels exist in both forms, so that the rare-branch
// code (a call, live ranges across it) costs the default form nothing (it cost 2 % as a run-time flag, measured).
template <bool EX = false>
CILQR_DEV int nearest_from_cell(const DeviceState& s, const double* __restrict__ lanes, int side, uint4 raw,
                                double px, double py) {
  const double* __restrict__ tab = lanes + (side ? s.nl * kLaneFields : 0);
  const int n = side ? s.nr : s.nl;
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
  const int cnt = (int)(w[0] & 0xffu);
  constexpr bool exact = EX;
  if (cnt == kGridFullScan) return nearest_segment_scan(tab, n, px, py, exact);
  double best = DBL_MAX;
  int bi = 0;
  bool suspect = false;
  int bcode = -1;                // EX: where the best distance was taken (segment_dist2_code)
  // Compact loop (not unrolled: this function is inlined at every disc of three kernels and an
  // unrolled 15-way test made them instruction-cache bound).  The trip count is the longest list
  // of the wave, tested with a ballot, so the loop is a scalar branch around straight-line
  // predicated code instead of a divergent loop with its exec-mask bookkeeping; a lane past the
  // end of its list tests row 0 and discards the result.
  // The list travels as a 128-bit shift register: byte 1 of the lowest word is the next candidate (byte 0 of the cell was the
  // count), one funnel shift per word and iteration instead of a four-way select on the trip counter.
  unsigned q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3];
#pragma unroll 1
  for (int k = 1; __builtin_amdgcn_ballot_w64(k <= cnt) != 0; ++k) {
    const bool in = k <= cnt;
    const int seg = in ? (int)((q0 >> 8) & 0xffu) : 0;
    q0 = __builtin_amdgcn_alignbit(q1, q0, 8);
    q1 = __builtin_amdgcn_alignbit(q2, q1, 8);
    q2 = __builtin_amdgcn_alignbit(q3, q2, 8);
    q3 >>= 8;
    double d2;
    bool take;
    if constexpr (exact) {
      int code;
      d2 = segment_dist2_code(tab + seg * kLaneFields, seg, px, py, &code);
      // bitwise, not short-circuit: three compares and two s_and are cheaper than the two nested divergent branches (exec
      // saved and restored around each) the compiler makes of `&&` here
      suspect |= in & near_tie(d2, best) & (code != bcode);
      take = in & (d2 < best);
      bcode = take ? code : bcode;
    } else {
      d2 = segment_dist2(tab + seg * kLaneFields, px, py);
      take = in & (d2 < best);
    }
    best = take ? d2 : best;
    bi = take ? seg : bi;
  }
  if constexpr (exact) {
    if (suspect) bi = nearest_by_reference_distance(tab, cnt, raw, 1, px, py);
  }
  return bi;
}
// `lanes`: the lane table (left rows then right rows), in global memory or staged in LDS.
template <bool EX = false>
CILQR_DEV int nearest_segment(const DeviceState& s, const double* __restrict__ lanes, int side, double px,
                              double py) {
  return nearest_from_cell<EX>(s, lanes, side, lane_cell_fetch(s, side, px, py), px, py);
}

// LDS exchange among the lanes of ONE wavefront (the wave-per-problem kernels): no s_barrier, only ordering
struct WaveSync {
  CILQR_DEV void operator()() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

// The tail kernel's view keeps a problem's tensors in LDS as far as they fit; the device functions reach them through the
// generic pointers of DeviceState (flat_load / flat_store: every access waits for the address check and shares a counter
// with the real LDS traffic).  Where a launch KNOWS that the tensors a phase touches are in LDS it says so, and the
// compiler's address-space inference turns the accesses behind the pointer into ds_read / ds_write.
template <bool InLds, class T>
CILQR_DEV void assume_lds(T* ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (InLds) __builtin_assume(__builtin_amdgcn_is_shared((const void*)ptr));
#else
  (void)ptr;
#endif
}

// Which list position a workgroup of the wavefront-per-problem kernels takes (k_backward_wave, k_init_guess_wave).  A wavefront fetches ONE pair (16 B) of each of its problem's rows per step (lin,
// term, gains at the problem's position, U at its slot); the other seven pairs of the same 128-byte line belong to the next
// seven positions.  Workgroups are dealt to the eight XCDs round-robin (block b on XCD b % 8: observed, not promised -- a
// wrong guess only costs speed), and every XCD has an L2 of its own: with position = blockIdx those eight wavefronts sat
// on eight XCDs and every line was fetched into eight L2s -- 2280 B per problem-step through FETCH_SIZE against 288 B read
// (rocprofv3 --pmc, a launch of 2703 problems; found when CILQR_OPT_WAVE_THRESHOLD went from 1024 to 3072 and the backward
// pass's counted bytes rose from 418 to 466 B per problem-step).  Block b = 8 k + x takes position
// (k / 8) * 64 + x * 8 + k % 8: eight consecutive positions -- one line of every row -- per XCD.  A bijection on [0, 64 m).
CILQR_DEV int xcd_local_position(int b) {
  const int x = b & 7, k = b >> 3;
  return ((k >> 3) << 6) + (x << 3) + (k & 7);
}

// where the per-iteration scratch of a slot (lin, term, gains) lives: see DeviceState::posn
CILQR_DEV int scratch_index(const DeviceState& s, int slot) { return s.posn ? s.posn[slot] : slot; }

// number of list entries a kernel of the solve loop has to process (see DeviceState::n_dev)
CILQR_DEV int active_count(const DeviceState& s, int n_host) {
  return s.n_dev ? min(*s.n_dev, n_host) : n_host;
}
// entries a pass over a device-side list takes: the list's count (n_ptr) less the `off` entries earlier passes took,
// at most n_max; no list count = the active list
CILQR_DEV int list_count(const DeviceState& s, const int* __restrict__ n_ptr, int off, int n_max) {
  if (n_ptr == nullptr) return active_count(s, n_max);
  const int left = *n_ptr - off;
  return left < 0 ? 0 : (left < n_max ? left : n_max);
}

// small helpers for the batch-fastest pair layout
CILQR_DEV double2 ld2(const double2* __restrict__ base, int row, int Bcap, int slot) {
  return base[(size_t)row * Bcap + slot];
}
CILQR_DEV void st2(double2* __restrict__ base, int row, int Bcap, int slot, double a, double b) {
  base[(size_t)row * Bcap + slot] = make_double2(a, b);
}

// current-iterate state/control of a slot
CILQR_DEV void load_x(const DeviceState& s, int buf, int i, int slot, double* x) {
  const double2* b = s.X + ((size_t)buf * s.p.K + i) * 3 * s.Bcap;
  const double2 p0 = b[slot], p1 = b[(size_t)s.Bcap + slot], p2 = b[(size_t)2 * s.Bcap + slot];
  x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y;
}
CILQR_DEV void store_x(const DeviceState& s, int buf, int i, int slot, const double* x) {
  double2* b = s.X + ((size_t)buf * s.p.K + i) * 3 * s.Bcap;
  b[slot] = make_double2(x[0], x[1]);
  b[(size_t)s.Bcap + slot] = make_double2(x[2], x[3]);
  b[(size_t)2 * s.Bcap + slot] = make_double2(x[4], x[5]);
}
CILQR_DEV void load_u(const DeviceState& s, int buf, int i, int slot, double* u) {
  const double2 q = s.U[((size_t)buf * s.p.N + i) * s.Bcap + slot];
  u[0] = q.x; u[1] = q.y;
}
CILQR_DEV void store_u(const DeviceState& s, int buf, int i, int slot, const double* u) {
  s.U[((size_t)buf * s.p.N + i) * s.Bcap + slot] = make_double2(u[0], u[1]);
}

// one point of TransformToTrajectory (cc:771-791): t x y theta v a delta kappa jerk delta_rate
CILQR_DEV void write_traj_point(const DeviceState& s, int buf, int i, int slot, double* __restrict__ o) {
  double x[6], u[2] = {0.0, 0.0};
  load_x(s, buf, i, slot, x);
  if (i < s.p.N) load_u(s, buf, i, slot, u);
  o[0] = i * s.p.dt;
  o[1] = x[0]; o[2] = x[1]; o[3] = x[2]; o[4] = x[3]; o[5] = x[4]; o[6] = x[5];
  o[7] = tan(x[5]) / s.p.wheel_base;
  o[8] = u[0]; o[9] = u[1];
}

// sum of the knot partials in index order -> c5 (total, J, dynamics, corridor, lane).  `cand`: which buffer of the
// slot the partials were computed on (0 = the iterate, 1 = the candidate) -- only the test-only reference-order build
// needs it, which re-evaluates the whole cost from that trajectory instead of summing partials.
CILQR_DEV void reduce_cost(const DeviceState& s, int slot, int cand, double* c5);

}  // namespace cilqr
