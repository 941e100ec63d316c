// Alternative init guess: the closed-loop tracker the reference keeps next to iqr() and recommends for
// better results (README.md:61-67; IlqrOptimizer::InitGuess, ilqr_optimizer.cc:107-139, commented out at cc:168).
//
// Reference behaviour (algorithm/ilqr/tracker.cc): Tracker::lqr cc:169-215 simulates the vehicle at sumulation_dt
// = 10 ms with RK4 (VehicleDynamic cc:83-135); every step, CalcaulateInitState cc:19-55 projects a previewed point on
// the coarse trajectory (DiscretizedTrajectory::GetProjection / EvaluateTime) and two 3-state LQR controllers give
// delta_rate and jerk (LateralControl cc:57-72, LongitudinalControl cc:74-81) from gains of the iterative DARE solver
// math::SolveLQRProblem (algorithm/math/linear_quadratic_regulator.cc:30-78), re-solved every step for the lateral
// model (it depends on the speed).  The state is kept whenever the clock passes a knot of the coarse trajectory.
//
// One lane per problem; the coarse trajectory is read batch-fastest from `goals` (knot 0 from `coarse0`, because
// goals_[0] is the start state) and `cstation`.  Products accumulate k = 0, 1, 2 like Eigen's coefficient-based
// product of small dynamic matrices; the DARE loop's stopping test is the reference's (fabs of the max coefficient of
// P_next - P against the tolerance).  The longitudinal gains do not depend on the state and are solved once.
#include "dev_model.hpp"

namespace cilqr {

namespace {

struct M3 {
  double m[9];
};
CILQR_DEV void mul33(const M3& a, const M3& b, M3& r) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = a.m[i * 3 + 0] * b.m[0 * 3 + j];
      s += a.m[i * 3 + 1] * b.m[1 * 3 + j];
      s += a.m[i * 3 + 2] * b.m[2 * 3 + j];
      r.m[i * 3 + j] = s;
    }
}
CILQR_DEV void row33(const double* v, const M3& m, double* out) {   // (1x3) (3x3)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double s = v[0] * m.m[0 * 3 + j];
    s += v[1] * m.m[1 * 3 + j];
    s += v[2] * m.m[2 * 3 + j];
    out[j] = s;
  }
}

// math::SolveLQRProblem, linear_quadratic_regulator.cc:30-78, with B 3x1, R 1x1, M = 0 -> K (1x3)
CILQR_DEV void solve_lqr(const M3& A, const double* B, const M3& Q, double R, double tolerance, int max_iter, double* K) {
  M3 AT;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) AT.m[i * 3 + j] = A.m[j * 3 + i];
  M3 P = Q;
  int num_iteration = 0;
  double diff = DBL_MAX;
#pragma unroll 1
  while (num_iteration++ < max_iter && diff > tolerance) {
    M3 ATP, ATPA;
    mul33(AT, P, ATP);
    mul33(ATP, A, ATPA);
    double ATPB[3], BTP[3], BTPA[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s = ATP.m[i * 3 + 0] * B[0];
      s += ATP.m[i * 3 + 1] * B[1];
      s += ATP.m[i * 3 + 2] * B[2];
      ATPB[i] = s + 0.0;
    }
    row33(B, P, BTP);
    double BTPB = BTP[0] * B[0];
    BTPB += BTP[1] * B[1];
    BTPB += BTP[2] * B[2];
    const double inv = 1.0 / (R + BTPB);
    row33(BTP, A, BTPA);
    double maxc = -DBL_MAX;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double left = ATPB[i] * inv;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double pn = (ATPA.m[i * 3 + j] - left * (BTPA[j] + 0.0)) + Q.m[i * 3 + j];
        maxc = fmax(maxc, pn - P.m[i * 3 + j]);
        ATP.m[i * 3 + j] = pn;   // ATP is dead: reuse it for P_next
      }
    }
    diff = fabs(maxc);
    P = ATP;
  }
  double BTP[3], BTPA[3];
  row33(B, P, BTP);
  double BTPB = BTP[0] * B[0];
  BTPB += BTP[1] * B[1];
  BTPB += BTP[2] * B[2];
  const double inv = 1.0 / (R + BTPB);
  row33(BTP, A, BTPA);
#pragma unroll
  for (int j = 0; j < 3; ++j) K[j] = inv * (BTPA[j] + 0.0);
}

CILQR_DEV double slerp(double a0, double t0, double a1, double t1, double t) {   // math_utils.h:208-225
  if (fabs(t1 - t0) <= kMathEps) return normalize_angle(a0);
  const double a0_n = normalize_angle(a0), a1_n = normalize_angle(a1);
  double d = a1_n - a0_n;
  if (d > kPi) d = d - 2 * kPi;
  else if (d < -kPi) d = d + 2 * kPi;
  const double r = (t - t0) / (t1 - t0);
  return normalize_angle(a0_n + d * r);
}

struct FollowPt {
  double s, x, y, theta, v;
};

}  // namespace

__global__ __launch_bounds__(64) void k_init_guess_tracker(DeviceState s, TrackerParams tp, int B) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= B) return;
  const Params& p = s.p;
  const int K = p.K, Bc = s.Bcap;
  // the coarse trajectory: knot 0 from coarse0 (goals_[0] holds the start state), time_i = i dt (dp_planner.cpp:236)
  auto follow = [&](int i) {
    FollowPt f;
    if (i == 0) {
      const double2 a = s.coarse0[slot], b = s.coarse0[(size_t)Bc + slot];
      f.x = a.x; f.y = a.y; f.theta = b.x; f.v = b.y;
    } else {
      const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
      const double2 a = gp[0], b = gp[(size_t)Bc];
      f.x = a.x; f.y = a.y; f.theta = b.x; f.v = b.y;
    }
    f.s = s.cstation[(size_t)i * Bc + slot];
    return f;
  };
  auto follow_xy = [&](int i, double& x, double& y) {
    const double2 a = (i == 0) ? s.coarse0[slot] : s.goals[(size_t)i * 3 * Bc + slot];
    x = a.x;
    y = a.y;
  };
  if (!tp.have_station) {   // no stations from the caller: accumulated chord length of the coarse points
    double px, py, acc = 0.0;
    follow_xy(0, px, py);
    s.cstation[slot] = 0.0;
    for (int i = 1; i < K; ++i) {
      double x, y;
      follow_xy(i, x, y);
      acc = acc + hypot_ref(x - px, y - py);
      s.cstation[(size_t)i * Bc + slot] = acc;
      px = x;
      py = y;
    }
  }
  // InitMatrix tracker.cc:137-167
  M3 latA, latQ, lonA, lonQ;
#pragma unroll
  for (int e = 0; e < 9; ++e) latA.m[e] = latQ.m[e] = lonA.m[e] = lonQ.m[e] = 0.0;
  latA.m[0] = 1.0; latA.m[4] = 1.0; latA.m[8] = 1.0;
  const double latB[3] = {0.0, 0.0, 1.0 * tp.dt};
  latQ.m[0] = tp.weight_l; latQ.m[4] = tp.weight_theta; latQ.m[8] = tp.weight_delta;
  lonA.m[0] = 1.0; lonA.m[4] = 1.0; lonA.m[8] = 1.0;
  lonA.m[1] = tp.dt;
  lonA.m[5] = -tp.dt;
  const double lonB[3] = {0.0, 0.0, 1.0 * tp.dt};
  lonQ.m[0] = tp.weight_s; lonQ.m[4] = tp.weight_v; lonQ.m[8] = tp.weight_a;
  double Klon[3];
  solve_lqr(lonA, lonB, lonQ, tp.weight_j, tp.tolerance, tp.max_num_iteration, Klon);   // constant over the simulation

  // start_state_: x, y, theta, velocity of goals_[0]; a = delta = 0 (ilqr_optimizer.cc:61, trajectory_planner.cpp:73-76)
  double cx, cy, cth, cv, ca = 0.0, cdl = 0.0;
  {
    const double2 a = s.goals[slot], b = s.goals[(size_t)Bc + slot];
    cx = a.x; cy = a.y; cth = b.x; cv = b.y;
  }
  {
    const double x0[6] = {cx, cy, cth, cv, 0.0, 0.0};
    store_x(s, 0, 0, slot, x0);
  }
  const double start_time = p.dt * 0, end_time = p.dt * (K - 1);
  double ctime = start_time;
  int i = 1;
#pragma unroll 1
  for (double t = start_time; t < end_time + kMathEps; t += tp.sim_dt) {
    // ---- CalcaulateInitState cc:19-55 ----
    double sn, cs;
    lean_sincos(cth, &sn, &cs);
    const double pvx = cx + cs * cv * tp.preview_time;
    const double pvy = cy + sn * cv * tp.preview_time;
    // GetProjection discretized_trajectory.cpp:165-197: nearest knot (first minimum), then the point at station
    // s_start + (v0 . v1) / |v1| between its two neighbours
    int idx = 0;
    {
      double nearest = DBL_MAX;
#pragma unroll 4
      for (int k = 0; k < K; ++k) {
        double fx, fy;
        follow_xy(k, fx, fy);
        const double dx = fx - pvx, dy = fy - pvy;
        const double d = dx * dx + dy * dy;
        if (d < nearest) {
          nearest = d;
          idx = k;
        }
      }
    }
    FollowPt proj = follow(idx);
    {
      const int i0 = max(0, idx - 1), i1 = min(K - 1, idx + 1);
      if (i0 < i1) {
        const FollowPt p0 = follow(i0), p1 = follow(i1);
        const double v0x = pvx - p0.x, v0y = pvy - p0.y;
        const double v1x = p1.x - p0.x, v1y = p1.y - p0.y;
        const double v1_norm = sqrt(v1x * v1x + v1y * v1y);
        const double dot = v0x * v1x + v0y * v1y;
        const double st = p0.s + dot / v1_norm;
        if (fabs(p1.s - p0.s) < kMathEps) {   // LinearInterpolateTrajectory cpp:66-89
          proj = p0;
        } else {
          const double w = (st - p0.s) / (p1.s - p0.s);
          proj.s = st;
          proj.x = (1 - w) * p0.x + w * p1.x;
          proj.y = (1 - w) * p0.y + w * p1.y;
          proj.theta = slerp(p0.theta, p0.s, p1.theta, p1.s, st);
          proj.v = (1 - w) * p0.v + w * p1.v;
        }
      }
    }
    double psn, pcs;
    lean_sincos(proj.theta, &psn, &pcs);
    const double ddx = cx - proj.x, ddy = cy - proj.y;
    const double lat0 = psn * ddx - pcs * ddy;
    const double lat1 = normalize_angle(proj.theta - cth);
    const double lat2 = cdl;
    // EvaluateTime(cur.time) cpp:130-141: first knot whose time is not below, interpolated from its predecessor
    double match_s, match_v;
    {
      const double time = ctime + 0.0;
      int it;
      if (time >= p.dt * (K - 1)) {
        it = K - 1;
      } else if (time < p.dt * 0) {
        it = 0;
      } else {
        it = min(K - 1, max(0, (int)(time / p.dt)));
        while (it > 0 && !(p.dt * (it - 1) < time)) --it;
        while (p.dt * it < time) ++it;
      }
      if (it == 0) it = 1;
      const FollowPt p0 = follow(it - 1), p1 = follow(it);
      const double t0 = p.dt * (it - 1), t1 = p.dt * it;
      if (fabs(t1 - t0) < kMathEps) {
        match_s = p0.s;
        match_v = p0.v;
      } else {
        const double w = (time - t0) / (t1 - t0);
        match_s = (1 - w) * p0.s + w * p1.s;
        match_v = (1 - w) * p0.v + w * p1.v;
      }
    }
    const double lon0 = match_s - proj.s, lon1 = match_v - cv, lon2 = ca;
    // ---- LateralControl cc:57-72 ----
    const double v_amend = fmax(2.0, cv);
    latA.m[1] = v_amend * 0.1;
    latA.m[5] = -v_amend / p.wheel_base * 0.1;
    double Klat[3];
    solve_lqr(latA, latB, latQ, tp.weight_delta_rate, tp.tolerance, tp.max_num_iteration, Klat);
    double delta_rate, jerk;
    {
      double a = Klat[0] * lat0;
      a += Klat[1] * lat1;
      a += Klat[2] * lat2;
      delta_rate = -a;
      double b = Klon[0] * lon0;   // LongitudinalControl cc:74-81
      b += Klon[1] * lon1;
      b += Klon[2] * lon2;
      jerk = -b;
    }
    delta_rate = fmax(p.delta_rate_min, fmin(p.delta_rate_max, delta_rate));
    jerk = fmax(p.jerk_min, fmin(p.jerk_max, jerk));
    // ---- VehicleDynamic cc:83-135: RK4 of (x, y, theta, v, delta, a) ----
    {
      const double h = tp.sim_dt, h2 = h / 2.0;
      double k1[4], k2[4], k3[4], k4[4];   // x', y', theta', (v' = a; delta' = delta_rate; a' = jerk)
      auto mode = [&](double th, double v, double dl, double* o) {   // vehicle_mode tracker.h:72-87
        double sn_, cs_;
        lean_sincos(th, &sn_, &cs_);
        o[0] = v * cs_;
        o[1] = v * sn_;
        o[2] = CILQR_OVER_L(p, v * lean_tan(dl));
      };
      mode(cth, cv, cdl, k1);
      const double a1 = ca, a2 = ca + jerk * h2, a3 = ca + jerk * h2, a4 = ca + jerk * h;
      mode(cth + k1[2] * h2, cv + a1 * h2, cdl + delta_rate * h2, k2);
      mode(cth + k2[2] * h2, cv + a2 * h2, cdl + delta_rate * h2, k3);
      mode(cth + k3[2] * h, cv + a3 * h, cdl + delta_rate * h, k4);
      const double nx = cx + (k1[0] + k2[0] * 2.0 + k3[0] * 2.0 + k4[0]) / 6.0 * h;
      const double ny = cy + (k1[1] + k2[1] * 2.0 + k3[1] * 2.0 + k4[1]) / 6.0 * h;
      const double nth = normalize_angle(cth + (k1[2] + k2[2] * 2.0 + k3[2] * 2.0 + k4[2]) / 6.0 * h);
      const double nv = fmax(0.0, cv + (a1 + a2 * 2.0 + a3 * 2.0 + a4) / 6.0 * h);
      const double ndl = normalize_angle(fmin(p.delta_max, fmax(p.delta_min, cdl + (delta_rate + delta_rate * 2.0 + delta_rate * 2.0 + delta_rate) / 6.0 * h)));
      const double na = fmin(p.max_acc, fmax(p.min_acc, ca + (jerk + jerk * 2.0 + jerk * 2.0 + jerk) / 6.0 * h));
      cx = nx; cy = ny; cth = nth; cv = nv; cdl = ndl; ca = na;
    }
    ctime = t;
    if (i < K && ctime > p.dt * i - kMathEps) {   // cc:199-202
      const double u[2] = {jerk, delta_rate};       // what trajectory.back() carries when the next knot is pushed
      store_u(s, 0, i - 1, slot, u);
      const double x6[6] = {cx, cy, cth, cv, ca, cdl};
      store_x(s, 0, i, slot, x6);
      ++i;
    }
  }
  // the reference gives up when the count differs ("tacker failed", cc:205-208); here the remaining knots repeat the
  // last state with zero controls (cannot happen for the reference's time grids: the comparisons carry 1e-10 of slack)
  for (; i < K; ++i) {
    const double u[2] = {0.0, 0.0};
    store_u(s, 0, i - 1, slot, u);
    const double x6[6] = {cx, cy, cth, cv, ca, cdl};
    store_x(s, 0, i, slot, x6);
  }
}

void launch_init_guess_tracker(const DeviceState& s, const TrackerParams& tp, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_init_guess_tracker, dim3((B + 63) / 64), dim3(64), 0, st, s, tp, B);
}

}  // namespace cilqr
