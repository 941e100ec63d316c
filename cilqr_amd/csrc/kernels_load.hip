// Load / prepare / init-guess / export kernels.
//
// Reference behaviour implemented here (algorithm/ilqr/ilqr_optimizer.cc):
//   TransformGoals cc:141-152, ShrinkConstraints cc:438-473, NormalizeHalfPlane cc:475-495,
//   iqr (init guess) cc:793-842, TransformToTrajectory cc:771-791,
//   OpenLoopRollout algorithm/slover/ilqr.h:363-370.
#include <cstdlib>

#include "dev_model.hpp"

namespace cilqr {

// ---------------------------------------------------------------------------------------------
// Corridor: problem-major [B][K][cmax_in][3] -> shrunk, normalised, batch-fastest
// [K][cmax][3][Bcap].  One block = 64 problems x one knot, transposed through LDS so both the
// HBM reads (runs of cmax_in*3 doubles) and the writes (64 consecutive slots) are contiguous.
// ---------------------------------------------------------------------------------------------
constexpr int kLoadPlaneChunk = 16;   // planes per LDS tile: 64 x (48 + 1) doubles = 24.5 KiB for any cmax
__global__ __launch_bounds__(256) void k_load_corridor(DeviceState s, int B, ProblemView in) {
  __shared__ double tile[64 * (kLoadPlaneChunk * 3 + 1)];
  const int i = blockIdx.y;
  const int b0 = blockIdx.x * 64;
  const int row = in.cmax_in * 3;
  constexpr int ld = kLoadPlaneChunk * 3 + 1;
  const int nb = min(64, B - b0);
  const int pb = threadIdx.x & 63;
  const int slot = b0 + pb;
  int cnt = 0;
  if (pb < nb) {
    const int raw = in.ccount[(size_t)slot * s.p.K + i];
    // A negative count is a knot whose corridor could not be built (cilqr_build_corridors codes -2..-4):
    // the reference aborts the whole Plan there (corridor.cc:78-81), so the problem is flagged and
    // never optimised (k_load_goals ran before this kernel and set status = 0).
    if (raw < 0 && (threadIdx.x >> 6) == 0) s.status[slot] = 6;   // CILQR_ST_NO_CORRIDOR
    cnt = max(0, min(raw, min(s.cmax, in.cmax_in)));
    if ((threadIdx.x >> 6) == 0) s.ccnt[(size_t)i * s.Bcap + slot] = cnt;
  }
  for (int c0 = 0; c0 < in.cmax_in; c0 += kLoadPlaneChunk) {
    const int w = min(kLoadPlaneChunk, in.cmax_in - c0) * 3;   // doubles of this chunk per problem
    __syncthreads();
    for (int e = threadIdx.x; e < nb * w; e += blockDim.x) {
      const int q = e / w, off = e - q * w;
      tile[q * ld + off] = in.corridor[((size_t)(b0 + q) * s.p.K + i) * row + c0 * 3 + off];
    }
    __syncthreads();
    if (pb >= nb) continue;
    for (int c = c0 + (threadIdx.x >> 6); c < min(cnt, c0 + kLoadPlaneChunk); c += 4) {
      double a = tile[pb * ld + (c - c0) * 3 + 0];
      double b = tile[pb * ld + (c - c0) * 3 + 1];
      double cc = tile[pb * ld + (c - c0) * 3 + 2];
      cc = cc - s.p.shrink_corridor * (a * a + b * b) / hypot_ref(a, b);   // cc:448
      const double nrm = hypot_ref(hypot_ref(a, b), cc);                       // cc:479
      double* o = s.cor + ((size_t)(i * s.cmax + c) * 3) * s.Bcap + slot;
      double pa = a / nrm, pbn = b / nrm, pc = cc / nrm;
      // A live plane with a NaN / Inf coefficient: in the reference every barrier value of it is non-finite
      // (barrier_function.h:104-113 on a NaN argument), the first TotalCost (cc:172) with it, every trial's z is NaN
      // (cc:255-258) and the solve leaves through lambda > 1e11 (cc:298-307).  The branch-free barrier of the cost kernels
      // takes max(-g, eps), which would DROP a NaN; such a plane is kept as (0, 0, -inf) instead -- g = +inf at every disc,
      // a barrier value of +inf: non-finite like the reference's, so every decision is the reference's (tested).
      if (!(__builtin_isfinite(pa) && __builtin_isfinite(pbn) && __builtin_isfinite(pc))) {
        pa = 0.0; pbn = 0.0; pc = -__builtin_huge_val();
      }
      o[0] = pa;
      o[(size_t)s.Bcap] = pbn;
      o[(size_t)2 * s.Bcap] = pc;
    }
  }
}

// goals_[i] from the coarse trajectory, goals_[0] from the start state (cc:141-152), and the
// per-problem solver state of Optimize() (cc:180-186).
__global__ __launch_bounds__(256) void k_load_goals(DeviceState s, int B, ProblemView in) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = s.p.K;
  if (t >= B * K) return;
  const int slot = t / K, i = t - slot * K;
  double g[6];
  if (i == 0) {
    const double* st = in.start + (size_t)slot * 4;
    g[0] = st[0]; g[1] = st[1]; g[2] = st[2]; g[3] = st[3]; g[4] = 0.0; g[5] = 0.0;
    s.lambda[slot] = 1.0;
    s.dlambda[slot] = 1.0;
    s.cost_old[slot] = 0.0;
    s.dcost[slot] = 0.0;
    s.iter[slot] = 0;
    s.status[slot] = 0;
    s.n_cost[slot] = 0;
    s.upd[slot] = 1;
    if (s.posn) s.posn[slot] = slot;      // the first active list is the identity
    s.acc_idx[slot] = -1;
    s.cur[slot] = 0;
    s.n_iter_trajs[slot] = 0;
    s.emit[slot] = 0;
    s.pid[slot] = slot;
    s.done_now[slot] = 0;
    s.act[slot] = slot;
    const double* c = in.coarse + (size_t)slot * K * 6;   // the tracker init guess follows the coarse trajectory itself
    s.coarse0[slot] = make_double2(c[0], c[1]);
    s.coarse0[(size_t)s.Bcap + slot] = make_double2(c[2], c[3]);
  } else {
    const double* c = in.coarse + ((size_t)slot * K + i) * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) g[e] = c[e];
  }
  if (in.station != nullptr) s.cstation[(size_t)i * s.Bcap + slot] = in.station[(size_t)slot * K + i];
  double2* o = s.goals + (size_t)i * 3 * s.Bcap + slot;
  o[0] = make_double2(g[0], g[1]);
  o[(size_t)s.Bcap] = make_double2(g[2], g[3]);
  o[(size_t)2 * s.Bcap] = make_double2(g[4], g[5]);
}

// Lane rows (a,b,c,sx,sy,ex,ey) -> shrunk/normalised plane + segment frame
// (cc:458-472, 484-494; LineSegment2d ctor line_segment2d.cpp:38-48).
__global__ void k_load_lanes(DeviceState s, const double* __restrict__ raw) {
  const int t = threadIdx.x;
  if (t >= s.nl + s.nr) return;
  const double* r = raw + t * 7;
  double a = r[0], b = r[1], c = r[2];
  c = c - s.p.shrink_lane * (a * a + b * b) / hypot_ref(a, b);
  const double nrm = hypot_ref(hypot_ref(a, b), c);
  const double sx = r[3], sy = r[4], ex = r[5], ey = r[6];
  const double dx = ex - sx, dy = ey - sy;
  const double len = hypot_ref(dx, dy);
  double* o = s.lanes + t * kLaneFields;
  o[0] = a / nrm; o[1] = b / nrm; o[2] = c / nrm;
  o[3] = sx; o[4] = sy;
  o[5] = (len <= kMathEps) ? 0.0 : dx / len;
  o[6] = (len <= kMathEps) ? 0.0 : dy / len;
  o[7] = len;
  o[8] = ex; o[9] = ey;
  // identities of the two end points within this side's table: the smallest index (start of segment k = 2k, its end =
  // 2k + 1) of a point with bitwise equal coordinates.  Consecutive segments of a polyline share a point, hence an id;
  // the exact-tie rule of the nearest-segment search compares these instead of coordinates (dev_model.hpp).
  const int first = (t < s.nl) ? 0 : s.nl, n_side = (t < s.nl) ? s.nl : s.nr;
  auto same = [](double ax, double ay, double bx, double by) {
    return __double_as_longlong(ax) == __double_as_longlong(bx) && __double_as_longlong(ay) == __double_as_longlong(by);
  };
  int sid = 2 * (t - first), eid = 2 * (t - first) + 1;
  for (int k = n_side - 1; k >= 0; --k) {           // downwards: the smallest matching index is written last
    const double* q = raw + (first + k) * 7;
    if (same(q[5], q[6], sx, sy)) sid = min(sid, 2 * k + 1);
    if (same(q[3], q[4], sx, sy)) sid = min(sid, 2 * k);
    if (same(q[5], q[6], ex, ey)) eid = min(eid, 2 * k + 1);
    if (same(q[3], q[4], ex, ey)) eid = min(eid, 2 * k);
  }
  o[10] = __longlong_as_double((long long)(unsigned)sid | ((long long)eid << 32));
}

// Candidate sets of the lane grid.  For a square with centre q and half-diagonal r, any point p of
// the square has d(q, s*) <= d(p, s*) + r <= d(p, s) + r <= d(q, s) + 2r for its nearest segment s*
// and every s, so {s : d(q, s) <= min_s d(q, s) + 2r (+ slack)} contains every possible answer
// (ties included).  Applied to the whole cell this bound keeps every segment within
// sqrt(2 d 2r) of the true Voronoi boundary -- about 2.5 candidates per cell a few metres from a
// lane.  So cells near the lanes are refined: the bound is applied to the four quarters of the
// square, recursively, and the cell's set is the union of its leaves' sets.  A square is a leaf
// when one candidate is left, when everything it could add is in the union already, or at
// kGridRefineLevels (12.5 cm squares for 1 m cells: at most 85 squares per cell, and deeper levels
// no longer shorten the longest list of a wave, which is what the kernels pay for).  Still
// conservative; ~1.75 candidates per cell instead of ~2.5, longest list of a wave 2 instead of 3.2.
// Indices are stored ascending so the scan order of the reference is kept.
constexpr int kGridRefineLevels = 3;
constexpr double kGridRefineRange = 12.0;   // cells farther than this from the lane keep the cell-level set

__global__ __launch_bounds__(256) void k_build_lane_grid(DeviceState s) {
  const int ncell = s.gnx * s.gny;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * ncell) return;
  const int side = t / ncell, cell = t - side * ncell;
  const int iy = cell / s.gnx, ix = cell - iy * s.gnx;
  const double h = 1.0 / s.ginv_h;
  const double qx = s.gx0 + (ix + 0.5) * h, qy = s.gy0 + (iy + 0.5) * h;
  const double* tab = s.lanes + (side ? s.nl * kLaneFields : 0);
  const int n = side ? s.nr : s.nl;
  double dmin = DBL_MAX;
  for (int k = 0; k < n; ++k) dmin = fmin(dmin, sqrt(segment_dist2(tab + k * kLaneFields, qx, qy)));
  const double thr = dmin + 1.4142135623730951 * h + 1e-6 * (1.0 + dmin);
  constexpr int kMaxList = kGridCellBytes - 1;
  unsigned char root[kMaxList];
  int cnt = 0;
  for (int k = 0; k < n; ++k) {
    if (sqrt(segment_dist2(tab + k * kLaneFields, qx, qy)) <= thr) {
      if (cnt < kMaxList) root[cnt] = (unsigned char)k;
      ++cnt;
    }
  }
  unsigned keep = (cnt <= kMaxList) ? ((1u << cnt) - 1u) : 0u;   // bit e: root[e] stays in the cell's list
  if (cnt > 1 && cnt <= kMaxList && dmin <= kGridRefineRange) {
    struct Node { double cx, cy; int level; unsigned mask; };
    Node stack[3 * kGridRefineLevels + 4];
    int sp = 0;
    unsigned got = 0u;
    stack[sp++] = Node{qx, qy, 0, keep};
    while (sp > 0) {
      const Node nd = stack[--sp];
      if ((got & nd.mask) == nd.mask) continue;          // nothing new can come from this square
      const double half = 0.5 * h / (double)(1 << nd.level);   // half side of this square
      double dm = DBL_MAX, dd[kMaxList];
      for (int e = 0; e < cnt; ++e) {
        dd[e] = (nd.mask >> e & 1u) ? sqrt(segment_dist2(tab + root[e] * kLaneFields, nd.cx, nd.cy)) : DBL_MAX;
        dm = fmin(dm, dd[e]);
      }
      const double th = dm + 2.0 * 1.4142135623730951 * half + 1e-6 * (1.0 + dm);
      unsigned m = 0u;
      for (int e = 0; e < cnt; ++e)
        if (dd[e] <= th) m |= 1u << e;
      if ((m & (m - 1u)) == 0u || nd.level == kGridRefineLevels) {
        got |= m;
        continue;
      }
      const double q = 0.5 * half;
      stack[sp++] = Node{nd.cx - q, nd.cy - q, nd.level + 1, m};
      stack[sp++] = Node{nd.cx + q, nd.cy - q, nd.level + 1, m};
      stack[sp++] = Node{nd.cx - q, nd.cy + q, nd.level + 1, m};
      stack[sp++] = Node{nd.cx + q, nd.cy + q, nd.level + 1, m};
    }
    keep = got;
  }
  unsigned char out[kGridCellBytes];
  for (int k = 0; k < kGridCellBytes; ++k) out[k] = 0;
  int kept = 0;
  if (cnt <= kMaxList)
    for (int e = 0; e < cnt; ++e)
      if (keep >> e & 1u) out[1 + kept++] = root[e];
  out[0] = (cnt <= kMaxList) ? (unsigned char)kept : (unsigned char)kGridFullScan;
  unsigned w[4] = {0, 0, 0, 0};
  for (int k = 0; k < kGridCellBytes; ++k) w[k >> 2] |= (unsigned)out[k] << ((k & 3) * 8);
  *reinterpret_cast<uint4*>(s.lgrid + ((size_t)side * ncell + cell) * kGridCellBytes) =
      make_uint4(w[0], w[1], w[2], w[3]);
}
void launch_build_lane_grid(const DeviceState& s, hipStream_t st) {
  const int n = 2 * s.gnx * s.gny;
  hipLaunchKernelGGL(k_build_lane_grid, dim3((n + 255) / 256), dim3(256), 0, st, s);
}

// test hook: nearest left/right segment of arbitrary points, through the grid or the full scan
__global__ void k_nearest_lane(DeviceState s, int n, const double* __restrict__ xy, int* __restrict__ left,
                               int* __restrict__ right, int use_grid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double px = xy[2 * t], py = xy[2 * t + 1];
  if (use_grid) {
    left[t] = s.exact_ties ? nearest_segment<true>(s, s.lanes, 0, px, py) : nearest_segment<false>(s, s.lanes, 0, px, py);
    right[t] = s.exact_ties ? nearest_segment<true>(s, s.lanes, 1, px, py) : nearest_segment<false>(s, s.lanes, 1, px, py);
  } else {
    left[t] = nearest_segment_scan(s.lanes, s.nl, px, py, s.exact_ties != 0);
    right[t] = nearest_segment_scan(s.lanes + s.nl * kLaneFields, s.nr, px, py, s.exact_ties != 0);
  }
}
void launch_nearest_lane(const DeviceState& s, int n, const double* xy, int* left, int* right, int use_grid,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_nearest_lane, dim3((n + 255) / 256), dim3(256), 0, st, s, n, xy, left, right, use_grid);
}

// test hook: the lean fp64 routines of dev_model.hpp on arbitrary inputs
__global__ void k_device_math(int fn, int n, const double* __restrict__ in, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double x = in[t];
  double sn, cs;
  lean_sincos(x, &sn, &cs);
  if (fn == 10) {   // NormalizeAngle as the rollouts take it: the straight-line form, the complete one if any lane of the wave needs it
    bool rare = false;
    double r = normalize_angle_common(x, rare);
    if (__builtin_amdgcn_ballot_w64(rare) != 0) {
      if (rare) r = normalize_angle(x);
    }
    out[t] = r;
    return;
  }
  out[t] = (fn == 0) ? log_pos(x, 0) : (fn == 1) ? fast_rcp(x)
         : (fn == 2) ? log_pos(__builtin_amdgcn_frexp_mant(x), __builtin_amdgcn_frexp_exp(x))
         : (fn == 3) ? sn : (fn == 4) ? cs : (fn == 5) ? lean_tan(x) : (fn == 6) ? normalize_angle(x)
         : (fn == 7) ? cos(x) : (fn == 8) ? sin(x) : hypot_ref(x, 1.0);
}
void launch_device_math(int fn, int n, const double* in, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_device_math, dim3((n + 255) / 256), dim3(256), 0, st, fn, n, in, out);
}

void launch_load(const DeviceState& s, int B, const ProblemView& in, const double* lanes_raw,
                 hipStream_t st) {
  // goals first: it resets the per-problem state that k_load_corridor may flag (status 6)
  const int n = B * s.p.K;
  hipLaunchKernelGGL(k_load_goals, dim3((n + 255) / 256), dim3(256), 0, st, s, B, in);
  dim3 g((B + 63) / 64, s.p.K);
  hipLaunchKernelGGL(k_load_corridor, g, dim3(256), 0, st, s, B, in);
  if (lanes_raw) {   // nullptr: the lane tables and their grid are those of the previous load
    hipLaunchKernelGGL(k_load_lanes, dim3(1), dim3(512), 0, st, s, lanes_raw);
    launch_build_lane_grid(s, st);
  }
}

// ---------------------------------------------------------------------------------------------
// Init guess: time-varying LQR along the goals + clamped closed-loop rollout (iqr, cc:793-842).
// One lane per problem: backward sweep stores K_i in the gains arena, forward sweep rolls out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_init_guess(DeviceState s, int B) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= B) return;
  const Params& p = s.p;
  const int N = p.N, Bc = s.Bcap;
  const double Qd[6] = {0.001, 0.001, 0.001, 0.001, 0.01, 0.005};  // cc:801-807
  const double R0 = 0.2, R1 = 0.05;                                // cc:811-813 (off-diagonals: 0)
  double P[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) P[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) P[e * 7] = Qd[e];
  const double dt = p.dt;
  const double zero_u[2] = {0.0, 0.0};
  for (int i = N - 1; i >= 0; --i) {
    double g[6];
    {
      const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
      const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
      g[0] = g0.x; g[1] = g0.y; g[2] = g1.x; g[3] = g1.y; g[4] = g2.x; g[5] = g2.y;
    }
    DynJac J;
    dynamics_jacobian(p, g, zero_u, J);
    double A[36], Bm[12];
#pragma unroll
    for (int e = 0; e < 36; ++e) A[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 12; ++e) Bm[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) A[e * 7] = 1.0;
    A[2] = J.a02; A[3] = J.a03; A[4] = J.a04; A[5] = J.a05;
    A[8] = J.a12; A[9] = J.a13; A[10] = J.a14; A[11] = J.a15;
    A[15] = J.a23; A[16] = J.a24; A[17] = J.a25;
    A[22] = dt;
    Bm[5] = J.b21; Bm[6] = 0.5 * dt * dt; Bm[8] = dt; Bm[11] = dt;
    // BtP = B^T P (2x6): an X^T Y product, terms added as sum6_xty says (dev_model.hpp)
    double BtP[12];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        BtP[r * 6 + c] = sum6_xty(Bm[0 * 2 + r] * P[0 * 6 + c], Bm[1 * 2 + r] * P[1 * 6 + c], Bm[2 * 2 + r] * P[2 * 6 + c],
                                  Bm[3 * 2 + r] * P[3 * 6 + c], Bm[4 * 2 + r] * P[4 * 6 + c], Bm[5 * 2 + r] * P[5 * 6 + c]);
    double M[4], BtPA[12];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        double acc = BtP[r * 6 + 0] * Bm[0 * 2 + c];
#pragma unroll
        for (int k = 1; k < 6; ++k) acc += BtP[r * 6 + k] * Bm[k * 2 + c];
        M[r * 2 + c] = acc;
      }
    M[0] = R0 + M[0]; M[1] = 0.0 + M[1]; M[2] = 0.0 + M[2]; M[3] = R1 + M[3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double acc = BtP[r * 6 + 0] * A[0 * 6 + c];
#pragma unroll
        for (int k = 1; k < 6; ++k) acc += BtP[r * 6 + k] * A[k * 6 + c];
        BtPA[r * 6 + c] = acc;
      }
    const double invdet = 1.0 / (M[0] * M[3] - M[2] * M[1]);
    const double inv[4] = {M[3] * invdet, -M[1] * invdet, -M[2] * invdet, M[0] * invdet};
    double Kg[12];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) Kg[r * 6 + c] = inv[r * 2 + 0] * BtPA[c] + inv[r * 2 + 1] * BtPA[6 + c];
    {
      double2* gp = s.gains + (size_t)i * kGainPairs * Bc + slot;
#pragma unroll
      for (int q = 0; q < 6; ++q) gp[(size_t)q * Bc] = make_double2(Kg[2 * q], Kg[2 * q + 1]);
    }
    // P = Q + (A^T P)(A - B K)     cc:823
    double AmBK[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        AmBK[r * 6 + c] = A[r * 6 + c] - (Bm[r * 2 + 0] * Kg[c] + Bm[r * 2 + 1] * Kg[6 + c]);
    double AtP[36];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)   // A^T P: X^T Y again
        AtP[r * 6 + c] = sum6_xty(A[0 * 6 + r] * P[0 * 6 + c], A[1 * 6 + r] * P[1 * 6 + c], A[2 * 6 + r] * P[2 * 6 + c],
                                  A[3 * 6 + r] * P[3 * 6 + c], A[4 * 6 + r] * P[4 * 6 + c], A[5 * 6 + r] * P[5 * 6 + c]);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double acc = AtP[r * 6 + 0] * AmBK[0 * 6 + c];
#pragma unroll
        for (int k = 1; k < 6; ++k) acc += AtP[r * 6 + k] * AmBK[k * 6 + c];
        P[r * 6 + c] = ((r == c) ? Qd[r] : 0.0) + acc;
      }
  }
  // closed-loop rollout with clamped controls   cc:830-841
  double x[6];
  {
    const double2* gp = s.goals + slot;
    const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
    x[0] = g0.x; x[1] = g0.y; x[2] = g1.x; x[3] = g1.y; x[4] = g2.x; x[5] = g2.y;
  }
  store_x(s, 0, 0, slot, x);
  for (int i = 0; i < N; ++i) {
    const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
    const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
    const double dx[6] = {x[0] - g0.x, x[1] - g0.y, x[2] - g1.x, x[3] - g1.y, x[4] - g2.x, x[5] - g2.y};
    const double2* kp = s.gains + (size_t)i * kGainPairs * Bc + slot;
    double u[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double2 k0 = kp[(size_t)(r * 3 + 0) * Bc], k1 = kp[(size_t)(r * 3 + 1) * Bc],
                    k2 = kp[(size_t)(r * 3 + 2) * Bc];
      double acc = (-k0.x) * dx[0];
      acc += (-k0.y) * dx[1];
      acc += (-k1.x) * dx[2];
      acc += (-k1.y) * dx[3];
      acc += (-k2.x) * dx[4];
      acc += (-k2.y) * dx[5];
      u[r] = acc;
    }
    u[0] = fmin(p.jerk_max, fmax(u[0], p.jerk_min));
    u[1] = fmin(p.delta_rate_max, fmax(u[1], p.delta_rate_min));
    store_u(s, 0, i, slot, u);
    dynamics(p, x, u, x);
    store_x(s, 0, i + 1, slot, x);
  }
}

// ---------------------------------------------------------------------------------------------
// The same init guess with ONE WAVEFRONT per problem, for small batches (the reference's own call is a batch of one,
// trajectory_planner.cpp:79-89): with one lane per problem a step of the LQR sweep is ~1300 fp64 instructions in a row
// on one lane (2.2 us; 0.185 ms of a 1.7 ms Plan were this kernel).  Here the Jacobians at the goals -- which do not
// depend on the sweep -- are evaluated for all steps side by side first, and every lane owns one output element of
// every stage of the sweep: B^T P | A^T P, then M | (B^T P) A, then K, then A - B K, then the new P; each element is the
// SAME dense 6-term dot product, in the same order (sum6_xty for A^T P and B^T P, index order for the rest), that k_init_guess
// writes out (the exact zeros and ones of A and B
// included), so the two kernels agree bit for bit (tests/test_gpu_parity.py).  The clamped closed-loop rollout stays
// one dependent chain (lane 0), with the gains read back from LDS instead of global memory.
// LDS: jac [N][12] | A 36 | B 12 | P 36 | BtP 12 | AtP 36 | M 4 | BtPA 12 | K 12 | AmBK 36 | Kall [N][12]
// ---------------------------------------------------------------------------------------------
constexpr int kIgFixed = 36 + 12 + 36 + 12 + 36 + 4 + 12 + 12 + 36;
__global__ __launch_bounds__(64) void k_init_guess_wave(DeviceState s, int B) {
  extern __shared__ double ig_lds[];
  // eight consecutive slots -- one 128-byte line of every row -- per XCD (a handful of problems: launched as they are)
  const int slot = (gridDim.x & 63u) == 0 ? xcd_local_position((int)blockIdx.x) : (int)blockIdx.x;
  if (slot >= B) return;
  const int lane = threadIdx.x;
  const Params& p = s.p;
  const int N = p.N, Bc = s.Bcap;
  const double dt = p.dt;
  const WaveSync sync{};
  double* jac = ig_lds;
  double* A = jac + (size_t)N * 12;
  double* Bm = A + 36;
  double* P = Bm + 12;
  double* BtP = P + 36;
  double* AtP = BtP + 12;
  double* M = AtP + 36;
  double* BtPA = M + 4;
  double* Kg = BtPA + 12;
  double* AmBK = Kg + 12;
  double* Kall = AmBK + 36;
  const double Qd[6] = {0.001, 0.001, 0.001, 0.001, 0.01, 0.005};  // cc:801-807
  const double R0 = 0.2, R1 = 0.05;                                // cc:811-813 (off-diagonals: 0)
  const double zero_u[2] = {0.0, 0.0};
  // ---- all Jacobians at the goals (vm:21-86), a step per lane ----
  for (int i = lane; i < N; i += 64) {
    double g[6];
    const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
    const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
    g[0] = g0.x; g[1] = g0.y; g[2] = g1.x; g[3] = g1.y; g[4] = g2.x; g[5] = g2.y;
    DynJac J;
    dynamics_jacobian(p, g, zero_u, J);
    double* o = jac + (size_t)i * 12;
    o[0] = J.a02; o[1] = J.a03; o[2] = J.a04; o[3] = J.a05; o[4] = J.a12; o[5] = J.a13; o[6] = J.a14; o[7] = J.a15;
    o[8] = J.a23; o[9] = J.a24; o[10] = J.a25; o[11] = J.b21;
  }
  if (lane < 36) P[lane] = (lane / 6 == lane % 6) ? Qd[lane / 6] : 0.0;
  // where a lane's entry of the dense A / B comes from: the Jacobian record (>= 0), or a constant
  int a_src = -1;
  double a_const = 0.0;
  if (lane < 36) {
    const int r = lane / 6, c = lane % 6;
    if (r == c) a_const = 1.0;
    else if (r == 0 && c >= 2) a_src = c - 2;
    else if (r == 1 && c >= 2) a_src = 4 + c - 2;
    else if (r == 2 && c >= 3) a_src = 8 + c - 3;
    else if (r == 3 && c == 4) a_const = dt;
  } else if (lane < 48) {
    const int e = lane - 36;   // B[k][q] at k * 2 + q
    if (e == 2 * 2 + 1) a_src = 11;
    else if (e == 3 * 2 + 0) a_const = 0.5 * dt * dt;
    else if (e == 4 * 2 + 0 || e == 5 * 2 + 1) a_const = dt;
  }
  // stage roles
  const int r6 = (lane < 36) ? lane / 6 : 0, c6 = (lane < 36) ? lane % 6 : 0;
  const bool s2_btp = lane >= 36 && lane < 48;           // B^T P (q, c)
  const int q2 = s2_btp ? (lane - 36) / 6 : 0, c2 = s2_btp ? (lane - 36) % 6 : 0;
  const bool s3_m = lane < 4, s3_bpa = lane >= 4 && lane < 16;
  const int m_r = lane >> 1, m_c = lane & 1, b_r = s3_bpa ? (lane - 4) / 6 : 0, b_c = s3_bpa ? (lane - 4) % 6 : 0;
  const int k_r = (lane < 12) ? lane / 6 : 0, k_c = (lane < 12) ? lane % 6 : 0;
  sync();
  for (int i = N - 1; i >= 0; --i) {
    // ---- dense A, B of this step ----
    if (lane < 36) A[lane] = (a_src >= 0) ? jac[(size_t)i * 12 + a_src] : a_const;
    else if (lane < 48) Bm[lane - 36] = (a_src >= 0) ? jac[(size_t)i * 12 + a_src] : a_const;
    sync();
    // ---- A^T P (36 lanes) | B^T P (12 lanes) ----
    if (lane < 36) {
      AtP[r6 * 6 + c6] = sum6_xty(A[0 * 6 + r6] * P[0 * 6 + c6], A[1 * 6 + r6] * P[1 * 6 + c6], A[2 * 6 + r6] * P[2 * 6 + c6],
                                  A[3 * 6 + r6] * P[3 * 6 + c6], A[4 * 6 + r6] * P[4 * 6 + c6], A[5 * 6 + r6] * P[5 * 6 + c6]);
    } else if (s2_btp) {
      BtP[q2 * 6 + c2] = sum6_xty(Bm[0 * 2 + q2] * P[0 * 6 + c2], Bm[1 * 2 + q2] * P[1 * 6 + c2], Bm[2 * 2 + q2] * P[2 * 6 + c2],
                                  Bm[3 * 2 + q2] * P[3 * 6 + c2], Bm[4 * 2 + q2] * P[4 * 6 + c2], Bm[5 * 2 + q2] * P[5 * 6 + c2]);
    }
    sync();
    // ---- M = R + (B^T P) B (4 lanes) | (B^T P) A (12 lanes) ----
    if (s3_m) {
      double acc = BtP[m_r * 6 + 0] * Bm[0 * 2 + m_c];
#pragma unroll
      for (int k = 1; k < 6; ++k) acc += BtP[m_r * 6 + k] * Bm[k * 2 + m_c];
      M[lane] = ((lane == 0) ? R0 : (lane == 3) ? R1 : 0.0) + acc;
    } else if (s3_bpa) {
      double acc = BtP[b_r * 6 + 0] * A[0 * 6 + b_c];
#pragma unroll
      for (int k = 1; k < 6; ++k) acc += BtP[b_r * 6 + k] * A[k * 6 + b_c];
      BtPA[b_r * 6 + b_c] = acc;
    }
    sync();
    // ---- K = M^-1 (B^T P A), closed-form inverse (cc:822) ----
    if (lane < 12) {
      const double invdet = 1.0 / (M[0] * M[3] - M[2] * M[1]);
      const double inv[4] = {M[3] * invdet, -M[1] * invdet, -M[2] * invdet, M[0] * invdet};
      const double kv = inv[k_r * 2 + 0] * BtPA[k_c] + inv[k_r * 2 + 1] * BtPA[6 + k_c];
      Kg[lane] = kv;
      Kall[(size_t)i * 12 + lane] = kv;
    }
    sync();
    if (lane < 6) s.gains[((size_t)i * kGainPairs + lane) * Bc + slot] = make_double2(Kg[2 * lane], Kg[2 * lane + 1]);
    // ---- A - B K ----
    if (lane < 36) AmBK[lane] = A[lane] - (Bm[r6 * 2 + 0] * Kg[c6] + Bm[r6 * 2 + 1] * Kg[6 + c6]);
    sync();
    // ---- P = Q + (A^T P)(A - B K)     cc:823 ----
    if (lane < 36) {
      double acc = AtP[r6 * 6 + 0] * AmBK[0 * 6 + c6];
#pragma unroll
      for (int k = 1; k < 6; ++k) acc += AtP[r6 * 6 + k] * AmBK[k * 6 + c6];
      P[lane] = ((r6 == c6) ? Qd[r6] : 0.0) + acc;
    }
    sync();
  }
  if (lane != 0) return;
  // closed-loop rollout with clamped controls   cc:830-841
  double x[6];
  {
    const double2* gp = s.goals + slot;
    const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
    x[0] = g0.x; x[1] = g0.y; x[2] = g1.x; x[3] = g1.y; x[4] = g2.x; x[5] = g2.y;
  }
  store_x(s, 0, 0, slot, x);
  for (int i = 0; i < N; ++i) {
    const double2* gp = s.goals + (size_t)i * 3 * Bc + slot;
    const double2 g0 = gp[0], g1 = gp[(size_t)Bc], g2 = gp[(size_t)2 * Bc];
    const double dx[6] = {x[0] - g0.x, x[1] - g0.y, x[2] - g1.x, x[3] - g1.y, x[4] - g2.x, x[5] - g2.y};
    const double* kp = Kall + (size_t)i * 12;
    double u[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double acc = (-kp[r * 6 + 0]) * dx[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) acc += (-kp[r * 6 + k]) * dx[k];
      u[r] = acc;
    }
    u[0] = fmin(p.jerk_max, fmax(u[0], p.jerk_min));
    u[1] = fmin(p.delta_rate_max, fmax(u[1], p.delta_rate_min));
    store_u(s, 0, i, slot, u);
    dynamics(p, x, u, x);
    store_x(s, 0, i + 1, slot, x);
  }
}

// Batches up to this size give every problem a wavefront (4096 problems are 4 waves per SIMD of ~0.1 ms each; the one-lane
// kernel needs 0.185 ms whatever the batch, and wins from there on).  CILQR_INIT_GUESS_WAVE=0 / 1: never / always (tests).
constexpr int kInitGuessWaveMax = 4096;
void launch_init_guess(const DeviceState& s, int B, hipStream_t st) {
  static const int forced = [] {
    const char* e = std::getenv("CILQR_INIT_GUESS_WAVE");
    return e ? std::atoi(e) : -1;
  }();
  const size_t lds = ((size_t)s.p.N * 24 + kIgFixed) * sizeof(double);
  const bool wave = forced >= 0 ? forced != 0 : B <= kInitGuessWaveMax;
  if (wave && lds <= 64 * 1024)
    hipLaunchKernelGGL(k_init_guess_wave, dim3(B >= 64 ? (B + 63) / 64 * 64 : B), dim3(64), lds, st, s, B);
  else
    hipLaunchKernelGGL(k_init_guess, dim3((B + 63) / 64), dim3(64), 0, st, s, B);
}

// ---------------------------------------------------------------------------------------------
// set / gather the iterate (stage API)
// ---------------------------------------------------------------------------------------------
__global__ void k_set_trajectory(DeviceState s, int B, const double* __restrict__ X,
                                 const double* __restrict__ U) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = s.p.K, N = s.p.N;
  if (t >= B * K) return;
  const int slot = t / K, i = t - slot * K;
  const int buf = s.cur[slot];
  store_x(s, buf, i, slot, X + ((size_t)slot * K + i) * 6);
  if (i < N) store_u(s, buf, i, slot, U + ((size_t)slot * N + i) * 2);
  if (i == 0) s.upd[slot] = 1;
}
void launch_set_trajectory(const DeviceState& s, int B, const double* X, const double* U, hipStream_t st) {
  const int n = B * s.p.K;
  hipLaunchKernelGGL(k_set_trajectory, dim3((n + 255) / 256), dim3(256), 0, st, s, B, X, U);
}

__global__ void k_gather_xu(DeviceState s, int B, int cand, double* __restrict__ X, double* __restrict__ U) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = s.p.K, N = s.p.N;
  if (t >= B * K) return;
  const int slot = t / K, i = t - slot * K;
  const int buf = s.cur[slot] ^ cand;
  double x[6], u[2];
  load_x(s, buf, i, slot, x);
  if (X) {
#pragma unroll
    for (int e = 0; e < 6; ++e) X[((size_t)slot * K + i) * 6 + e] = x[e];
  }
  if (U && i < N) {
    load_u(s, buf, i, slot, u);
    U[((size_t)slot * N + i) * 2 + 0] = u[0];
    U[((size_t)slot * N + i) * 2 + 1] = u[1];
  }
}
void launch_gather_xu(const DeviceState& s, int B, int cand, double* X, double* U, hipStream_t st) {
  const int n = B * s.p.K;
  hipLaunchKernelGGL(k_gather_xu, dim3((n + 255) / 256), dim3(256), 0, st, s, B, cand, X, U);
}

// dst[b*dst_stride + dst_off + 2*r + {0,1}] = src[r][b].{x,y}
__global__ void k_gather_pairs(const double2* __restrict__ src, int rows, int Bcap, int B,
                               double* __restrict__ dst, int dst_stride, int dst_off) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * rows) return;
  const int b = t / rows, r = t - b * rows;
  const double2 v = src[(size_t)r * Bcap + b];
  dst[(size_t)b * dst_stride + dst_off + 2 * r] = v.x;
  dst[(size_t)b * dst_stride + dst_off + 2 * r + 1] = v.y;
}
void launch_gather_pairs(const double2* src, int rows_pairs, int Bcap, int B, double* dst,
                         int dst_stride, int dst_off, hipStream_t st) {
  const int n = B * rows_pairs;
  hipLaunchKernelGGL(k_gather_pairs, dim3((n + 255) / 256), dim3(256), 0, st, src, rows_pairs, Bcap, B,
                     dst, dst_stride, dst_off);
}
__global__ void k_gather_scalar(const double* __restrict__ src, int rows, int Bcap, int B,
                                double* __restrict__ dst, int dst_stride, int dst_off) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * rows) return;
  const int b = t / rows, r = t - b * rows;
  dst[(size_t)b * dst_stride + dst_off + r] = src[(size_t)r * Bcap + b];
}
void launch_gather_scalar(const double* src, int rows, int Bcap, int B, double* dst, int dst_stride,
                          int dst_off, hipStream_t st) {
  const int n = B * rows;
  hipLaunchKernelGGL(k_gather_scalar, dim3((n + 255) / 256), dim3(256), 0, st, src, rows, Bcap, B, dst,
                     dst_stride, dst_off);
}

// ---------------------------------------------------------------------------------------------
// Export: TransformToTrajectory (cc:771-791) into problem-major [B][K][10]
// ---------------------------------------------------------------------------------------------
// final trajectory (cc:238,285,303,319) of every slot that terminated in the last k_update
__global__ void k_export_done(DeviceState s, int n, double* __restrict__ traj) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = s.p.K;
  if (t >= n * K) return;
  const int i = t / n, j = t - i * n;
  if (j >= active_count(s, n)) return;
  const int slot = s.act[j];
  if (!s.done_now[slot]) return;
  write_traj_point(s, s.cur[slot], i, slot, traj + ((size_t)s.pid[slot] * K + i) * 10);
}
void launch_export_done(const DeviceState& s, int n_act, double* traj, hipStream_t st) {
  const int n = n_act * s.p.K;
  if (n == 0) return;
  hipLaunchKernelGGL(k_export_done, dim3((n + 255) / 256), dim3(256), 0, st, s, n_act, traj);
}

// Re-pack the surviving problems: entry j of src's NEXT active list moves to slot j of dst.
// Moved: the current iterate (into buffer 0), goals, corridor planes + counts, the scalar state.
// Not moved: the linearisation / gains (recomputed: upd is forced to 1, which reproduces the
// same values because they depend only on the iterate), and everything indexed by problem.
__global__ __launch_bounds__(256) void k_compact(DeviceState a, DeviceState b, int n_max) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= min(*a.n_next, n_max)) return;   // survivors counted by k_update
  const int i = blockIdx.y;   // knot
  const int K = a.p.K, N = a.p.N;
  const int src = a.act_next[j];
  const int buf = a.cur[src];
  {
    const double2* x = a.X + ((size_t)buf * K + i) * 3 * a.Bcap + src;
    double2* o = b.X + ((size_t)i) * 3 * b.Bcap + j;
    o[0] = x[0];
    o[(size_t)b.Bcap] = x[(size_t)a.Bcap];
    o[(size_t)2 * b.Bcap] = x[(size_t)2 * a.Bcap];
    if (i < N) b.U[(size_t)i * b.Bcap + j] = a.U[((size_t)buf * N + i) * a.Bcap + src];
    const double2* g = a.goals + (size_t)i * 3 * a.Bcap + src;
    double2* go = b.goals + (size_t)i * 3 * b.Bcap + j;
    go[0] = g[0];
    go[(size_t)b.Bcap] = g[(size_t)a.Bcap];
    go[(size_t)2 * b.Bcap] = g[(size_t)2 * a.Bcap];
    const int cnt = a.ccnt[(size_t)i * a.Bcap + src];
    b.ccnt[(size_t)i * b.Bcap + j] = cnt;
    const double* c = a.cor + (size_t)i * a.cmax * 3 * a.Bcap + src;
    double* co = b.cor + (size_t)i * b.cmax * 3 * b.Bcap + j;
    for (int e = 0; e < cnt * 3; ++e) co[(size_t)e * b.Bcap] = c[(size_t)e * a.Bcap];
  }
  if (i == 0) {
    b.cur[j] = 0;
    b.pid[j] = a.pid[src];
    b.lambda[j] = a.lambda[src];
    b.dlambda[j] = a.dlambda[src];
    b.cost_old[j] = a.cost_old[src];
    b.dcost[j] = a.dcost[src];
    b.upd[j] = 1;
    b.acc_idx[j] = -1;
    b.emit[j] = 0;
    b.done_now[j] = 0;
    b.act[j] = j;
    if (b.posn) b.posn[j] = j;
  }
}
void launch_compact(const DeviceState& src, const DeviceState& dst, int n_max, hipStream_t st) {
  if (n_max == 0) return;
  dim3 g((n_max + 255) / 256, src.p.K);
  hipLaunchKernelGGL(k_compact, g, dim3(256), 0, st, src, dst, n_max);
}

// iter_trajs: append the current iterate of every listed slot whose emit flag is set
__global__ void k_export_iter_traj(DeviceState s, const int* __restrict__ list, int n,
                                   double* __restrict__ out, int cap) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = s.p.K;
  if (t >= n * K) return;
  const int i = t / n, j = t - i * n;
  if (list && j >= active_count(s, n)) return;
  const int slot = list ? list[j] : j;
  if (!s.emit[slot]) return;
  const int pb = s.pid[slot];
  const int idx = s.n_iter_trajs[pb] - 1;  // already counted by the kernel that set emit
  if (idx >= cap) return;
  write_traj_point(s, s.cur[slot], i, slot, out + (((size_t)pb * cap + idx) * K + i) * 10);
}
void launch_export_iter_traj(const DeviceState& s, const int* list, int n, double* iter_trajs,
                             int max_iter_trajs, hipStream_t st) {
  const int tot = n * s.p.K;
  if (tot == 0) return;
  hipLaunchKernelGGL(k_export_iter_traj, dim3((tot + 255) / 256), dim3(256), 0, st, s, list, n,
                     iter_trajs, max_iter_trajs);
}

// cost history + counters, problem-major.  Rows >= n_cost are left untouched.
__global__ void k_export_hist(DeviceState s, int B, double* __restrict__ hist, int* __restrict__ n_cost,
                              int* __restrict__ status, int* __restrict__ n_iter,
                              int* __restrict__ n_iter_trajs, signed char* __restrict__ alpha_trace) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= B) return;
  const int nc = s.n_cost[slot];
  if (n_cost) n_cost[slot] = nc;
  if (status) status[slot] = s.status[slot];
  if (n_iter) n_iter[slot] = s.iter[slot];
  if (n_iter_trajs) n_iter_trajs[slot] = s.n_iter_trajs[slot];
  if (hist) {
    double* o = hist + (size_t)slot * (s.p.max_iter + 1) * 5;
    for (int r = 0; r < nc; ++r)
#pragma unroll
      for (int c = 0; c < 5; ++c) o[r * 5 + c] = s.hist[((size_t)r * 5 + c) * s.Pcap + slot];
  }
  if (alpha_trace) {   // [B][max_iter]; iterations that never ran: -3
    signed char* o = alpha_trace + (size_t)slot * s.p.max_iter;
    const int ni = s.iter[slot];
    for (int r = 0; r < s.p.max_iter; ++r) o[r] = (r < ni) ? s.atrace[(size_t)r * s.Pcap + slot] : (signed char)-3;
  }
}
void launch_export_hist(const DeviceState& s, int B, double* cost_hist, int* n_cost, int* status,
                        int* n_iter, int* n_iter_trajs, signed char* alpha_trace, hipStream_t st) {
  hipLaunchKernelGGL(k_export_hist, dim3((B + 255) / 256), dim3(256), 0, st, s, B, cost_hist, n_cost,
                     status, n_iter, n_iter_trajs, alpha_trace);
}

// ---------------------------------------------------------------------------------------------
// Ragged export of the Cost history (large batches in host memory): offsets, then the live rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_hist_row_offsets(DeviceState s, int B, long long* __restrict__ off) {
  __shared__ long long part[1024];
  const int t = threadIdx.x, nt = blockDim.x;
  const int per = (B + nt - 1) / nt;
  const int lo = min(B, t * per), hi = min(B, lo + per);
  long long sum = 0;
  for (int b = lo; b < hi; ++b) sum += s.n_cost[b];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    long long run = 0;
    for (int i = 0; i < nt; ++i) {
      const long long v = part[i];
      part[i] = run;
      run += v;
    }
    off[B] = run;
  }
  __syncthreads();
  long long run = part[t];
  for (int b = lo; b < hi; ++b) {
    off[b] = run;
    run += s.n_cost[b];
  }
}
// a thread per (problem, row slot): the threads of a wave read one row of 64 neighbouring problems (contiguous in the
// batch-fastest history) and write it where the problem's rows start
__global__ void k_export_hist_rows(DeviceState s, int B, const long long* __restrict__ off, double* __restrict__ rows) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= B) return;
  const int nc = s.n_cost[slot];
  double* o = rows + off[slot] * 5;
  for (int r = 0; r < nc; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) o[r * 5 + c] = s.hist[((size_t)r * 5 + c) * s.Pcap + slot];
}
void launch_export_hist_rows(const DeviceState& s, int B, long long* off, double* rows, hipStream_t st) {
  hipLaunchKernelGGL(k_hist_row_offsets, dim3(1), dim3(1024), 0, st, s, B, off);
  hipLaunchKernelGGL(k_export_hist_rows, dim3((B + 255) / 256), dim3(256), 0, st, s, B, off, rows);
}

// ---------------------------------------------------------------------------------------------
// Open-loop rollout: x_{k+1} = f(x_k, u_k) for B independent (x0, U) pairs, problem-major I/O.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rollout(Params p, int B, const double* __restrict__ x0,
                                                const double* __restrict__ U, double* __restrict__ X) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double x[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) x[e] = x0[(size_t)b * 6 + e];
  double* xo = X + (size_t)b * p.K * 6;
#pragma unroll
  for (int e = 0; e < 6; ++e) xo[e] = x[e];
  for (int i = 0; i < p.N; ++i) {
    const double u[2] = {U[((size_t)b * p.N + i) * 2], U[((size_t)b * p.N + i) * 2 + 1]};
    dynamics(p, x, u, x);
#pragma unroll
    for (int e = 0; e < 6; ++e) xo[(i + 1) * 6 + e] = x[e];
  }
}
void launch_rollout(const Params& p, int B, const double* x0, const double* U, double* X, hipStream_t st) {
  hipLaunchKernelGGL(k_rollout, dim3((B + 63) / 64), dim3(64), 0, st, p, B, x0, U, X);
}

}  // namespace cilqr

namespace cilqr {

// ---------------------------------------------------------------------------------------------
// stage_read: expand the compact linearisation / gains into dense problem-major tensors
// ---------------------------------------------------------------------------------------------
__global__ void k_expand(DeviceState s, int B, int tensor, double* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = s.p.K, N = s.p.N, Bc = s.Bcap;
  if (t >= B * K) return;
  const int slot = t / K, i = t - slot * K;
  const double dt = s.p.dt;
  const bool term = (i == N);
  double2 w[kLinPairs];
  if (!term) {
    const double2* q = s.lin + (size_t)i * kLinPairs * Bc + scratch_index(s, slot);
    for (int r = 0; r < kLinPairs; ++r) w[r] = q[(size_t)r * Bc];
  } else {
    const double2* q = s.term + scratch_index(s, slot);
    for (int r = 0; r < 3; ++r) w[kRowLx + r] = q[(size_t)r * Bc];
    for (int r = 0; r < 6; ++r) w[kRowH + r] = q[(size_t)(3 + r) * Bc];
  }
  switch (tensor) {
    case 7: {  // A [B][N][6][6]
      if (term) return;
      double* o = dst + ((size_t)slot * N + i) * 36;
      for (int e = 0; e < 36; ++e) o[e] = 0.0;
      for (int e = 0; e < 6; ++e) o[e * 7] = 1.0;
      o[2] = w[0].x; o[3] = w[0].y; o[4] = w[1].x; o[5] = w[1].y;
      o[8] = w[2].x; o[9] = w[2].y; o[10] = w[3].x; o[11] = w[3].y;
      o[15] = w[4].x; o[16] = w[4].y; o[17] = w[5].x;
      o[22] = dt;
      break;
    }
    case 8: {  // B [B][N][6][2]
      if (term) return;
      double* o = dst + ((size_t)slot * N + i) * 12;
      for (int e = 0; e < 12; ++e) o[e] = 0.0;
      o[5] = w[5].y; o[6] = 0.5 * dt * dt; o[8] = dt; o[11] = dt;
      break;
    }
    case 9: {  // lx [B][K][6]
      double* o = dst + ((size_t)slot * K + i) * 6;
      o[0] = w[6].x; o[1] = w[6].y; o[2] = w[7].x; o[3] = w[7].y; o[4] = w[8].x; o[5] = w[8].y;
      break;
    }
    case 10: {  // lu [B][N][2]
      if (term) return;
      double* o = dst + ((size_t)slot * N + i) * 2;
      o[0] = w[9].x; o[1] = w[9].y;
      break;
    }
    case 11: {  // lxx [B][K][6][6]
      double* o = dst + ((size_t)slot * K + i) * 36;
      for (int e = 0; e < 36; ++e) o[e] = 0.0;
      o[0] = w[10].x; o[1] = w[10].y; o[2] = w[11].x;
      o[6] = w[11].y; o[7] = w[12].x; o[8] = w[12].y;
      o[12] = w[13].x; o[13] = w[13].y; o[14] = w[14].x;
      o[21] = w[14].y; o[28] = w[15].x; o[35] = w[15].y;
      break;
    }
    case 12: {  // luu [B][N][2][2]
      if (term) return;
      double* o = dst + ((size_t)slot * N + i) * 4;
      o[0] = w[16].x; o[1] = 0.0; o[2] = 0.0; o[3] = w[16].y;
      break;
    }
    case 13: {  // K [B][N][2][6]
      if (term) return;
      const double2* g = s.gains + (size_t)i * kGainPairs * Bc + scratch_index(s, slot);
      double* o = dst + ((size_t)slot * N + i) * 12;
      for (int r = 0; r < 6; ++r) {
        const double2 v = g[(size_t)r * Bc];
        o[2 * r] = v.x; o[2 * r + 1] = v.y;
      }
      break;
    }
    case 14: {  // k [B][N][2]
      if (term) return;
      const double2 v = s.gains[((size_t)i * kGainPairs + 6) * Bc + scratch_index(s, slot)];
      double* o = dst + ((size_t)slot * N + i) * 2;
      o[0] = v.x; o[1] = v.y;
      break;
    }
    default: break;
  }
}
void launch_expand(const DeviceState& s, int B, int tensor, double* dst, hipStream_t st) {
  const int n = B * s.p.K;
  hipLaunchKernelGGL(k_expand, dim3((n + 255) / 256), dim3(256), 0, st, s, B, tensor, dst);
}

}  // namespace cilqr
