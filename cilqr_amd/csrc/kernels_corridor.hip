// Safe-corridor construction for a batch of (problem, knot) pairs: the producer of the per-knot
// half-planes the CILQR cost consumes.  Behaviour follows (not copied from)
//   Corridor::BuildCorridorConstraints  algorithm/ilqr/corridor.cc:58-87   (the loop over knots)
//   Corridor::AddCorridorPoints         corridor.cc:89-120
//   Corridor::BuildCorridor             corridor.cc:122-263
// with cv::convexHull (OpenCV, not part of the reference tree) replaced by a monotone-chain hull
// on float32 points that drops collinear points, as OpenCV's Sklansky scan does.  The reference's
// mixed float32 / float64 arithmetic is kept statement by statement (the hulls run on
// cv::Point2f), and so are its quirks: safe_radius = norm of the LAST point inside `radius`
// (cc:169-171); (OriginIndex - 1) % size in unsigned 64-bit arithmetic (cc:203).
//
// Mapping: every knot of every problem is independent, so one lane builds one corridor (64 corridors
// share one instruction stream; spreading a corridor over a wave would cost ~10x the instructions,
// the hull scans being sequential).  A lane's working set (<= MAXP points, three small hulls) is indexed with
// data-dependent indices.  What decides the kernel's time is the number of DEPENDENT trips to memory a wave makes and
// how many cache lines a trip touches (round 6 counters: a wave waits on s_waitcnt for 87 % of its life; an access with a
// per-lane index into the private segment touches 64 lines, and the L1 of a CU holds a fraction of one wave's arrays):
//   * the point arrays (flip / vd / dual, 8 B x 57) stay in the private segment (scratch memory; built with
//     -disable-promote-alloca-to-vector, see the Makefile): they are read with per-lane indices only where a chain pops
//     past its LDS window and where a stage gathers hull vertices;
//   * the rank sorts run on keys in registers (no load in the n^2 loop, NaN test on the key loads);
//   * the monotone chains keep their two topmost points in registers and the next WIN levels in an LDS ring, so a pop is an
//     LDS read; the next point of a chain is requested a step ahead;
//   * the small index arrays (sorted order, chain stack = hull, second hull's copy) are LDS rows [index][lane];
//   * occupancy is whatever the registers allow (the LDS of a wave is 11.4 KB: 14 fit a CU, 12 run at 159 VGPRs).
// 65536 x 51 knots (28.5 obstacle points + 8 box points each): 23.4 ms in round 5, 15.7 with the register sort, 6.4 now.
// The kernel is a once-per-solve prologue (3.3 M corridors for 65536 x 51 knots).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "state.hpp"

namespace cilqr {

namespace {

constexpr double kEps = 1e-10;  // algorithm/math/vec2d.h:33

struct P2f {
  float x, y;
};
typedef __attribute__((address_space(3))) unsigned long long lds_float2;
typedef __attribute__((address_space(3))) unsigned char lds_u8;   // a pointer to it is an LDS address: ds_read / ds_write, never a flat access

// A lane's small index arrays (the sorted order, the chain's stack) in one of two homes.  In the lane's private segment an
// access is a trip to scratch memory, and a write with a per-lane index scatters over 64 cache lines.  As rows of LDS
// (row = index, column = lane: a wave reading with its uniform loop index reads one 64-byte row; per-lane indices
// collide only on equal (row mod 4, lane / 4)) it is an LDS access; indices past the rows go to a private array.
template <typename Idx>
struct PrivateIdx {
  Idx* a;
  __device__ __forceinline__ int get(int i) const { return (int)a[i]; }
  __device__ __forceinline__ void set(int i, int v) const { a[i] = (Idx)v; }
};
template <typename Idx, int ROWS>
struct LdsIdx {
  lds_u8* rows;   // + lane
  Idx* past;      // entries ROWS, ROWS + 1, ... (nullptr: there are none)
  __device__ __forceinline__ int get(int i) const { return (i < ROWS || past == nullptr) ? (int)rows[i * 64] : (int)past[i]; }
  __device__ __forceinline__ void set(int i, int v) const {
    if (i < ROWS || past == nullptr) rows[i * 64] = (unsigned char)v;
    else past[i] = (Idx)v;
  }
};

// Order-preserving image of a float32: as unsigned integers the images compare exactly as the floats do (-0 folded onto
// +0 first; NaNs have no place in that order -- waves that hold one sort by float comparisons).
__device__ inline uint32_t ordered_bits(float v) {
  const uint32_t u = __float_as_uint(v + 0.0f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// A loop the COMPILER'S FRONT END unrolls (template recursion; f is called with std::integral_constant<int, B> ... <E - 1>).
// `#pragma unroll` unrolls in a pass that runs after the last scalar-replacement pass, so an array indexed by the loop
// variable stayed in scratch memory even though every index had become a constant (measured: 464 B of scratch, the 57 sort
// keys of the rank sorts); indexed by template constants from the start it becomes registers.
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// rank(i) = number of points that sort before point i by (x, y, index) -- the stable sort of the monotone chain, through
// ranks.  The points of a lane live in scratch memory; read in the inner loop of an n^2 count they were 90 % of the kernel's
// memory traffic (2 n^2 loads per hull: 66 GB per 65536 x 51 corridors).  Here every lane loads its points ONCE into
// registers, as 64-bit keys (x image high, y image low: one integer compare per pair), and the double loop is unrolled
// by the front end (static_for), so every key is a register the compiler names: no load in it at all.  Lanes hold different counts: keys past a
// lane's count are all-ones (they sort behind everything and are never stored); `nmax`, the largest count of the wave, ends
// the unrolled loops early with scalar branches.
template <int CAP, typename SetOrder>
__device__ __forceinline__ bool rank_sort_in_registers(const P2f* p, int n, int nmax, const SetOrder& set_order) {
  uint64_t key[CAP];
  bool plain = true;
  static_for<0, CAP>([&](auto C) {
    constexpr int c = decltype(C)::value;
    key[c] = ~0ull;
    if (c < nmax && c < n) {
      const P2f q = p[c];
      plain = plain && (q.x == q.x) && (q.y == q.y);
      key[c] = ((uint64_t)ordered_bits(q.x) << 32) | ordered_bits(q.y);
    }
  });
  if (__all(plain ? 1 : 0) == 0) return false;   // a NaN somewhere in the wave: nothing written, the caller sorts by float comparisons
  static_for<0, CAP>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if (i < nmax) {      // (wave-uniform: a scalar branch around the rest of the body)
      int rank = 0;
      static_for<0, (CAP + 7) / 8>([&](auto JB) {
        constexpr int jb = decltype(JB)::value * 8;
        if (jb < nmax) {
          static_for<jb, (jb + 8 < CAP ? jb + 8 : CAP)>([&](auto J) {
            constexpr int j = decltype(J)::value;   // equal points: the earlier index first
            if constexpr (j != i) rank += (j < i) ? (key[j] <= key[i] ? 1 : 0) : (key[j] < key[i] ? 1 : 0);
          });
        }
      });
      if (i < n) set_order(rank, i);
    }
  });
  return true;
}

// strictly convex hull of p[0..n): indices into p, counter-clockwise (y up) from the
// lexicographically smallest point; `order` and `h` are caller-provided work arrays of n and 2n
// entries (exact predicates keep at most n + 1 on the stack; with float32 rounding a nearly
// collinear point can survive in both chains, and 2n - 1 pushes is the hard bound).  Returns the
// number of hull vertices.  CAP > 0: the register sort above for waves whose counts all fit CAP.
template <int CAP, int WIN, typename Ord, typename Stk>
__device__ __forceinline__ int hull_indices(const P2f* p, int n, const Ord& order, const Stk& h, lds_float2* win) {
  const int lane = threadIdx.x & 63;
  auto set_order = [&](int r, int v) { order.set(r, v); };
  bool in_registers = false;
  if (CAP > 0) {
    if (__all(n <= CAP ? 1 : 0) != 0) {
      // largest count among the lanes that are HERE (the call sits inside divergent branches: a butterfly of shuffles would
      // read lanes that are not), bit by bit through votes -- the result is the same scalar in every lane
      int nmax = 0;
#pragma unroll
      for (int b = 6; b >= 0; --b) {
        const int cand = nmax | (1 << b);
        if (__any(n >= cand ? 1 : 0)) nmax = cand;
      }
      nmax = __builtin_amdgcn_readfirstlane(nmax);
      in_registers = rank_sort_in_registers<(CAP > 0 ? CAP : 1)>(p, n, nmax, set_order);
    }
  }
  if (!in_registers) {
    // the same ranks by float comparisons, points read from scratch: n^2 comparisons, but the loads of the inner loop do
    // not depend on each other (an insertion sort is a chain of dependent memory round trips).  A NaN coordinate compares
    // false with everything, so two points can land on one rank and leave another rank without a point: every entry is set
    // first, so that what the scans read for such a rank is a point of this lane (what the reference's cv::convexHull makes of
    // a NaN is undefined; here it is at least the same corridor on every run and in every instantiation)
    // Eight points are ranked per pass over the others: one load of p[j] serves eight comparisons.
    for (int i = 0; i < n; ++i) set_order(i, 0);
    constexpr int R = CAP > 0 ? 1 : 8;   // (behind a register sort this path is the rare fallback: kept narrow, its registers count towards the kernel's)
    for (int i0 = 0; i0 < n; i0 += R) {
      P2f q[R];
      int rank[R];
      static_for<0, R>([&](auto U) {
        constexpr int u = decltype(U)::value;
        q[u] = p[min(i0 + u, n - 1)];
        rank[u] = 0;
      });
      for (int j = 0; j < n; ++j) {
        const P2f r = p[j];
        static_for<0, R>([&](auto U) {
          constexpr int u = decltype(U)::value;
          rank[u] += (r.x < q[u].x || (r.x == q[u].x && (r.y < q[u].y || (r.y == q[u].y && j < i0 + u)))) ? 1 : 0;
        });
      }
      static_for<0, R>([&](auto U) {
        constexpr int u = decltype(U)::value;
        if (i0 + u < n) set_order(rank[u], i0 + u);
      });
    }
  }
  // The monotone chain.  The two topmost stack points stay in registers (a = p[h[k-2]], b = p[h[k-1]]): the test of a
  // new point reads nothing.  A pop needs the point two below the new top: the points of the topmost WIN levels are
  // kept in LDS as well (a ring: level l in row l % WIN, column = lane, so any per-lane level is conflict-free), written
  // at every push; `lo` is the lowest level whose row has not been overwritten by a push WIN levels higher -- below it
  // (a run of more than WIN - 2 pops) the point comes from the lane's arrays, two dependent trips to memory, as every
  // pop did before.  The next point of the chain is requested one trip ahead.
  auto turn = [](const P2f& o, const P2f& m, const P2f& q) {
    const float ax = m.x - o.x, ay = m.y - o.y;
    const float bx = q.x - o.x, by = q.y - o.y;
    const float t1 = ax * by, t2 = ay * bx;
    return t1 - t2;
  };
  int k = 0, lo = 0;
  P2f a = P2f{0.0f, 0.0f}, b = P2f{0.0f, 0.0f};
  auto level_point = [&](int l) -> P2f {
    if (WIN > 0 && l >= lo) {
      const unsigned long long v = win[(l % (WIN > 0 ? WIN : 1)) * 64 + lane];
      return P2f{__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32))};
    }
    return p[h.get(l)];
  };
  auto push = [&](int oi, const P2f& q) {
    h.set(k, oi);
    if (WIN > 0) {
      win[(k % (WIN > 0 ? WIN : 1)) * 64 + lane] = ((unsigned long long)__float_as_uint(q.y) << 32) | __float_as_uint(q.x);
      lo = max(lo, k - WIN + 1);
    }
    ++k;
    a = b;
    b = q;
  };
  int on = 0;
  P2f qn = P2f{0.0f, 0.0f};
  if (n > 0) {
    on = order.get(0);
    qn = p[on];
  }
  for (int i = 0; i < n; ++i) {
    const int oi = on;
    const P2f q = qn;
    if (i + 1 < n) {
      on = order.get(i + 1);
      qn = p[on];
    }
    while (k >= 2 && turn(a, b, q) <= 0.0f) {
      --k;
      b = a;
      if (k >= 2) a = level_point(k - 2);
    }
    push(oi, q);
  }
  if (n >= 2) {
    on = order.get(n - 2);
    qn = p[on];
  }
  for (int i = n - 2, t = k + 1; i >= 0; --i) {
    const int oi = on;
    const P2f q = qn;
    if (i >= 1) {
      on = order.get(i - 1);
      qn = p[on];
    }
    while (k >= t && turn(a, b, q) <= 0.0f) {
      --k;
      b = a;
      if (k >= 2) a = level_point(k - 2);
    }
    push(oi, q);
  }
  if (k > 1) --k;
  if (k == 2) {
    const P2f h0 = p[h.get(0)], h1 = p[h.get(1)];
    if (h0.x == h1.x && h0.y == h1.y) k = 1;
  }
  return k;
}
template <typename Stk>
__device__ __forceinline__ void make_clockwise(const Stk& h, int k) {
  for (int a = 1, b = k - 1; a < b; ++a, --b) {
    const int t = h.get(a);
    h.set(a, h.get(b));
    h.set(b, t);
  }
}

}  // namespace

// knots [n][3] = x, y, theta; points [n][pmax][2]; count [n]; out corridor [n][cmax][3] (rows past
// the count zeroed), ccount [n]
// (m >= 3 half-planes, or -1 no points, -2 fewer than 4 flipped points, -3 more than cmax
// half-planes, -4 degenerate hull); *n_failed counts the knots with a negative code.
template <int MAXP, typename Idx, int BIG, int SMALL, int WIN, bool ORD>
__global__ __launch_bounds__(64) void k_build_corridors(int n, CorridorParams cp, const double* __restrict__ knots,
                                                        const double* __restrict__ points,
                                                        const int* __restrict__ count, int pmax,
                                                        double* __restrict__ corridor, int* __restrict__ ccount,
                                                        int cmax, int* __restrict__ n_failed,
                                                        double* __restrict__ polygons) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  __shared__ unsigned long long win_rows[(WIN > 0 ? WIN : 1) * 64];
  lds_float2* win = (lds_float2*)win_rows;
  constexpr int kOrdRows = MAXP + 1, kStkRows = MAXP + 2;   // a chain's stack holds at most n + 1 entries unless float32 rounding keeps a point in both chains
  __shared__ unsigned char order_bytes[ORD ? kOrdRows * 64 : 64];
  __shared__ unsigned char stack_bytes[ORD ? kStkRows * 64 : 64];
  static_assert(!ORD || sizeof(Idx) == 1, "the LDS rows of the sorted order hold bytes");
  const double ox = knots[3 * t], oy = knots[3 * t + 1], theta = knots[3 * t + 2];
  // private working set (scratch), kept small: it is what the kernel's memory traffic consists of.
  // `flip` is dead once the first hull is known and is reused for the dual points; the kept points
  // are remembered by their index in the input list and re-read (or re-derived) when needed.
  P2f flip[MAXP + 1], vd[MAXP + 1];
  P2f* dual = flip;
  Idx src[MAXP], order_priv[ORD ? 1 : MAXP + 1], hull_priv[2 * (MAXP + 1)], v2_priv[ORD ? 1 : MAXP + 1];
  // the sorted order, the chain's stack = the hull, and the second hull's copy (v2: alive only between two hulls, in the
  // rows of the sorted order): LDS rows in the kernels that have them, private arrays in the others
  using OrdT = typename std::conditional<ORD, LdsIdx<Idx, kOrdRows>, PrivateIdx<Idx>>::type;
  using StkT = typename std::conditional<ORD, LdsIdx<Idx, kStkRows>, PrivateIdx<Idx>>::type;
  OrdT order, v2;
  StkT hull;
  if constexpr (ORD) {
    order = OrdT{(lds_u8*)order_bytes + (threadIdx.x & 63), nullptr};
    v2 = order;
    hull = StkT{(lds_u8*)stack_bytes + (threadIdx.x & 63), hull_priv};
  } else {
    order = OrdT{order_priv};
    v2 = OrdT{v2_priv};
    hull = StkT{hull_priv};
  }
  int code = 0;
  int nf = 0;
  double safe_radius = cp.radius;
  const int np = min(max(count[t], 0), pmax);
  const double* pp = points + (size_t)t * pmax * 2;
  // box points of AddCorridorPoints (cc:89-120): both ends of each edge, or six samples per edge
  const double ch = cos(theta), sh = sin(theta);
  const double dx1 = ch * cp.max_axis_x, dy1 = sh * cp.max_axis_x;
  const double dx2 = sh * cp.max_axis_y, dy2 = -ch * cp.max_axis_y;
  const int per_edge = cp.per_edge, nbox = 4 * cp.per_edge;
  const double ratio_step = 1.0 / (per_edge == 2 ? 1.0 : 5.0);
  auto input_point = [&](int i, double& x, double& y) {   // obstacle points, then the box points
    if (i < np) {
      x = pp[2 * i];
      y = pp[2 * i + 1];
    } else {
      // corners: +dx1 +dx2, +dx1 -dx2, -dx1 -dx2, -dx1 +dx2; edge e runs from corner e to e + 1
      const int e = (i - np) / per_edge, kth = (i - np) - e * per_edge;
      const int k0 = e, k1 = (e + 1) & 3;
      const double s1a = (k0 < 2) ? 1.0 : -1.0, s2a = (k0 == 0 || k0 == 3) ? 1.0 : -1.0;
      const double s1b = (k1 < 2) ? 1.0 : -1.0, s2b = (k1 == 0 || k1 == 3) ? 1.0 : -1.0;
      const double ax = ox + s1a * dx1 + s2a * dx2, ay = oy + s1a * dy1 + s2a * dy2;
      const double bx = ox + s1b * dx1 + s2b * dx2, by = oy + s1b * dy1 + s2b * dy2;
      double ratio = 0.0;   // the reference's loop variable: ratio += ratio_step, in floating point (cc:113)
      for (int q = 0; q < kth; ++q) ratio += ratio_step;
      x = ax * (1 - ratio) + bx * ratio;
      y = ay * (1 - ratio) + by * ratio;
    }
  };
  {
    // filter cc:136-149 and sphere flip cc:154-177
    auto consider = [&](int i, double x, double y) {
      const double dx = x - ox, dy = y - oy;
      if (fabs(dx) > cp.max_diff_x || fabs(dy) > cp.max_diff_y) return;
      const double norm2 = sqrt(dx * dx + dy * dy);
      if (fabs(norm2) < kEps) return;
      if (norm2 < cp.radius) safe_radius = norm2;
      src[nf] = (Idx)i;
      flip[nf].x = (float)(dx + 2 * (cp.radius - norm2) * dx / norm2);
      flip[nf].y = (float)(dy + 2 * (cp.radius - norm2) * dy / norm2);
      ++nf;
    };
    // the obstacle points (lanes hold different counts; the next point is requested while this one is worked on), then
    // the box points, the same number in every lane: two loops, so that no trip of the first carries the box arithmetic
    double xn = 0.0, yn = 0.0;
    if (np > 0) {
      xn = pp[0];
      yn = pp[1];
    }
    for (int i = 0; i < np; ++i) {
      const double x = xn, y = yn;
      if (i + 1 < np) {
        xn = pp[2 * i + 2];
        yn = pp[2 * i + 3];
      }
      consider(i, x, y);
    }
    for (int i = np; i < np + nbox; ++i) {
      double x, y;
      input_point(i, x, y);
      consider(i, x, y);
    }
  }
  int m = 0;
  if (nf < 4) {
    code = -2;
  } else {
    flip[nf] = P2f{0.0f, 0.0f};
    const int n1 = hull_indices<BIG, WIN>(flip, nf + 1, order, hull, win);  // cc:184
    if (n1 < 3 || n1 > nf + 1) {   // more vertices than points: float32 predicates disagreed (degenerate)
      code = -4;
    } else {
      // star-shaped polygon through the visible points cc:186-198
      int origin_index = -1;
      for (int i = 0; i < n1; ++i) {
        if (hull.get(i) == nf) {
          origin_index = i;
          vd[i] = P2f{(float)ox, (float)oy};
        } else {
          double x, y;
          input_point(src[hull.get(i)], x, y);
          vd[i] = P2f{(float)x, (float)y};
        }
      }
      double ix = ox, iy = oy;  // cc:200-216
      if (origin_index >= 0) {
        const uint64_t sz = (uint64_t)n1;
        const int last = (int)(((uint64_t)(int64_t)(origin_index - 1)) % sz);
        const int next = (int)(((uint64_t)(int64_t)(origin_index + 1)) % sz);
        const int vl = hull.get(last), vn = hull.get(next);
        double lx = ox, ly = oy, nx = ox, ny = oy;
        if (vl != nf) input_point(src[vl], lx, ly);
        if (vn != nf) input_point(src[vn], nx, ny);
        const double dx = (lx + ox + nx) / 3 - ox;
        const double dy = (ly + oy + ny) / 3 - oy;
        const double d = sqrt(dx * dx + dy * dy);
        ix = 0.99 * safe_radius * dx / d + ox;
        iy = 0.99 * safe_radius * dy / d + oy;
      }
      const int n2 = hull_indices<SMALL, WIN>(vd, n1, order, hull, win);  // cc:218
      if (n2 < 3 || n2 > n1) {
        code = -4;
      } else {
        for (int j = 0; j < n2; ++j) v2.set(j, hull.get(j));
        // one half-plane per star vertex, normal of the hull edge it hides behind  cc:220-239
        int nt = 0;
        for (int j = 0; j < n2; ++j) {
          const int j1 = (j + 1 == n2) ? 0 : j + 1;
          const int vj = v2.get(j), vj1 = v2.get(j1);
          const float rx = vd[vj1].x - vd[vj].x, ry = vd[vj1].y - vd[vj].y;
          float n0 = ry, nn1 = -rx;
          const float z = n0 * n0 + nn1 * nn1;
          if (z > 0.0f) {
            const float s = sqrtf(z);
            n0 = n0 / s;
            nn1 = nn1 / s;
          }
          int idx = vj;
          int guard = 0;
          while (idx != vj1 && guard++ <= n1 && nt < MAXP + 1) {
            const double c = (vd[idx].x - ix) * n0 + (vd[idx].y - iy) * nn1;
            const float cf = (float)c;
            dual[nt].x = n0 / cf;
            dual[nt].y = nn1 / cf;
            ++nt;
            idx = (idx + 1 == n1) ? 0 : idx + 1;
          }
        }
        const int n3 = hull_indices<SMALL, WIN>(dual, nt, order, hull, win);  // cc:241-242
        if (n3 < 3 || n3 > nt) {
          code = -4;
        } else if (n3 > cmax) {
          code = -3;
        } else {
          make_clockwise(hull, n3);
          m = n3;
          double* out = corridor + (size_t)t * cmax * 3;
          double qx0 = 0.0, qy0 = 0.0, qxp = 0.0, qyp = 0.0;
          for (int i = 0; i <= m; ++i) {  // polygon vertices cc:244-249, half-planes cc:251-261
            double qx, qy;
            if (i < m) {
              const P2f a = dual[hull.get(i)], b = dual[hull.get((i + 1 == m) ? 0 : i + 1)];
              const float rx = b.x - a.x, ry = b.y - a.y;
              const float t1 = ry * a.x, t2 = rx * a.y;
              const double c = t1 - t2;
              qx = ix + ry / c;
              qy = iy - rx / c;
            } else {
              qx = qx0;
              qy = qy0;
            }
            if (polygons != nullptr && i < m) {
              polygons[((size_t)t * cmax + i) * 2] = qx;
              polygons[((size_t)t * cmax + i) * 2 + 1] = qy;
            }
            if (i == 0) {
              qx0 = qx;
              qy0 = qy;
            } else {
              const double rx = qx - qxp, ry = qy - qyp;
              const double c = -ry * qxp + rx * qyp;
              out[3 * (i - 1)] = -ry;
              out[3 * (i - 1) + 1] = rx;
              out[3 * (i - 1) + 2] = c;
            }
            qxp = qx;
            qyp = qy;
          }
        }
      }
    }
  }
  {
    double* out = corridor + (size_t)t * cmax * 3;
    for (int i = 3 * m; i < 3 * cmax; ++i) out[i] = 0.0;   // rows past the count: zeros
    if (polygons != nullptr)
      for (int i = 2 * m; i < 2 * cmax; ++i) polygons[(size_t)t * cmax * 2 + i] = 0.0;
  }
  ccount[t] = code < 0 ? code : m;
  if (code < 0) atomicAdd(n_failed, 1);
}

void launch_build_corridors(int n, const CorridorParams& cp, const double* knots, const double* points,
                            const int* count, int pmax, double* corridor, int* ccount, int cmax, int* n_failed,
                            double* polygons, hipStream_t st) {
  // Occupancy.  The two kernels with the LDS window run as many waves as their registers allow (11-12 per CU at 142
  // VGPRs): a wave waits on memory for most of its life and, with the pops of the chains served from LDS, another
  // wave in flight is time gained (65536 x 51 knots: 10.0 / 8.9 / 8.4 ms at 8 / 10 / 12 waves per CU).  The generic
  // kernel, every access of which goes to scratch memory, is capped at 16 waves per CU by unused dynamic LDS:
  // with more, the scratch of the waves in flight streams through HBM on every access (20 % slower, measured).
  constexpr int lds_pad_generic = 10000;
  // three capacities: a lane's scratch working set scales with it
  const int need = pmax + 4 * cp.per_edge;
  const dim3 grid((n + 63) / 64), block(64);
  if (need <= 56)
    hipLaunchKernelGGL((k_build_corridors<56, unsigned char, 57, 40, 8, true>), grid, block, 0, st, n, cp, knots,
                       points, count, pmax, corridor, ccount, cmax, n_failed, polygons);
  else if (need <= 96)
    hipLaunchKernelGGL((k_build_corridors<96, unsigned char, 0, 40, 16, false>), grid, block, 0, st, n, cp, knots,
                       points, count, pmax, corridor, ccount, cmax, n_failed, polygons);
  else   // is_multiple_sample scenes: six samples per obstacle edge and per box edge (every sort on the generic path: the
         // kernel the two above are held against bit for bit, tests/test_corridor.py)
    hipLaunchKernelGGL((k_build_corridors<kCorMaxPts, unsigned short, 0, 0, 0, false>), grid, block, lds_pad_generic, st, n, cp,
                       knots, points, count, pmax, corridor, ccount, cmax, n_failed, polygons);
}

}  // namespace cilqr
