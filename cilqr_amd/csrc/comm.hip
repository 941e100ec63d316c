// Multi-GPU results gather behind the C-ABI (include/cilqr.h, "multi-GPU" section).
//
// The reference is a single process (nothing to mirror).  Problems are independent, so
// a batch shards contiguously over one process per GPU with no data-path communication during the
// solve; the only exchange is the gather of the results to one rank.  It goes through RCCL directly
// (grouped ncclSend / ncclRecv into the root over its xGMI links: point-to-point, no ring), so the
// C++ host the boundary is built for can shard without PyTorch:
//
//   message of a rank = header (batch, live rows, rank, max_iter + 1) | [B][K][8] trajectory columns (time and kappa
//   are functions of the others and are rebuilt on the root, TransformToTrajectory cc:771-791) | n_cost, status,
//   n_iter | the LIVE Cost rows only (~9 of the 201 rows per problem), in a region of 32 rows per problem -- all
//   fp64, ONE message of a fixed size per rank, so nothing has to be agreed on before it is posted: one grouped
//   exchange and one host synchronisation per gather.  (More than 32 live rows per problem on average: the rest
//   follows in a second exchange both sides know about from the header.  The first gather of a communicator, and
//   any gather whose batch differs from the last one's, is preceded by a 16-byte exchange in which the root checks
//   that every rank holds the same batch: a mismatch is an error on every rank, not a hang.)
//
// RCCL is loaded with dlopen on the first cilqr_comm_* call: libcilqr_hip.so has no link-time
// dependency on it, and a process that already holds an RCCL (PyTorch bundles one) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "dev_model.hpp"
#include "solver_priv.hpp"

using namespace cilqr;

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  // an RCCL that is already part of the process first, then the ROCm one
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) return nullptr;
#define SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f))
  SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd);
  SYM(GetErrorString);
#undef SYM
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
    r.lib = nullptr;
    return nullptr;
  }
  return &r;
}

#define NCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) {                                                                        \
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s -> %s", #expr,                  \
                    R->GetErrorString ? R->GetErrorString(r_) : "rccl error");                      \
      return CILQR_ERR_DEVICE;                                                                      \
    }                                                                                               \
  } while (0)

// inside ncclGroupStart .. ncclGroupEnd: close the group before leaving, or the communicator stays in an open group
#define NCCL_TRY_G(expr)                                                                            \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) {                                                                        \
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s -> %s", #expr,                  \
                    R->GetErrorString ? R->GetErrorString(r_) : "rccl error");                      \
      (void)R->GroupEnd();                                                                          \
      return CILQR_ERR_DEVICE;                                                                      \
    }                                                                                               \
  } while (0)

constexpr int kTravelCols = 8;   // x y theta v a delta jerk delta_rate

// exclusive prefix sum of n_cost (one block; B is at most a few hundred thousand) -> off[B], total
__global__ __launch_bounds__(1024) void k_row_offsets(const int* __restrict__ n_cost, int B, long long* __restrict__ off,
                                                       long long* __restrict__ total) {
  __shared__ long long part[1024];
  const int t = threadIdx.x, nt = blockDim.x;
  const int per = (B + nt - 1) / nt;
  const int lo = min(B, t * per), hi = min(B, lo + per);
  long long s = 0;
  for (int b = lo; b < hi; ++b) s += n_cost[b];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    long long run = 0;
    for (int i = 0; i < nt; ++i) {
      const long long v = part[i];
      part[i] = run;
      run += v;
    }
    *total = run;
  }
  __syncthreads();
  long long run = part[t];
  for (int b = lo; b < hi; ++b) {
    off[b] = run;
    run += n_cost[b];
  }
}

// payload layout (doubles): traj8 [B][K][8] | rows [R][5] | n_cost [B] | status [B] | n_iter [B]
__global__ void k_pack_traj(const double* __restrict__ traj, int B, int K, double* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * K) return;
  const double* r = traj + t * CILQR_TRAJ_FIELDS;
  double* o = out + t * kTravelCols;
  o[0] = r[1]; o[1] = r[2]; o[2] = r[3]; o[3] = r[4]; o[4] = r[5]; o[5] = r[6]; o[6] = r[8]; o[7] = r[9];
}
// the rows of the window [lo, hi) of this rank's packed row sequence, to rows[(row - lo)]; n_cost / status / n_iter as doubles
__global__ void k_pack_rows(const double* __restrict__ hist, const int* __restrict__ n_cost, const int* __restrict__ status,
                            const int* __restrict__ n_iter, const long long* __restrict__ off, int B, int M1, long long lo,
                            long long hi, double* __restrict__ rows, double* __restrict__ ints) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nc = n_cost[b];
  const double* h = hist + (size_t)b * M1 * CILQR_COST_FIELDS;
  const long long first = off[b];
  for (int r = 0; r < nc; ++r) {
    const long long g = first + r;
    if (g < lo || g >= hi) continue;
    double* o = rows + (g - lo) * CILQR_COST_FIELDS;
    for (int e = 0; e < CILQR_COST_FIELDS; ++e) o[e] = h[r * CILQR_COST_FIELDS + e];
  }
  if (ints != nullptr) {
    ints[b] = (double)nc;
    ints[(size_t)B + b] = (double)status[b];
    ints[(size_t)2 * B + b] = n_iter ? (double)n_iter[b] : 0.0;
  }
}
// header of a rank's message: batch, live rows, rank, rows of a dense history (all exact in a double)
__global__ void k_pack_header(const long long* __restrict__ total, int B, int rank, int M1, double* __restrict__ hdr) {
  if (threadIdx.x == 0) {
    hdr[0] = (double)B;
    hdr[1] = (double)*total;
    hdr[2] = (double)rank;
    hdr[3] = (double)M1;
  }
}
// root: the headers of all blocks against what this rank holds; rows of every rank to `rows_out`, verdict to `ok`
__global__ void k_check_headers(const double* __restrict__ own, const double* __restrict__ recv, size_t block_doubles, int W,
                                int root, int B, int M1, long long* __restrict__ rows_out, long long* __restrict__ ok) {
  if (threadIdx.x != 0) return;
  long long good = 1;
  for (int p = 0; p < W; ++p) {
    const double* hdr = (p == root) ? own : recv + (size_t)(p - (p > root ? 1 : 0)) * block_doubles;
    const long long rows = (long long)hdr[1];
    rows_out[p] = rows;
    if ((long long)hdr[0] != B || (long long)hdr[2] != p || (long long)hdr[3] != M1 || rows < 0 || rows > (long long)B * M1) good = 0;
  }
  *ok = good;
}
__global__ void k_unpack_ints(const double* __restrict__ ints, int B, int* __restrict__ n_cost, int* __restrict__ status,
                              int* __restrict__ n_iter) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  n_cost[b] = (int)ints[b];
  status[b] = (int)ints[(size_t)B + b];
  if (n_iter) n_iter[b] = (int)ints[(size_t)2 * B + b];
}
__global__ void k_unpack_traj(const double* __restrict__ in, int B, int K, double dt, double wheel_base,
                              double* __restrict__ traj) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * K) return;
  const int i = (int)(t % K);
  const double* r = in + t * kTravelCols;
  double* o = traj + t * CILQR_TRAJ_FIELDS;
  o[0] = i * dt;                                   // the same expressions as write_traj_point (cc:771-791)
  o[1] = r[0]; o[2] = r[1]; o[3] = r[2]; o[4] = r[3]; o[5] = r[4]; o[6] = r[5];
  o[7] = tan(r[5]) / wheel_base;
  o[8] = r[6]; o[9] = r[7];
}
// rows of the window [lo, hi) of a block's packed row sequence (rows[(row - lo)]) into the dense history
__global__ void k_unpack_rows(const double* __restrict__ rows, const int* __restrict__ n_cost,
                              const long long* __restrict__ off, int B, int M1, long long lo, long long hi,
                              double* __restrict__ hist) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nc = n_cost[b];
  const long long first = off[b];
  double* o = hist + (size_t)b * M1 * CILQR_COST_FIELDS;
  for (int r = 0; r < nc; ++r) {
    const long long g = first + r;
    if (g < lo || g >= hi) continue;
    const double* src = rows + (g - lo) * CILQR_COST_FIELDS;
    for (int e = 0; e < CILQR_COST_FIELDS; ++e) o[r * CILQR_COST_FIELDS + e] = src[e];
  }
}

int grow_dev(void** p, size_t* have, size_t need) {
  if (need <= *have) return CILQR_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  *have = 0;
  need += need / 8;
  HIP_TRY(hipMalloc(p, need));
  *have = need;
  return CILQR_OK;
}

}  // namespace

struct cilqr_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  void* send = nullptr;     // this rank's payload
  size_t send_bytes = 0;
  void* recv = nullptr;     // root: the payloads of all ranks, back to back
  size_t recv_bytes = 0;
  long long* off = nullptr;       // [capacity] row offsets
  // device: (live rows, batch) of every rank [2 p], of this rank [2 world], the root's verdict [2 world + 2]
  long long* totals = nullptr;
  long long* h_totals = nullptr;  // pinned copy
  size_t off_cap = 0;
  int agreed_batch = -1;          // the batch every rank was seen to hold (first use, and whenever this rank's changes)
};

void cilqr_comm_release(cilqr_solver* h) {
  cilqr_comm* c = h->comm;
  if (c == nullptr) return;
  Rccl* R = rccl();
  if (c->comm && R) (void)R->CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  if (c->recv) (void)hipFree(c->recv);
  if (c->off) (void)hipFree(c->off);
  if (c->totals) (void)hipFree(c->totals);
  if (c->h_totals) (void)hipHostFree(c->h_totals);
  delete c;
  h->comm = nullptr;
}

extern "C" {

int cilqr_comm_unique_id(uint8_t* id) {
  if (id == nullptr) return CILQR_ERR_NULL;
  Rccl* R = rccl();
  if (R == nullptr) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "librccl.so.1 could not be loaded: %s", dlerror());
    return CILQR_ERR_DEVICE;
  }
  static_assert(CILQR_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  NCCL_TRY(R->GetUniqueId(&u));
  std::memcpy(id, u.internal, CILQR_UNIQUE_ID_BYTES);
  return CILQR_OK;
}

int cilqr_comm_create(cilqr_handle h, const uint8_t* id, int32_t rank, int32_t world) {
  if (h == nullptr || id == nullptr) return CILQR_ERR_NULL;
  if (world < 1 || rank < 0 || rank >= world) return CILQR_ERR_ARG;
  if (h->comm != nullptr) return CILQR_ERR_STATE;
  Rccl* R = rccl();
  if (R == nullptr) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "librccl.so.1 could not be loaded: %s", dlerror());
    return CILQR_ERR_DEVICE;
  }
  HIP_TRY(hipSetDevice(h->device));
  cilqr_comm* c = new (std::nothrow) cilqr_comm();
  if (c == nullptr) return CILQR_ERR_DEVICE;
  c->rank = rank;
  c->world = world;
  ncclUniqueId u;
  std::memcpy(u.internal, id, CILQR_UNIQUE_ID_BYTES);
  h->comm = c;
  ncclResult_t r_ = R->CommInitRank(&c->comm, world, u, rank);
  if (r_ != ncclSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->totals), (size_t)(4 * world + 8) * sizeof(long long)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&c->h_totals), (size_t)(4 * world + 8) * sizeof(long long), hipHostMallocDefault) != hipSuccess) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "ncclCommInitRank(rank %d of %d) -> %s", rank, world,
                  (r_ != ncclSuccess && R->GetErrorString) ? R->GetErrorString(r_) : "allocation failed");
    cilqr_comm_release(h);
    return CILQR_ERR_DEVICE;
  }
  return CILQR_OK;
}

int cilqr_comm_destroy(cilqr_handle h) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (h->comm == nullptr) return CILQR_ERR_STATE;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  cilqr_comm_release(h);
  return CILQR_OK;
}

int cilqr_comm_info(cilqr_handle h, int32_t* rank, int32_t* world) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (h->comm == nullptr) return CILQR_ERR_STATE;
  if (rank) *rank = h->comm->rank;
  if (world) *world = h->comm->world;
  return CILQR_OK;
}

int cilqr_gather_results(cilqr_handle h, int32_t batch, const cilqr_solution_batch* local, int32_t root,
                         cilqr_solution_batch* gathered) {
  if (h == nullptr || local == nullptr) return CILQR_ERR_NULL;
  cilqr_comm* c = h->comm;
  if (c == nullptr) return CILQR_ERR_STATE;
  if (batch <= 0 || root < 0 || root >= c->world) return CILQR_ERR_ARG;
  if (local->memory != CILQR_MEM_DEVICE) return CILQR_ERR_ARG;
  if (local->traj == nullptr || local->cost_hist == nullptr || local->n_cost == nullptr || local->status == nullptr)
    return CILQR_ERR_NULL;
  const bool is_root = c->rank == root;
  if (is_root) {
    if (gathered == nullptr) return CILQR_ERR_NULL;
    if (gathered->memory != CILQR_MEM_DEVICE) return CILQR_ERR_ARG;
    if (gathered->traj == nullptr || gathered->cost_hist == nullptr || gathered->n_cost == nullptr ||
        gathered->status == nullptr)
      return CILQR_ERR_NULL;
  }
  Rccl* R = rccl();
  if (R == nullptr) return CILQR_ERR_DEVICE;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int B = batch, K = h->cfg.n_steps + 1, M1 = h->cfg.max_iter + 1, W = c->world;
  const size_t nb = (size_t)(B + 255) / 256;
  if ((size_t)B * (size_t)W > c->off_cap) {
    if (c->off) HIP_TRY(hipFree(c->off));
    c->off = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->off), (size_t)B * W * sizeof(long long)));
    c->off_cap = (size_t)B * W;
  }
  // device scalars: [0, W) live rows of every rank (root), [W] this rank's, [W + 1] verdict, [W + 2 ...] the agreement exchange
  long long* rows_of = c->totals;
  long long* own_total = c->totals + W;
  long long* verdict = c->totals + W + 1;
  // 0. first use (or another batch than last time): every rank tells the root its batch, the root answers with a verdict --
  //    before anything is sized from that number, so that a rank that holds another batch is an error on every rank
  if (W > 1 && c->agreed_batch != B) {
    long long* said = c->totals + W + 2;         // [W] on the root, [0] elsewhere
    c->h_totals[0] = B;
    HIP_TRY(hipMemcpyAsync(said + c->rank % W, c->h_totals, sizeof(long long), hipMemcpyHostToDevice, st));
    NCCL_TRY(R->GroupStart());
    if (is_root) {
      for (int p = 0; p < W; ++p)
        if (p != root) NCCL_TRY_G(R->Recv(said + p, 1, ncclInt64, p, c->comm, st));
    } else {
      NCCL_TRY_G(R->Send(said + c->rank % W, 1, ncclInt64, root, c->comm, st));
    }
    NCCL_TRY(R->GroupEnd());
    long long ok0 = 1;
    if (is_root) {
      HIP_TRY(hipMemcpyAsync(c->h_totals, said, (size_t)W * sizeof(long long), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (int p = 0; p < W; ++p)
        if (p != root && c->h_totals[p] != B) ok0 = 0;
      c->h_totals[W] = ok0;
      HIP_TRY(hipMemcpyAsync(verdict, c->h_totals + W, sizeof(long long), hipMemcpyHostToDevice, st));
    }
    NCCL_TRY(R->GroupStart());
    if (is_root) {
      for (int p = 0; p < W; ++p)
        if (p != root) NCCL_TRY_G(R->Send(verdict, 1, ncclInt64, p, c->comm, st));
    } else {
      NCCL_TRY_G(R->Recv(verdict, 1, ncclInt64, root, c->comm, st));
    }
    NCCL_TRY(R->GroupEnd());
    if (!is_root) {
      HIP_TRY(hipMemcpyAsync(c->h_totals + W, verdict, sizeof(long long), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      ok0 = c->h_totals[W];
    }
    if (ok0 != 1) {
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "cilqr_gather_results: the ranks do not hold the same batch");
      return CILQR_ERR_ARG;
    }
    c->agreed_batch = B;
  }
  // 1. this rank's message, a fixed number of doubles: header | traj8 | ints | rows (a region of `cap` rows)
  const long long cap = std::min<long long>((long long)B * M1, (long long)B * 32);
  const size_t n_traj8 = (size_t)B * K * kTravelCols, n_ints = (size_t)3 * B;
  const size_t block = 4 + n_traj8 + n_ints + (size_t)cap * CILQR_COST_FIELDS;
  int rc = grow_dev(&c->send, &c->send_bytes, block * sizeof(double));
  if (rc != CILQR_OK) return rc;
  double* sp = static_cast<double*>(c->send);
  double* s_traj = sp + 4;
  double* s_ints = s_traj + n_traj8;
  double* s_rows = s_ints + n_ints;
  hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, st, local->n_cost, B, c->off, own_total);
  hipLaunchKernelGGL(k_pack_header, dim3(1), dim3(64), 0, st, own_total, B, c->rank, M1, sp);
  hipLaunchKernelGGL(k_pack_traj, dim3(((size_t)B * K + 255) / 256), dim3(256), 0, st, local->traj, B, K, s_traj);
  hipLaunchKernelGGL(k_pack_rows, dim3(nb), dim3(256), 0, st, local->cost_hist, local->n_cost, local->status,
                     local->n_iter, c->off, B, M1, 0LL, cap, s_rows, s_ints);
  HIP_TRY(hipGetLastError());
  // 2. the gather: one message per rank, straight into the root (block p of the receive buffer: rank p, the root's own skipped)
  auto slot_of = [&](int p) { return (size_t)(p - (p > root ? 1 : 0)); };
  if (is_root && W > 1) {
    rc = grow_dev(&c->recv, &c->recv_bytes, (size_t)(W - 1) * block * sizeof(double));
    if (rc != CILQR_OK) return rc;
  }
  if (W > 1) {
    NCCL_TRY(R->GroupStart());
    if (is_root) {
      for (int p = 0; p < W; ++p)
        if (p != root) NCCL_TRY_G(R->Recv(static_cast<double*>(c->recv) + slot_of(p) * block, block, ncclFloat64, p, c->comm, st));
    } else {
      NCCL_TRY_G(R->Send(sp, block, ncclFloat64, root, c->comm, st));
    }
    NCCL_TRY(R->GroupEnd());
  }
  // 3. unpack on the root, blocks in rank order (everything sized on the device: no host round trip in between)
  auto unpack = [&](int p, const double* rows, long long lo, long long hi, bool all) {
    const double* src = (p == root) ? sp : static_cast<const double*>(c->recv) + slot_of(p) * block;
    const size_t b0 = (size_t)p * B;
    int* g_nc = gathered->n_cost + b0;
    long long* off_p = c->off + b0;
    if (all) {
      hipLaunchKernelGGL(k_unpack_ints, dim3(nb), dim3(256), 0, st, src + 4 + n_traj8, B, g_nc, gathered->status + b0,
                         gathered->n_iter ? gathered->n_iter + b0 : nullptr);
      hipLaunchKernelGGL(k_unpack_traj, dim3(((size_t)B * K + 255) / 256), dim3(256), 0, st, src + 4, B, K, h->cfg.dt,
                         h->cfg.wheel_base, gathered->traj + b0 * K * CILQR_TRAJ_FIELDS);
      hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, st, g_nc, B, off_p, verdict + 1 + W + p);   // (total unused)
    }
    hipLaunchKernelGGL(k_unpack_rows, dim3(nb), dim3(256), 0, st, rows ? rows : src + 4 + n_traj8 + n_ints, g_nc, off_p, B, M1,
                       lo, hi, gathered->cost_hist + b0 * M1 * CILQR_COST_FIELDS);
  };
  if (is_root) {
    hipLaunchKernelGGL(k_check_headers, dim3(1), dim3(64), 0, st, sp, static_cast<const double*>(c->recv), block, W, root, B, M1,
                       rows_of, verdict);
    for (int p = 0; p < W; ++p) unpack(p, nullptr, 0LL, cap, true);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_totals, c->totals, (size_t)(W + 2) * sizeof(long long), hipMemcpyDeviceToHost, st));
  } else {
    HIP_TRY(hipMemcpyAsync(c->h_totals + W, own_total, sizeof(long long), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));      // the one host synchronisation of a gather
  if (is_root && c->h_totals[W + 1] != 1) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "cilqr_gather_results: a rank's message does not carry this batch");
    return CILQR_ERR_ARG;
  }
  // 4. more live rows than the region holds (over 32 per problem on average): the rest in a second exchange -- both sides
  //    know from the same number (the sender its own total, the root the headers')
  bool more = false;
  if (is_root) {
    for (int p = 0; p < W; ++p) more = more || c->h_totals[p] > cap;
  } else {
    more = c->h_totals[W] > cap;
  }
  if (more) {
    const long long own_more = std::max<long long>(0, (is_root ? c->h_totals[root] : c->h_totals[W]) - cap);
    std::vector<size_t> at(W + 1, 0);
    if (is_root)
      for (int p = 0; p < W; ++p) at[p + 1] = at[p] + (size_t)std::max<long long>(0, c->h_totals[p] - cap) * CILQR_COST_FIELDS;
    void* extra = nullptr;      // (a path for unusual batches: allocated and freed here)
    const size_t extra_doubles = is_root ? at[W] : (size_t)own_more * CILQR_COST_FIELDS;
    HIP_TRY(hipMalloc(&extra, (extra_doubles + 1) * sizeof(double)));
    double* ex = static_cast<double*>(extra);
    if (own_more > 0)
      hipLaunchKernelGGL(k_pack_rows, dim3(nb), dim3(256), 0, st, local->cost_hist, local->n_cost, local->status, local->n_iter,
                         c->off + (is_root ? (size_t)root * B : 0), B, M1, cap, cap + own_more, ex + (is_root ? at[root] : 0),
                         static_cast<double*>(nullptr));
    ncclResult_t gr = R->GroupStart();
    if (gr == ncclSuccess) {
      if (is_root) {
        for (int p = 0; p < W && gr == ncclSuccess; ++p)
          if (p != root && at[p + 1] > at[p]) gr = R->Recv(ex + at[p], at[p + 1] - at[p], ncclFloat64, p, c->comm, st);
      } else if (own_more > 0) {
        gr = R->Send(ex, (size_t)own_more * CILQR_COST_FIELDS, ncclFloat64, root, c->comm, st);
      }
      const ncclResult_t ge = R->GroupEnd();
      if (gr == ncclSuccess) gr = ge;
    }
    if (gr == ncclSuccess && is_root)
      for (int p = 0; p < W; ++p)
        if (at[p + 1] > at[p]) unpack(p, ex + at[p], cap, c->h_totals[p], false);
    const hipError_t se = hipStreamSynchronize(st);
    (void)hipFree(extra);
    if (gr != ncclSuccess || se != hipSuccess) {
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "cilqr_gather_results: the exchange of the rows beyond the region failed");
      return CILQR_ERR_DEVICE;
    }
  }
  return CILQR_OK;
}

}  // extern "C"
