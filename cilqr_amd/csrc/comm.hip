// Multi-GPU results gather behind the C-ABI (include/cilqr.h, "multi-GPU" section).
//
// The reference is a single process (nothing to mirror).  Problems are independent, so
// a batch shards contiguously over one process per GPU with no data-path communication during the
// solve; the only exchange is the gather of the results to one rank.  It goes through RCCL directly
// (grouped ncclSend / ncclRecv into the root over its xGMI links: point-to-point, no ring), so the
// C++ host the boundary is built for can shard without PyTorch:
//
//   payload of a rank = [B][K][8] trajectory columns (time and kappa are functions of the others and
//   are rebuilt on the root, TransformToTrajectory cc:771-791) | the LIVE Cost rows only (ragged: ~9
//   of the 201 rows per problem) | n_cost, status, n_iter -- all fp64, one message per rank.
//   An 8-byte message per rank first tells the root how many rows follow.
//
// RCCL is loaded with dlopen on the first cilqr_comm_* call: libcilqr_hip.so has no link-time
// dependency on it, and a process that already holds an RCCL (PyTorch bundles one) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

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
__global__ void k_pack_rows(const double* __restrict__ hist, const int* __restrict__ n_cost, const int* __restrict__ status,
                            const int* __restrict__ n_iter, const long long* __restrict__ off, int B, int M1,
                            double* __restrict__ rows, double* __restrict__ ints) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nc = n_cost[b];
  const double* h = hist + (size_t)b * M1 * CILQR_COST_FIELDS;
  double* o = rows + off[b] * CILQR_COST_FIELDS;
  for (int e = 0; e < nc * CILQR_COST_FIELDS; ++e) o[e] = h[e];
  ints[b] = (double)nc;
  ints[(size_t)B + b] = (double)status[b];
  ints[(size_t)2 * B + b] = n_iter ? (double)n_iter[b] : 0.0;
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
__global__ void k_unpack_rows(const double* __restrict__ rows, const int* __restrict__ n_cost,
                              const long long* __restrict__ off, int B, int M1, double* __restrict__ hist) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nc = n_cost[b];
  const double* r = rows + off[b] * CILQR_COST_FIELDS;
  double* o = hist + (size_t)b * M1 * CILQR_COST_FIELDS;
  for (int e = 0; e < nc * CILQR_COST_FIELDS; ++e) o[e] = r[e];
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
      hipMalloc(reinterpret_cast<void**>(&c->totals), (size_t)(2 * world + 4) * sizeof(long long)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&c->h_totals), (size_t)(2 * world + 4) * sizeof(long long), hipHostMallocDefault) != hipSuccess) {
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
  // 1. (live rows, batch) of this rank, told to the root; the root checks that every rank holds the same batch and
  //    a plausible row count, and tells every rank its verdict before anything is sized from those numbers
  long long* own_meta = c->totals + 2 * W;
  long long* verdict = c->totals + 2 * W + 2;
  hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, st, local->n_cost, B, c->off, own_meta);
  c->h_totals[2 * W + 1] = B;
  HIP_TRY(hipMemcpyAsync(own_meta + 1, c->h_totals + 2 * W + 1, sizeof(long long), hipMemcpyHostToDevice, st));
  NCCL_TRY(R->GroupStart());
  if (is_root) {
    for (int p = 0; p < W; ++p)
      if (p != root) NCCL_TRY_G(R->Recv(c->totals + 2 * p, 2, ncclInt64, p, c->comm, st));
  } else {
    NCCL_TRY_G(R->Send(own_meta, 2, ncclInt64, root, c->comm, st));
  }
  NCCL_TRY(R->GroupEnd());
  HIP_TRY(hipMemcpyAsync(c->h_totals, c->totals, (size_t)(2 * W + 1) * sizeof(long long), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  c->h_totals[2 * c->rank] = c->h_totals[2 * W];
  c->h_totals[2 * c->rank + 1] = B;
  long long ok = 1;
  if (is_root) {
    for (int p = 0; p < W; ++p) {
      const long long rows_p = c->h_totals[2 * p], batch_p = c->h_totals[2 * p + 1];
      if (batch_p != B || rows_p < 0 || rows_p > (long long)B * M1) ok = 0;
    }
    c->h_totals[2 * W + 2] = ok;
    if (W > 1) HIP_TRY(hipMemcpyAsync(verdict, c->h_totals + 2 * W + 2, sizeof(long long), hipMemcpyHostToDevice, st));
  }
  if (W > 1) {
    NCCL_TRY(R->GroupStart());
    if (is_root) {
      for (int p = 0; p < W; ++p)
        if (p != root) NCCL_TRY_G(R->Send(verdict, 1, ncclInt64, p, c->comm, st));
    } else {
      NCCL_TRY_G(R->Recv(verdict, 1, ncclInt64, root, c->comm, st));
    }
    NCCL_TRY(R->GroupEnd());
    if (!is_root) HIP_TRY(hipMemcpyAsync(c->h_totals + 2 * W + 2, verdict, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ok = c->h_totals[2 * W + 2];
  }
  if (ok != 1) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "cilqr_gather_results: the ranks do not hold the same batch");
    return CILQR_ERR_ARG;
  }
  const size_t own_rows = (size_t)c->h_totals[2 * W];
  const size_t fixed = (size_t)B * K * kTravelCols + (size_t)3 * B;   // doubles besides the rows
  const size_t own = fixed + own_rows * CILQR_COST_FIELDS;
  // 2. pack
  int rc = grow_dev(&c->send, &c->send_bytes, own * sizeof(double));
  if (rc != CILQR_OK) return rc;
  double* sp = static_cast<double*>(c->send);
  double* s_rows = sp + (size_t)B * K * kTravelCols;
  double* s_ints = s_rows + own_rows * CILQR_COST_FIELDS;
  hipLaunchKernelGGL(k_pack_traj, dim3(((size_t)B * K + 255) / 256), dim3(256), 0, st, local->traj, B, K, sp);
  hipLaunchKernelGGL(k_pack_rows, dim3(nb), dim3(256), 0, st, local->cost_hist, local->n_cost, local->status,
                     local->n_iter, c->off, B, M1, s_rows, s_ints);
  HIP_TRY(hipGetLastError());
  // 3. the gather: one message per rank, straight into the root
  std::vector<size_t> at(W + 1, 0);
  if (is_root) {
    for (int p = 0; p < W; ++p)
      at[p + 1] = at[p] + ((p == root) ? 0 : fixed + (size_t)c->h_totals[2 * p] * CILQR_COST_FIELDS);
    rc = grow_dev(&c->recv, &c->recv_bytes, (at[W] + 1) * sizeof(double));
    if (rc != CILQR_OK) return rc;
  }
  NCCL_TRY(R->GroupStart());
  if (is_root) {
    for (int p = 0; p < W; ++p)
      if (p != root)
        NCCL_TRY_G(R->Recv(static_cast<double*>(c->recv) + at[p], at[p + 1] - at[p], ncclFloat64, p, c->comm, st));
  } else {
    NCCL_TRY_G(R->Send(sp, own, ncclFloat64, root, c->comm, st));
  }
  NCCL_TRY(R->GroupEnd());
  // 4. unpack on the root, blocks in rank order
  if (is_root) {
    for (int p = 0; p < W; ++p) {
      const double* src = (p == root) ? sp : static_cast<const double*>(c->recv) + at[p];
      const double* rows = src + (size_t)B * K * kTravelCols;
      const double* ints = rows + (size_t)c->h_totals[2 * p] * CILQR_COST_FIELDS;
      const size_t b0 = (size_t)p * B;
      int* g_nc = gathered->n_cost + b0;
      hipLaunchKernelGGL(k_unpack_ints, dim3(nb), dim3(256), 0, st, ints, B, g_nc, gathered->status + b0,
                         gathered->n_iter ? gathered->n_iter + b0 : nullptr);
      hipLaunchKernelGGL(k_unpack_traj, dim3(((size_t)B * K + 255) / 256), dim3(256), 0, st, src, B, K, h->cfg.dt,
                         h->cfg.wheel_base, gathered->traj + b0 * K * CILQR_TRAJ_FIELDS);
      long long* off_p = c->off + b0;
      hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, st, g_nc, B, off_p, own_meta);
      hipLaunchKernelGGL(k_unpack_rows, dim3(nb), dim3(256), 0, st, rows, g_nc, off_p, B, M1,
                         gathered->cost_hist + b0 * M1 * CILQR_COST_FIELDS);
    }
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(st));
  return CILQR_OK;
}

}  // extern "C"
