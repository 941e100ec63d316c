// One host process, several GPUs: cilqr_multi_* (include/cilqr.h).
//
// The caller this boundary is built for is ONE process (the reference's planning node owns its planner by value:
// algorithm/planning_node.cc:9-31, trajectory_planner.h:49).  Problems are independent, so a batch held by that
// process is cut into contiguous shards, one per listed device, every shard is solved by its own handle (its own
// stream and host threads: cilqr_submit) and lands in the caller's arrays at the shard's offset -- no copy between
// devices, no collective, no second process.  cilqr_comm_* / cilqr_gather_results (comm.hip) stay the form for one
// process per GPU.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "solver_priv.hpp"

struct cilqr_multi {
  std::vector<cilqr_handle> shard;   // one handle per listed device (a device may be listed more than once)
  std::vector<int> device;
  int capacity = 0;                  // problems per call, all shards together
  int max_iter = 0;
};

namespace {
// shard k of n over `total` problems: the first total % n shards hold one problem more (cilqr_amd/distributed.py: shard_range)
void shard_range(int total, int k, int n, int* lo, int* hi) {
  const int base = total / n, rem = total % n;
  *lo = k * base + (k < rem ? k : rem);
  *hi = *lo + base + (k < rem ? 1 : 0);
}
}  // namespace

extern "C" {

int cilqr_multi_create(const cilqr_config* cfg, const int32_t* devices, int32_t n_devices, int32_t batch_capacity,
                       int32_t cmax, int32_t max_lane_segments, cilqr_multi_handle* out) {
  if (cfg == nullptr || devices == nullptr || out == nullptr) return CILQR_ERR_NULL;
  *out = nullptr;
  if (n_devices < 1 || batch_capacity < n_devices) return CILQR_ERR_ARG;
  cilqr_device_guard keep_callers_device;
  cilqr_multi* m = new (std::nothrow) cilqr_multi();
  if (m == nullptr) return CILQR_ERR_DEVICE;
  m->capacity = batch_capacity;
  m->max_iter = cfg->max_iter;
  const int per = (batch_capacity + n_devices - 1) / n_devices;
  for (int k = 0; k < n_devices; ++k) {
    cilqr_handle h = nullptr;
    const int rc = cilqr_create(cfg, devices[k], per, cmax, max_lane_segments, &h);
    if (rc != CILQR_OK) {
      cilqr_multi_destroy(m);
      return rc;
    }
    m->shard.push_back(h);
    m->device.push_back(devices[k]);
  }
  // A shard is submitted through cilqr_submit, whose default speculation threshold (2048) is the one tuned for several
  // solves SHARING a GPU.  A device listed once is its shard's alone: those shards take the threshold of the synchronous
  // call (8192, ~1 % faster there).  A device listed several times does share, and keeps the default.
  for (int k = 0; k < n_devices; ++k) {
    int listed = 0;
    for (int q = 0; q < n_devices; ++q) listed += devices[q] == devices[k];
    m->shard[(size_t)k]->alone_on_device = (listed == 1);
  }
  *out = m;
  return CILQR_OK;
}

int cilqr_multi_destroy(cilqr_multi_handle m) {
  if (m == nullptr) return CILQR_ERR_NULL;
  cilqr_device_guard keep_callers_device;
  for (cilqr_handle h : m->shard) (void)cilqr_destroy(h);
  delete m;
  return CILQR_OK;
}

int cilqr_multi_shards(cilqr_multi_handle m, int32_t batch, int32_t* first_problem, int32_t* device, int32_t max_shards) {
  if (m == nullptr) return CILQR_ERR_NULL;
  const int n = (int)m->shard.size();
  if (batch < 0) return CILQR_ERR_ARG;
  for (int k = 0; k < n && k < max_shards; ++k) {
    int lo, hi;
    shard_range(batch, k, n, &lo, &hi);
    if (first_problem) first_problem[k] = lo;
    if (device) device[k] = m->device[k];
  }
  return n;
}

int cilqr_multi_set_option(cilqr_multi_handle m, int32_t option, int64_t value) {
  if (m == nullptr) return CILQR_ERR_NULL;
  for (cilqr_handle h : m->shard) {
    const int rc = cilqr_set_option(h, option, value);
    if (rc != CILQR_OK) return rc;
  }
  return CILQR_OK;
}

int64_t cilqr_multi_device_bytes(cilqr_multi_handle m) {
  if (m == nullptr) return 0;
  int64_t b = 0;
  for (cilqr_handle h : m->shard) b += cilqr_device_bytes(h);
  return b;
}

int cilqr_multi_solve(cilqr_multi_handle m, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  if (m == nullptr || in == nullptr || out == nullptr) return CILQR_ERR_NULL;
  if (in->batch <= 0) return CILQR_ERR_ARG;
  if (in->batch > m->capacity) return CILQR_ERR_CAPACITY;
  if (in->n_lane_groups > 1) return CILQR_ERR_ARG;   // one lane table per call: the groups would straddle the shards
  const int n = (int)m->shard.size();
  const size_t K = (size_t)in->n_knots, M1 = (size_t)m->max_iter + 1;
  std::vector<cilqr_problem_batch> pin(n);
  std::vector<cilqr_solution_batch> pout(n);
  std::vector<char> submitted(n, 0);
  int rc = CILQR_OK;
  for (int k = 0; k < n && rc == CILQR_OK; ++k) {
    int b0, b1;
    shard_range(in->batch, k, n, &b0, &b1);
    if (b1 <= b0) continue;
    cilqr_problem_batch& pi = pin[k];
    pi = *in;
    pi.batch = b1 - b0;
    pi.start = in->start ? in->start + (size_t)b0 * 4 : nullptr;
    pi.coarse = in->coarse ? in->coarse + (size_t)b0 * K * 6 : nullptr;
    pi.corridor = in->corridor ? in->corridor + (size_t)b0 * K * in->cmax * 3 : nullptr;
    pi.corridor_count = in->corridor_count ? in->corridor_count + (size_t)b0 * K : nullptr;
    pi.coarse_station = in->coarse_station ? in->coarse_station + (size_t)b0 * K : nullptr;
    cilqr_solution_batch& po = pout[k];
    po = *out;
    po.traj = out->traj ? out->traj + (size_t)b0 * K * CILQR_TRAJ_FIELDS : nullptr;
    po.cost_hist = out->cost_hist ? out->cost_hist + (size_t)b0 * M1 * CILQR_COST_FIELDS : nullptr;
    po.n_cost = out->n_cost ? out->n_cost + b0 : nullptr;
    po.status = out->status ? out->status + b0 : nullptr;
    po.n_iter = out->n_iter ? out->n_iter + b0 : nullptr;
    po.iter_trajs = out->iter_trajs ? out->iter_trajs + (size_t)b0 * out->max_iter_trajs * K * CILQR_TRAJ_FIELDS : nullptr;
    po.n_iter_trajs = out->n_iter_trajs ? out->n_iter_trajs + b0 : nullptr;
    po.alpha_trace = out->alpha_trace ? out->alpha_trace + (size_t)b0 * m->max_iter : nullptr;
    rc = cilqr_submit(m->shard[k], &pi, &po);   // the shard's own host threads drive its device from here on
    if (rc == CILQR_OK) submitted[k] = 1;
  }
  for (int k = 0; k < n; ++k) {
    if (!submitted[k]) continue;
    const int r = cilqr_wait(m->shard[k]);   // every submitted shard is collected, also after an error
    if (rc == CILQR_OK && r != CILQR_OK) rc = r;
  }
  return rc;
}

}  // extern "C"
