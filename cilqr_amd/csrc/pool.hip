// A stream of batches on ONE GPU: cilqr_pool_* (include/cilqr.h).
//
// A handle overlaps the latency-bound end of a solve with the bulk of the next one (cilqr_submit, two solves in flight).
// What is left idle then is inside the bulk itself: a backward pass or a rollout is one lane per problem, a chain of N
// dependent steps that uses a fraction of the chip once the active set has shrunk -- exactly the room another solve's
// cost kernels need.  A pool owns n handles on one device and deals the submitted batches out round-robin; the
// handles' first stages drift apart by themselves and fill each other's gaps (measured on the bench workload, 65536
// problems per batch: 1.69-1.72 M solves/s with one handle, 1.94-1.99 M with two, 1.97-2.00 M with three).
// Results are the handle's own: bit-identical to cilqr_solve_batch.  Nothing here touches the device: the pool is
// bookkeeping over cilqr_create / cilqr_submit / cilqr_wait.
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "solver_priv.hpp"

struct cilqr_pool {
  std::vector<cilqr_handle> handle;
  long long submitted = 0;   // solves dealt out so far: solve s runs on handle s % n
  long long collected = 0;   // solves waited for so far (always the oldest first)
};

extern "C" {

int cilqr_pool_create(const cilqr_config* cfg, int32_t device, int32_t n_handles, int32_t batch_capacity, int32_t cmax,
                      int32_t max_lane_segments, cilqr_pool_handle* out) {
  if (cfg == nullptr || out == nullptr) return CILQR_ERR_NULL;
  *out = nullptr;
  if (n_handles < 1 || n_handles > 16) return CILQR_ERR_ARG;
  cilqr_device_guard keep_callers_device;
  cilqr_pool* p = new (std::nothrow) cilqr_pool();
  if (p == nullptr) return CILQR_ERR_DEVICE;
  for (int k = 0; k < n_handles; ++k) {
    cilqr_handle h = nullptr;
    const int rc = cilqr_create(cfg, device, batch_capacity, cmax, max_lane_segments, &h);
    if (rc != CILQR_OK) {
      cilqr_pool_destroy(p);
      return rc;
    }
    p->handle.push_back(h);
  }
  *out = p;
  return CILQR_OK;
}

int cilqr_pool_destroy(cilqr_pool_handle p) {
  if (p == nullptr) return CILQR_ERR_NULL;
  cilqr_device_guard keep_callers_device;
  while (p->collected < p->submitted) (void)cilqr_pool_wait(p);   // nothing is left running on arrays the caller frees next
  for (cilqr_handle h : p->handle) (void)cilqr_destroy(h);
  delete p;
  return CILQR_OK;
}

// handle k of the pool, for what a pool has no call of its own for (stage entry points, profiling, a synchronous solve);
// use it only while nothing is in flight on the pool
cilqr_handle cilqr_pool_handle_at(cilqr_pool_handle p, int32_t k) {
  if (p == nullptr || k < 0 || k >= (int32_t)p->handle.size() || p->collected < p->submitted) return nullptr;
  return p->handle[(size_t)k];
}

int32_t cilqr_pool_depth(cilqr_pool_handle p) { return p ? kJobRing * (int32_t)p->handle.size() : 0; }

int cilqr_pool_set_option(cilqr_pool_handle p, int32_t option, int64_t value) {
  if (p == nullptr) return CILQR_ERR_NULL;
  if (p->collected < p->submitted) return CILQR_ERR_STATE;
  for (cilqr_handle h : p->handle) {
    const int rc = cilqr_set_option(h, option, value);
    if (rc != CILQR_OK) return rc;
  }
  return CILQR_OK;
}

int64_t cilqr_pool_device_bytes(cilqr_pool_handle p) {
  if (p == nullptr) return 0;
  int64_t b = 0;
  for (cilqr_handle h : p->handle) b += cilqr_device_bytes(h);
  return b;
}

int cilqr_pool_submit(cilqr_pool_handle p, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  if (p == nullptr || in == nullptr || out == nullptr) return CILQR_ERR_NULL;
  const long long n = (long long)p->handle.size();
  if (p->submitted - p->collected >= kJobRing * n) return CILQR_ERR_STATE;   // the oldest solve has to be waited for first
  const int rc = cilqr_submit(p->handle[(size_t)(p->submitted % n)], in, out);
  if (rc == CILQR_OK) ++p->submitted;
  return rc;
}

int cilqr_pool_wait(cilqr_pool_handle p) {
  if (p == nullptr) return CILQR_ERR_NULL;
  if (p->collected >= p->submitted) return CILQR_ERR_STATE;
  const long long n = (long long)p->handle.size();
  const int rc = cilqr_wait(p->handle[(size_t)(p->collected % n)]);   // a handle returns its solves oldest first, too
  ++p->collected;
  return rc;
}

int cilqr_pool_get_profile(cilqr_pool_handle p, cilqr_profile* out) {
  if (p == nullptr || out == nullptr) return CILQR_ERR_NULL;
  if (p->collected == 0) return CILQR_ERR_STATE;
  const long long n = (long long)p->handle.size();
  return cilqr_get_profile(p->handle[(size_t)((p->collected - 1) % n)], out);   // of the solve the last wait collected
}

}  // extern "C"
