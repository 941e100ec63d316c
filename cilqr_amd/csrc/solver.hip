// Host side of the C-ABI (include/cilqr.h): HBM arena, lockstep driver of Optimize()
// (algorithm/ilqr/ilqr_optimizer.cc:154-320) over the whole batch, stage entry points.
//
// Nothing here computes on the CPU: the host only sizes grids, launches kernels on one HIP
// stream and reads back the active-problem count once per lockstep iteration.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <pthread.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "build_id.h"
#include "solver_priv.hpp"

using namespace cilqr;

void cilqr_comm_release(cilqr_solver* h);   // comm.hip

thread_local char g_last_hip_error[256] = "";

namespace {

template <typename T>
int dev_alloc(cilqr_solver* h, T** p, size_t count) {
  void* q = nullptr;
  const size_t bytes = count * sizeof(T);
  HIP_TRY(hipMalloc(&q, bytes ? bytes : 256));
  h->allocs.push_back(q);
  h->bytes += (int64_t)bytes;
  *p = static_cast<T*>(q);
  return CILQR_OK;
}

constexpr int64_t kTailMaxProblems = 8192;
constexpr size_t kSmallTransfer = (size_t)4 << 20;   // host batches up to this many bytes travel as one pinned block each way   // CILQR_OPT_TAIL_THRESHOLD is clamped to this

// the lazily grown blocks of a handle (staging, tail workspaces).  Their sizes are owned by the thread that grows them;
// what another thread may ask for at any time -- cilqr_device_bytes -- is the atomic sum kept beside them (found by the
// ThreadSanitizer run of tools/tsan_run.sh: the sizes themselves used to be read there)
int grow(cilqr_solver* h, void** p, size_t* have, size_t need) {
  if (need <= *have) return CILQR_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  h->grown_bytes.fetch_sub((int64_t)*have, std::memory_order_relaxed);
  *have = 0;
  HIP_TRY(hipMalloc(p, need));
  *have = need;
  h->grown_bytes.fetch_add((int64_t)need, std::memory_order_relaxed);
  return CILQR_OK;
}

void fill_params(const cilqr_config& c, Params* p) {
  std::memset(p, 0, sizeof(*p));
  p->N = c.n_steps;
  p->K = c.n_steps + 1;
  p->num_of_disc = c.num_of_disc;
  p->max_iter = c.max_iter;
  p->dt = c.dt;
  p->wheel_base = c.wheel_base;
  p->inv_wheel_base = 1.0 / c.wheel_base;
  p->w_jerk = c.w_jerk; p->w_delta_rate = c.w_delta_rate;
  p->w_x = c.w_x; p->w_y = c.w_y; p->w_theta = c.w_theta;
  p->w_v = c.w_v; p->w_a = c.w_a; p->w_delta = c.w_delta;
  p->abs_tol = c.abs_cost_tol; p->rel_tol = c.rel_cost_tol;
  p->max_velocity = c.max_velocity; p->min_acc = c.min_acceleration; p->max_acc = c.max_acceleration;
  p->jerk_min = c.jerk_min; p->jerk_max = c.jerk_max;
  p->delta_min = c.delta_min; p->delta_max = c.delta_max;
  p->delta_rate_min = c.delta_rate_min; p->delta_rate_max = c.delta_rate_max;
  p->bar_r = 1.0 / c.barrier_t;                       // barrier_function.h:85
  p->bar_eps = c.barrier_eps;
  p->bar_inv_eps = 1.0 / c.barrier_eps;
  p->bar_half_r_inv_eps2 = 0.5 * p->bar_r / (c.barrier_eps * c.barrier_eps);
  p->bar_rlogeps = p->bar_r * std::log(c.barrier_eps);
  // CalculateDiscRadius cc:97-104 and the disc offsets of cc:556-565
  const double length = c.front_hang + c.wheel_base + c.rear_hang;
  const double disc_radius = std::hypot(c.width / 2.0, length / 2.0 / c.num_of_disc);
  const double L = (c.rear_hang + c.wheel_base + c.front_hang) / c.num_of_disc;
  for (int j = 0; j < c.num_of_disc && j < kMaxDiscs; ++j) p->disc_off[j] = L * (j - 0.5) - c.rear_hang;
  p->shrink_corridor = disc_radius + c.safe_margin;
  p->shrink_lane = disc_radius;
}

int check_problem(const cilqr_solver* h, const cilqr_problem_batch* in) {
  if (in == nullptr) return CILQR_ERR_NULL;
  if (in->batch <= 0) return CILQR_ERR_ARG;
  if (in->n_lane_groups > 1) return CILQR_ERR_ARG;   // grouped lane tables: cilqr_solve_batch only (one table per load)
  if (in->corridor == nullptr || in->corridor_count == nullptr || in->left_lane == nullptr ||
      in->right_lane == nullptr || in->n_left <= 0 || in->n_right <= 0 || in->cmax <= 0)
    return CILQR_ERR_CONSTRAINTS;                                 // cc:68-73
  if (in->start == nullptr || in->coarse == nullptr) return CILQR_ERR_NULL;
  if (in->n_knots != h->cfg.n_steps + 1) return CILQR_ERR_KNOTS;  // cc:75-78
  if (in->batch > h->capacity || in->cmax > h->cmax || in->n_left > h->smax || in->n_right > h->smax)
    return CILQR_ERR_CAPACITY;
  if (in->memory != CILQR_MEM_HOST && in->memory != CILQR_MEM_DEVICE) return CILQR_ERR_ARG;
  return CILQR_OK;
}

// the main arena as solve `js` sees it: its own problem-indexed tensors, lane tables and counters
DeviceState main_view(const cilqr_solver* h, const cilqr_job_set& js) {
  DeviceState v = h->ds;
  v.hist = js.hist; v.iter = js.iter; v.status = js.status; v.n_cost = js.n_cost;
  v.n_iter_trajs = js.n_iter_trajs; v.atrace = js.atrace;
  v.lanes = js.lanes; v.lgrid = js.lgrid;
  return v;
}

// pinned host blocks of a job set, grown like the device blocks (never while a copy into them is in flight: the solve that
// owns the set is the only user)
int grow_pinned(void** p, size_t* have, size_t need) {
  if (need <= *have) return CILQR_OK;
  if (*p) HIP_TRY(hipHostFree(*p));
  *p = nullptr;
  *have = 0;
  HIP_TRY(hipHostMalloc(p, need, hipHostMallocDefault));
  *have = need;
  return CILQR_OK;
}

// Host arrays travel on two streams of the handle's own, created when the first large host batch shows up: HIP maps streams
// onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 unless the environment says otherwise) in the order they are created, and a
// handle that only ever sees device arrays should not spend queues on streams it never uses (round 3 measured what two
// handles on one queue cost: they run like one; with the copy streams created at cilqr_create one run of round 6 gave 1.84
// instead of 2.00 M solves/s -- not reproduced once they existed lazily, r06 log 2).
int io_streams(cilqr_solver* h) {
  std::lock_guard<std::mutex> lk(h->io_mu);
  if (h->stream_in == nullptr) HIP_TRY(hipStreamCreateWithFlags(&h->stream_in, hipStreamNonBlocking));
  if (h->stream_out == nullptr) HIP_TRY(hipStreamCreateWithFlags(&h->stream_out, hipStreamNonBlocking));
  return CILQR_OK;
}

// An input buffer for a solve's host arrays: a free one, or one whose last user's load kernels are known to have run.
// `block`: wait until one is given back (the transfer thread); otherwise -1 when both belong to solves.
int acquire_in_buffer(cilqr_solver* h, bool block) {
  std::unique_lock<std::mutex> lk(h->mu);
  for (;;) {
    for (int k = 0; k < 2; ++k) {
      cilqr_in_buffer& b = h->in_bufs[k];
      if (b.state == 1) continue;
      const bool wait_loaded = b.state == 2;
      b.state = 1;
      lk.unlock();
      if (wait_loaded && hipEventSynchronize(b.loaded) != hipSuccess) return -2;
      return k;
    }
    if (!block || h->quit) return -1;
    h->cv.wait(lk);
  }
}
void release_in_buffer(cilqr_solver* h, int k, hipStream_t st) {   // behind the kernels on `st` that read it
  if (k < 0) return;
  const bool recorded = hipEventRecord(h->in_bufs[k].loaded, st) == hipSuccess;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->in_bufs[k].state = recorded ? 2 : 0;
  }
  h->cv.notify_all();
}

size_t input_payload_bytes(const cilqr_solver* h, const cilqr_problem_batch* in) {
  const size_t B = (size_t)in->batch, K = (size_t)in->n_knots;
  const bool want_station = h->cfg.init_guess == CILQR_INIT_TRACKER && in->coarse_station != nullptr;
  return (B * 4 + B * K * 6 + B * K * in->cmax * 3 + (want_station ? B * K : 0)) * sizeof(double) + B * K * sizeof(int);
}

// Where the kernels find the problem-major inputs: the caller's own arrays (device memory), or a copy of them in the job
// set's staging block, enqueued on `st` (host memory).  A small batch travels through one pinned block in one copy; a large
// one array by array -- from pageable memory the runtime's own staged path sustains 51 GB/s on these boxes (57 from pinned
// memory; tools/host_path_probe.cc), so nothing is gained by copying through a ring of our own first.
int stage_inputs(cilqr_solver* h, const cilqr_problem_batch* in, cilqr_in_buffer* ib, hipStream_t st, ProblemView* out_pv) {
  const int B = in->batch, K = in->n_knots;
  ProblemView pv;
  pv.cmax_in = in->cmax;
  const size_t n_start = (size_t)B * 4, n_coarse = (size_t)B * K * 6,
               n_cor = (size_t)B * K * in->cmax * 3, n_cnt = (size_t)B * K;
  const bool want_station = h->cfg.init_guess == CILQR_INIT_TRACKER && in->coarse_station != nullptr;
  const size_t n_sta = want_station ? (size_t)B * K : 0;
  pv.station = nullptr;
  if (in->memory == CILQR_MEM_HOST) {
    const size_t bytes = (n_start + n_coarse + n_cor + n_sta) * sizeof(double) + n_cnt * sizeof(int) + 1024;
    const size_t payload = (n_start + n_coarse + n_cor + n_sta) * sizeof(double) + n_cnt * sizeof(int);
    // (a small batch lands in a block of its own: it is staged by the solving thread on the solve's stream, so the next
    // small batch's copy is ordered behind this one's load kernels by the stream itself)
    if (payload > kSmallTransfer && ib == nullptr) return CILQR_ERR_STATE;
    const int g = payload <= kSmallTransfer ? grow(h, &h->in_small, &h->in_small_bytes, kSmallTransfer + 1024) : grow(h, &ib->p, &ib->bytes, bytes);
    if (g != CILQR_OK) return g;
    double* d = static_cast<double*>(payload <= kSmallTransfer ? h->in_small : ib->p);
    if (payload <= kSmallTransfer) {
      // a small batch (the drop-in call is a batch of one): the five arrays go through ONE pinned block and ONE copy --
      // five pageable copies cost ~10 us each before the first kernel can start
      if (h->in_pinned == nullptr) {
        HIP_TRY(hipHostMalloc(&h->in_pinned, kSmallTransfer, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&h->in_pinned_ev, hipEventDisableTiming));
      } else {
        HIP_TRY(hipEventSynchronize(h->in_pinned_ev));   // the previous load's copy has left the block
      }
      char* q = static_cast<char*>(h->in_pinned);
      std::memcpy(q, in->start, n_start * 8); q += n_start * 8;
      std::memcpy(q, in->coarse, n_coarse * 8); q += n_coarse * 8;
      std::memcpy(q, in->corridor, n_cor * 8); q += n_cor * 8;
      if (want_station) { std::memcpy(q, in->coarse_station, n_sta * 8); q += n_sta * 8; }
      std::memcpy(q, in->corridor_count, n_cnt * 4);
      HIP_TRY(hipMemcpyAsync(d, h->in_pinned, payload, hipMemcpyHostToDevice, st));
      HIP_TRY(hipEventRecord(h->in_pinned_ev, st));
      pv.start = d; d += n_start;
      pv.coarse = d; d += n_coarse;
      pv.corridor = d; d += n_cor;
      if (want_station) { pv.station = d; d += n_sta; }
      pv.ccount = reinterpret_cast<const int*>(d);
    } else {
      HIP_TRY(hipMemcpyAsync(d, in->start, n_start * 8, hipMemcpyHostToDevice, st));
      pv.start = d; d += n_start;
      HIP_TRY(hipMemcpyAsync(d, in->coarse, n_coarse * 8, hipMemcpyHostToDevice, st));
      pv.coarse = d; d += n_coarse;
      HIP_TRY(hipMemcpyAsync(d, in->corridor, n_cor * 8, hipMemcpyHostToDevice, st));
      pv.corridor = d; d += n_cor;
      if (want_station) {
        HIP_TRY(hipMemcpyAsync(d, in->coarse_station, n_sta * 8, hipMemcpyHostToDevice, st));
        pv.station = d; d += n_sta;
      }
      HIP_TRY(hipMemcpyAsync(d, in->corridor_count, n_cnt * 4, hipMemcpyHostToDevice, st));
      pv.ccount = reinterpret_cast<const int*>(d);
    }
  } else {
    pv.start = in->start; pv.coarse = in->coarse; pv.corridor = in->corridor;
    pv.ccount = in->corridor_count;
    if (want_station) pv.station = in->coarse_station;
  }
  *out_pv = pv;
  return CILQR_OK;
}

// inputs staged (unless `staged` says the transfer thread has done it: the caller has made `st` wait for its event) +
// prepare kernels, on stream `st`, into the arena `*v` (whose lane geometry is filled in here)
int load_kernels(cilqr_solver* h, const cilqr_problem_batch* in, cilqr_job_set& js, DeviceState* v, hipStream_t st, const ProblemView& pv);

int do_load(cilqr_solver* h, const cilqr_problem_batch* in, cilqr_job_set& js, DeviceState* v, hipStream_t st,
            const ProblemView* staged = nullptr, int staged_buf = -1) {
  int rc = check_problem(h, in);
  if (rc != CILQR_OK) { release_in_buffer(h, staged_buf, st); return rc; }
  HIP_TRY(hipSetDevice(h->device));
  ProblemView pv;
  int buf = staged_buf;
  if (staged) {
    pv = *staged;
  } else {
    if (in->memory == CILQR_MEM_HOST && input_payload_bytes(h, in) > kSmallTransfer) {
      // (never refused: the transfer thread uploads for a solve only once every solve before it is past its load, so at
      // most one buffer is ever ahead of the solve that stages here)
      buf = acquire_in_buffer(h, false);
      if (buf < 0) return buf == -1 ? CILQR_ERR_STATE : CILQR_ERR_DEVICE;
    }
    rc = stage_inputs(h, in, buf >= 0 ? &h->in_bufs[buf] : nullptr, st, &pv);
  }
  if (rc == CILQR_OK) rc = load_kernels(h, in, js, v, st, pv);
  release_in_buffer(h, buf, st);
  return rc;
}

int load_kernels(cilqr_solver* h, const cilqr_problem_batch* in, cilqr_job_set& js, DeviceState* v, hipStream_t st, const ProblemView& pv) {
  const int B = in->batch;
  const bool want_station = h->cfg.init_guess == CILQR_INIT_TRACKER && in->coarse_station != nullptr;
  h->tracker.have_station = want_station ? 1 : 0;
  // The lane tables of consecutive solves are usually the same road: their device image and the lane grid built
  // from them (0.35 ms per solve) are kept while the caller's tables do not change by a bit.
  const size_t n_lane_d = (size_t)(in->n_left + in->n_right) * 7;
  bool same_lanes = js.lane_cache_nl == in->n_left && js.lane_cache_nr == in->n_right && js.lane_cache.size() == n_lane_d &&
                    std::memcmp(js.lane_cache.data(), in->left_lane, (size_t)in->n_left * 7 * 8) == 0 &&
                    std::memcmp(js.lane_cache.data() + (size_t)in->n_left * 7, in->right_lane, (size_t)in->n_right * 7 * 8) == 0;
  if (!same_lanes) {
    // the cache key is committed only once the upload and the grid build were enqueued without error (below):
    // until then a failure must not let the next load believe the device image is current
    js.lane_cache_nl = -1;
    js.lane_cache_nr = -1;
    js.lane_cache.resize(n_lane_d);
    std::memcpy(js.lane_cache.data(), in->left_lane, (size_t)in->n_left * 7 * 8);
    std::memcpy(js.lane_cache.data() + (size_t)in->n_left * 7, in->right_lane, (size_t)in->n_right * 7 * 8);
    // from the handle's own copy: the caller's arrays need not outlive the call
    HIP_TRY(hipMemcpyAsync(js.lanes_raw, js.lane_cache.data(), n_lane_d * 8, hipMemcpyHostToDevice, st));
  }
  v->nl = in->n_left;
  v->nr = in->n_right;
  {  // lane grid geometry: bounding box of the segment end points + 60 m, cells >= 1 m
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int side = 0; side < 2; ++side) {
      const double* tab = side ? in->right_lane : in->left_lane;
      const int n = side ? in->n_right : in->n_left;
      for (int k = 0; k < n; ++k)
        for (int e = 0; e < 2; ++e)
          for (int c = 0; c < 2; ++c) {
            const double val = tab[k * 7 + 3 + 2 * e + c];
            if (val < lo[c]) lo[c] = val;
            if (val > hi[c]) hi[c] = val;
          }
    }
    const double margin = 60.0;  // rejected line-search candidates overshoot far off the road
    double w = (hi[0] - lo[0]) + 2 * margin, hgt = (hi[1] - lo[1]) + 2 * margin;
    if (!(w > 0.0) || !(hgt > 0.0) || !std::isfinite(w) || !std::isfinite(hgt)) return CILQR_ERR_ARG;
#ifndef CILQR_GRID_MIN_CELL
#define CILQR_GRID_MIN_CELL 1.0
#endif
    double cell = CILQR_GRID_MIN_CELL;
    while (std::ceil(w / cell) * std::ceil(hgt / cell) > (double)kGridMaxCells) cell *= 1.25;
    v->gx0 = lo[0] - margin;
    v->gy0 = lo[1] - margin;
    v->ginv_h = 1.0 / cell;
    v->gnx = (int)std::ceil(w / cell);
    v->gny = (int)std::ceil(hgt / cell);
  }
  launch_load(*v, B, pv, same_lanes ? nullptr : js.lanes_raw, st);
  HIP_TRY(hipGetLastError());
  js.lane_cache_nl = in->n_left;
  js.lane_cache_nr = in->n_right;
  return CILQR_OK;
}

// solve on a given job: the three parts of a solve (see cilqr_job)
int job_begin(cilqr_solver* h, cilqr_job& j);
int job_iterate(cilqr_solver* h, cilqr_job& j, int stage);
int job_finish(cilqr_solver* h, cilqr_job& j);
void release_fin(cilqr_solver* h, cilqr_job& j);

// ------------------------------------------------------------------------------------------
// Host waits.  hipEventSynchronize / hipStreamSynchronize SPIN: the waiting thread keeps a core at 100 % until the GPU
// signals.  Right for the synchronous call -- its thread has nothing else to do, and the drop-in Plan is a 1 ms call where a
// late wake-up is a measurable share -- and wrong for submitted solves: the two workers of a handle sit in such a wait
// nearly all the time (the host runs two iterations ahead of the GPU), so a pool of two handles burned 4.4 cores per rank
// (measured, profiles/r05_bench_host_wait_spin.json) and eight ranks would have needed 35 of the 16 cores the GPU boxes grant.
// Submitted solves therefore poll: a query, a short spin for waits that are nearly over, then naps.  A nap that ends late
// costs nothing as long as it is shorter than an iteration -- the stream still holds the next one (kLead = 2).
// CILQR_HOST_WAIT=spin / nap overrides the choice for both kinds of call (measurement hook).
// ------------------------------------------------------------------------------------------
constexpr int kWaitSpinUs = 20, kWaitNapUs = 50;
constexpr int kWaitSpinShortUs = 250, kShortIterationProblems = 2048;   // the nap policy by work size
int host_wait_override() {
  static const int v = [] {
    const char* e = std::getenv("CILQR_HOST_WAIT");
    if (e == nullptr) return -1;
    return (e[0] == 'n' || e[0] == '1') ? 1 : 0;
  }();
  return v;
}
int wait_event(hipEvent_t ev, bool relaxed, int spin_us = kWaitSpinUs) {
  const int ov = host_wait_override();
  if (ov >= 0) relaxed = ov != 0;
  if (!relaxed) {
    HIP_TRY(hipEventSynchronize(ev));
    return CILQR_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return CILQR_OK;
    if (e != hipErrorNotReady) {
      HIP_TRY(e);
      return CILQR_ERR_DEVICE;
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us))
      std::this_thread::sleep_for(std::chrono::microseconds(kWaitNapUs));
  }
}
// everything enqueued on `st` so far; `scratch` is an event of the caller's that nothing else is waiting on
int wait_stream(hipStream_t st, hipEvent_t scratch, bool relaxed) {
  const int ov = host_wait_override();
  if (ov >= 0) relaxed = ov != 0;
  if (!relaxed || scratch == nullptr) {
    HIP_TRY(hipStreamSynchronize(st));
    return CILQR_OK;
  }
  HIP_TRY(hipEventRecord(scratch, st));
  return wait_event(scratch, true);
}

}  // namespace

bool cilqr_timer::on() const { return h->profiling; }
bool cilqr_timer::wants(int k) const { return h->profiling && (h->profiling_level != 2 || k == 1); }   // level 2: the backward launches only
int cilqr_timer::reserve() {
  if (next + 2 > js->ev.size()) {
    const size_t old = js->ev.size();
    js->ev.resize(old + 64);
    for (size_t i = old; i < js->ev.size(); ++i)
      if (hipEventCreate(&js->ev[i]) != hipSuccess) return -1;
  }
  return 0;
}
// roctx ranges around the phases of a solve (tracing), for `rocprofv3 --marker-trace --kernel-trace`: the host-side
// span in which a phase's kernels are enqueued, named cilqr:quadratize / backward / linesearch / other / tail.  Off unless
// CILQR_ROCTX=1 is in the environment when the first solve starts (the library is dlopened then; absent = no ranges).
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
const Roctx& roctx() {
  static const Roctx r = [] {
    Roctx x;
    const char* e = std::getenv("CILQR_ROCTX");
    if (e == nullptr || e[0] != '1') return x;
    // rocprofv3 listens to the SDK's roctx (librocprofiler-sdk-roctx); the roctracer-era libroctx64 is the fallback
    void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib == nullptr) return x;
    x.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    x.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    if (x.push == nullptr || x.pop == nullptr) x.push = nullptr, x.pop = nullptr;
    return x;
  }();
  return r;
}
constexpr const char* kPhaseName[5] = {"cilqr:quadratize", "cilqr:backward", "cilqr:linesearch", "cilqr:other", "cilqr:tail"};
}  // namespace
static void cilqr_phase_mark_begin(int k) {
  if (roctx().push) (void)roctx().push(kPhaseName[k >= 0 && k < 5 ? k : 3]);
}
static void cilqr_phase_mark_end() {
  if (roctx().pop) (void)roctx().pop();
}

int cilqr_timer::begin(int k) {
  marked = roctx().push != nullptr;
  if (marked) cilqr_phase_mark_begin(k);
  open = wants(k);
  if (!open) return 0;
  if (reserve()) return -1;
  kind.push_back(k);
  return hipEventRecord(js->ev[next++], stream) == hipSuccess ? 0 : -1;
}
// an event pair for a kernel that stamps its own start and end (hipExtLaunchKernelGGL); nullptrs when off
int cilqr_timer::pair(int k, hipEvent_t* a, hipEvent_t* b) {
  *a = nullptr; *b = nullptr;
  if (!wants(k)) return 0;
  if (reserve()) return -1;
  kind.push_back(k);
  *a = js->ev[next++];
  *b = js->ev[next++];
  return 0;
}
int cilqr_timer::end() {
  if (marked) { cilqr_phase_mark_end(); marked = false; }
  if (!open) return 0;
  open = false;
  return hipEventRecord(js->ev[next++], stream) == hipSuccess ? 0 : -1;
}
void cilqr_timer::resolve(cilqr_profile* p) {
  if (!on()) return;
  size_t nb = 0;
  for (size_t i = 0; i < kind.size(); ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, js->ev[2 * i], js->ev[2 * i + 1]);
    switch (kind[i]) {
      case 0: p->quadratize_ms += ms; break;
      case 1:
        if (nb < live_flags.size() && !live_flags[nb]) { ++nb; p->other_ms += ms; break; }  // run-ahead no-op
        p->backward_ms += ms;
        if (nb < full_flags.size() && full_flags[nb]) { p->backward_full_ms += ms; p->backward_full_launches += 1; }
        ++nb;
        break;
      case 2: p->linesearch_ms += ms; break;
      case 4: p->tail_ms += ms; break;
      default: p->other_ms += ms; break;
    }
  }
  if (!kind.empty() && h->profiling_level == 1) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, js->ev[0], js->ev[next - 1]);
    p->total_ms = ms;
  }
}

extern "C" {

int cilqr_abi_version(void) { return CILQR_ABI_VERSION; }

const char* cilqr_build_id(void) { return CILQR_BUILD_ID; }

const char* cilqr_error_string(int code) {
  switch (code) {
    case CILQR_OK: return "ok";
    case CILQR_ERR_NULL: return "null pointer";
    case CILQR_ERR_CONSTRAINTS: return "ilqr input constraints error";
    case CILQR_ERR_KNOTS: return "ilqr input coarse_traj error";
    case CILQR_ERR_CAPACITY: return "batch / cmax / lane segments exceed the capacity given to cilqr_create";
    case CILQR_ERR_DEVICE: return g_last_hip_error[0] ? g_last_hip_error : "HIP runtime error";
    case CILQR_ERR_ARG: return "invalid argument";
    case CILQR_ERR_STATE: return "stage called out of order";
    case CILQR_ERR_NO_PATH: return "DP failed";
    default: return "unknown error";
  }
}

int cilqr_default_config(cilqr_config* c, int32_t n_steps) {
  if (c == nullptr) return CILQR_ERR_NULL;
  std::memset(c, 0, sizeof(*c));
  c->n_steps = n_steps;
  c->num_of_disc = 5;       // planner_config.h:58
  c->max_iter = 200;        // :63
  c->init_guess = CILQR_INIT_IQR;   // cc:168-169
  c->dt = 0.1;              // :94
  c->safe_margin = 0.2;     // :59
  c->w_jerk = 1; c->w_delta_rate = 1;               // :46-47
  c->w_x = 0.5; c->w_y = 0.5; c->w_theta = 1e-3;    // :49-51
  c->w_v = 0.0; c->w_a = 0.0; c->w_delta = 0.0;     // :52-54
  c->abs_cost_tol = 1e-2; c->rel_cost_tol = 1e-2;   // :65-66
  c->front_hang = 0.96; c->wheel_base = 1.0; c->rear_hang = 0.929; c->width = 1.942;  // vehicle_param.h:26-41
  c->max_velocity = 20.0;                           // :46
  c->min_acceleration = -5.0; c->max_acceleration = 5.0;   // :51-52
  c->jerk_min = -10.0; c->jerk_max = 10.0;          // :57-58
  c->delta_min = -40.0 / 180 * M_PI; c->delta_max = 40.0 / 180 * M_PI;        // :60-61
  c->delta_rate_min = c->delta_min / 3.0; c->delta_rate_max = c->delta_max / 3.0;  // :63-64
  c->barrier_t = 5.0; c->barrier_eps = 0.01;        // barrier_function.h:144-145
  return CILQR_OK;
}

int cilqr_create(const cilqr_config* cfg, int32_t device, int32_t batch_capacity, int32_t cmax,
                 int32_t max_lane_segments, cilqr_handle* out) {
  if (cfg == nullptr || out == nullptr) return CILQR_ERR_NULL;
  *out = nullptr;
  if (cfg->init_guess != CILQR_INIT_IQR && cfg->init_guess != CILQR_INIT_TRACKER) return CILQR_ERR_ARG;
  if (cfg->n_steps < 1 || cfg->num_of_disc < 1 || cfg->num_of_disc > CILQR_MAX_DISCS ||
      cfg->max_iter < 1 || batch_capacity < 1 || cmax < 1 || max_lane_segments < 1 ||
      max_lane_segments > CILQR_MAX_LANE_SEGMENTS || !(cfg->dt > 0.0) || !(cfg->barrier_t > 0.0) ||
      !(cfg->barrier_eps > 0.0))
    return CILQR_ERR_ARG;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "device %d not present (%d visible)", device, ndev);
    return CILQR_ERR_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  cilqr_solver* h = new (std::nothrow) cilqr_solver();
  if (h == nullptr) return CILQR_ERR_DEVICE;
  h->cfg = *cfg;
  h->device = device;
  // slots are padded to a multiple of 64 so that every row of every tensor starts 512 B aligned
  const int Bc = ((batch_capacity + 63) / 64) * 64;
  h->Bcap = Bc;
  h->capacity = batch_capacity;
  h->cmax = cmax;
  h->smax = max_lane_segments;
  std::memset(&h->ds, 0, sizeof(h->ds));
  std::memset(&h->prof, 0, sizeof(h->prof));
  DeviceState& d = h->ds;
  d.Bcap = Bc;
  d.cmax = cmax;
  d.exact_ties = 1;   // CILQR_OPT_EXACT_LANE_TIES: the reference's tie rule (cc:605-618) is the default; twin / fin copy it below
  fill_params(*cfg, &d.p);
  {
    cilqr_tracker_config tc;
    cilqr_default_tracker_config(&tc);
    (void)cilqr_set_tracker_config(h, &tc);
  }
  const size_t N = cfg->n_steps, K = N + 1, B = Bc;
  d.Pcap = Bc;
  int rc = CILQR_OK;
  // tensors the survivor re-packing moves (k_compact): every arena and every twin has its own
  auto alloc_moved = [&](DeviceState& t, size_t cap) {
#define ALLOCM(field, count) \
  if (rc == CILQR_OK) rc = dev_alloc(h, &t.field, (size_t)(count))
    ALLOCM(X, 2 * K * 3 * cap);
    ALLOCM(U, 2 * N * cap);
    ALLOCM(cur, cap);
    ALLOCM(goals, K * 3 * cap);
    ALLOCM(cor, K * cmax * 3 * cap);
    ALLOCM(ccnt, K * cap);
    ALLOCM(lambda, cap); ALLOCM(dlambda, cap); ALLOCM(cost_old, cap); ALLOCM(dcost, cap);
    ALLOCM(upd, cap); ALLOCM(acc_idx, cap); ALLOCM(emit, cap); ALLOCM(pid, cap); ALLOCM(done_now, cap);
    ALLOCM(act, cap); ALLOCM(act_next, cap); ALLOCM(posn, cap);
#undef ALLOCM
  };
  // per-iteration scratch of an arena (shared with its twin)
  auto alloc_scratch = [&](DeviceState& t, size_t cap) {
#define ALLOCS(field, count) \
  if (rc == CILQR_OK) rc = dev_alloc(h, &t.field, (size_t)(count))
    ALLOCS(lin, N * kLinPairs * cap);
    ALLOCS(term, (size_t)kTermPairs * cap);
    ALLOCS(gains, N * kGainPairs * cap);
    ALLOCS(dV, 2 * cap);
    ALLOCS(gnorm, cap);
    ALLOCS(part, K * kPartPairs * cap);
    ALLOCS(trial, 5 * cap);
    // Candidate arena: all eleven step sizes of every slot for small arenas; FOUR per slot from 32768 slots on -- what the
    // pre-rolled rounds need, re-strided afterwards for the rest (kernels_search.hip, spec_view): 1.5 GB instead of 4.1 GB
    // at B = 65536, N = 50.  CILQR_SPEC_ROWS (4..11) forces the count (tests drive the four-row layout with small batches).
    t.spec_cap = (int)cap;
    t.spec_rows = (cap >= 32768) ? 4 : kNumAlpha;
    if (const char* e = std::getenv("CILQR_SPEC_ROWS")) t.spec_rows = std::min(kNumAlpha, std::max(4, std::atoi(e)));
    ALLOCS(Xs, (size_t)t.spec_rows * K * 3 * cap);
    ALLOCS(Us, (size_t)t.spec_rows * N * cap);
    ALLOCS(parts, (size_t)t.spec_rows * K * kPartPairs * cap);
    ALLOCS(spec_tot, (size_t)kNumAlpha * 5 * cap);
    ALLOCS(pend, (size_t)(kNumAlpha + 1) * cap);
    ALLOCS(counters, 64);
#undef ALLOCS
  };
  alloc_moved(d, B);
  alloc_scratch(d, B);
  if (rc == CILQR_OK) rc = dev_alloc(h, &d.coarse0, 2 * B);
  if (rc == CILQR_OK) rc = dev_alloc(h, &d.cstation, K * B);
  // twin arena for re-packing the survivors (k_compact)
  h->twin = d;
  alloc_moved(h->twin, B);
  // finishing arena + twin: a solve moves here once at most fin_cap problems are left (job_iterate)
  size_t fin_want = 8192;
  if (const char* fe = std::getenv("CILQR_FIN_CAP")) fin_want = (size_t)std::max(64, std::atoi(fe)) / 64 * 64;   // tuning experiments
  h->fin_cap = (int)std::min<size_t>(B, fin_want);
  h->fin_threshold = h->fin_cap;
  h->fin = d;
  h->fin.Bcap = h->fin_cap;
  alloc_moved(h->fin, (size_t)h->fin_cap);
  alloc_scratch(h->fin, (size_t)h->fin_cap);
  h->fin_twin = h->fin;
  alloc_moved(h->fin_twin, (size_t)h->fin_cap);
  // what a solve in flight owns (two sets: see cilqr_job_set)
  for (int k = 0; k < 2 && rc == CILQR_OK; ++k) {
    cilqr_job_set& js = h->sets[k];
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.hist, (size_t)(cfg->max_iter + 1) * 5 * B);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.iter, B);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.status, B);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.n_cost, B);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.n_iter_trajs, B);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.atrace, (size_t)cfg->max_iter * B);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.lanes, (size_t)2 * max_lane_segments * kLaneFields);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.lgrid, (size_t)2 * kGridMaxCells * kGridCellBytes);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.lanes_raw, (size_t)2 * max_lane_segments * 7);
    if (rc == CILQR_OK) rc = dev_alloc(h, &js.tail_iter_dev, 4);
    if (rc == CILQR_OK && hipHostMalloc(reinterpret_cast<void**>(&js.h_count), (size_t)(cfg->max_iter + 64) * sizeof(int),
                                        hipHostMallocMapped) != hipSuccess)
      rc = CILQR_ERR_DEVICE;
    if (rc == CILQR_OK && hipHostGetDevicePointer(reinterpret_cast<void**>(&js.h_count_dev), js.h_count, 0) != hipSuccess)
      rc = CILQR_ERR_DEVICE;
    if (rc == CILQR_OK && hipEventCreateWithFlags(&js.handoff, hipEventDisableTiming) != hipSuccess) rc = CILQR_ERR_DEVICE;
    if (rc == CILQR_OK && hipEventCreateWithFlags(&js.sync_ev, hipEventDisableTiming) != hipSuccess) rc = CILQR_ERR_DEVICE;
    if (rc == CILQR_OK && hipEventCreateWithFlags(&js.exported, hipEventDisableTiming) != hipSuccess) rc = CILQR_ERR_DEVICE;
  }
  for (cilqr_in_buffer& b : h->in_bufs) {
    if (rc == CILQR_OK && (hipEventCreateWithFlags(&b.ready, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&b.loaded, hipEventDisableTiming) != hipSuccess))
      rc = CILQR_ERR_DEVICE;
  }
  if (rc == CILQR_OK) {   // the stage API works on the main arena with the first set
    const DeviceState v = main_view(h, h->sets[0]);
    d.hist = v.hist; d.iter = v.iter; d.status = v.status; d.n_cost = v.n_cost; d.n_iter_trajs = v.n_iter_trajs;
    d.atrace = v.atrace; d.lanes = v.lanes; d.lgrid = v.lgrid;
  }
  if (rc == CILQR_OK) rc = dev_alloc(h, &h->lambda_stage, B);
  if (rc == CILQR_OK && hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
    rc = CILQR_ERR_DEVICE;
  if (rc == CILQR_OK) {
    int lo = 0, hi = 0;   // numerically lower = higher priority
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const char* pe = std::getenv("CILQR_FIN_PRIORITY");   // tuning experiments: 0 = same priority as the first stage
    const int prio = (pe && pe[0] == '0') ? lo : hi;
    if (hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio) != hipSuccess) rc = CILQR_ERR_DEVICE;
  }
  if (rc != CILQR_OK) {
    cilqr_destroy(h);
    return rc;
  }
  h->stream = h->own_stream;
  *out = h;
  return CILQR_OK;
}

int cilqr_destroy(cilqr_handle h) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (h->workers_started) {
    {
      // nothing is left running on arrays the caller frees next: the solves submitted and not collected finish first
      std::unique_lock<std::mutex> lk(h->mu);
      h->cv.wait(lk, [&] {
        for (int k = 0; k < h->job_count; ++k)
          if (h->jobs[(h->job_head + k) % kJobRing].phase != 5) return false;
        return true;
      });
      h->quit = true;
    }
    h->cv.notify_all();
    h->worker1.join();
    h->worker2.join();
    h->worker_io.join();
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->stream_in) (void)hipStreamSynchronize(h->stream_in);
  if (h->stream_out) (void)hipStreamSynchronize(h->stream_out);
  cilqr_comm_release(h);
  for (void* p : h->allocs) (void)hipFree(p);
  for (cilqr_in_buffer& b : h->in_bufs) {
    if (b.p) (void)hipFree(b.p);
    if (b.ready) (void)hipEventDestroy(b.ready);
    if (b.loaded) (void)hipEventDestroy(b.loaded);
  }
  if (h->in_small) (void)hipFree(h->in_small);
  if (h->in_pinned) (void)hipHostFree(h->in_pinned);
  if (h->in_pinned_ev) (void)hipEventDestroy(h->in_pinned_ev);
  if (h->cor_fail) (void)hipFree(h->cor_fail);
  if (h->cor_fail_host) (void)hipHostFree(h->cor_fail_host);
  if (h->cor_done) (void)hipEventDestroy(h->cor_done);
  if (h->tail_ws) (void)hipFree(h->tail_ws);
  if (h->tail_ws1) (void)hipFree(h->tail_ws1);
  for (cilqr_job_set& js : h->sets) {
    if (js.out_stage) (void)hipFree(js.out_stage);
    if (js.row_off) (void)hipFree(js.row_off);
    if (js.out_pinned) (void)hipHostFree(js.out_pinned);
    if (js.host_counts) (void)hipHostFree(js.host_counts);
    if (js.host_rows) (void)hipHostFree(js.host_rows);
    if (js.exported) (void)hipEventDestroy(js.exported);
    if (js.h_count) (void)hipHostFree(js.h_count);
    for (hipEvent_t e : js.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : js.iter_ev) (void)hipEventDestroy(e);
    if (js.handoff) (void)hipEventDestroy(js.handoff);
    if (js.sync_ev) (void)hipEventDestroy(js.sync_ev);
  }
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream_in) (void)hipStreamDestroy(h->stream_in);
  if (h->stream_out) (void)hipStreamDestroy(h->stream_out);
  delete h;
  return CILQR_OK;
}

int cilqr_set_stream(cilqr_handle h, void* hip_stream) {
  if (h == nullptr) return CILQR_ERR_NULL;
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return CILQR_OK;
}

int cilqr_set_option(cilqr_handle h, int32_t option, int64_t value) {
  if (h == nullptr) return CILQR_ERR_NULL;
  switch (option) {
    case CILQR_OPT_COMPACTION:   // 0 = off, 1 = default, 2..100 = re-pack at that occupancy percentage
      if (value < 0 || value > 100) return CILQR_ERR_ARG;
      h->compaction = value != 0;
      if (value >= 2) h->compact_percent = (int)value;
      return CILQR_OK;
    case CILQR_OPT_SPEC_THRESHOLD:
      if (value < 0) return CILQR_ERR_ARG;
      h->spec_threshold = (int)std::min<int64_t>(value, spec_open_capacity(h->ds));
      h->spec_threshold_submit = h->spec_threshold;   // an explicit choice holds for both kinds of call
      return CILQR_OK;
    case CILQR_OPT_SEQ_ROUNDS:
      if (value < 1 || value > kNumAlpha) return CILQR_ERR_ARG;
      h->seq_rounds = (int)value;
      return CILQR_OK;
    case CILQR_OPT_TEAM_THRESHOLD:
      if (value < 0) return CILQR_ERR_ARG;
      h->team_threshold = (int)(value > 0x7fffffff ? 0x7fffffff : value);
      return CILQR_OK;
    case CILQR_OPT_ROUND_GROUP:
      if (value != 1 && value != 2 && value != 4) return CILQR_ERR_ARG;
      h->round_group = (int)value;
      return CILQR_OK;
    case CILQR_OPT_WAVE_THRESHOLD:
      if (value < 0) return CILQR_ERR_ARG;
      h->wave_threshold = (int)(value > 0x7fffffff ? 0x7fffffff : value);
      return CILQR_OK;
    case CILQR_OPT_TAIL_THRESHOLD:
      if (value < 0) return CILQR_ERR_ARG;
      h->tail_threshold = (int)(value > kTailMaxProblems ? kTailMaxProblems : value);
      h->tail_threshold_submit = h->tail_threshold;   // an explicit choice holds for both kinds of call
      return CILQR_OK;
    case CILQR_OPT_EXACT_LANE_TIES:
      if (value != 0 && value != 1) return CILQR_ERR_ARG;
      h->ds.exact_ties = h->twin.exact_ties = h->fin.exact_ties = h->fin_twin.exact_ties = (int)value;
      return CILQR_OK;
    case CILQR_OPT_FINISH_THRESHOLD:
      if (value < 0) return CILQR_ERR_ARG;
      h->fin_threshold = (int)(value > h->fin_cap ? h->fin_cap : value);
      return CILQR_OK;
    default:
      return CILQR_ERR_ARG;
  }
}

int cilqr_get_option(cilqr_handle h, int32_t option, int64_t* value, int64_t* value_submitted) {
  if (h == nullptr || value == nullptr) return CILQR_ERR_NULL;
  int64_t v = 0, vs = 0;
  switch (option) {
    case CILQR_OPT_COMPACTION: v = vs = h->compaction ? h->compact_percent : 0; break;
    case CILQR_OPT_SPEC_THRESHOLD: v = h->spec_threshold; vs = h->spec_threshold_submit; break;
    case CILQR_OPT_SEQ_ROUNDS: v = vs = h->seq_rounds; break;
    case CILQR_OPT_TEAM_THRESHOLD: v = vs = h->team_threshold; break;
    case CILQR_OPT_ROUND_GROUP: v = vs = h->round_group; break;
    case CILQR_OPT_WAVE_THRESHOLD: v = vs = h->wave_threshold; break;
    case CILQR_OPT_TAIL_THRESHOLD: v = h->tail_threshold; vs = h->tail_threshold_submit; break;
    case CILQR_OPT_EXACT_LANE_TIES: v = vs = h->ds.exact_ties; break;
    case CILQR_OPT_FINISH_THRESHOLD: v = vs = h->fin_threshold; break;
    default: return CILQR_ERR_ARG;
  }
  *value = v;
  if (value_submitted) *value_submitted = vs;
  return CILQR_OK;
}

void cilqr_default_tracker_config(cilqr_tracker_config* c) {
  if (c == nullptr) return;
  c->weight_l = 1e-1; c->weight_theta = 1e-12; c->weight_delta = 1e-12; c->weight_delta_rate = 0.1; c->preview_time = 0.2;   // :18-25
  c->weight_s = 5.0 * 1e-1; c->weight_v = 1e-12; c->weight_a = 1e-12; c->weight_j = 0.1;                                     // :27-34
  c->sumulation_dt = 0.01; c->dt = 0.1; c->tolerance = 0.01; c->max_num_iteration = 150;                                    // :37-40
  c->reserved0 = 0;
}

int cilqr_set_tracker_config(cilqr_handle h, const cilqr_tracker_config* c) {
  if (h == nullptr || c == nullptr) return CILQR_ERR_NULL;
  if (!(c->sumulation_dt > 0.0) || !(c->dt > 0.0) || !(c->tolerance >= 0.0) || c->max_num_iteration < 1) return CILQR_ERR_ARG;
  TrackerParams& t = h->tracker;
  t.weight_l = c->weight_l; t.weight_theta = c->weight_theta; t.weight_delta = c->weight_delta;
  t.weight_delta_rate = c->weight_delta_rate; t.preview_time = c->preview_time;
  t.weight_s = c->weight_s; t.weight_v = c->weight_v; t.weight_a = c->weight_a; t.weight_j = c->weight_j;
  t.sim_dt = c->sumulation_dt; t.dt = c->dt; t.tolerance = c->tolerance; t.max_num_iteration = c->max_num_iteration;
  t.have_station = 0;
  return CILQR_OK;
}

int cilqr_set_profiling(cilqr_handle h, int32_t enable) {
  if (h == nullptr) return CILQR_ERR_NULL;
  h->profiling = enable != 0;
  h->profiling_level = (enable == 2) ? 2 : 1;
  return CILQR_OK;
}

int cilqr_get_profile(cilqr_handle h, cilqr_profile* out) {
  if (h == nullptr || out == nullptr) return CILQR_ERR_NULL;
  *out = h->prof;
  return CILQR_OK;
}

int64_t cilqr_device_bytes(cilqr_handle h) {
  if (h == nullptr) return 0;
  return h->bytes + h->grown_bytes.load(std::memory_order_relaxed);   // arenas (fixed at create) + staging and tail workspaces
}

static int solve_sync(cilqr_solver* h, const cilqr_problem_batch* in, cilqr_solution_batch* out);

int cilqr_solve_batch(cilqr_handle h, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  if (h == nullptr || out == nullptr) return CILQR_ERR_NULL;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->job_count != 0) return CILQR_ERR_STATE;   // submitted solves not collected yet (cilqr_wait)
  }
  return solve_sync(h, in, out);
}

// one synchronous solve (both stages on the handle's stream), lane groups one after the other
static int solve_groups(cilqr_solver* h, cilqr_job& j, const cilqr_problem_batch* in, cilqr_solution_batch* out);

static int solve_one(cilqr_solver* h, cilqr_job& j, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  if (in) j.in = *in; else std::memset(&j.in, 0, sizeof(j.in));
  j.out = *out;
  j.upload = 0;          // staged inline, on the solve's own stream
  j.zero_inline = true;
  int rc = (in == nullptr) ? CILQR_ERR_NULL : job_begin(h, j);
  if (rc == CILQR_OK) rc = job_iterate(h, j, 1);
  if (rc == CILQR_OK && j.handed) rc = job_iterate(h, j, 2);
  if (rc == CILQR_OK) rc = job_finish(h, j);
  else {   // nothing of a failed solve may still be running when its buffers go back to the caller
    (void)hipStreamSynchronize(j.st1);
    if (j.st2 != j.st1) (void)hipStreamSynchronize(j.st2);
  }
  release_fin(h, j);
  return rc;
}

static int solve_sync(cilqr_solver* h, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  cilqr_job& j = h->jobs[0];
  j.set = 0;
  j.spec_threshold = h->spec_threshold;
  j.tail_threshold = h->tail_threshold;
  j.st1 = h->stream;
  j.st2 = h->stream;
  j.relaxed_wait = false;    // the caller's own thread waits, and the call's latency is what it is measured by (wait_event)
  const int rc = solve_groups(h, j, in, out);
  std::lock_guard<std::mutex> lk(h->mu);
  h->prof = j.prof;
  return rc;
}

static int solve_groups(cilqr_solver* h, cilqr_job& j, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  if (in == nullptr || in->n_lane_groups <= 1) return solve_one(h, j, in, out);   // j.prof: published by the caller's thread
  // problems grouped by lane table: one solve per group on contiguous sub-ranges of every array
  if (in->lane_group_start == nullptr || in->lane_group_left == nullptr || in->lane_group_right == nullptr ||
      in->left_lane == nullptr || in->right_lane == nullptr)
    return CILQR_ERR_CONSTRAINTS;
  if (in->lane_group_start[0] != 0 || in->lane_group_start[in->n_lane_groups] != in->batch) return CILQR_ERR_ARG;
  const size_t K = (size_t)in->n_knots, M1 = (size_t)h->cfg.max_iter + 1;
  size_t lrow = 0, rrow = 0;
  cilqr_profile acc;
  std::memset(&acc, 0, sizeof(acc));
  for (int g = 0; g < in->n_lane_groups; ++g) {
    const int b0 = in->lane_group_start[g], b1 = in->lane_group_start[g + 1];
    if (b1 < b0 || in->lane_group_left[g] <= 0 || in->lane_group_right[g] <= 0) return CILQR_ERR_ARG;
    if (b1 > b0) {
      cilqr_problem_batch pi = *in;
      pi.n_lane_groups = 0;
      pi.batch = b1 - b0;
      pi.start = in->start ? in->start + (size_t)b0 * 4 : nullptr;
      pi.coarse = in->coarse ? in->coarse + (size_t)b0 * K * 6 : nullptr;
      pi.corridor = in->corridor ? in->corridor + (size_t)b0 * K * in->cmax * 3 : nullptr;
      pi.corridor_count = in->corridor_count ? in->corridor_count + (size_t)b0 * K : nullptr;
      pi.coarse_station = in->coarse_station ? in->coarse_station + (size_t)b0 * K : nullptr;
      pi.n_left = in->lane_group_left[g];
      pi.n_right = in->lane_group_right[g];
      pi.left_lane = in->left_lane + lrow * CILQR_LANE_FIELDS;
      pi.right_lane = in->right_lane + rrow * CILQR_LANE_FIELDS;
      cilqr_solution_batch po = *out;
      po.traj = out->traj ? out->traj + (size_t)b0 * K * CILQR_TRAJ_FIELDS : nullptr;
      po.cost_hist = out->cost_hist ? out->cost_hist + (size_t)b0 * M1 * CILQR_COST_FIELDS : nullptr;
      po.n_cost = out->n_cost ? out->n_cost + b0 : nullptr;
      po.status = out->status ? out->status + b0 : nullptr;
      po.n_iter = out->n_iter ? out->n_iter + b0 : nullptr;
      po.iter_trajs = out->iter_trajs ? out->iter_trajs + (size_t)b0 * out->max_iter_trajs * K * CILQR_TRAJ_FIELDS : nullptr;
      po.n_iter_trajs = out->n_iter_trajs ? out->n_iter_trajs + b0 : nullptr;
      po.alpha_trace = out->alpha_trace ? out->alpha_trace + (size_t)b0 * h->cfg.max_iter : nullptr;
      const int rc = solve_one(h, j, &pi, &po);
      if (rc != CILQR_OK) return rc;
      acc.iterations += j.prof.iterations;
      acc.backward_launches += j.prof.backward_launches;
      acc.backward_ms += j.prof.backward_ms; acc.quadratize_ms += j.prof.quadratize_ms;
      acc.linesearch_ms += j.prof.linesearch_ms; acc.other_ms += j.prof.other_ms; acc.total_ms += j.prof.total_ms;
      acc.backward_problem_steps += j.prof.backward_problem_steps;
      acc.backward_full_launches += j.prof.backward_full_launches;
      acc.backward_full_ms += j.prof.backward_full_ms;
      acc.tail_ms += j.prof.tail_ms;
      acc.tail_problems += j.prof.tail_problems;
    }
    lrow += (size_t)in->lane_group_left[g];
    rrow += (size_t)in->lane_group_right[g];
  }
  j.prof = acc;
  return CILQR_OK;
}

}  // extern "C"

namespace {

// the fields of `src` that belong to the solve rather than to an arena: problem-indexed tensors, lane tables
void adopt_job_fields(DeviceState* t, const DeviceState& src) {
  t->hist = src.hist; t->iter = src.iter; t->status = src.status; t->n_cost = src.n_cost;
  t->n_iter_trajs = src.n_iter_trajs; t->atrace = src.atrace;
  t->lanes = src.lanes; t->lgrid = src.lgrid;
  t->nl = src.nl; t->nr = src.nr;
  t->exact_ties = src.exact_ties;
  t->gx0 = src.gx0; t->gy0 = src.gy0; t->ginv_h = src.ginv_h; t->gnx = src.gnx; t->gny = src.gny;
  t->Pcap = src.Pcap;
}

// `t` = arena `own` with the twin's copies of what k_compact moves
DeviceState twin_of(const DeviceState& own, const DeviceState& tw) {
  DeviceState t = own;
  t.X = tw.X; t.U = tw.U; t.cur = tw.cur; t.goals = tw.goals; t.cor = tw.cor; t.ccnt = tw.ccnt;
  t.lambda = tw.lambda; t.dlambda = tw.dlambda; t.cost_old = tw.cost_old; t.dcost = tw.dcost;
  t.upd = tw.upd; t.acc_idx = tw.acc_idx; t.emit = tw.emit; t.pid = tw.pid; t.done_now = tw.done_now;
  t.act = tw.act; t.act_next = tw.act_next; t.posn = tw.posn;
  return t;
}

// bytes of the output staging block in front of the iterates, and whether host outputs of this size travel through the
// small pinned block (one copy, handed out on the host) or ragged (job_finish)
size_t out_head_bytes(const cilqr_solver* h, int B, const cilqr_solution_batch* out) {
  const size_t K = (size_t)h->cfg.n_steps + 1, M = (size_t)h->cfg.max_iter;
  const size_t n_traj = (size_t)B * K * 10, n_hist = (size_t)B * (M + 1) * 5, n_at = out->alpha_trace ? (size_t)B * M : 0;
  return (n_traj + n_hist) * 8 + (((size_t)4 * B * 4 + n_at + 7) & ~(size_t)7);
}
bool host_out_is_big(const cilqr_solver* h, int B, const cilqr_solution_batch* out) {
  if (out->memory != CILQR_MEM_HOST) return false;
  const size_t K = (size_t)h->cfg.n_steps + 1;
  const size_t n_itr = out->iter_trajs ? (size_t)B * (size_t)std::max(out->max_iter_trajs, 0) * K * 10 : 0;
  return out_head_bytes(h, B, out) + n_itr * 8 > kSmallTransfer;
}

// Arguments, staging, load, init guess and its cost (cc:64-78, 141-173), on the first stage's stream.
int job_begin(cilqr_solver* h, cilqr_job& j) {
  const cilqr_problem_batch* in = &j.in;
  const cilqr_solution_batch* out = &j.out;
  if (out->traj == nullptr || out->cost_hist == nullptr || out->n_cost == nullptr || out->status == nullptr)
    return CILQR_ERR_NULL;                                                     // cc:64-66
  if (out->memory != CILQR_MEM_HOST && out->memory != CILQR_MEM_DEVICE) return CILQR_ERR_ARG;
  if (out->iter_trajs != nullptr && (out->max_iter_trajs <= 0 || out->n_iter_trajs == nullptr)) return CILQR_ERR_ARG;
  HIP_TRY(hipSetDevice(h->device));
  cilqr_job_set& js = h->sets[j.set];
  hipStream_t st = j.st1;
  j.tm = cilqr_timer();
  j.tm.h = h; j.tm.js = &js; j.tm.stream = st;
  std::memset(&j.prof, 0, sizeof(j.prof));
  j.handed = false; j.owns_fin = false; j.tail_used = false; j.tail_n = 0; j.it = 0;
  j.bwd_iter.clear();
  if (j.tm.begin(3)) return CILQR_ERR_DEVICE;
  j.gmain = main_view(h, js);
  int rc;
  int upload_state;
  {   // (under the lock: the transfer thread moves the field while it works -- found by the ThreadSanitizer run of round 6)
    std::unique_lock<std::mutex> lk(h->mu);
    // the transfer thread has been copying this solve's arrays since it was submitted (worker_io_main): wait until every
    // copy is enqueued and the event behind the last one recorded, then let the stream wait for that event
    if (j.upload != 0) h->cv.wait(lk, [&] { return j.upload == 3 || j.upload < 0; });
    upload_state = j.upload;
  }
  if (upload_state != 0) {
    if (upload_state < 0) {
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s", j.err_text);
      {
        std::lock_guard<std::mutex> lk(h->mu);
        j.past_load = true;
      }
      h->cv.notify_all();
      return j.upload_rc;
    }
    HIP_TRY(hipStreamWaitEvent(st, h->in_bufs[j.in_buf].ready, 0));
    rc = do_load(h, in, js, &j.gmain, st, &j.pv, j.in_buf);
    j.in_buf = -1;   // (given back behind the load kernels, whatever happened)
  } else {
    rc = do_load(h, in, js, &j.gmain, st);
  }
  {   // (the transfer thread uploads for the next solve only now: worker_io_main)
    std::lock_guard<std::mutex> lk(h->mu);
    j.past_load = true;
  }
  h->cv.notify_all();
  if (rc != CILQR_OK) return rc;
  h->stage = 0;   // the main arena no longer holds what cilqr_stage_load put there
  const int B = in->batch, K = in->n_knots, M = h->cfg.max_iter;
  j.B = B;
  // two views of the device state: `d` is the arena the active problems live in, `o` the other one
  j.d = j.gmain;
  j.o = twin_of(j.gmain, h->twin);

  // output staging when the caller's buffers live in host memory
  j.o_traj = out->traj; j.o_hist = out->cost_hist; j.o_it = out->iter_trajs;
  j.o_nc = out->n_cost; j.o_st = out->status; j.o_ni = out->n_iter; j.o_nit = out->n_iter_trajs;
  j.o_at = reinterpret_cast<signed char*>(out->alpha_trace);
  j.n_traj = (size_t)B * K * 10; j.n_hist = (size_t)B * (M + 1) * 5;
  j.n_itr = out->iter_trajs ? (size_t)B * out->max_iter_trajs * K * 10 : 0;
  j.n_at = out->alpha_trace ? (size_t)B * M : 0;
  j.n_head = 0;
  j.small_out = false;
  j.big_out = false;
  if (out->memory == CILQR_MEM_HOST) {
    // staging block: traj | cost_hist | n_cost, status, n_iter, n_iter_trajs | alpha_trace | (8-byte pad) | iter_trajs --
    // the iterates last, so that a small batch can fetch everything else (and the first few iterates) in one short copy
    j.n_head = out_head_bytes(h, B, out);
    const size_t bytes = j.n_head + j.n_itr * 8 + 1024;
    rc = grow(h, &js.out_stage, &js.out_stage_bytes, bytes);
    if (rc != CILQR_OK) return rc;
    double* p = static_cast<double*>(js.out_stage);
    j.o_traj = p; p += j.n_traj;
    j.o_hist = p; p += j.n_hist;
    int* q = reinterpret_cast<int*>(p);
    j.o_nc = q; j.o_st = q + B; j.o_ni = q + 2 * B; j.o_nit = q + 3 * B;
    j.o_at = out->alpha_trace ? reinterpret_cast<signed char*>(q + 4 * B) : nullptr;
    j.o_it = out->iter_trajs ? reinterpret_cast<double*>(static_cast<char*>(js.out_stage) + j.n_head) : nullptr;
    j.big_out = host_out_is_big(h, B, out);
    j.small_out = !j.big_out;
    // The staging buffer is reused between solves.  Its Cost rows reach the caller through the live rows only, on either
    // path (a small batch is handed out row by row in job_finish, a large one downloads them packed); the iterates of a large
    // batch travel as the dense block they are, so the entries the kernels do not write (>= n_iter_trajs) are cleared here
    if (j.big_out && j.n_itr) HIP_TRY(hipMemsetAsync(j.o_it, 0, j.n_itr * 8, st));
    if (j.big_out) {
      rc = grow(h, reinterpret_cast<void**>(&js.row_off), &js.row_off_bytes, ((size_t)B + 1) * sizeof(long long));
      if (rc == CILQR_OK) rc = grow_pinned(&js.host_counts, &js.host_counts_bytes, (size_t)4 * B * sizeof(int));
      if (rc != CILQR_OK) return rc;
    }
  }

  if (h->cfg.init_guess == CILQR_INIT_TRACKER) launch_init_guess_tracker(j.d, h->tracker, B, st);   // cc:168 (InitGuess)
  else launch_init_guess(j.d, B, st);                  // cc:169
  launch_cost_only(j.d, nullptr, B, 0, st);            // cc:172
  launch_init_cost_commit(j.d, B, st);                 // cc:170,173
  if (j.o_it) launch_export_iter_traj(j.d, nullptr, B, j.o_it, out->max_iter_trajs, st);
  if (j.tm.end()) return CILQR_ERR_DEVICE;

  if ((int)js.iter_ev.size() < M) {
    const size_t old = js.iter_ev.size();
    js.iter_ev.resize(M);
    for (size_t i = old; i < js.iter_ev.size(); ++i)
      HIP_TRY(hipEventCreateWithFlags(&js.iter_ev[i], hipEventDisableTiming));
  }
  // The tail of the batch (kernels_tail.hip) needs a private arena per problem; sized before the first kernel so
  // that no allocation falls into the solve.  (Grown only while no other solve is finishing: see cilqr_submit.)
  if (j.tail_threshold > 0) {
    const size_t need = (size_t)std::min(B, j.tail_threshold) * tail_workspace_bytes(j.d);
    if (need > h->tail_ws_bytes) {
      std::unique_lock<std::mutex> lk(h->mu);
      h->cv.wait(lk, [h] { return !h->fin_busy || h->quit; });
      if (h->quit) return CILQR_ERR_STATE;   // the handle is being destroyed
      rc = grow(h, &h->tail_ws, &h->tail_ws_bytes, need);
      if (rc != CILQR_OK) return rc;
    }
    rc = grow(h, &h->tail_ws1, &h->tail_ws1_bytes, need);   // only ever used by the first stage (this thread)
    if (rc != CILQR_OK) return rc;
  }
  launch_init_counters(j.d, B, st);
  // rows >= n_cost of the caller's cost_hist are zero on every host path (include/cilqr.h).  A large batch receives its live
  // rows only, so the array is cleared on the host -- by the transfer thread for a submitted solve, here (behind the first
  // kernels, which keep the GPU busy meanwhile) for the synchronous call
  if (j.big_out && j.zero_inline) std::memset(out->cost_hist, 0, j.n_hist * 8);
  j.n_hint = B;   // upper bound of the active count of the iteration being enqueued
  j.span = B;     // slots occupied in the current arena (upper bound)
  return CILQR_OK;
}

__global__ void k_seed_counters(int* __restrict__ dst, const int* __restrict__ src_count, int ring_entry) {
  if (threadIdx.x < 64) dst[threadIdx.x] = (threadIdx.x == ring_entry) ? *src_count : 0;
}

// The lockstep iterations of Optimize() (cc:201-319).  stage 1: in the main arena, until at most fin_threshold
// problems are left -- then the survivors are copied into the finishing arena and the function returns with
// j.handed set; stage 2: the rest, there, on j.st2, down to the tail kernel.
int job_iterate(cilqr_solver* h, cilqr_job& j, int stage) {
  HIP_TRY(hipSetDevice(h->device));
  cilqr_job_set& js = h->sets[j.set];
  const int M = h->cfg.max_iter, B = j.B;
  hipStream_t st = (stage == 1) ? j.st1 : j.st2;
  DeviceState& d = j.d;
  DeviceState& o = j.o;
  if (stage == 2) {
    if (j.st2 != j.st1) HIP_TRY(hipStreamWaitEvent(j.st2, js.handoff, 0));
    j.tm.stream = st;
  }
  // The host runs kLead iterations ahead of the GPU: before enqueueing iteration `it` it waits only
  // for the active count produced by iteration it - kLead (normally long finished), which bounds
  // the grids of iteration `it`; the kernels clamp to the exact device-side count (n_dev).
  constexpr int kLead = 2;
#ifdef CILQR_REF_ORDER
  const int tail_threshold = 0;   // the test-only build re-evaluates whole trajectories in the reference's order
#else
  // (a horizon whose per-step rows no longer fit beside the tail kernel's fixed LDS block stays in the lockstep loop)
  const int tail_threshold = tail_supported(d) ? j.tail_threshold : 0;
#endif
  int& it = j.it;
  int& n_hint = j.n_hint;
  for (; it < M; ++it) {                               // cc:201
    if (it >= kLead) {
      // (few problems left: an iteration is ~100 us, and a nap that ends late costs a whole one -- spin longer before napping)
      if (int wrc = wait_event(js.iter_ev[it - kLead], j.relaxed_wait, n_hint < kShortIterationProblems ? kWaitSpinShortUs : kWaitSpinUs)) return wrc;
      n_hint = js.h_count[it - kLead];
      if (n_hint == 0) break;                          // iterations it-kLead+1 .. it-1 were no-ops
    }
    bool hand_over = false;
    // (a synchronous call whose problems all go to the tail kernel next has nothing to gain from moving them first:
    // no other solve is waiting for the main arena -- 17 us of a batch-of-one Plan)
    const bool straight_to_tail = j.st1 == j.st2 && n_hint <= tail_threshold;
    if (stage == 1 && h->fin_threshold > 0 && n_hint <= h->fin_threshold && !straight_to_tail) {
      // few enough problems left: they continue in the finishing arena, the main arena is free for the next solve.
      // Only if that arena is free -- while the solve before this one still finishes there, this one keeps iterating
      // where it is and asks again next iteration (small batches then run side by side, one in each arena).
      std::lock_guard<std::mutex> lk(h->mu);
      if (!h->fin_busy) {
        h->fin_busy = true;
        j.owns_fin = true;      // released by release_fin, also when an error ends the solve before the hand-over is complete
        hand_over = true;
      }
    }
    if (hand_over) {
      // The active list of iteration `it` is d.act, its exact length entry it % 3 of the ring of counts.
      if (j.tm.begin(3)) return CILQR_ERR_DEVICE;
      DeviceState a = d;
      a.act_next = d.act;
      a.n_next = d.counters + kCntActive + it % 3;
      DeviceState f = h->fin;
      adopt_job_fields(&f, j.gmain);
      launch_compact(a, f, n_hint, st);
      hipLaunchKernelGGL(k_seed_counters, dim3(1), dim3(64), 0, st, f.counters, a.n_next, kCntActive + it % 3);
      HIP_TRY(hipGetLastError());
      if (j.tm.end()) return CILQR_ERR_DEVICE;
      HIP_TRY(hipEventRecord(js.handoff, st));
      d = f;
      o = twin_of(f, h->fin_twin);
      j.span = n_hint;
      j.handed = true;
      j.owns_fin = true;
      return CILQR_OK;
    }
    // iteration `it` reads entry it % 3 of the ring of active counts, counts its survivors into
    // the next entry and clears the one after that
    d.n_dev = d.counters + kCntActive + it % 3;
    d.n_next = d.counters + kCntActive + (it + 1) % 3;
    d.n_clear = d.counters + kCntActive + (it + 2) % 3;
    d.h_count_dev = js.h_count_dev + it;
    o.n_dev = d.n_dev; o.n_next = d.n_next; o.n_clear = d.n_clear; o.h_count_dev = d.h_count_dev;
    if (n_hint <= tail_threshold) {
      // few problems left: each gets a workgroup that runs all its remaining iterations (cc:201-319) in one launch
      // a solve that was never handed over has the tail workspace of the first stage to itself
      void* ws = (stage == 1 && !j.handed) ? h->tail_ws1 : h->tail_ws;
      if (j.tm.begin(4)) return CILQR_ERR_DEVICE;
      HIP_TRY(hipMemsetAsync(js.tail_iter_dev, 0, sizeof(int), st));
      launch_tail(d, ws, n_hint, j.o_traj, j.o_it, j.out.max_iter_trajs, js.tail_iter_dev, st);
      HIP_TRY(hipMemcpyAsync(js.h_count + M + 32, js.tail_iter_dev, sizeof(int), hipMemcpyDeviceToHost, st));
      if (j.tm.end()) return CILQR_ERR_DEVICE;
      j.tail_used = true;
      j.tail_n = n_hint;
      break;
    }
    if (j.tm.begin(0)) return CILQR_ERR_DEVICE;
    // every active problem is linearised every iteration: the rows of `lin` belong to POSITIONS of this iteration's active
    // list (DeviceState::posn), so a problem whose line search was rejected (1.3 % of the iterations; cc:296-308 keeps its
    // linearisation) computes the same numbers again instead of finding them at last iteration's position
    launch_quadratize(d, d.act, n_hint, 0, st);        // cc:203-214
    hipEvent_t eb0, eb1;
    if (j.tm.end() || j.tm.pair(1, &eb0, &eb1)) return CILQR_ERR_DEVICE;
    cilqr_phase_mark_begin(1);
    launch_backward(d, d.act, n_hint, nullptr, h->team_threshold, h->wave_threshold, st, eb0, eb1);    // cc:218
    cilqr_phase_mark_end();
    if (j.tm.begin(2)) return CILQR_ERR_DEVICE;
    j.bwd_iter.push_back(it);
    launch_linesearch(d, n_hint, j.spec_threshold, h->seq_rounds, h->round_group, st);  // cc:235-270
    launch_update(d, n_hint, st);                      // cc:272-308
    launch_export_done(d, n_hint, j.o_traj, st);       // cc:238,285,303,319
    if (j.o_it) launch_export_iter_traj(d, d.act, n_hint, j.o_it, j.out.max_iter_trajs, st);
    if (j.tm.end()) return CILQR_ERR_DEVICE;
    HIP_TRY(hipEventRecord(js.iter_ev[it], st));
    if (h->compaction && (int64_t)100 * n_hint <= (int64_t)h->compact_percent * j.span) {
      // the survivors have thinned out: re-pack them densely (k_compact reads the exact count)
      if (j.tm.begin(3)) return CILQR_ERR_DEVICE;
      launch_compact(d, o, n_hint, st);
      if (j.tm.end()) return CILQR_ERR_DEVICE;
      DeviceState t = d; d = o; o = t;
      j.span = n_hint;
    } else {
      int* t = d.act; d.act = d.act_next; d.act_next = t;
    }
  }
  (void)B;
  return CILQR_OK;
}

// Everything enqueued: wait, export the problem-indexed results, copy out, resolve the profile.
int job_finish(cilqr_solver* h, cilqr_job& j) {
  cilqr_job_set& js = h->sets[j.set];
  const cilqr_solution_batch* out = &j.out;
  const int M = h->cfg.max_iter, B = j.B;
  hipStream_t st = j.handed ? j.st2 : j.st1;
  j.tm.stream = st;
  // (no wait here: the export and the copy out are enqueued behind the solve's last kernel, ONE host round trip for all of it)
  if (j.tm.begin(3)) return CILQR_ERR_DEVICE;
  const bool big_out = out->memory == CILQR_MEM_HOST && j.big_out;
  launch_export_hist(j.gmain, B, big_out ? nullptr : j.o_hist, j.o_nc, j.o_st, j.o_ni, j.o_nit, j.o_at, st);
  if (big_out) launch_export_hist_rows(j.gmain, B, js.row_off, j.o_hist, st);   // the live rows, packed, where the dense block used to go
  if (j.tm.end()) return CILQR_ERR_DEVICE;
  HIP_TRY(hipGetLastError());
  const bool small_out = out->memory == CILQR_MEM_HOST && j.small_out;
  const size_t one_traj = ((size_t)h->cfg.n_steps + 1) * 10 * 8;                 // bytes of one iterate
  const size_t per = out->iter_trajs ? (size_t)out->max_iter_trajs * one_traj : 0;   // iterate block of one problem
  // a batch of one (the drop-in call) fetches its first iterates with the head: 8-9 exist on average, the capacity is
  // max_iter + 1 (820 KB, 20 us of copy for 36 KB of data)
  constexpr int kFirstIterates = 16;
  const size_t first = (small_out && B == 1 && out->iter_trajs) ? (size_t)std::min(kFirstIterates, out->max_iter_trajs) * one_traj : 0;
  if (small_out) {
    // a small batch: the head of the staging block in one copy into pinned memory, handed out on the host below (seven
    // pageable copies cost ~25 us each after the last kernel); the iterates that exist follow once their counts are known
    if (js.out_pinned == nullptr) HIP_TRY(hipHostMalloc(&js.out_pinned, kSmallTransfer, hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(js.out_pinned, js.out_stage, j.n_head + first, hipMemcpyDeviceToHost, st));
  } else if (big_out) {
    // A large batch: everything is downloaded on the handle's download stream, so that the kernels of the NEXT solve's
    // finishing stage do not queue behind 0.3 GB of copies.  First the counts (they size the packed rows), then the
    // trajectories straight into the caller's array; the finishing arena goes back as soon as the last kernel is known to
    // be done, i.e. before the copies are
    if (int src = io_streams(h)) return src;
    hipStream_t so = h->stream_out;
    HIP_TRY(hipEventRecord(js.exported, st));
    HIP_TRY(hipStreamWaitEvent(so, js.exported, 0));
    HIP_TRY(hipMemcpyAsync(js.host_counts, j.o_nc, (size_t)4 * B * 4, hipMemcpyDeviceToHost, so));   // n_cost | status | n_iter | n_iter_trajs
    HIP_TRY(hipEventRecord(js.sync_ev, so));
    if (int wrc = wait_event(js.sync_ev, j.relaxed_wait)) return wrc;
    release_fin(h, j);
    const int32_t* nc = static_cast<const int32_t*>(js.host_counts);
    size_t rows = 0;
    for (int b = 0; b < B; ++b) rows += (size_t)std::min(std::max(nc[b], 0), M + 1);
    if (int grc = grow_pinned(&js.host_rows, &js.host_rows_bytes, std::max(rows + rows / 2, (size_t)B * 16) * 5 * 8)) return grc;
    HIP_TRY(hipMemcpyAsync(js.host_rows, j.o_hist, rows * 5 * 8, hipMemcpyDeviceToHost, so));
    HIP_TRY(hipMemcpyAsync(out->traj, j.o_traj, j.n_traj * 8, hipMemcpyDeviceToHost, so));
    if (out->iter_trajs) HIP_TRY(hipMemcpyAsync(out->iter_trajs, j.o_it, j.n_itr * 8, hipMemcpyDeviceToHost, so));
    if (out->alpha_trace) HIP_TRY(hipMemcpyAsync(out->alpha_trace, j.o_at, j.n_at, hipMemcpyDeviceToHost, so));
    st = so;
  }
  if (int wrc = wait_stream(st, js.sync_ev, j.relaxed_wait)) return wrc;
  int it = j.it;
  {  // lockstep iterations that had work, and the problem-steps each backward launch covered
    int used = 0;
    for (int i = 0; i < it; ++i) {
      const int n_in = (i == 0) ? B : js.h_count[i - 1];
      if (n_in > 0) used = i + 1;
    }
    for (int i : j.bwd_iter) {
      const int n_in = (i == 0) ? B : js.h_count[i - 1];
      if (n_in <= 0) continue;
      j.prof.backward_launches += 1;
      j.prof.backward_problem_steps += (int64_t)n_in * h->cfg.n_steps;
    }
    j.tm.full_flags.assign(j.bwd_iter.size(), 0);
    j.tm.live_flags.assign(j.bwd_iter.size(), 0);
    for (size_t k = 0; k < j.bwd_iter.size(); ++k) {
      const int n_in = (j.bwd_iter[k] == 0) ? B : js.h_count[j.bwd_iter[k] - 1];
      j.tm.full_flags[k] = (n_in == B);
      j.tm.live_flags[k] = (n_in > 0);
    }
    it = used;
    if (j.tail_used) {
      it = std::max(it, js.h_count[M + 32]);
      j.prof.tail_problems = j.tail_n;   // upper bound (the count the host knew when it enqueued the tail)
    }
  }
  j.prof.iterations = it;
  if (big_out) {
    if (j.io_busy) {   // the transfer thread clears the caller's cost_hist behind the upload: done long ago, but make sure
      std::unique_lock<std::mutex> lk(h->mu);
      h->cv.wait(lk, [&] { return !j.io_busy; });
    }
    const char* q = static_cast<const char*>(js.host_counts);
    const int32_t* nc = reinterpret_cast<const int32_t*>(q);
    const size_t row = 5 * 8, block = (size_t)(M + 1) * row;
    const char* src = static_cast<const char*>(js.host_rows);
    for (int b = 0; b < B; ++b) {   // live rows into the caller's dense array; the rest of it is zero already
      const size_t live = (size_t)std::min(std::max(nc[b], 0), M + 1) * row;
      std::memcpy(reinterpret_cast<char*>(out->cost_hist) + (size_t)b * block, src, live);
      src += live;
    }
    std::memcpy(out->n_cost, q, (size_t)B * 4);
    std::memcpy(out->status, q + (size_t)B * 4, (size_t)B * 4);
    if (out->n_iter) std::memcpy(out->n_iter, q + (size_t)2 * B * 4, (size_t)B * 4);
    if (out->iter_trajs) std::memcpy(out->n_iter_trajs, q + (size_t)3 * B * 4, (size_t)B * 4);
  }
  if (small_out) {   // same layout as the staging block (job_begin)
    char* pin = static_cast<char*>(js.out_pinned);
    const char* q = pin + (j.n_traj + j.n_hist) * 8;
    const int32_t* nc = reinterpret_cast<const int32_t*>(q);
    const int32_t* nit = reinterpret_cast<const int32_t*>(q + (size_t)3 * B * 4);
    if (out->iter_trajs) {
      // the iterates that exist and did not come with the head (the capacity is max_iter + 1 in the drop-in adapter)
      bool more = false;
      for (int b = 0; b < B; ++b) {
        const size_t used = (size_t)std::min(std::max(nit[b], 0), out->max_iter_trajs) * one_traj;
        const size_t have = (b == 0) ? std::min(first, used) : 0;
        if (used > have) {
          HIP_TRY(hipMemcpyAsync(pin + j.n_head + (size_t)b * per + have, reinterpret_cast<const char*>(j.o_it) + (size_t)b * per + have,
                                 used - have, hipMemcpyDeviceToHost, st));
          more = true;
        }
      }
      if (more) { if (int wrc = wait_stream(st, js.sync_ev, j.relaxed_wait)) return wrc; }
    }
    std::memcpy(out->traj, pin, j.n_traj * 8);
    {  // cost rows: the live ones; the rest of the caller's array is zero, as on every host path
      const size_t row = 5 * 8, block = (size_t)(M + 1) * row;
      std::memset(out->cost_hist, 0, j.n_hist * 8);
      for (int b = 0; b < B; ++b) {
        const size_t live = (size_t)std::min(std::max(nc[b], 0), M + 1) * row;
        std::memcpy(reinterpret_cast<char*>(out->cost_hist) + (size_t)b * block, pin + j.n_traj * 8 + (size_t)b * block, live);
      }
    }
    if (out->iter_trajs) {
      for (int b = 0; b < B; ++b) {
        const size_t used = (size_t)std::min(std::max(nit[b], 0), out->max_iter_trajs) * one_traj;
        std::memcpy(reinterpret_cast<char*>(out->iter_trajs) + (size_t)b * per, pin + j.n_head + (size_t)b * per, used);
      }
    }
    std::memcpy(out->n_cost, q, (size_t)B * 4);
    std::memcpy(out->status, q + (size_t)B * 4, (size_t)B * 4);
    if (out->n_iter) std::memcpy(out->n_iter, q + (size_t)2 * B * 4, (size_t)B * 4);
    if (out->iter_trajs) std::memcpy(out->n_iter_trajs, q + (size_t)3 * B * 4, (size_t)B * 4);
    if (out->alpha_trace) std::memcpy(out->alpha_trace, q + (size_t)4 * B * 4, j.n_at);
  }
  j.tm.resolve(&j.prof);
  return CILQR_OK;
}

// the finishing arena is free again (also after an error on the way)
void release_fin(cilqr_solver* h, cilqr_job& j) {
  if (!j.owns_fin) return;
  j.owns_fin = false;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    h->fin_busy = false;
  }
  h->cv.notify_all();
}

// ------------------------------------------------------------------------------------------
// asynchronous solves: two jobs in flight on one handle.  Worker 1 drives the first stage of a solve on the
// handle's stream, worker 2 the finishing stage on a second (high-priority) stream, so the few hundred
// stragglers of solve i finish while the bulk of solve i+1 is being iterated.
// ------------------------------------------------------------------------------------------
// A solve is over (done, or failed anywhere on the way): the transfer thread has let go of it, its input buffer -- if its
// load never got to give it back -- is free again, and the solve behind it may upload.
void job_io_settle(cilqr_solver* h, cilqr_job& j) {
  {
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv.wait(lk, [&] { return !j.io_busy; });
  }
  if (j.in_buf >= 0) {
    release_in_buffer(h, j.in_buf, j.st1);
    j.in_buf = -1;
  }
  {
    std::lock_guard<std::mutex> lk(h->mu);
    j.past_load = true;
  }
  h->cv.notify_all();
}

void worker1_main(cilqr_solver* h) {
  (void)pthread_setname_np(pthread_self(), "cilqr-stage1");
  (void)hipSetDevice(h->device);
  std::unique_lock<std::mutex> lk(h->mu);
  for (;;) {
    cilqr_job* job = nullptr;
    h->cv.wait(lk, [&] {
      if (h->quit) return true;
      for (int k = 0; k < h->job_count; ++k) {   // oldest first
        cilqr_job& c = h->jobs[(h->job_head + k) % kJobRing];
        if (c.phase == 1) {       // the oldest queued solve starts as soon as one of the two job sets is free
          if (h->set_busy[0] && h->set_busy[1]) return false;
          job = &c;
          return true;
        }
      }
      return false;
    });
    if (h->quit) return;
    job->phase = 2;
    job->set = h->set_busy[0] ? 1 : 0;
    h->set_busy[job->set] = true;
    lk.unlock();
    cilqr_job& j = *job;
    int rc;
    bool to_stage2 = false;
    if (j.in.n_lane_groups > 1) {
      // grouped lane tables: the groups are solved one after the other on this thread, nothing overlaps
      j.st2 = j.st1;
      cilqr_problem_batch in = j.in;
      cilqr_solution_batch out = j.out;
      rc = solve_groups(h, j, &in, &out);
    } else {
      rc = job_begin(h, j);
      if (rc == CILQR_OK) rc = job_iterate(h, j, 1);
      if (rc == CILQR_OK && j.handed) to_stage2 = true;
      else {
        if (rc == CILQR_OK) rc = job_finish(h, j);
        else (void)hipStreamSynchronize(j.st1);
        release_fin(h, j);
      }
    }
    if (!to_stage2) job_io_settle(h, j);   // (a failed solve: nothing of it may still touch the caller's arrays)
    lk.lock();
    j.rc = rc;
    if (rc != CILQR_OK) std::snprintf(j.err_text, sizeof(j.err_text), "%s", g_last_hip_error);
    j.phase = to_stage2 ? 3 : 5;
    if (!to_stage2) h->set_busy[j.set] = false;
    h->cv.notify_all();
  }
}

void worker2_main(cilqr_solver* h) {
  (void)pthread_setname_np(pthread_self(), "cilqr-finish");
  (void)hipSetDevice(h->device);
  std::unique_lock<std::mutex> lk(h->mu);
  for (;;) {
    cilqr_job* job = nullptr;
    h->cv.wait(lk, [&] {
      if (h->quit) return true;
      for (int k = 0; k < h->job_count; ++k) {
        cilqr_job& c = h->jobs[(h->job_head + k) % kJobRing];
        if (c.phase == 3) { job = &c; return true; }
      }
      return false;
    });
    if (h->quit) return;
    job->phase = 4;
    lk.unlock();
    cilqr_job& j = *job;
    int rc = job_iterate(h, j, 2);
    if (rc == CILQR_OK) rc = job_finish(h, j);
    else (void)hipStreamSynchronize(j.st2);
    release_fin(h, j);
    job_io_settle(h, j);
    lk.lock();
    j.rc = rc;
    if (rc != CILQR_OK) std::snprintf(j.err_text, sizeof(j.err_text), "%s", g_last_hip_error);
    j.phase = 5;
    h->set_busy[j.set] = false;
    h->cv.notify_all();
  }
}

// Host arrays of submitted solves.  A handle accepts one solve more than it keeps in flight (kJobRing), and that solve's
// inputs travel while the solve in front of it iterates: on a stream of their own, into whichever of the two input buffers
// is free, as soon as every older solve is past its load -- so the 1.46 GB of a bench batch (26 ms of PCIe at 57 GB/s) never
// leave the main arena idle.  From pageable memory hipMemcpyAsync returns when the copy is done, which is why a thread of its
// own issues it.  Behind the upload the same thread clears the caller's dense cost_hist (rows >= n_cost are zero on every host
// path; the live rows arrive packed and are scattered in job_finish).
void worker_io_main(cilqr_solver* h) {
  (void)pthread_setname_np(pthread_self(), "cilqr-transfer");
  (void)hipSetDevice(h->device);
  std::unique_lock<std::mutex> lk(h->mu);
  for (;;) {
    cilqr_job* job = nullptr;
    h->cv.wait(lk, [&] {
      if (h->quit) return true;
      bool older_past_load = true;
      for (int k = 0; k < h->job_count; ++k) {   // oldest first
        cilqr_job& c = h->jobs[(h->job_head + k) % kJobRing];
        if (c.io_busy && !c.io_taken && (c.upload != 1 || older_past_load)) { job = &c; return true; }
        older_past_load = older_past_load && c.past_load;
      }
      return false;
    });
    if (h->quit) return;
    cilqr_job& j = *job;
    j.io_taken = true;
    const bool upload = j.upload == 1;
    if (upload) j.upload = 2;
    lk.unlock();
    int rc = CILQR_OK;
    if (upload) {
      rc = check_problem(h, &j.in);
      if (rc == CILQR_OK) rc = io_streams(h);
      int buf = -1;
      if (rc == CILQR_OK) {
        buf = acquire_in_buffer(h, true);
        if (buf < 0) rc = CILQR_ERR_DEVICE;
      }
      if (rc == CILQR_OK) rc = stage_inputs(h, &j.in, &h->in_bufs[buf], h->stream_in, &j.pv);
      if (rc == CILQR_OK && hipEventRecord(h->in_bufs[buf].ready, h->stream_in) != hipSuccess) rc = CILQR_ERR_DEVICE;
      if (rc != CILQR_OK && buf >= 0) {   // nothing will read it
        (void)hipStreamSynchronize(h->stream_in);
        release_in_buffer(h, buf, h->stream_in);
        buf = -1;
      }
      lk.lock();
      j.in_buf = buf;
      j.upload_rc = rc;
      if (rc != CILQR_OK) std::snprintf(j.err_text, sizeof(j.err_text), "%s", g_last_hip_error);
      j.upload = (rc == CILQR_OK) ? 3 : -1;
      h->cv.notify_all();
      lk.unlock();
    }
    if (j.zero_by_io && j.out.cost_hist != nullptr)
      std::memset(j.out.cost_hist, 0, (size_t)j.in.batch * ((size_t)h->cfg.max_iter + 1) * 5 * 8);
    lk.lock();
    j.io_busy = false;
    h->cv.notify_all();
  }
}

}  // namespace

extern "C" {

int cilqr_submit(cilqr_handle h, const cilqr_problem_batch* in, cilqr_solution_batch* out) {
  if (h == nullptr || in == nullptr || out == nullptr) return CILQR_ERR_NULL;
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->job_count >= kJobRing) return CILQR_ERR_STATE;   // two in flight and one queued: collect the oldest first (cilqr_wait)
  if (!h->workers_started) {
    h->worker1 = std::thread(worker1_main, h);
    h->worker2 = std::thread(worker2_main, h);
    h->worker_io = std::thread(worker_io_main, h);
    h->workers_started = true;
  }
  const int slot = (h->job_head + h->job_count) % kJobRing;
  cilqr_job& j = h->jobs[slot];
  j.in = *in;
  j.out = *out;
  j.set = 0;            // taken when the first stage starts (worker1_main)
  j.in_buf = -1;
  j.past_load = false;
  j.spec_threshold = h->alone_on_device ? h->spec_threshold : h->spec_threshold_submit;
  j.tail_threshold = h->alone_on_device ? h->tail_threshold : h->tail_threshold_submit;
  j.st1 = h->stream;
  j.st2 = h->stream2;
  j.relaxed_wait = true;     // a worker thread waits for this solve, and other solves want the cores (wait_event)
  // host arrays of a large batch: the transfer thread starts on them now (worker_io_main)
  const bool plain = in->n_lane_groups <= 1 && in->batch > 0 && in->batch <= h->capacity;
  j.upload = (plain && in->memory == CILQR_MEM_HOST && check_problem(h, in) == CILQR_OK &&
              input_payload_bytes(h, in) > kSmallTransfer) ? 1 : 0;
  j.zero_by_io = plain && out->cost_hist != nullptr && host_out_is_big(h, in->batch, out);
  j.zero_inline = !j.zero_by_io;
  j.io_busy = j.upload == 1 || j.zero_by_io;
  j.io_taken = false;
  j.upload_rc = CILQR_OK;
  j.rc = CILQR_OK;
  j.err_text[0] = 0;
  j.phase = 1;
  h->job_count += 1;
  h->cv.notify_all();
  return CILQR_OK;
}

int cilqr_wait(cilqr_handle h) {
  if (h == nullptr) return CILQR_ERR_NULL;
  std::unique_lock<std::mutex> lk(h->mu);
  if (h->job_count == 0) return CILQR_ERR_STATE;
  cilqr_job& j = h->jobs[h->job_head];
  h->cv.wait(lk, [&] { return j.phase == 5; });
  const int rc = j.rc;
  h->prof = j.prof;
  if (rc != CILQR_OK) std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s", j.err_text);   // the caller's cilqr_error_string
  j.phase = 0;
  h->job_head = (h->job_head + 1) % kJobRing;
  h->job_count -= 1;
  return rc;
}

// ------------------------------------------------------------------------------------------
// stages
// ------------------------------------------------------------------------------------------
int cilqr_stage_load(cilqr_handle h, const cilqr_problem_batch* in) {
  if (h == nullptr) return CILQR_ERR_NULL;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->job_count != 0) return CILQR_ERR_STATE;   // the main arena belongs to the submitted solves
  }
  h->stage = 0;
  const int rc = do_load(h, in, h->sets[0], &h->ds, h->stream);
  if (rc != CILQR_OK) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->B = in->batch;
  h->stage = 1;
  return CILQR_OK;
}

int cilqr_stage_init_guess(cilqr_handle h) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 1)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  if (h->cfg.init_guess == CILQR_INIT_TRACKER) launch_init_guess_tracker(h->ds, h->tracker, h->B, h->stream);
  else launch_init_guess(h->ds, h->B, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stage = 1 | 2;
  return CILQR_OK;
}

static int to_device(cilqr_solver* h, const void* src, size_t bytes, int memory, void** tmp, const void** dev) {
  *tmp = nullptr;
  if (memory == CILQR_MEM_DEVICE) { *dev = src; return CILQR_OK; }
  HIP_TRY(hipMalloc(tmp, bytes ? bytes : 256));
  HIP_TRY(hipMemcpyAsync(*tmp, src, bytes, hipMemcpyHostToDevice, h->stream));
  *dev = *tmp;
  return CILQR_OK;
}

int cilqr_stage_set_trajectory(cilqr_handle h, const double* X, const double* U, int32_t memory) {
  if (h == nullptr || X == nullptr || U == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 1)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  const int B = h->B, K = h->cfg.n_steps + 1, N = h->cfg.n_steps;
  void *tx = nullptr, *tu = nullptr;
  const void *dx = nullptr, *du = nullptr;
  int rc = to_device(h, X, (size_t)B * K * 6 * 8, memory, &tx, &dx);
  if (rc == CILQR_OK) rc = to_device(h, U, (size_t)B * N * 2 * 8, memory, &tu, &du);
  if (rc == CILQR_OK) {
    launch_set_trajectory(h->ds, B, static_cast<const double*>(dx), static_cast<const double*>(du), h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = CILQR_ERR_DEVICE;
  }
  if (tx) (void)hipFree(tx);
  if (tu) (void)hipFree(tu);
  if (rc == CILQR_OK) h->stage = 1 | 2;
  return rc;
}

static int from_device(cilqr_solver* h, double* dst, size_t count, int memory, double** dev, void** tmp) {
  *tmp = nullptr;
  if (memory == CILQR_MEM_DEVICE) { *dev = dst; return CILQR_OK; }
  HIP_TRY(hipMalloc(tmp, count ? count * 8 : 256));
  *dev = static_cast<double*>(*tmp);
  (void)h;
  return CILQR_OK;
}
static int finish_from_device(cilqr_solver* h, double* dst, size_t count, int memory, void* tmp) {
  int rc = CILQR_OK;
  if (memory == CILQR_MEM_HOST && tmp) {
    if (hipMemcpyAsync(dst, tmp, count * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = CILQR_ERR_DEVICE;
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess) rc = CILQR_ERR_DEVICE;
  if (tmp) (void)hipFree(tmp);
  if (rc == CILQR_OK && hipGetLastError() != hipSuccess) rc = CILQR_ERR_DEVICE;
  return rc;
}

int cilqr_stage_total_cost(cilqr_handle h, double* cost5, int32_t memory) {
  if (h == nullptr || cost5 == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 2)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  const int B = h->B;
  double* dev = nullptr;
  void* tmp = nullptr;
  int rc = from_device(h, cost5, (size_t)B * 5, memory, &dev, &tmp);
  if (rc != CILQR_OK) return rc;
  launch_cost_only(h->ds, nullptr, B, 0, h->stream);
  launch_gather_scalar(h->ds.trial, 5, h->ds.Bcap, B, dev, 5, 0, h->stream);
  return finish_from_device(h, cost5, (size_t)B * 5, memory, tmp);
}

int cilqr_stage_quadratize(cilqr_handle h) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 2)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  launch_quadratize(h->ds, nullptr, h->B, 0, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stage |= 4;
  return CILQR_OK;
}

int cilqr_stage_backward(cilqr_handle h, const double* lambda, int32_t memory) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 4)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  const double* dl = nullptr;
  if (lambda != nullptr) {
    if (memory == CILQR_MEM_HOST) {
      HIP_TRY(hipMemcpyAsync(h->lambda_stage, lambda, (size_t)h->B * 8, hipMemcpyHostToDevice, h->stream));
      dl = h->lambda_stage;
    } else {
      dl = lambda;
    }
  }
  launch_backward(h->ds, nullptr, h->B, dl, h->team_threshold, h->wave_threshold, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stage |= 8;
  return CILQR_OK;
}

int cilqr_stage_forward(cilqr_handle h, double alpha) {
  if (h == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 8)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  launch_forward(h->ds, nullptr, h->B, alpha, 0, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CILQR_OK;
}

int cilqr_stage_read(cilqr_handle h, int32_t tensor, double* dst, int32_t memory) {
  if (h == nullptr || dst == nullptr) return CILQR_ERR_NULL;
  if (!(h->stage & 1)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  const DeviceState& d = h->ds;
  const int B = h->B, N = h->cfg.n_steps, K = N + 1, Bc = d.Bcap;
  size_t count = 0;
  switch (tensor) {
    case CILQR_T_GOALS: count = (size_t)B * K * 6; break;
    case CILQR_T_CORRIDOR: count = (size_t)B * K * d.cmax * 3; break;
    case CILQR_T_LANES: count = (size_t)(d.nl + d.nr) * 3; break;
    case CILQR_T_X: case CILQR_T_XCAND: case CILQR_T_LX: count = (size_t)B * K * 6; break;
    case CILQR_T_U: case CILQR_T_UCAND: case CILQR_T_LU: case CILQR_T_KFF: count = (size_t)B * N * 2; break;
    case CILQR_T_A: count = (size_t)B * N * 36; break;
    case CILQR_T_B: case CILQR_T_KFB: count = (size_t)B * N * 12; break;
    case CILQR_T_LXX: count = (size_t)B * K * 36; break;
    case CILQR_T_LUU: count = (size_t)B * N * 4; break;
    case CILQR_T_DV: count = (size_t)B * 2; break;
    case CILQR_T_GNORM: count = (size_t)B; break;
    default: return CILQR_ERR_ARG;
  }
  if (tensor >= CILQR_T_X && tensor <= CILQR_T_UCAND && !(h->stage & 2)) return CILQR_ERR_STATE;
  if (tensor >= CILQR_T_A && tensor <= CILQR_T_LUU && !(h->stage & 4)) return CILQR_ERR_STATE;
  if (tensor >= CILQR_T_KFB && !(h->stage & 8)) return CILQR_ERR_STATE;
  if (tensor == CILQR_T_LANES) {
    std::vector<double> tab((size_t)(d.nl + d.nr) * kLaneFields);
    HIP_TRY(hipMemcpyAsync(tab.data(), d.lanes, tab.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<double> abc((size_t)(d.nl + d.nr) * 3);
    for (int s = 0; s < d.nl + d.nr; ++s)
      for (int e = 0; e < 3; ++e) abc[(size_t)s * 3 + e] = tab[(size_t)s * kLaneFields + e];
    if (memory == CILQR_MEM_HOST) std::memcpy(dst, abc.data(), abc.size() * 8);
    else HIP_TRY(hipMemcpy(dst, abc.data(), abc.size() * 8, hipMemcpyHostToDevice));
    return CILQR_OK;
  }
  double* dev = nullptr;
  void* tmp = nullptr;
  int rc = from_device(h, dst, count, memory, &dev, &tmp);
  if (rc != CILQR_OK) return rc;
  hipStream_t st = h->stream;
  switch (tensor) {
    case CILQR_T_GOALS: launch_gather_pairs(d.goals, K * 3, Bc, B, dev, K * 6, 0, st); break;
    case CILQR_T_CORRIDOR: launch_gather_scalar(d.cor, K * d.cmax * 3, Bc, B, dev, K * d.cmax * 3, 0, st); break;
    case CILQR_T_X: launch_gather_xu(d, B, 0, dev, nullptr, st); break;
    case CILQR_T_U: launch_gather_xu(d, B, 0, nullptr, dev, st); break;
    case CILQR_T_XCAND: launch_gather_xu(d, B, 1, dev, nullptr, st); break;
    case CILQR_T_UCAND: launch_gather_xu(d, B, 1, nullptr, dev, st); break;
    case CILQR_T_DV: launch_gather_scalar(d.dV, 2, Bc, B, dev, 2, 0, st); break;
    case CILQR_T_GNORM: launch_gather_scalar(d.gnorm, 1, Bc, B, dev, 1, 0, st); break;
    default: launch_expand(d, B, tensor, dev, st); break;
  }
  return finish_from_device(h, dst, count, memory, tmp);
}

int cilqr_stage_nearest_lane(cilqr_handle h, int32_t n, const double* xy, int32_t* left, int32_t* right,
                             int32_t use_grid, int32_t memory) {
  if (h == nullptr || xy == nullptr || left == nullptr || right == nullptr) return CILQR_ERR_NULL;
  if (n <= 0) return CILQR_ERR_ARG;
  if (!(h->stage & 1)) return CILQR_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  void *t0 = nullptr, *tl = nullptr;
  const void* dxy = nullptr;
  int rc = to_device(h, xy, (size_t)n * 2 * 8, memory, &t0, &dxy);
  int* dl = left; int* dr = right;
  if (rc == CILQR_OK && memory == CILQR_MEM_HOST) {
    if (hipMalloc(&tl, (size_t)n * 2 * sizeof(int)) != hipSuccess) rc = CILQR_ERR_DEVICE;
    dl = static_cast<int*>(tl); dr = dl + n;
  }
  if (rc == CILQR_OK) {
    launch_nearest_lane(h->ds, n, static_cast<const double*>(dxy), dl, dr, use_grid, h->stream);
    if (memory == CILQR_MEM_HOST) {
      if (hipMemcpyAsync(left, dl, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
          hipMemcpyAsync(right, dr, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess)
        rc = CILQR_ERR_DEVICE;
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = CILQR_ERR_DEVICE;
  }
  if (t0) (void)hipFree(t0);
  if (tl) (void)hipFree(tl);
  return rc;
}

// ------------------------------------------------------------------------------------------
// corridor producer
// ------------------------------------------------------------------------------------------
void cilqr_default_corridor_config(cilqr_corridor_config* c) {
  if (c == nullptr) return;
  c->max_diff_x = 25.0; c->max_diff_y = 25.0; c->radius = 150.0;   // planner_config.h:77-79
  c->max_axis_x = 10.0; c->max_axis_y = 10.0;                      // planner_config.h:81-82
  c->lane_segment_length = 5.0;                                    // planner_config.h:85
  c->is_multiple_sample = 0;                                       // planner_config.h:76
  c->reserved0 = 0;
}

int cilqr_build_corridors(cilqr_handle h, const cilqr_corridor_config* cfg, int32_t batch, int32_t n_knots,
                          const double* knots, const double* points, const int32_t* point_count,
                          int32_t max_points, double* corridor, int32_t* corridor_count, int32_t cmax,
                          int32_t memory, int32_t* n_failed, double* polygons) {
  if (h == nullptr || cfg == nullptr || knots == nullptr || point_count == nullptr || corridor == nullptr ||
      corridor_count == nullptr)
    return CILQR_ERR_NULL;                                          // corridor.cc:29-35
  if (points == nullptr && max_points > 0) return CILQR_ERR_NULL;
  if (batch <= 0 || n_knots <= 0 || cmax < 3 || max_points < 0) return CILQR_ERR_ARG;   // empty trajectory cc:24-27
  if (max_points + (cfg->is_multiple_sample ? 24 : 8) > kCorMaxPts) return CILQR_ERR_CAPACITY;
  if (memory != CILQR_MEM_HOST && memory != CILQR_MEM_DEVICE) return CILQR_ERR_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)batch * n_knots;
  const size_t b_knots = n * 3 * 8, b_pts = n * (size_t)max_points * 2 * 8, b_cnt = n * 4;
  const size_t b_cor = n * (size_t)cmax * 3 * 8, b_poly = polygons ? n * (size_t)cmax * 2 * 8 : 0;
  CorridorParams cp{cfg->max_diff_x, cfg->max_diff_y, cfg->radius, cfg->max_axis_x, cfg->max_axis_y,
                    cfg->is_multiple_sample ? 6 : 2};
  void *t_in = nullptr, *t_out = nullptr;
  int rc = CILQR_OK;
  // the failure counter and its landing place on the host belong to the handle: a hipMalloc / hipFree per call is a
  // device-wide synchronisation, i.e. a producer that runs beside solves in flight (other handles, a pool) would wait for
  // all of them
  if (h->cor_fail == nullptr) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->cor_fail), 256));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->cor_fail_host), 64, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&h->cor_done, hipEventDisableTiming));
  }
  int* t_fail = h->cor_fail;
  // (a producer beside solves in flight -- bench.py: end_to_end -- runs on the handle's own stream at normal priority: on a
  // low-priority stream the call took 25 ms instead of 16.5 and the pipeline lost 1.5 %, r06 log 7)
  hipStream_t cst = h->stream;
  const double *d_knots = knots, *d_pts = points;
  const int* d_cnt = point_count;
  double* d_cor = corridor;
  double* d_poly = polygons;
  int* d_ccnt = corridor_count;
  if (hipMemsetAsync(t_fail, 0, 4, cst) != hipSuccess) rc = CILQR_ERR_DEVICE;
  if (rc == CILQR_OK && memory == CILQR_MEM_HOST) {
    const size_t o_pts = (b_knots + 255) / 256 * 256, o_cnt = o_pts + (b_pts + 255) / 256 * 256;
    if (hipMalloc(&t_in, o_cnt + b_cnt + 256) != hipSuccess || hipMalloc(&t_out, b_cor + 512 + b_cnt + b_poly) != hipSuccess) {
      rc = CILQR_ERR_DEVICE;
    } else {
      char* bi = static_cast<char*>(t_in);
      char* bo = static_cast<char*>(t_out);
      if (hipMemcpyAsync(bi, knots, b_knots, hipMemcpyHostToDevice, cst) != hipSuccess ||
          (b_pts && hipMemcpyAsync(bi + o_pts, points, b_pts, hipMemcpyHostToDevice, cst) != hipSuccess) ||
          hipMemcpyAsync(bi + o_cnt, point_count, b_cnt, hipMemcpyHostToDevice, cst) != hipSuccess)
        rc = CILQR_ERR_DEVICE;
      d_knots = reinterpret_cast<const double*>(bi);
      d_pts = reinterpret_cast<const double*>(bi + o_pts);
      d_cnt = reinterpret_cast<const int*>(bi + o_cnt);
      d_cor = reinterpret_cast<double*>(bo);
      d_ccnt = reinterpret_cast<int*>(bo + (b_cor + 255) / 256 * 256);
      if (polygons) d_poly = reinterpret_cast<double*>(bo + (b_cor + 255) / 256 * 256 + (b_cnt + 255) / 256 * 256);
    }
  }
  int failed = 0;
  if (rc == CILQR_OK) {
    launch_build_corridors((int)n, cp, d_knots, d_pts, d_cnt, max_points, d_cor, d_ccnt, cmax, t_fail, d_poly, cst);
    if (hipGetLastError() != hipSuccess) rc = CILQR_ERR_DEVICE;
    if (rc == CILQR_OK && memory == CILQR_MEM_HOST) {
      if (hipMemcpyAsync(corridor, d_cor, b_cor, hipMemcpyDeviceToHost, cst) != hipSuccess ||
          hipMemcpyAsync(corridor_count, d_ccnt, b_cnt, hipMemcpyDeviceToHost, cst) != hipSuccess ||
          (polygons && hipMemcpyAsync(polygons, d_poly, b_poly, hipMemcpyDeviceToHost, cst) != hipSuccess))
        rc = CILQR_ERR_DEVICE;
    }
    if (rc == CILQR_OK &&
        hipMemcpyAsync(h->cor_fail_host, t_fail, 4, hipMemcpyDeviceToHost, cst) != hipSuccess)
      rc = CILQR_ERR_DEVICE;
    // a large batch is milliseconds of kernel time: the caller's thread naps through it instead of spinning (it usually has
    // solves in flight whose worker threads want the cores); a small one is waited for the short way
    if (rc == CILQR_OK && n >= (size_t)1 << 18) {
      if (hipEventRecord(h->cor_done, cst) != hipSuccess || wait_event(h->cor_done, true) != CILQR_OK) rc = CILQR_ERR_DEVICE;
    } else if (hipStreamSynchronize(cst) != hipSuccess) {
      rc = CILQR_ERR_DEVICE;
    }
    if (rc == CILQR_OK) failed = *h->cor_fail_host;
  }
  if (t_in) (void)hipFree(t_in);
  if (t_out) (void)hipFree(t_out);
  if (n_failed) *n_failed = failed;
  return rc;
}

int cilqr_lane_constraints(const double* boundary, int32_t n, double segment_length, int32_t is_left,
                           double* rows, int32_t max_rows) {
  if (boundary == nullptr || rows == nullptr) return CILQR_ERR_NULL;
  if (n < 1 || max_rows < 1) return CILQR_ERR_ARG;
  // LaneBoundarySample corridor.cc:298-311: keep a point once it is a segment length from the last kept one
  int m = 0;              // rows written
  double lx = boundary[0], ly = boundary[1];
  for (int i = 0; i < n; ++i) {
    const double x = boundary[2 * i], y = boundary[2 * i + 1];
    if (std::hypot(x - lx, y - ly) >= segment_length - 1e-10) {
      if (m >= max_rows) return CILQR_ERR_CAPACITY;
      // Cal{Left,Right}LaneConstraints cc:265-296: the left barrier runs from the new point back to
      // the previous one, the right barrier forward; HalfPlaneConstraint cc:313-321
      const double ax = is_left ? x : lx, ay = is_left ? y : ly;
      const double bx = is_left ? lx : x, by = is_left ? ly : y;
      const double a = by - ay, b = -(bx - ax);
      double* r = rows + 7 * (size_t)m;
      r[0] = a; r[1] = b; r[2] = a * ax + b * ay;
      r[3] = ax; r[4] = ay; r[5] = bx; r[6] = by;
      ++m;
      lx = x; ly = y;
    }
  }
  if (m < 1) return CILQR_ERR_CONSTRAINTS;   // fewer than two sampled points  cc:273-275
  return m;
}

int cilqr_device_math(cilqr_handle h, int32_t fn, int32_t n, const double* in, double* out) {
  if (h == nullptr || in == nullptr || out == nullptr) return CILQR_ERR_NULL;
  if (n <= 0 || fn < 0 || fn > 10) return CILQR_ERR_ARG;
  HIP_TRY(hipSetDevice(h->device));
  double* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 16) != hipSuccess) return CILQR_ERR_DEVICE;
  int rc = CILQR_OK;
  if (hipMemcpyAsync(d, in, (size_t)n * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = CILQR_ERR_DEVICE;
  if (rc == CILQR_OK) {
    launch_device_math(fn, n, d, d + n, h->stream);
    if (hipMemcpyAsync(out, d + n, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
      rc = CILQR_ERR_DEVICE;
  }
  (void)hipFree(d);
  return rc;
}

int cilqr_open_loop_rollout(cilqr_handle h, int32_t batch, const double* x0, const double* U, double* X,
                            int32_t memory) {
  if (h == nullptr || x0 == nullptr || U == nullptr || X == nullptr) return CILQR_ERR_NULL;
  if (batch <= 0) return CILQR_ERR_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const int N = h->cfg.n_steps, K = N + 1;
  void *t0 = nullptr, *tu = nullptr, *tx = nullptr;
  const void *d0 = nullptr, *du = nullptr;
  double* dx = nullptr;
  int rc = to_device(h, x0, (size_t)batch * 6 * 8, memory, &t0, &d0);
  if (rc == CILQR_OK) rc = to_device(h, U, (size_t)batch * N * 2 * 8, memory, &tu, &du);
  if (rc == CILQR_OK) rc = from_device(h, X, (size_t)batch * K * 6, memory, &dx, &tx);
  if (rc == CILQR_OK) {
    launch_rollout(h->ds.p, batch, static_cast<const double*>(d0), static_cast<const double*>(du), dx, h->stream);
    rc = finish_from_device(h, X, (size_t)batch * K * 6, memory, tx);
    tx = nullptr;
  }
  if (t0) (void)hipFree(t0);
  if (tu) (void)hipFree(tu);
  if (tx) (void)hipFree(tx);
  return rc;
}

}  // extern "C"
