// Private to cilqr_amd/csrc: the solver handle behind `cilqr_handle` (include/cilqr.h) and the
// error-reporting macro shared by the translation units that implement the C-ABI.
#pragma once
#include <atomic>

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/cilqr.h"
#include "state.hpp"

extern thread_local char g_last_hip_error[256];

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s -> %s", #expr, \
                    hipGetErrorString(e_));                                        \
      return CILQR_ERR_DEVICE;                                                     \
    }                                                                              \
  } while (0)

struct cilqr_comm;   // comm.hip: RCCL communicator + staging of cilqr_gather_results

// Entry points that create or destroy handles on SEVERAL devices (cilqr_multi_*, cilqr_pool_*) leave the calling thread's
// current HIP device as they found it.
struct cilqr_device_guard {
  int dev = -1;
  cilqr_device_guard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~cilqr_device_guard() { if (dev >= 0) (void)hipSetDevice(dev); }
  cilqr_device_guard(const cilqr_device_guard&) = delete;
  cilqr_device_guard& operator=(const cilqr_device_guard&) = delete;
};

// What ONE solve in flight owns besides the arenas: the tensors indexed by PROBLEM (they outlive the hand-over of
// the survivors to the finishing arena and are read by the final export), the lane tables it was loaded with,
// its host-visible iteration counters and its events.  A handle has two of these, so that the finishing stage of
// solve i and the first stage of solve i+1 can be in flight together.
struct cilqr_job_set {
  double* hist = nullptr;
  int *iter = nullptr, *status = nullptr, *n_cost = nullptr, *n_iter_trajs = nullptr;
  signed char* atrace = nullptr;
  double* lanes = nullptr;
  unsigned char* lgrid = nullptr;
  double* lanes_raw = nullptr;      // device [2*smax][7]
  std::vector<double> lane_cache;   // the lane tables whose device image and grid are current (left rows, then right rows)
  int lane_cache_nl = -1, lane_cache_nr = -1;
  int* h_count = nullptr;           // pinned, written by k_update through h_count_dev
  int* h_count_dev = nullptr;
  int* tail_iter_dev = nullptr;     // largest iteration count reached inside the tail kernel
  // host arrays (CILQR_MEM_HOST) on the way out: the staging of ONE solve in flight (the way in: cilqr_in_buffer)
  void* out_stage = nullptr;        // problem-major results on the device when the caller's buffers are host memory
  size_t out_stage_bytes = 0;
  void* out_pinned = nullptr;       // small host batches: the staging block lands here in one copy
  long long* row_off = nullptr;     // device [B + 1]: first packed Cost row of every problem, total (large host batches)
  size_t row_off_bytes = 0;
  void* host_counts = nullptr;      // pinned: n_cost | status | n_iter | n_iter_trajs of a large host batch
  size_t host_counts_bytes = 0;
  void* host_rows = nullptr;        // pinned: its LIVE Cost rows, packed (scattered into the caller's dense array on the host)
  size_t host_rows_bytes = 0;
  hipEvent_t exported = nullptr;    // recorded on the solve's stream behind its last kernel (what the download stream waits for)
  std::vector<hipEvent_t> iter_ev;  // one per lockstep iteration (count read-back)
  std::vector<hipEvent_t> ev;       // profiling
  hipEvent_t handoff = nullptr;     // survivors copied into the finishing arena (recorded on the first stage's stream)
  hipEvent_t sync_ev = nullptr;     // what a relaxed host wait for a whole stream polls (solver.hip: wait_stream)
};

struct cilqr_timer {  // event pairs around kernels / phases, resolved after the final sync
  cilqr_solver* h = nullptr;
  cilqr_job_set* js = nullptr;
  hipStream_t stream = nullptr;     // where the next events are recorded (changes at the hand-over)
  size_t next = 0;
  std::vector<int> kind;  // 0 quad, 1 backward, 2 linesearch, 3 other, 4 tail
  std::vector<char> full_flags, live_flags;  // per backward launch: covered the whole batch / had work
  bool open = false;   // the last begin() recorded an event, so the matching end() must too
  bool marked = false; // the last begin() opened a roctx range (CILQR_ROCTX=1), so the matching end() closes it
  bool on() const;
  bool wants(int k) const;
  int reserve();
  int begin(int k);
  int pair(int k, hipEvent_t* a, hipEvent_t* b);
  int end();
  void resolve(cilqr_profile* p);
};

// Host arrays on the way in: a problem-major copy of the caller's inputs on the device.  A buffer belongs to a solve from the
// start of its upload until its load kernels are enqueued; two of them, so that the arrays of the NEXT solve travel while this
// one iterates (solver.hip: worker_io_main).
struct cilqr_in_buffer {
  void* p = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr;    // recorded on the upload stream behind the last input copy
  hipEvent_t loaded = nullptr;   // recorded on the solve's stream behind the load kernels that read the buffer
  int state = 0;                 // 0 free; 1 owned by a solve; 2 given back: free once `loaded` has happened
};

// Solves a handle accepts before the oldest is collected: two in flight (first stage / finishing stage) and one more, queued,
// whose host arrays are uploaded meanwhile -- without it the main arena would sit idle for the length of an upload between
// two solves (26 ms of a 33 ms step on the bench workload, measured: 1.40 M solves/s from host arrays against 1.98 M).
constexpr int kJobRing = 3;

// One solve on its way through the handle: first stage (load, init guess, the lockstep iterations over the bulk of
// the batch, in the main arena), hand-over of the survivors, finishing stage (remaining lockstep iterations + the
// per-problem tail kernel in the small finishing arena, final export).
struct cilqr_job {
  cilqr_problem_batch in;
  cilqr_solution_batch out;
  int set = 0;              // the cilqr_job_set it runs on: taken when its first stage starts, given back when it is done
  int spec_threshold = 0;   // the threshold of this solve (cilqr_solver::spec_threshold or spec_threshold_submit)
  int tail_threshold = 0;   // likewise (cilqr_solver::tail_threshold or tail_threshold_submit)
  int phase = 0;            // 0 free, 1 queued, 2 first stage, 3 waiting for the finishing stage, 4 finishing, 5 done
  int rc = CILQR_OK;
  char err_text[256] = "";  // what the worker thread's g_last_hip_error held when rc was set (that variable is thread-local)
  hipStream_t st1 = nullptr, st2 = nullptr;
  cilqr::DeviceState gmain;   // main arena with this job's problem-indexed tensors and lane tables
  cilqr::DeviceState d, o;    // the arena the active problems live in, and its twin
  int B = 0, it = 0, n_hint = 0, span = 0;
  bool handed = false;      // the survivors live in the finishing arena
  bool owns_fin = false;    // this solve holds the finishing arena / the tail workspace (cilqr_solver::fin_busy)
  bool tail_used = false;
  int tail_n = 0;
  std::vector<int> bwd_iter;  // iteration index of every profiled backward launch
  cilqr_timer tm;
  cilqr_profile prof;
  // where the kernels write the results (the caller's device buffers, or the staging of the job set)
  double *o_traj = nullptr, *o_hist = nullptr, *o_it = nullptr;
  int *o_nc = nullptr, *o_st = nullptr, *o_ni = nullptr, *o_nit = nullptr;
  signed char* o_at = nullptr;
  size_t n_traj = 0, n_hist = 0, n_itr = 0, n_at = 0;
  size_t n_head = 0;        // bytes of the staging block in front of the iterates (traj, cost_hist, counts, alpha_trace, pad)
  bool small_out = false;   // host outputs small enough to travel through the pinned block
  bool relaxed_wait = false;  // the host waits of this solve poll and nap instead of spinning (solver.hip: wait_event)
  // Host inputs of a submitted solve travel ahead of it: the handle's transfer thread (solver.hip: worker_io_main) copies
  // them into the job set's staging on a stream of its own as soon as the solve is SUBMITTED, i.e. while the solve before it
  // still iterates; the first stage then only waits for the event behind the last copy.  upload: 0 = not delegated (device
  // inputs, a small batch, the synchronous call: staged inline by the solving thread), 1 = queued, 2 = being copied,
  // 3 = every copy enqueued and the event recorded, -1 = failed (upload_rc).
  int upload = 0;
  int upload_rc = CILQR_OK;
  cilqr::ProblemView pv;      // where the staged inputs lie (filled by whoever staged them)
  bool past_load = false;     // its inputs are staged and its load kernels enqueued (or it failed before): the transfer thread may
                              // upload for the solve behind it
  int in_buf = -1;            // the cilqr_in_buffer they lie in (-1: the caller's own device arrays)
  bool io_busy = false;       // the transfer thread still works for this solve (upload, zero fill of the caller's cost_hist)
  bool io_taken = false;      // ... and has picked it up
  bool zero_by_io = false;    // large host outputs of a submitted solve: the transfer thread clears the caller's cost_hist
  bool zero_inline = false;   // large host outputs of a solve without the transfer thread: the solving thread clears cost_hist
  bool big_out = false;       // host outputs that travel ragged: trajectory straight into the caller's array, live Cost rows packed
};

struct cilqr_solver {
  cilqr_config cfg;
  int device = 0;
  int Bcap = 0, capacity = 0, cmax = 0, smax = 0;
  cilqr::DeviceState ds;      // arena A (also what the stage API works on)
  cilqr::DeviceState twin;    // arena B: only the fields k_compact moves are its own, the rest alias ds
  // finishing arena (capacity fin_cap slots) and its twin: where a solve continues once few enough problems are
  // left, so that the main arena is free for the next solve (cilqr_submit) while the stragglers finish
  cilqr::DeviceState fin, fin_twin;
  int fin_cap = 0;
  int fin_threshold = 0;     // hand the survivors over at this active count (CILQR_OPT_FINISH_THRESHOLD); 0 = never
  bool compaction = true;
  int compact_percent = 75;  // re-pack when the survivors fill at most this share of the occupied slots
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // finishing stage of asynchronous solves (high priority: short, latency-bound kernels)
  hipStream_t stream_in = nullptr;   // host arrays: uploads of submitted solves (the transfer thread), beside the solves' kernels
  hipStream_t stream_out = nullptr;  // host arrays: downloads of finished solves, so that the next solve's kernels need not queue behind them
  std::vector<void*> allocs;
  int64_t bytes = 0;
  std::atomic<int64_t> grown_bytes{0};   // staging blocks + tail workspaces, grown by the solving threads (solver.hip: grow)
  void* in_small = nullptr;         // small host batches: their device block (the large ones: in_bufs)
  size_t in_small_bytes = 0;
  void* in_pinned = nullptr;        // small host batches: the input arrays leave from here in one copy
  hipEvent_t in_pinned_ev = nullptr;
  cilqr_job_set sets[2];
  bool set_busy[2] = {false, false};   // (under mu) a submitted solve runs on the set
  cilqr_in_buffer in_bufs[2];
  double* lambda_stage = nullptr;
  int B = 0;               // problems loaded
  int stage = 0;           // bit0 loaded, bit1 iterate, bit2 quadratized, bit3 gains
  int spec_threshold = 8192;  // active sets up to this size evaluate all 11 step sizes at once (cilqr_solve_batch)
  // ... and for solves submitted with cilqr_submit: other solves share the GPU then, and eleven candidates per problem
  // where two or three would do is throughput taken from them (measured: pool of two 1.90 -> 1.98 M solves/s, one handle
  // with two solves in flight 1.67 -> 1.72 M; the sequential call loses 1 % at 2048, hence the two defaults).  Round 5, re-swept
  // beside the tail threshold below after the backward mappings had moved: 1024 with 128 there, pool of two 2.03-2.06 -> 2.07 M on two boxes
  int spec_threshold_submit = 1024;
  bool alone_on_device = false;   // a shard of cilqr_multi_*: its device is the caller's alone, submitted solves take spec_threshold
  int team_threshold = 4096;  // active sets up to this size run the backward pass with 8 lanes per problem
  int round_group = 2;        // step sizes costed per sequential round (1, 2 or 4)
  int wave_threshold = 3072;  // active sets up to this size run the backward pass with a wavefront per problem (1024 until
                              // round 5: re-swept after the wave form had become a third faster, tools/bwd_forms_sweep.py)
  int seq_rounds = 4;         // larger sets: this many round-by-round trials, then the rest at once
  // active sets up to this size leave the lockstep loop: one workgroup per problem (kernels_tail.hip).  Like the speculation
  // threshold it depends on company: a solve that has the GPU to itself (cilqr_solve_batch) is shortest when up to 1024 problems
  // finish there (the kernel then runs four rounds of workgroups: 1.27 -> 1.31 M solves/s); beside other solves those workgroups
  // hold a CU each for milliseconds and take it from the neighbours' bulk kernels (pool 1.97 -> 1.93 M), so submitted solves keep
  // fewer: 256 until round 5, 128 since (profiles/r05_cost_kernel_experiments.txt 6)
  int tail_threshold = 1024;
  int tail_threshold_submit = 128;
  void* tail_ws = nullptr;    // private arenas of the tail's problems (lazily grown)
  size_t tail_ws_bytes = 0;
  void* tail_ws1 = nullptr;   // the same for a solve that reaches the tail without having been handed over (first stage)
  size_t tail_ws1_bytes = 0;
  // asynchronous submit / wait: two jobs in flight, one worker thread per stage
  std::thread worker1, worker2, worker_io;
  std::mutex io_mu;           // creation of stream_in / stream_out (solver.hip: io_streams)
  std::mutex mu;
  std::condition_variable cv;
  bool workers_started = false, quit = false;
  bool fin_busy = false;      // the finishing arena holds a solve
  cilqr_job jobs[kJobRing];   // ring of submitted solves, oldest at job_head
  int job_head = 0;           // oldest job not yet collected by cilqr_wait
  int job_count = 0;          // submitted and not yet collected
  // profiling
  bool profiling = false;
  int profiling_level = 1;
  cilqr_profile prof;         // of the last solve that completed
  cilqr_comm* comm = nullptr;   // multi-GPU results gather (cilqr_comm_create)
  int* cor_fail = nullptr;        // cilqr_build_corridors: failure counter on the device, where it lands on the host, its event
  int* cor_fail_host = nullptr;
  hipEvent_t cor_done = nullptr;
  cilqr::TrackerParams tracker;   // CILQR_INIT_TRACKER
};
