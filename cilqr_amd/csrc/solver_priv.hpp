// Private to cilqr_amd/csrc: the solver handle behind `cilqr_handle` (include/cilqr.h) and the
// error-reporting macro shared by the translation units that implement the C-ABI.
#pragma once

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

struct cilqr_solver {
  cilqr_config cfg;
  int device = 0;
  int Bcap = 0, capacity = 0, cmax = 0, smax = 0;
  cilqr::DeviceState ds;      // arena A (also what the stage API works on)
  cilqr::DeviceState twin;    // arena B: only the fields k_compact moves are its own, the rest alias ds
  bool compaction = true;
  int compact_percent = 75;  // re-pack when the survivors fill at most this share of the occupied slots
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  int64_t bytes = 0;
  // staging (lazily grown): problem-major copies of host inputs / outputs on the device
  void* in_stage = nullptr;
  size_t in_stage_bytes = 0;
  void* out_stage = nullptr;
  size_t out_stage_bytes = 0;
  double* lanes_raw = nullptr;  // device [2*smax][7]
  std::vector<double> lane_cache;   // the lane tables whose device image and grid are current (left rows, then right rows)
  int lane_cache_nl = -1, lane_cache_nr = -1;
  double* lambda_stage = nullptr;
  int* h_count = nullptr;  // pinned, written by k_update through h_count_dev
  int* h_count_dev = nullptr;
  int B = 0;               // problems loaded
  int stage = 0;           // bit0 loaded, bit1 iterate, bit2 quadratized, bit3 gains
  int spec_threshold = 8192;  // active sets up to this size evaluate all 11 step sizes at once
  int team_threshold = 4096;  // active sets up to this size run the backward pass with 8 lanes per problem
  int round_group = 2;        // step sizes costed per sequential round (1, 2 or 4)
  int wave_threshold = 1024;  // active sets up to this size run the backward pass with a wavefront per problem
  int seq_rounds = 4;         // larger sets: this many round-by-round trials, then the rest at once
  int tail_threshold = 256;   // active sets up to this size leave the lockstep loop: one workgroup per problem (kernels_tail.hip)
  void* tail_ws = nullptr;    // private arenas of the tail's problems (lazily grown)
  size_t tail_ws_bytes = 0;
  int* tail_iter_dev = nullptr;   // largest iteration count reached inside the tail kernel
  // asynchronous submit / wait: one worker thread per handle, one job in flight
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  bool worker_started = false, job_pending = false, job_done = false, quit = false;
  cilqr_problem_batch job_in;
  cilqr_solution_batch job_out;
  int job_rc = CILQR_OK;
  // profiling
  bool profiling = false;
  int profiling_level = 1;
  std::vector<hipEvent_t> ev;
  std::vector<hipEvent_t> iter_ev;  // one per lockstep iteration (count read-back)
  cilqr_profile prof;
  cilqr_comm* comm = nullptr;   // multi-GPU results gather (cilqr_comm_create)
  cilqr::TrackerParams tracker;   // CILQR_INIT_TRACKER
};
