// Backward pass: Riccati-like gain recursion, one lane per problem (the roofline kernel).
//
// Reference behaviour: IlqrOptimizer::Backward (algorithm/ilqr/ilqr_optimizer.cc:334-390) and
// CalGradientNorm (cc:322-332).  Quirks kept:
//   * K, k and the Vx/Vxx updates use the un-regularised Quu except inside the inverse (cc:361-380);
//   * delta_V_ is accumulated from Qu / Quu re-evaluated with the UPDATED Vx / Vxx, because the
//     reference holds them as lazy Eigen `auto` expressions (cc:348-352 vs cc:383-384);
//   * Vxx is symmetrised in place, column-major, without a temporary (cc:381);
//   * products associate left to right; a coefficient of `X.transpose() * Y` (A^T Vx, A^T Vxx, B^T ...) adds its six terms as
//     the reference's SSE2 build does, (t0 + (t2 + t4)) + (t1 + (t3 + t5)); every other dot product runs k = 0..5 in order
//     (dev_model.hpp: sum6_xty).
//
// Memory: per step a lane reads 17 double2 (the non-constant entries of A, B, lx, lu, lxx, luu;
// see state.hpp) + 1 double2 of U, and writes 7 double2 of gains: 18 KiB in + 7 KiB out per
// wave-step, every access 16 B/lane and 1 KiB contiguous per wave; the reads of the next step go straight into LDS
// (global_load_lds_dwordx4) while this step computes.  Structural zeros/ones of A and
// B are skipped at compile time; adding an exact zero never changes an IEEE sum, so the results
// are those of the dense recursion.
#include <hip/hip_ext.h>

#include "backward_core.hpp"

namespace cilqr {

__global__ __launch_bounds__(64) void k_backward_team(DeviceState s, const int* __restrict__ list, int n,
                                                      const double* __restrict__ lambda_override) {
  using namespace team;
  __shared__ double lds[(64 / kLanes) * kStride];
  const int lane = threadIdx.x;
  const int n_act = active_count(s, n);
  const int jraw = blockIdx.x * (64 / kLanes) + lane / kLanes;
  if ((int)(blockIdx.x * (64 / kLanes)) >= n_act) return;   // whole block idle (uniform)
  const bool live = jraw < n_act;
  const int j = live ? jraw : n_act - 1;       // idle teams shadow the last problem, stores masked
  const int slot = list ? list[j] : j;
  const double lambda = lambda_override ? lambda_override[slot] : s.lambda[slot];
  backward_team_problem(s, slot, lambda, live, lane & (kLanes - 1), lds + (lane / kLanes) * kStride, BlockSync{});
}

__global__ __launch_bounds__(64, 1) void k_backward(DeviceState s, const int* __restrict__ list, int n,
                                                     const double* __restrict__ lambda_override) {
  __shared__ double2 stage[2 * (kLinPairs + 1) * 64];   // the next step's operands, double-buffered (backward_core.hpp)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= active_count(s, n)) return;
  const int slot = list ? list[j] : j;
  const double lambda = lambda_override ? lambda_override[slot] : s.lambda[slot];
  backward_problem<true>(s, slot, lambda, stage);
}

// one wavefront per problem: kernels_backward_wave.hip (a file of its own: it is scheduled for ILP, this one is not)
bool launch_backward_wave(const DeviceState& s, const int* list, int n, const double* lambda_override, hipStream_t st,
                          hipEvent_t ev_start, hipEvent_t ev_stop);

// wave_threshold: active sets up to this size give every problem a wavefront; team_threshold: up to this size,
// eight lanes; larger ones, one lane (the HBM-bound form)
void launch_backward(const DeviceState& s, const int* list, int n, const double* lambda_override,
                     int team_threshold, int wave_threshold, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n == 0) return;
  // ev_start / ev_stop (profiling): the kernel's OWN start and end (dispatch timestamps, what rocprofv3 reports) --
  // events recorded around the launch would also count the time the dispatch waits for CUs that other streams hold
  if (n <= wave_threshold && launch_backward_wave(s, list, n, lambda_override, st, ev_start, ev_stop))
    return;
  if (n <= team_threshold)
    hipExtLaunchKernelGGL(k_backward_team, dim3((n + 7) / 8), dim3(64), 0, st, ev_start, ev_stop, 0, s, list, n, lambda_override);
  else
    hipExtLaunchKernelGGL(k_backward, dim3((n + 63) / 64), dim3(64), 0, st, ev_start, ev_stop, 0, s, list, n, lambda_override);
}

}  // namespace cilqr
