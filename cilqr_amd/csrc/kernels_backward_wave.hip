// Backward pass with one wavefront per problem (backward_core.hpp: backward_wave_problem), for launches of at most
// CILQR_OPT_WAVE_THRESHOLD problems.  Reference behaviour: IlqrOptimizer::Backward (algorithm/ilqr/ilqr_optimizer.cc:334-390),
// CalGradientNorm (cc:322-332); see kernels_backward.hip for the quirks that are kept.
//
// A file of its own because of how it is compiled: a launch this small leaves a wavefront alone on its SIMD, walking N
// dependent steps of five exchange stages each, and the order of the instructions inside a stage is all that fills the
// gaps -- this file is built with LLVM's max-ILP scheduling strategy (Makefile: ILPFLAGS), which the lane-per-problem
// kernel of kernels_backward.hip (HBM-bound, 350 registers) does not gain from.
#include <hip/hip_ext.h>

#include "backward_core.hpp"

namespace cilqr {

// (which list position a workgroup takes: dev_model.hpp, xcd_local_position)
__global__ __launch_bounds__(64) void k_backward_wave(DeviceState s, const int* __restrict__ list, int n,
                                                      const double* __restrict__ lambda_override) {
  extern __shared__ double lds[];   // wave::lds_doubles(N)
  const int j = (gridDim.x & 63u) == 0 ? xcd_local_position((int)blockIdx.x) : (int)blockIdx.x;   // small launches: as they are
  if (j >= active_count(s, n)) return;
  const int slot = list ? list[j] : j;
  const double lambda = lambda_override ? lambda_override[slot] : s.lambda[slot];
  backward_wave_problem(s, slot, lambda, (int)threadIdx.x, lds, WaveSync{});
}

// false: the horizon's per-step rows do not fit the 64 KiB of LDS a launch gets without asking (the caller takes the
// eight-lane kernel instead)
bool launch_backward_wave(const DeviceState& s, const int* list, int n, const double* lambda_override, hipStream_t st,
                          hipEvent_t ev_start, hipEvent_t ev_stop) {
  const size_t lds = wave::lds_doubles(s.p.N) * sizeof(double);
  if (lds > 64 * 1024) return false;
  hipExtLaunchKernelGGL(k_backward_wave, dim3(n >= 64 ? (n + 63) / 64 * 64 : n), dim3(64), lds, st, ev_start, ev_stop, 0, s, list, n, lambda_override);
  return true;
}

}  // namespace cilqr
