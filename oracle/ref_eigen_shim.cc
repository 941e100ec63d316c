// oracle/_ref, second library: the reference's vehicle model and barrier functions THEMSELVES, for an image that
// has Eigen 3.4 (this one does not: the target `ref_eigen` of oracle/Makefile does nothing unless an Eigen tree is
// found, and nothing here stands in for Eigen).  algorithm/ilqr/vehicle_model.cc and barrier_function.h need Eigen
// only -- unlike ilqr_optimizer.cc (ROS, OpenCV) -- so with Eigen present they build from the reference's own files
// with g++ alone, and tests/test_reference_pins.py then holds
//   oracle_dynamics / oracle_dynamics_jacobian                   against VehicleModel::Dynamics / DynamicsJacbian  (vm:20-124)
//   oracle_barrier_value / _jacobian / _hessian                  against RelaxBarrierFunction<N>                   (bf:80-146)
// bit for bit (the first check to run when Eigen appears; 8(a)-3, -4, -6).  Test infrastructure only.
// This file is the only code of this repository in that library; it has never been compiled here (no Eigen).
#include <Eigen/Eigen>

#include "algorithm/ilqr/barrier_function.h"
#include "algorithm/ilqr/vehicle_model.h"

namespace {
template <std::size_t N>
void barrier_jacobian(double t, double eps, double g, const double* dg, double* out) {
  planning::RelaxBarrierFunction<N> b;
  b.SetParam(t);
  b.SetEpsilon(eps);
  Eigen::Matrix<double, N, 1> d;
  for (std::size_t i = 0; i < N; ++i) d(i, 0) = dg[i];
  const Eigen::Matrix<double, N, 1> r = b.Jacbian(g, d);
  for (std::size_t i = 0; i < N; ++i) out[i] = r(i, 0);
}
template <std::size_t N>
void barrier_hessian(double t, double eps, double g, const double* dg, const double* ddg, double* out) {
  planning::RelaxBarrierFunction<N> b;
  b.SetParam(t);
  b.SetEpsilon(eps);
  Eigen::Matrix<double, N, 1> d;
  Eigen::Matrix<double, N, N> dd = Eigen::Matrix<double, N, N>::Zero();
  for (std::size_t i = 0; i < N; ++i) d(i, 0) = dg[i];
  if (ddg)
    for (std::size_t i = 0; i < N; ++i)
      for (std::size_t j = 0; j < N; ++j) dd(i, j) = ddg[i * N + j];
  const Eigen::Matrix<double, N, N> r = ddg ? b.Hessian(g, d, dd) : b.Hessian(g, d);
  for (std::size_t i = 0; i < N; ++i)
    for (std::size_t j = 0; j < N; ++j) out[i * N + j] = r(i, j);
}
}  // namespace

extern "C" {

// VehicleModel with the reference's default IlqrConfig; the vehicle's wheel base is the one parameter it reads
void* ref_model_create(double wheel_base, double horizon, double dt) {
  planning::IlqrConfig config;
  planning::VehicleParam param;
  param.wheel_base = wheel_base;
  return new planning::VehicleModel(config, param, horizon, dt);
}
void ref_model_destroy(void* m) { delete static_cast<planning::VehicleModel*>(m); }

void ref_dynamics(void* m, const double* x, const double* u, double* xn) {            // vm:88-124
  planning::State s, n;
  planning::Control c;
  for (int i = 0; i < 6; ++i) s(i, 0) = x[i];
  c(0, 0) = u[0];
  c(1, 0) = u[1];
  static_cast<planning::VehicleModel*>(m)->Dynamics(s, c, &n);
  for (int i = 0; i < 6; ++i) xn[i] = n(i, 0);
}
// A: 6 x 6, B: 6 x 2, row-major
void ref_dynamics_jacobian(void* m, const double* x, const double* u, double* A, double* B) {   // vm:20-86
  planning::State s;
  planning::Control c;
  for (int i = 0; i < 6; ++i) s(i, 0) = x[i];
  c(0, 0) = u[0];
  c(1, 0) = u[1];
  planning::SystemMatrix a;
  planning::InputMatrix b;
  static_cast<planning::VehicleModel*>(m)->DynamicsJacbian(s, c, &a, &b);
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) A[i * 6 + j] = a(i, j);
    for (int j = 0; j < 2; ++j) B[i * 2 + j] = b(i, j);
  }
}

double ref_barrier_value(double t, double eps, double g) {                              // bf:104-113
  planning::RelaxBarrierFunction<6> b;
  b.SetParam(t);
  b.SetEpsilon(eps);
  return b.value(g);
}
// n = 6 (state constraints) or 2 (control constraints); ddg may be null (the reference's default argument)
void ref_barrier_jacobian(double t, double eps, double g, const double* dg, int n, double* out) {   // bf:115-125
  if (n == 6) barrier_jacobian<6>(t, eps, g, dg, out);
  else barrier_jacobian<2>(t, eps, g, dg, out);
}
void ref_barrier_hessian(double t, double eps, double g, const double* dg, const double* ddg, int n, double* out) {   // bf:127-140
  if (n == 6) barrier_hessian<6>(t, eps, g, dg, ddg, out);
  else barrier_hessian<2>(t, eps, g, dg, ddg, out);
}

}  // extern "C"
