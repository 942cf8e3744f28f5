// dual.hpp -- forward-mode AD number with ONE directional derivative.
//
// The reference differentiates the residual with Sacado SFad<W>: every solution field at an integration point
// carries W = dofs-per-element derivative slots (src/preferences.hpp:57-75).  On the GPU the chain rule is split:
//   d res_i / d u_j = sum_q sum_{k,m} T_k(i,q) * [dF_k/dU_m](q) * T_m(j,q)
// where U are the solution fields at the point (values, gradients, divergence, time derivatives), F the point
// fluxes the physics module multiplies with the test-function slots T_k, and dF/dU a small dense matrix per point.
// Each column of dF/dU is one directional derivative of the module's point function, evaluated with this type by a
// different thread -- Sacado-style forward AD of width 1, parallel over directions instead of over an array.
#pragma once
#include <hip/hip_runtime.h>

namespace mha {

struct Dual {
  double v, d;
};

__host__ __device__ __forceinline__ Dual mk(double v, double d = 0.0) { return Dual{v, d}; }
__host__ __device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__host__ __device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__host__ __device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__host__ __device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__host__ __device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const double r = 1.0 / b.v, q = a.v * r;
  return {q, (a.d - q * b.d) * r};
}
__host__ __device__ __forceinline__ Dual operator+(Dual a, double b) { return {a.v + b, a.d}; }
__host__ __device__ __forceinline__ Dual operator-(Dual a, double b) { return {a.v - b, a.d}; }
__host__ __device__ __forceinline__ Dual operator*(Dual a, double b) { return {a.v * b, a.d * b}; }
__host__ __device__ __forceinline__ Dual operator*(double b, Dual a) { return {a.v * b, a.d * b}; }
__host__ __device__ __forceinline__ Dual operator/(Dual a, double b) { const double r = 1.0 / b; return {a.v * r, a.d * r}; }
__host__ __device__ __forceinline__ Dual operator/(double a, Dual b) {
  const double r = 1.0 / b.v, q = a * r;
  return {q, -q * b.d * r};
}
__host__ __device__ __forceinline__ Dual &operator+=(Dual &a, Dual b) { a.v += b.v; a.d += b.d; return a; }
__host__ __device__ __forceinline__ Dual &operator-=(Dual &a, Dual b) { a.v -= b.v; a.d -= b.d; return a; }
__host__ __device__ __forceinline__ Dual dsqrt(Dual a) {
  const double s = sqrt(a.v);
  return {s, a.d / (2.0 * s)};
}

}  // namespace mha
