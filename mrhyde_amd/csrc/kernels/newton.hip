// newton.hip -- the three vector kernels of the Newton driver (newton.hpp): infinity norm, u += alpha du, and the strong-
// Dirichlet lifting u[row] = value on fixed rows (SolverManager::setDirichlet, src/managers/solverManager.cpp:1876-1957).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "launch.hpp"

namespace mha {
namespace {

// |v|_inf into *out (an unsigned 64-bit word holding the bits of a non-negative double: its order is the doubles' order)
__global__ __launch_bounds__(256) void norm_inf_kernel(int64_t n, const double *__restrict__ v, unsigned long long *out) {
  double m = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = fabs(v[i]);
    m = (a > m || a != a) ? a : m;  // a NaN residual must show in the norm
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(m, o);
    m = (t > m || t != t) ? t : m;
  }
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long bits = (m != m) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(m);
    atomicMax(out, bits);
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(int64_t n, double alpha, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void lift_kernel(int64_t n, const uint8_t *__restrict__ fixed, const double *__restrict__ vals,
                                                   double scalar, double *__restrict__ u) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (fixed[i]) u[i] = vals ? vals[i] : scalar;
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

}  // namespace

void launch_norm_inf(int64_t n, const double *v, unsigned long long *out_bits, hipStream_t stream) {
  MHA_HIP(hipMemsetAsync(out_bits, 0, sizeof(unsigned long long), stream));
  if (n <= 0) return;
  hipLaunchKernelGGL(norm_inf_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, v, out_bits);
  MHA_HIP(hipGetLastError());
}

void launch_axpy(int64_t n, double alpha, const double *x, double *y, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, alpha, x, y);
  MHA_HIP(hipGetLastError());
}

void launch_dirichlet_lift(int64_t n, const uint8_t *fixed, const double *vals, double scalar, double *u, hipStream_t stream) {
  if (n <= 0 || !fixed) return;
  hipLaunchKernelGGL(lift_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, fixed, vals, scalar, u);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
