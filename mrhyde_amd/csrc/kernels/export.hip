// export.hip -- pack / unpack of the shared-row Export(ADD) (export_plan.hpp; reference:
// src/interfaces/linearAlgebraInterface.hpp:296-337, Tpetra::Export with ADD combine mode).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "launch.hpp"

namespace mha {
namespace {

__global__ __launch_bounds__(256) void export_pack_kernel(const double *__restrict__ src, const int32_t *__restrict__ idx,
                                                          int64_t n, double *__restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// targets of ONE neighbour are distinct (a value entry / a row arrives once per neighbour): plain read-modify-write
__global__ __launch_bounds__(256) void export_unpack_add_kernel(const double *__restrict__ src, const int32_t *__restrict__ tgt,
                                                                int64_t n, double *__restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t t = tgt[i];
    if (t >= 0) dst[t] += src[i];
  }
}

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

}  // namespace

void launch_export_pack(const double *src, const int32_t *idx, int64_t n, double *dst, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(export_pack_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, idx, n, dst);
  MHA_HIP(hipGetLastError());
}

void launch_export_unpack_add(const double *src, const int32_t *tgt, int64_t n, double *dst, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(export_unpack_add_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, tgt, n, dst);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
