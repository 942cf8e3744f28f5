// scatter.hip -- gather / scatter / Dirichlet-diagonal kernels.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

// scatterJac + scatterRes from the dense element arrays
// (reference: src/managers/assemblyManager.cpp:3916-3932, 3960-3977): one lane per (elem,row,col).
__global__ __launch_bounds__(256) void scatter_local_kernel(BlockDev b, const double *__restrict__ local_J,
                                                            const double *__restrict__ local_res,
                                                            double *res, double *crs_vals, int local_base) {
  const int n = b.n;
  const size_t per = (size_t)n * n;
  const size_t total = (size_t)b.e_count * per;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total;
       k += (size_t)gridDim.x * blockDim.x) {
    const size_t idx = k + (size_t)b.e_begin * per;
    const int e = (int)(idx / per);
    const int rc = (int)(idx - (size_t)e * per);
    const int r = rc / n, c = rc - r * n;
    const int32_t *L = b.lids + (size_t)e * n;
    const int row = L[r];
    if (b.fixed && b.fixed[row]) continue;
    if (crs_vals && local_J) {
      const int p = find_col(b.colind, b.rowptr[row], b.rowptr[row + 1], L[c]);
      if (p >= 0) atomicAdd(crs_vals + p, local_J[idx - (size_t)local_base * per]);
    }
    if (c == 0 && res && local_res) atomicAdd(res + row, local_res[(size_t)(e - local_base) * n + r]);
  }
}

// updateJacDBC: diagonal of every fixed row := 1 (reference: assemblyManager.cpp:1166-1179)
__global__ __launch_bounds__(256) void dbc_diag_kernel(BlockDev b, double *crs_vals) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < b.nrows; row += gridDim.x * blockDim.x) {
    if (!b.fixed[row]) continue;
    const int p = find_col(b.colind, b.rowptr[row], b.rowptr[row + 1], row);
    if (p >= 0) crs_vals[p] = 1.0;
  }
}

// performGather: data(e,dof) = vec(LIDs(e,offsets(dof)))  (reference: assemblyManager.cpp:3633-3641)
__global__ __launch_bounds__(256) void gather_kernel(BlockDev b, const double *__restrict__ vec, double *out) {
  const size_t total = (size_t)b.nelem * b.n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx / b.n), dof = (int)(idx - (size_t)e * b.n);
    out[idx] = vec[b.lids[(size_t)e * b.n + b.offsets[dof]]];
  }
}

inline int grid_for(size_t total, int block) {
  const size_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

}  // namespace

void launch_scatter_local(const BlockDev &b, const double *local_J, const double *local_res, double *res,
                          double *crs_vals, int local_base, hipStream_t stream) {
  const size_t total = (size_t)b.e_count * b.n * b.n;
  if (total == 0) return;
  hipLaunchKernelGGL(scatter_local_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, b, local_J, local_res,
                     res, crs_vals, local_base);
  MHA_HIP(hipGetLastError());
}

void launch_dbc_diag(const BlockDev &b, double *crs_vals, hipStream_t stream) {
  if (!b.fixed) return;
  hipLaunchKernelGGL(dbc_diag_kernel, dim3(grid_for(b.nrows, 256)), dim3(256), 0, stream, b, crs_vals);
  MHA_HIP(hipGetLastError());
}

void launch_gather(const BlockDev &b, const double *vec, double *elem_vals, hipStream_t stream) {
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for((size_t)b.nelem * b.n, 256)), dim3(256), 0, stream, b, vec,
                     elem_vals);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
