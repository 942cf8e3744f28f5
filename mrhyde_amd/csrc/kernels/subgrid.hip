// subgrid.hip -- the small kernels of the subgrid (DtN / HDG) sub-iteration driver.
//
// SubGridDtN_Solver::nonlinearSolver (src/subgrid/subgridDtN_solver.cpp:909-1041) iterates, per macro element, Newton
// steps on the interior unknowns with the trace held fixed: assembleJacobianResidual (:681-903), the infinity norm of
// the residual against the initial one, a direct solve, sol += du.  With one HDG element per subgrid the interior
// unknowns are element-local, so every element runs its own loop; the device driver (AssemblyManager::subgridSolve)
// launches a fixed number of passes without ever synchronising with the host, and these kernels carry the loop state:
//   combine: blocks[e][0:ni][0:ni] += volume block, res[e][0:ni] += volume residual (both in flattened (variable, dof)
//            order; the volume arrays come in LID-position order), then the reference's bookkeeping per element --
//            pass 0: resnorm_initial = |res_u|_inf, scaled = 1 (0 if the norm is 0); later: scaled = norm / initial;
//            an element stays in its loop while scaled > tol (:944, :990)
//   update:  sol += du for the elements still in their loop (:1027-1032).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "launch.hpp"

namespace mha {
namespace {

__global__ __launch_bounds__(256) void subgrid_combine_kernel(int64_t nelem, int ni, int n, const int32_t *offsets,
                                                              const double *local_J, const double *local_res,
                                                              double *blocks, double *res, int pass, double tol,
                                                              double *rn0, double *scaled, int32_t *iters, int32_t *active) {
  // one wavefront per element
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (e >= nelem) return;
  double *B = blocks + e * n * n, *r = res + e * n;
  const double *LJ = local_J + e * ni * ni, *LR = local_res + e * ni;
  for (int k = lane; k < ni * ni; k += 64) {
    const int i = k / ni, j = k - i * ni;
    B[(size_t)i * n + j] += LJ[(size_t)offsets[i] * ni + offsets[j]];
  }
  double nrm = 0.0;
  for (int i = lane; i < ni; i += 64) {
    const double v = r[i] + LR[offsets[i]];
    r[i] = v;
    nrm = fmax(nrm, fabs(v));
  }
  for (int o = 32; o > 0; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o));
  if (lane != 0 || pass < 0) return;  // pass < 0: the closing assembly, no bookkeeping
  if (pass == 0) {
    rn0[e] = nrm;
    scaled[e] = nrm > 0.0 ? 1.0 : 0.0;
    iters[e] = 1;
    active[e] = scaled[e] > tol ? 1 : 0;
  } else if (active[e]) {  // still in its loop: this pass counts, and decides about the next solve
    const double s = nrm / rn0[e];
    scaled[e] = s;
    iters[e] += 1;
    active[e] = s > tol ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void subgrid_update_kernel(int64_t nelem, int ni, const int32_t *lids, const int32_t *offsets,
                                                             const double *du, const int32_t *active, double *u) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nelem * ni; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / ni;
    const int i = (int)(t - e * ni);
    if (active[e]) u[lids[e * ni + offsets[i]]] += du[t];
  }
}

}  // namespace

void launch_subgrid_combine(int64_t nelem, int ni, int n, const int32_t *offsets, const double *local_J, const double *local_res,
                            double *blocks, double *res, int pass, double tol, double *rn0, double *scaled, int32_t *iters,
                            int32_t *active, hipStream_t stream) {
  if (nelem <= 0) return;
  hipLaunchKernelGGL(subgrid_combine_kernel, dim3((unsigned)((nelem + 3) / 4)), dim3(256), 0, stream, nelem, ni, n, offsets,
                     local_J, local_res, blocks, res, pass, tol, rn0, scaled, iters, active);
  MHA_HIP(hipGetLastError());
}

void launch_subgrid_update(int64_t nelem, int ni, const int32_t *lids, const int32_t *offsets, const double *du,
                           const int32_t *active, double *u, hipStream_t stream) {
  if (nelem <= 0) return;
  const int64_t total = nelem * ni;
  hipLaunchKernelGGL(subgrid_update_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, stream,
                     nelem, ni, lids, offsets, du, active, u);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
