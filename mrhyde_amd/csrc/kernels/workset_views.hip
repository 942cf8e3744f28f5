// workset_views.hip -- materialises the Workset field views of one workset on the device:
// basis(elem,dof,pt,0), basis_grad(elem,dof,pt,dim), wts(elem,pt), x/y/z(elem,pt).
// Device counterpart of Group::computeBasis -> DiscretizationInterface::getPhysicalVolumetricBasis
// and getPhysicalIntegrationData (reference: src/tools/group.cpp:134-243,
// src/interfaces/discretizationInterface.cpp:732-776, 898-981).  The fused assembly kernels never
// read these arrays (they recompute geometry on chip); the views exist so that functors and tests
// written against the reference's Workset API have the same data to look at.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

template <int DIM>
__global__ __launch_bounds__(256) void workset_views_kernel(BlockDev b, int e0, int ne, WorksetViewsDev v) {
  constexpr int NN = 1 << DIM;
  const int nq = b.nq, n = b.n;
  const int total = ne * nq;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int el = idx / nq, q = idx - el * nq;
    const double *xn = b.nodes + (size_t)(e0 + el) * NN * DIM;
    double J[DIM * DIM], Ji[DIM * DIM], det, x[DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double s = 0.0;
        for (int k = 0; k < NN; ++k) s += xn[k * DIM + r] * b.nodegrad[(k * nq + q) * DIM + c];
        J[r * DIM + c] = s;
      }
      double s = 0.0;
      for (int k = 0; k < NN; ++k) s += xn[k * DIM + r] * b.nodeval[k * nq + q];
      x[r] = s;
    }
    invert<DIM>(J, Ji, det);
    if (v.wts) v.wts[idx] = b.ref_wts[q] * det;
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      if (v.xyz[d]) v.xyz[d][idx] = x[d];
    for (int f = 0; f < n; ++f) {
      const size_t o = ((size_t)el * n + f) * nq + q;
      if (v.basis) v.basis[o] = b.ref_basis[f * nq + q];
      if (v.basis_grad) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < DIM; ++k) s += Ji[k * DIM + d] * b.ref_grad[(f * nq + q) * DIM + k];
          v.basis_grad[o * DIM + d] = s;
        }
      }
    }
  }
}

}  // namespace

void launch_workset_views(const BlockDev &b, int e0, int ne, const WorksetViewsDev &v, hipStream_t stream) {
  if (ne <= 0) return;
  const int total = ne * b.nq;
  const int grid = (total + 255) / 256;
  if (b.dim == 2) hipLaunchKernelGGL(workset_views_kernel<2>, dim3(grid), dim3(256), 0, stream, b, e0, ne, v);
  else hipLaunchKernelGGL(workset_views_kernel<3>, dim3(grid), dim3(256), 0, stream, b, e0, ne, v);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
