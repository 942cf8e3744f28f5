// swhdg_side.hip -- batch evaluation of shallowwaterHybridized's side terms with their derivatives.
//
// One thread per side integration point: the flux vector of the trace state, the stabilisation / boundary term, the
// interface flux computeFlux leaves in wkset->flux, and -- by forward AD with Dual numbers, one direction at a time --
// its derivatives with respect to the interior state S and the trace state Sh (what the reference gets from the SFad
// arithmetic of the subgrid solver).  Reference lines: swhdg_side.hpp.
#include <hip/hip_runtime.h>

#include "../../../include/mrhyde_amd.h"
#include "launch.hpp"
#include "swhdg_side.hpp"

namespace mha {
namespace {

__global__ __launch_bounds__(256) void swhdg_side_kernel(SwhSideArgs a) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.npts; p += (int64_t)gridDim.x * blockDim.x) {
    double S[3], Sh[3], Sinf[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) { S[i] = a.S[p * 3 + i]; Sh[i] = a.Shat[p * 3 + i]; }
    if (a.Sinf)
      for (int i = 0; i < 3; ++i) Sinf[i] = a.Sinf[p * 3 + i];
    const double nx = a.normals[p * 2], ny = a.normals[p * 2 + 1];
    const bool roe = a.roe != 0;
    if (a.fluxvec) {
      double F[3][2];
      swh_flux_vector(Sh, a.g, F);
#pragma unroll
      for (int i = 0; i < 3; ++i) { a.fluxvec[(p * 3 + i) * 2] = F[i][0]; a.fluxvec[(p * 3 + i) * 2 + 1] = F[i][1]; }
    }
    if (a.term) {
      double t[3];
      if (a.side_type == MHA_SWH_INTERFACE) swh_stab_term(S, Sh, nx, ny, a.g, roe, t);
      else swh_boundary_term(a.side_type, S, Sh, Sinf, nx, ny, a.g, t);
#pragma unroll
      for (int i = 0; i < 3; ++i) a.term[p * 3 + i] = t[i];
    }
    if (a.iflux) {
      double f[3];
      swh_interface_flux(a.side_type, roe, S, Sh, Sinf, nx, ny, a.g, f);
#pragma unroll
      for (int i = 0; i < 3; ++i) a.iflux[p * 3 + i] = f[i];
    }
    if (a.d_dS || a.d_dShat) {
      for (int dir = 0; dir < 6; ++dir) {  // directions: S_0..S_2, Sh_0..Sh_2
        double *dst = dir < 3 ? a.d_dS : a.d_dShat;
        if (!dst) continue;
        Dual dS[3], dSh[3], f[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { dS[i] = mk(S[i], dir == i ? 1.0 : 0.0); dSh[i] = mk(Sh[i], dir == 3 + i ? 1.0 : 0.0); }
        swh_interface_flux(a.side_type, roe, dS, dSh, Sinf, nx, ny, a.g, f);
        const int col = dir % 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) dst[(p * 3 + i) * 3 + col] = f[i].d;
      }
    }
    if (a.L) {
      double L[3][3], lam[3], R[3][3];
      swh_eigendecomp(Sh, nx, ny, a.g, L, lam, R);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        a.lam[p * 3 + i] = lam[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) { a.L[(p * 3 + i) * 3 + j] = L[i][j]; a.R[(p * 3 + i) * 3 + j] = R[i][j]; }
      }
    }
  }
}

}  // namespace

void launch_swhdg_side(const SwhSideArgs &a, hipStream_t stream) {
  if (a.npts <= 0) return;
  const int64_t g = (a.npts + 255) / 256;
  const int grid = (int)(g > 256 * 16 ? 256 * 16 : g);
  hipLaunchKernelGGL(swhdg_side_kernel, dim3(grid), dim3(256), 0, stream, a);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
