// porous_boundary.hip -- porousMixed::boundaryResidual (reference: src/physics/porousMixed.cpp:345-432).
//
// "Dirichlet" on p is natural in the mixed form: res(off_u(dof)) += p_D w_side (v_dof . n) with v the HDIV side basis
// (getPhysicalBoundaryBasis, discretizationInterface.cpp:1880-1918: J phi / detJ at the side points, times the
// orientation sign).  The data does not depend on the solution: residual only, no Jacobian block.
// One thread per (entry, u dof); the surface is O(N^{d-1}) entries.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "launch.hpp"
#include "side_geometry.hpp"

namespace mha {
namespace {

template <int DIM>
__global__ __launch_bounds__(256) void porous_boundary_kernel(BlockDev b, SideTablesDev st, BoundaryDev bd,
                                                              VarLayoutDev vl, ElemOut out) {
  constexpr int NN = 1 << DIM, NU = 2 * DIM;
  const int total = bd.num * NU, nqs = st.nqs, n = vl.n_tot, u0 = vl.varptr[1];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / NU, dof = idx - k * NU, e = bd.elem[k], s = bd.side[k];
    const int c = dof >> 1, hi = dof & 1;  // raw In_FEM function: (1 -/+ x_c)/2 e_c
    const double *xn = b.nodes + (size_t)e * NN * DIM;
    const double sg = vl.orient ? (double)vl.orient[(size_t)e * n + u0 + dof] : 1.0;
    double r = 0.0;
    for (int q = 0; q < nqs; ++q) {
      double Ji[DIM * DIM], nrm[DIM], w, x[DIM], J[DIM * DIM], det;
      side_point<DIM>(xn, st, s, q, Ji, nrm, w, x);
      side_point_J<DIM>(xn, st, s, q, J, x);
      invert<DIM>(J, Ji, det);
      const double xc = st.ip[(s * nqs + q) * DIM + c];
      const double phi = hi ? 0.5 * (1.0 + xc) : 0.5 * (1.0 - xc);
      double vdotn = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) vdotn += J[d * DIM + c] * nrm[d];
      vdotn *= sg * phi / det;
      r += eval_func<DIM, true>(bd.data, k, q, nqs, x, nrm) * w * vdotn;
    }
    const int row = b.lids[(size_t)e * n + b.offsets[u0 + dof]];
    if (out.res && !(b.fixed && b.fixed[row])) unsafeAtomicAdd(out.res + row, -r);
  }
}

// porousMixed::computeFlux (src/physics/porousMixed.cpp:440-500): flux(elem, auxp, pt) = u . n at the side points, u from
// the HDIV side basis (Piola at the side points, times the orientation sign); d flux / d u_j = (v_j . n).  One thread per
// (entry, side point).
template <int DIM>
__global__ __launch_bounds__(256) void porous_flux_kernel(BlockDev b, SideTablesDev st, BoundaryDev bd, VarLayoutDev vl,
                                                          TimeDev tm) {
  constexpr int NN = 1 << DIM, NU = 2 * DIM;
  const int nqs = st.nqs, total = bd.num * nqs, n = vl.n_tot, u0 = vl.varptr[1];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / nqs, q = idx - k * nqs, e = bd.elem[k], s = bd.side[k];
    const double *xn = b.nodes + (size_t)e * NN * DIM;
    double Ji[DIM * DIM], nrm[DIM], w, x[DIM], J[DIM * DIM], det;
    side_point<DIM>(xn, st, s, q, Ji, nrm, w, x);
    side_point_J<DIM>(xn, st, s, q, J, x);
    invert<DIM>(J, Ji, det);
    double f = 0.0;
    if (bd.dflux_du)
      for (int j = 0; j < n; ++j) bd.dflux_du[(size_t)idx * n + j] = 0.0;
    for (int dof = 0; dof < NU; ++dof) {
      const int c = dof >> 1, hi = dof & 1;
      const double sg = vl.orient ? (double)vl.orient[(size_t)e * n + u0 + dof] : 1.0;
      const double xc = st.ip[(s * nqs + q) * DIM + c];
      const double phi = hi ? 0.5 * (1.0 + xc) : 0.5 * (1.0 - xc);
      double vdotn = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) vdotn += J[d * DIM + c] * nrm[d];
      vdotn *= sg * phi / det;
      const int row = b.lids[(size_t)e * n + b.offsets[u0 + dof]];
      const double cu = tm.u[row];
      double ue = cu;
      if (tm.transient) {  // Workset::computeSolnTransientSeeded (workset.cpp:589-623)
        const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
        double beta_u = (1.0 - tm.alpha_u) * cp[0];
        for (int st_ = 0; st_ < tm.stage; ++st_) beta_u += tm.stage_ratio[st_] * (cs[st_] - cp[0]);
        ue = tm.alpha_u * cu + beta_u;
      }
      f += ue * vdotn;
      if (bd.dflux_du) bd.dflux_du[(size_t)idx * n + u0 + dof] = tm.alpha_u * vdotn;
    }
    bd.flux[idx] = f;
    if (bd.dflux_daux) bd.dflux_daux[idx] = 0.0;
  }
}

}  // namespace

void launch_porous_flux(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const VarLayoutDev &vl,
                        const TimeDev &tm, hipStream_t stream) {
  if (bd.num <= 0) return;
  const int total = bd.num * st.nqs, grid = (total + 255) / 256;
  if (b.dim == 2) hipLaunchKernelGGL(porous_flux_kernel<2>, dim3(grid), dim3(256), 0, stream, b, st, bd, vl, tm);
  else hipLaunchKernelGGL(porous_flux_kernel<3>, dim3(grid), dim3(256), 0, stream, b, st, bd, vl, tm);
  MHA_HIP(hipGetLastError());
}

void launch_porous_boundary(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const VarLayoutDev &vl,
                            const ElemOut &out, hipStream_t stream) {
  if (bd.num <= 0) return;
  const int total = bd.num * 2 * b.dim, grid = (total + 255) / 256;
  if (b.dim == 2) hipLaunchKernelGGL(porous_boundary_kernel<2>, dim3(grid), dim3(256), 0, stream, b, st, bd, vl, out);
  else hipLaunchKernelGGL(porous_boundary_kernel<3>, dim3(grid), dim3(256), 0, stream, b, st, bd, vl, out);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
