// var_views.hip -- per-variable Workset views of multi-variable blocks, solution fields, and the "Flux" condition.
//
// Replaces, for ONE variable of a block (HGRAD of any order, HVOL, lowest-order HDIV) on a list of elements (a
// workset's range, or the (element, local side) entries of a boundary group):
//   DiscretizationInterface::getPhysicalVolumetricBasis   src/interfaces/discretizationInterface.cpp:898-1127
//       (HGRADtransformVALUE/GRAD :955-971, HVOL :1000-1005, HDIVtransformVALUE/DIV :1011-1053, orientation signs)
//   DiscretizationInterface::getPhysicalBoundaryBasis     src/interfaces/discretizationInterface.cpp:1810-1955
//   Workset::getBasis / getBasisGrad / getBasisDiv / getBasisSide   src/tools/workset.hpp:241-293
//   Workset::computeSoln (fields at the points, ".val()" part)      src/tools/workset.cpp:1017-1190
//   Workset::getSolutionField                                        src/tools/workset.hpp:229
//   PhysicsInterface::fluxConditions                                 src/interfaces/physicsInterface.cpp:1702-1762
// The fused assembly kernels never read these arrays (geometry and fields are recomputed on chip); they exist so code
// written against the reference's Workset API finds the same data.  None of this is on the timed path.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

// one thread per (entry, point): Jacobian once, then every dof of the variable
template <int DIM>
__global__ __launch_bounds__(256) void var_views_kernel(BlockDev b, VarPointsDev t, const int32_t *elem,
                                                        const int32_t *side, int e0, int num, VarViewsDev out) {
  constexpr int NN = 1 << DIM;
  const int np = t.npts, card = t.card, total = num * np;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / np, q = idx - k * np;
    const int e = elem ? elem[k] : e0 + k, s = side ? side[k] : 0;
    const double *xn = b.nodes + (size_t)e * NN * DIM;
    double J[DIM * DIM], Ji[DIM * DIM], det;
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double sum = 0.0;
        for (int v = 0; v < NN; ++v) sum += xn[v * DIM + r] * t.nodegrad[((size_t)(s * NN + v) * np + q) * DIM + c];
        J[r * DIM + c] = sum;
      }
    invert<DIM>(J, Ji, det);
    for (int f = 0; f < card; ++f) {
      const double sg = t.orient ? (double)t.orient[(size_t)e * t.n_tot + t.var_off + f] : 1.0;
      const size_t ro = (size_t)(s * card + f) * np + q, o = ((size_t)k * card + f) * np + q;
      if (t.type == MHA_BASIS_HDIV) {
        // HDIVtransformVALUE: J phi / detJ; HDIVtransformDIV: div / detJ; then the orientation sign
        if (out.basis) {
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            double sum = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) sum += J[d * DIM + c] * t.val[ro * DIM + c];
            out.basis[o * DIM + d] = sg * sum / det;
          }
        }
        if (out.div && t.div) out.div[o] = sg * t.div[ro] / det;
      } else {
        if (out.basis) out.basis[o] = t.val[ro];  // HGRADtransformVALUE (HVOL uses the same transform)
        if (out.grad && t.grad) {
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            double sum = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) sum += Ji[c * DIM + d] * t.grad[ro * DIM + c];
            out.grad[o * DIM + d] = sum;
          }
        }
      }
    }
  }
}

// solution fields of one variable at the points of the current workset: one thread per (element, point)
__global__ __launch_bounds__(256) void var_fields_kernel(BlockDev b, TimeDev tm, int e0, int num, int card, int var_off,
                                                         int np, int ncomp, int dim, const double *basis,
                                                         const double *grad, const double *div, VarFieldsDev out) {
  const int total = num * np;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / np, q = idx - k * np;
    const int32_t *L = b.lids + (size_t)(e0 + k) * b.n;
    double val[3] = {0, 0, 0}, dot[3] = {0, 0, 0}, g[3] = {0, 0, 0}, dv = 0.0;
    for (int f = 0; f < card; ++f) {
      const int row = L[b.offsets[var_off + f]];
      const double cu = tm.u[row];
      double ue = cu, ud = 0.0;
      if (tm.transient) {  // Workset::computeSolnTransientSeeded, value part (workset.cpp:589-623)
        const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
        double beta_u = (1.0 - tm.alpha_u) * cp[0];
        for (int st = 0; st < tm.stage; ++st) beta_u += tm.stage_ratio[st] * (cs[st] - cp[0]);
        double beta_t = 0.0;
        for (int st = 1; st < tm.nsteps + 1; ++st) beta_t += tm.bdf[st] * cp[st - 1];
        beta_t *= tm.timewt;
        ue = tm.alpha_u * cu + beta_u;
        ud = tm.alpha_t * cu + beta_t;
      }
      const size_t o = ((size_t)k * card + f) * np + q;
      for (int c = 0; c < ncomp; ++c) {
        const double bv = basis[o * ncomp + c];
        val[c] += ue * bv;
        dot[c] += ud * bv;
      }
      if (grad)
        for (int d = 0; d < dim; ++d) g[d] += ue * grad[o * dim + d];
      if (div) dv += ue * div[o];
    }
    for (int c = 0; c < ncomp; ++c) {
      if (out.val[c]) out.val[c][idx] = val[c];
      if (out.dot[c]) out.dot[c][idx] = dot[c];
    }
    for (int d = 0; d < dim; ++d)
      if (out.grad[d]) out.grad[d][idx] = g[d];
    if (out.div) out.div[idx] = dv;
  }
}

// fluxConditions: res(elem, off(dof)) += -flux(elem,pt) * wts(elem,pt) * basis(elem,dof,pt,0); the global vector
// receives -res.val() and fixed rows are skipped by the scatter (assemblyManager.cpp:3943-3978)
template <int DIM, bool EXPR>
__global__ __launch_bounds__(256) void flux_condition_kernel(BlockDev b, FuncDesc flux, const int32_t *elem, int num,
                                                             int card, int var_off, int nqs, int ncomp,
                                                             const double *wts, const double *x0, const double *x1,
                                                             const double *x2, const double *n0, const double *n1,
                                                             const double *n2, const double *basis, double *res) {
  const int total = num * card;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / card, f = idx - k * card;
    const int row = b.lids[(size_t)elem[k] * b.n + b.offsets[var_off + f]];
    if (b.fixed && b.fixed[row]) continue;
    double r = 0.0;
    for (int q = 0; q < nqs; ++q) {
      const size_t p = (size_t)k * nqs + q;
      double x[3] = {x0[p], x1[p], DIM == 3 ? x2[p] : 0.0}, nrm[3] = {n0[p], n1[p], DIM == 3 ? n2[p] : 0.0};
      const double fv = eval_func<DIM, EXPR>(flux, k, q, nqs, x, nrm);
      r += fv * wts[p] * basis[(((size_t)k * card + f) * nqs + q) * ncomp];
    }
    unsafeAtomicAdd(res + row, r);
  }
}

__global__ __launch_bounds__(256) void negate_kernel(double *a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = -a[i];
}

}  // namespace

void launch_var_views(const BlockDev &b, const VarPointsDev &t, const int32_t *elem, const int32_t *side, int e0, int num,
                      const VarViewsDev &out, hipStream_t stream) {
  if (num <= 0) return;
  const int grid = (num * t.npts + 255) / 256;
  if (b.dim == 2) hipLaunchKernelGGL(var_views_kernel<2>, dim3(grid), dim3(256), 0, stream, b, t, elem, side, e0, num, out);
  else hipLaunchKernelGGL(var_views_kernel<3>, dim3(grid), dim3(256), 0, stream, b, t, elem, side, e0, num, out);
  MHA_HIP(hipGetLastError());
}

void launch_var_fields(const BlockDev &b, const TimeDev &tm, int e0, int num, int card, int var_off, int npts, int ncomp,
                       const double *basis, const double *grad, const double *div, const VarFieldsDev &out,
                       hipStream_t stream) {
  if (num <= 0) return;
  const int grid = (num * npts + 255) / 256;
  hipLaunchKernelGGL(var_fields_kernel, dim3(grid), dim3(256), 0, stream, b, tm, e0, num, card, var_off, npts, ncomp,
                     b.dim, basis, grad, div, out);
  MHA_HIP(hipGetLastError());
}

void launch_flux_condition(const BlockDev &b, const FuncDesc &flux, const int32_t *elem, int num, int card, int var_off,
                           int nqs, int ncomp, const double *wts, const double *const xyz[3], const double *const nrm[3],
                           const double *basis, double *res, hipStream_t stream) {
  if (num <= 0) return;
  const int grid = (num * card + 255) / 256;
  const bool expr = flux.kind == MHA_FUNC_EXPRESSION;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, b, flux, elem, num, card, var_off, nqs, ncomp, wts, xyz[0],
                       xyz[1], xyz[2], nrm[0], nrm[1], nrm[2], basis, res);
  };
  if (b.dim == 2) { if (expr) go(flux_condition_kernel<2, true>); else go(flux_condition_kernel<2, false>); }
  else { if (expr) go(flux_condition_kernel<3, true>); else go(flux_condition_kernel<3, false>); }
  MHA_HIP(hipGetLastError());
}

void launch_negate(double *a, size_t n, hipStream_t stream) {
  if (n == 0) return;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(negate_kernel, dim3(grid), dim3(256), 0, stream, a, n);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
