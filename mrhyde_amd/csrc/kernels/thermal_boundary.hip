// thermal_boundary.hip -- boundary-group kernels: side integration data and thermal::boundaryResidual.
//
// Replaces, for one boundary group (entries = (element, local side) with one side name and BC type):
//   DiscretizationInterface::getPhysicalBoundaryIntegrationData / getPhysicalBoundaryBasis
//       src/interfaces/discretizationInterface.cpp:1608-1790, 1810-1955
//   AssemblyManager::performBoundaryGather, updateWorksetBoundary, the boundary loop of assembleJacRes
//       src/managers/assemblyManager.cpp:3650-3700, 5646-5710, 2518-2638
//   Workset::getSideElementSize                         src/tools/workset.cpp:2682-2696
//   thermal::boundaryResidual (Neumann, weak Dirichlet, interface) src/physics/thermal.cpp:172-281
//   thermal::computeFlux                                  src/physics/thermal.cpp:288-347
//
// One wave per boundary entry.  A side has at most 16 integration points and an element at most 32 dofs, so the
// whole entry lives in a few KB of LDS; the Sacado derivative array is produced in closed form:
//   Neumann:         res_i = -sum_q g w N_i                                   (no Jacobian)
//   weak Dirichlet:  res_i = sum_q kappa w [ (epen/h)(T-g) N_i - (grad T.n) N_i - sf (T-g) (grad N_i.n) ]
//                    d res_i / d u_j = alpha_u sum_q kappa w [ (epen/h) N_i N_j - N_i (grad N_j.n) - sf (grad N_i.n) N_j ]
// with epen = 10 (thermal.cpp:236), h = (sum_q w)^(1/(dim-1)), sf = form_param.  The surface is O(N^{d-1}) against
// the volume's O(N^d): this kernel is never the roofline; it scatters with atomics + the CRS column search.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "launch.hpp"
#include "side_geometry.hpp"

namespace mha {
namespace {

constexpr int kBndMaxN = 32, kBndMaxQ = 16, kBndWaves = 4;

template <int DIM>
__global__ __launch_bounds__(256) void boundary_views_kernel(BlockDev b, SideTablesDev st, BoundaryDev bd,
                                                             BoundaryViewsDev v) {
  constexpr int NN = 1 << DIM;
  const int nqs = st.nqs, n = b.n, total = bd.num * nqs;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / nqs, q = idx - k * nqs, s = bd.side[k];
    const double *xn = b.nodes + (size_t)bd.elem[k] * NN * DIM;
    double Ji[DIM * DIM], nrm[DIM], w, x[DIM];
    side_point<DIM>(xn, st, s, q, Ji, nrm, w, x);
    if (v.wts) v.wts[idx] = w;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      if (v.xyz[d]) v.xyz[d][idx] = x[d];
      if (v.nrm[d]) v.nrm[d][idx] = nrm[d];
    }
    for (int f = 0; f < n; ++f) {
      const size_t o = ((size_t)k * n + f) * nqs + q;
      if (v.basis) v.basis[o] = st.basis[(s * n + f) * nqs + q];
      if (v.basis_grad) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double sum = 0.0;
#pragma unroll
          for (int c = 0; c < DIM; ++c) sum += Ji[c * DIM + d] * st.grad[((s * n + f) * nqs + q) * DIM + c];
          v.basis_grad[o * DIM + d] = sum;
        }
      }
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(64 * kBndWaves) void thermal_boundary_kernel(BlockDev b, SideTablesDev st, BoundaryDev bd,
                                                                          TimeDev tm, ElemOut out) {
  constexpr int NN = 1 << DIM;
  __shared__ double s_ua[kBndWaves][kBndMaxN];
  __shared__ double s_w[kBndWaves][kBndMaxQ], s_g[kBndWaves][kBndMaxQ], s_kap[kBndWaves][kBndMaxQ];
  __shared__ double s_T[kBndWaves][kBndMaxQ], s_gTn[kBndWaves][kBndMaxQ];
  __shared__ double s_Ji[kBndWaves][kBndMaxQ * DIM * DIM], s_nrm[kBndWaves][kBndMaxQ * DIM];
  __shared__ double s_bgn[kBndWaves][kBndMaxN * kBndMaxQ];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * kBndWaves + wv;
  const bool active = k < bd.num;  // inactive waves still take part in the block barriers
  const int n = b.n, nqs = st.nqs;
  const int e = active ? bd.elem[k] : 0, s = active ? bd.side[k] : 0;
  // "interface" is the weak-Dirichlet branch with the trace ("aux e") as the data (thermal.cpp:227-243); computeFlux
  // needs the same side fields
  const bool weak = bd.bc_type == MHA_BC_WEAK_DIRICHLET || bd.bc_type == MHA_BC_INTERFACE || bd.flux != nullptr;
  const int32_t *L = b.lids + (size_t)e * n;

  // A. side geometry + data (lane = side point); gather + seeding value (lane = basis dof)
  if (active && lane < nqs) {
    double Ji[DIM * DIM], nrm[DIM], w, x[DIM];
    side_point<DIM>(b.nodes + (size_t)e * NN * DIM, st, s, lane, Ji, nrm, w, x);
    s_w[wv][lane] = w;
    s_g[wv][lane] = eval_func<DIM, true>(bd.data, k, lane, nqs, x, nrm);
    s_kap[wv][lane] = eval_func<DIM, true>(bd.diff, k, lane, nqs, x, nrm);
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) s_Ji[wv][lane * DIM * DIM + i] = Ji[i];
#pragma unroll
    for (int d = 0; d < DIM; ++d) s_nrm[wv][lane * DIM + d] = nrm[d];
  }
  int row = -1;
  if (active && lane < n) {
    row = L[b.offsets[lane]];
    const double cu = tm.u[row];
    double ue = cu;
    if (tm.transient) {  // Workset::computeSolnTransientSeeded (workset.cpp:589-623)
      const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int st_ = 0; st_ < tm.stage; ++st_) beta_u += tm.stage_ratio[st_] * (cs[st_] - cp[0]);
      ue = tm.alpha_u * cu + beta_u;
    }
    s_ua[wv][lane] = ue;
  }
  __syncthreads();

  double h = 1.0;
  if (weak) {
    // B. (grad N_dof . n)(q), physical gradient = J^{-T} grad_ref
    if (active) {
      double vol = 0.0;
      for (int q = 0; q < nqs; ++q) vol += s_w[wv][q];
      h = (DIM == 2) ? vol : sqrt(vol);  // vol^(1/(dim-1)), getSideElementSize
      for (int idx = lane; idx < n * nqs; idx += 64) {
        const int dof = idx / nqs, q = idx - dof * nqs;
        const double *gr = st.grad + ((size_t)(s * n + dof) * nqs + q) * DIM;
        double sum = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double pg = 0.0;
#pragma unroll
          for (int c = 0; c < DIM; ++c) pg += s_Ji[wv][q * DIM * DIM + c * DIM + d] * gr[c];
          sum += pg * s_nrm[wv][q * DIM + d];
        }
        s_bgn[wv][dof * nqs + q] = sum;
      }
    }
    __syncthreads();
    // C. side fields T and grad T . n (lane = side point)
    if (active && lane < nqs) {
      double T = 0.0, gTn = 0.0;
      for (int dof = 0; dof < n; ++dof) {
        T += s_ua[wv][dof] * st.basis[(s * n + dof) * nqs + lane];
        gTn += s_ua[wv][dof] * s_bgn[wv][dof * nqs + lane];
      }
      s_T[wv][lane] = T;
      s_gTn[wv][lane] = gTn;
    }
    __syncthreads();
  }
  if (!active) return;
  const double epen = 10.0, sf = bd.form_param;
  if (bd.flux) {
    // computeFlux: flux = (epen / h) kappa (lambda - T) + sf kappa grad T . n with sf = 1 (forward runs, thermal.cpp:293-296),
    // lambda = "aux e" at the side points (here: the group's data function)
    if (lane < nqs) {
      const double kap = s_kap[wv][lane];
      bd.flux[(size_t)k * nqs + lane] = epen / h * kap * (s_g[wv][lane] - s_T[wv][lane]) + kap * s_gTn[wv][lane];
      if (bd.dflux_daux) bd.dflux_daux[(size_t)k * nqs + lane] = epen / h * kap;
    }
    if (bd.dflux_du)
      for (int idx = lane; idx < n * nqs; idx += 64) {
        const int q = idx / n, j = idx - q * n;
        bd.dflux_du[((size_t)k * nqs + q) * n + j] =
            tm.alpha_u * s_kap[wv][q] * (-epen / h * st.basis[(s * n + j) * nqs + q] + s_bgn[wv][j * nqs + q]);
      }
    return;
  }

  // D. residual rows (lane = basis dof)
  if (lane < n) {
    double r = 0.0;
    for (int q = 0; q < nqs; ++q) {
      const double w = s_w[wv][q], N = st.basis[(s * n + lane) * nqs + q], g = s_g[wv][q];
      if (!weak) {
        r += -g * w * N;
      } else {
        const double kap = s_kap[wv][q], dT = s_T[wv][q] - g;
        r += epen / h * kap * dT * w * N - kap * s_gTn[wv][q] * w * N - sf * kap * dT * w * s_bgn[wv][lane * nqs + q];
      }
    }
    if (out.res && !(b.fixed && b.fixed[row])) unsafeAtomicAdd(out.res + row, -r);
  }
  // E. Jacobian entries (weak Dirichlet only): lanes sweep the n x n block
  if (weak && out.compute_jacobian && out.crs_vals) {
    for (int idx = lane; idx < n * n; idx += 64) {
      const int i = idx / n, j = idx - i * n;
      const int ri = L[b.offsets[i]];
      if (b.fixed && b.fixed[ri]) continue;
      double a = 0.0;
      for (int q = 0; q < nqs; ++q) {
        const double Ni = st.basis[(s * n + i) * nqs + q], Nj = st.basis[(s * n + j) * nqs + q];
        a += s_kap[wv][q] * s_w[wv][q] *
             (epen / h * Ni * Nj - Ni * s_bgn[wv][j * nqs + q] - sf * s_bgn[wv][i * nqs + q] * Nj);
      }
      const int p = find_col(b.colind, b.rowptr[ri], b.rowptr[ri + 1], L[b.offsets[j]]);
      if (p >= 0) unsafeAtomicAdd(out.crs_vals + p, tm.alpha_u * a);
    }
  }
}

}  // namespace

bool thermal_boundary_supported(int n, int nqs) { return n <= kBndMaxN && nqs <= kBndMaxQ; }

void launch_boundary_views(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const BoundaryViewsDev &v,
                           hipStream_t stream) {
  if (bd.num <= 0) return;
  const int total = bd.num * st.nqs, grid = (total + 255) / 256;
  if (b.dim == 2) hipLaunchKernelGGL(boundary_views_kernel<2>, dim3(grid), dim3(256), 0, stream, b, st, bd, v);
  else hipLaunchKernelGGL(boundary_views_kernel<3>, dim3(grid), dim3(256), 0, stream, b, st, bd, v);
  MHA_HIP(hipGetLastError());
}

void launch_thermal_boundary(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const TimeDev &tm,
                             const ElemOut &out, hipStream_t stream) {
  if (bd.num <= 0) return;
  MHA_REQUIRE(thermal_boundary_supported(b.n, st.nqs), MHA_ERR_INVALID,
              "boundary kernel supports at most " << kBndMaxN << " dofs per element and " << kBndMaxQ
                                                  << " side integration points (got " << b.n << ", " << st.nqs << ")");
  const int grid = (bd.num + kBndWaves - 1) / kBndWaves;
  if (b.dim == 2)
    hipLaunchKernelGGL(thermal_boundary_kernel<2>, dim3(grid), dim3(64 * kBndWaves), 0, stream, b, st, bd, tm, out);
  else
    hipLaunchKernelGGL(thermal_boundary_kernel<3>, dim3(grid), dim3(64 * kBndWaves), 0, stream, b, st, bd, tm, out);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
