// thermal_element.hip -- per-element thermal volume residual + Jacobian for gfx950.
//
// One wavefront (64 lanes) owns one element; a workgroup carries EPB elements and shares the
// reference tables in LDS.  The kernel fuses what the reference runs as ~10 Kokkos launches per
// workset:
//   performGather                         src/managers/assemblyManager.cpp:3598-3643
//   computeSolnSteadySeeded/TransientSeeded  src/tools/workset.cpp:823-859, 559-623
//   getPhysicalVolumetricBasis / IntegrationData  src/interfaces/discretizationInterface.cpp:732-776, 898-981
//   evaluateSolutionField (e, e_t, grad(e)[.])   src/tools/workset.cpp:937-1062
//   thermal::volumeResidual               src/physics/thermal.cpp:71-165
//   updateJac/updateRes or the fused scatter  assemblyManager.cpp:7412-7455, 7115-7152, 4031-4145
// The Sacado derivative array of res(e,i) is produced in closed form: with u_AD = alpha_u*u + beta_u
// and udot_AD = alpha_t*u + beta_t seeded at slot off(j),
//   res(e,i).dx(off(j)) = sum_q [ rho*cp*alpha_t*N_j*w*N_i + kappa*alpha_u*(grad N_j . grad N_i)*w ]
// which is exactly what the width-W forward-mode sweep of the reference accumulates.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

template <int DIM, int P, int NQ1>
struct Shape {
  static constexpr int M = P + 1;
  static constexpr int N = (DIM == 2) ? M * M : M * M * M;
  static constexpr int NQ = (DIM == 2) ? NQ1 * NQ1 : NQ1 * NQ1 * NQ1;
  static constexpr int NN = 1 << DIM;
  // LDS doubles shared by the workgroup: ref basis, ref grad, node grad, node val, ref wts
  static constexpr int SHARED = N * NQ + N * NQ * DIM + NN * NQ * DIM + NN * NQ + NQ;
  // LDS doubles per wave
  static constexpr int O_XN = 0;                          // nodes       [NN][DIM]
  static constexpr int O_UE = O_XN + NN * DIM;            // u_eval      [N]
  static constexpr int O_UD = O_UE + N;                   // u_dot       [N]
  static constexpr int O_JI = O_UD + N;                   // J^{-1}      [NQ][DIM*DIM]
  static constexpr int O_KQ = O_JI + NQ * DIM * DIM;      // kappa*w     [NQ]
  static constexpr int O_MQ = O_KQ + NQ;                  // rho*cp*w    [NQ]
  static constexpr int O_RQ = O_MQ + NQ;                  // (rho cp T_t - f) w   [NQ]
  static constexpr int O_FX = O_RQ + NQ;                  // kappa*w*grad T  [NQ][DIM]
  static constexpr int O_G = O_FX + NQ * DIM;             // physical grads [N][NQ][DIM]
  static constexpr int PER_WAVE = O_G + N * NQ * DIM;
};

template <int DIM, int P, int NQ1, int EPB, bool EXPR>
__global__ __launch_bounds__(64 * EPB) void thermal_element_kernel(BlockDev b, ThermalDev ph, ElemOut out) {
  using S = Shape<DIM, P, NQ1>;
  constexpr int N = S::N, NQ = S::NQ, NN = S::NN;
  extern __shared__ double smem[];
  double *s_basis = smem;
  double *s_grad = s_basis + N * NQ;
  double *s_ng = s_grad + N * NQ * DIM;
  double *s_nv = s_ng + NN * NQ * DIM;
  double *s_w = s_nv + NN * NQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *wv = s_w + NQ + wave * S::PER_WAVE;

  for (int i = tid; i < N * NQ; i += 64 * EPB) s_basis[i] = b.ref_basis[i];
  for (int i = tid; i < N * NQ * DIM; i += 64 * EPB) s_grad[i] = b.ref_grad[i];
  for (int i = tid; i < NN * NQ * DIM; i += 64 * EPB) s_ng[i] = b.nodegrad[i];
  for (int i = tid; i < NN * NQ; i += 64 * EPB) s_nv[i] = b.nodeval[i];
  for (int i = tid; i < NQ; i += 64 * EPB) s_w[i] = b.ref_wts[i];

  const int el = blockIdx.x * EPB + wave;
  const int e = b.e_begin + el;
  const bool active = el < b.e_count;
  const TimeDev &tm = ph.time;

  // A. nodes + gather + seeding values
  if (active) {
    for (int i = lane; i < NN * DIM; i += 64) wv[S::O_XN + i] = b.nodes[(size_t)e * NN * DIM + i];
    for (int dof = lane; dof < N; dof += 64) {
      const int row = b.lids[(size_t)e * N + b.offsets[dof]];
      const double cu = tm.u[row];
      double ue = cu, ud = 0.0;
      if (tm.transient) {
        const double *cp = tm.u_prev + (size_t)row * tm.nsteps;
        const double *cs = tm.u_stage + (size_t)row * tm.nstages;
        double beta_u = (1.0 - tm.alpha_u) * cp[0];
        for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
        double beta_t = 0.0;
        for (int s = 1; s < tm.nsteps + 1; ++s) beta_t += tm.bdf[s] * cp[s - 1];
        beta_t *= tm.timewt;
        ue = tm.alpha_u * cu + beta_u;
        ud = tm.alpha_t * cu + beta_t;
      }
      wv[S::O_UE + dof] = ue;
      wv[S::O_UD + dof] = ud;
    }
  }
  __syncthreads();

  // B. geometry at the integration points + coefficient functions
  if (active) {
    for (int q = lane; q < NQ; q += 64) {
      double J[DIM * DIM], Ji[DIM * DIM], det;
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double s = 0.0;
#pragma unroll
          for (int v = 0; v < NN; ++v) s += wv[S::O_XN + v * DIM + r] * s_ng[(v * NQ + q) * DIM + c];
          J[r * DIM + c] = s;
        }
      invert<DIM>(J, Ji, det);
      double x[3] = {0, 0, 0};
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < NN; ++v) s += wv[S::O_XN + v * DIM + d] * s_nv[v * NQ + q];
        x[d] = s;
      }
      const double w = s_w[q] * det;
      const double kap = eval_func<DIM, EXPR>(ph.diff, e, q, NQ, x);
      const double rc = eval_func<DIM, EXPR>(ph.rho, e, q, NQ, x) * eval_func<DIM, EXPR>(ph.cp, e, q, NQ, x);
      const double f = eval_func<DIM, EXPR>(ph.source, e, q, NQ, x);
#pragma unroll
      for (int k = 0; k < DIM * DIM; ++k) wv[S::O_JI + q * DIM * DIM + k] = Ji[k];
      wv[S::O_KQ + q] = kap * w;
      wv[S::O_MQ + q] = rc * w;
      wv[S::O_RQ + q] = -f * w;  // completed in phase D
    }
  }
  __syncthreads();

  // C. physical gradients  G(j,q,:) = J^{-T} grad_ref(j,q,:)   (HGRADtransformGRAD)
  if (active) {
    for (int idx = lane; idx < N * NQ; idx += 64) {
      const int q = idx % NQ;
      const double *Ji = wv + S::O_JI + q * DIM * DIM;
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) s += Ji[k * DIM + d] * s_grad[idx * DIM + k];
        wv[S::O_G + idx * DIM + d] = s;
      }
    }
  }
  __syncthreads();

  // D. solution fields at the integration points
  if (active) {
    for (int q = lane; q < NQ; q += 64) {
      double g[DIM], tt = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) g[d] = 0.0;
      for (int j = 0; j < N; ++j) {
        const double uj = wv[S::O_UE + j];
#pragma unroll
        for (int d = 0; d < DIM; ++d) g[d] += uj * wv[S::O_G + (j * NQ + q) * DIM + d];
        tt += wv[S::O_UD + j] * s_basis[j * NQ + q];
      }
      const double kw = wv[S::O_KQ + q];
#pragma unroll
      for (int d = 0; d < DIM; ++d) wv[S::O_FX + q * DIM + d] = kw * g[d];
      wv[S::O_RQ + q] += wv[S::O_MQ + q] * tt;
    }
  }
  __syncthreads();

  if (!active) return;
  const int32_t *L = b.lids + (size_t)e * N;

  // E. residual rows
  for (int i = lane; i < N; i += 64) {
    double r = 0.0;
    for (int q = 0; q < NQ; ++q) {
      r += wv[S::O_RQ + q] * s_basis[i * NQ + q];
#pragma unroll
      for (int d = 0; d < DIM; ++d) r += wv[S::O_FX + q * DIM + d] * wv[S::O_G + (i * NQ + q) * DIM + d];
    }
    const int slot = b.offsets[i];
    if (out.local_res) out.local_res[(size_t)(e - out.local_base) * N + slot] -= r;
    if (out.res) {
      const int row = L[slot];
      if (!(b.fixed && b.fixed[row])) atomicAdd(out.res + row, -r);
    }
  }

  // F. Jacobian entries
  if (out.compute_jacobian > 0) {
    const double au = tm.alpha_u, at = tm.alpha_t;
    for (int idx = lane; idx < N * N; idx += 64) {
      const int i = idx / N, j = idx - i * N;
      double v = 0.0;
      for (int q = 0; q < NQ; ++q) {
        double gg = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          gg += wv[S::O_G + (i * NQ + q) * DIM + d] * wv[S::O_G + (j * NQ + q) * DIM + d];
        v += wv[S::O_KQ + q] * au * gg;
        if (at != 0.0) v += wv[S::O_MQ + q] * at * s_basis[i * NQ + q] * s_basis[j * NQ + q];
      }
      const int si = b.offsets[i], sj = b.offsets[j];
      if (out.local_J) out.local_J[((size_t)(e - out.local_base) * N + si) * N + sj] += v;
      if (out.crs_vals) {
        const int row = L[si];
        if (!(b.fixed && b.fixed[row])) {
          const int p = find_col(b.colind, b.rowptr[row], b.rowptr[row + 1], L[sj]);
          if (p >= 0) atomicAdd(out.crs_vals + p, v);
        }
      }
    }
  }
}

template <int DIM, int P, int NQ1, int EPB>
void launch_one(const BlockDev &b, const ThermalDev &ph, const ElemOut &out, hipStream_t stream) {
  using S = Shape<DIM, P, NQ1>;
  const size_t lds = sizeof(double) * (S::SHARED + (size_t)EPB * S::PER_WAVE);
  if (b.e_count <= 0) return;
  const int grid = (b.e_count + EPB - 1) / EPB;
  auto go = [&](auto kern) {
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * EPB), lds, stream, b, ph, out);
    MHA_HIP(hipGetLastError());
  };
  if (has_expression(ph)) go(thermal_element_kernel<DIM, P, NQ1, EPB, true>);
  else go(thermal_element_kernel<DIM, P, NQ1, EPB, false>);
}

}  // namespace

bool thermal_element_supported(int dim, int order, int nq1) {
  return (dim == 2 && ((order == 1 && nq1 == 2) || (order == 2 && nq1 == 3) || (order == 4 && nq1 == 5))) ||
         (dim == 3 && ((order == 1 && nq1 == 2) || (order == 2 && nq1 == 3)));
}

void launch_thermal_element(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                            const ElemOut &out, hipStream_t stream) {
  if (dim == 2 && order == 1 && nq1 == 2) return launch_one<2, 1, 2, 4>(b, ph, out, stream);
  if (dim == 2 && order == 2 && nq1 == 3) return launch_one<2, 2, 3, 4>(b, ph, out, stream);
  if (dim == 2 && order == 4 && nq1 == 5) return launch_one<2, 4, 5, 4>(b, ph, out, stream);
  if (dim == 3 && order == 1 && nq1 == 2) return launch_one<3, 1, 2, 4>(b, ph, out, stream);
  if (dim == 3 && order == 2 && nq1 == 3) return launch_one<3, 2, 3, 2>(b, ph, out, stream);
  MHA_REQUIRE(false, MHA_ERR_INVALID,
              "thermal element kernel: unsupported (dim,order,points/dir) = (" << dim << "," << order << "," << nq1 << ")");
}

}  // namespace mha
