// thermal_affine_residual.hip -- K1 of the affine fast path: the element residual, one THREAD per element.
//
// thermal::volumeResidual (src/physics/thermal.cpp:125-163) with the gather / seeding values of
// src/tools/workset.cpp:823-859, 559-792 and the scatter of -res.val() (src/managers/assemblyManager.cpp:4075-4094),
// for affine elements with element-wise constant kappa, rho, c_p:
//   r_i = sum_q w_q detJ [ (rho c_p T_t - f) N_i + kappa (J^-1 J^-T grad_ref T) . grad_ref N_i ].
// The tensor basis has as many integration points per direction as dofs (order + 1), so the element polynomial is
// carried by its VALUES at the points: nodal -> point values by one 1-D transform per direction (in place), reference
// gradients at the points by the 1-D collocation derivative  D[q][q'] = sum_i phi_i'(xi_q) (Phi^-1)[q'][i], the
// transposed operations on the way back.  ~1e3 FMAs and ~60 live doubles per element instead of the n x nq = 729
// four-term products (and 32 lanes) of the lane-per-dof form this replaces; all 64 lanes of a wavefront do useful work.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int cpow(int b, int e) { return e == 0 ? 1 : b * cpow(b, e - 1); }
constexpr int kK1tThreads = 256;

// index of the tensor entry `pt` with its digit in direction D replaced by v (digits base M, direction 0 fastest)
template <int M, int D>
__device__ __forceinline__ constexpr int with_digit(int pt, int v) {
  constexpr int S = cpow(M, D);
  return pt - ((pt / S) % M) * S + v * S;
}

// v <- T applied along direction D, in place.  FWD: out[q] = sum_i T[i*M + q] in[i];  !FWD: out[i] = sum_q T[i*M + q] in[q]
template <int DIM, int M, int D, bool FWD>
__device__ __forceinline__ void apply1d(double *v, const double *T) {
  constexpr int N = cpow(M, DIM), S = cpow(M, D);
#pragma unroll
  for (int base = 0; base < N; ++base) {
    if ((base / S) % M != 0) continue;  // one pass per line
    double in[M], out[M];
#pragma unroll
    for (int a = 0; a < M; ++a) in[a] = v[base + a * S];
#pragma unroll
    for (int o = 0; o < M; ++o) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < M; ++a) s += (FWD ? T[a * M + o] : T[o * M + a]) * in[a];
      out[o] = s;
    }
#pragma unroll
    for (int o = 0; o < M; ++o) v[base + o * S] = out[o];
  }
}

template <int DIM, int P, bool TR, bool EXPR>
__global__ __launch_bounds__(kK1tThreads) void thermal_affine_residual_kernel(BlockDev b, ThermalDev ph,
                                                                              const double *__restrict__ geo,
                                                                              AffineTables1D tab, double *res) {
  constexpr int M = P + 1, N = cpow(M, DIM);
  const int idx = blockIdx.x * kK1tThreads + threadIdx.x;
  if (idx >= b.e_count) return;
  const int e = b.e_begin + idx;
  const int32_t *L = b.lids + (size_t)e * N;
  const TimeDev &tm = ph.time;

  // performGather + computeSoln*Seeded values, basis (tensor) order
  double U[N], Ud[TR ? N : 1];
#pragma unroll
  for (int ib = 0; ib < N; ++ib) {
    const int row = L[b.offsets[ib]];
    const double cu = tm.u[row];
    double ue = cu;
    if constexpr (TR) {
      const double *cp = tm.u_prev + (size_t)row * tm.nsteps;
      const double *cs = tm.u_stage + (size_t)row * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
      double beta_t = 0.0;
      for (int s = 1; s < tm.nsteps + 1; ++s) beta_t += tm.bdf[s] * cp[s - 1];
      beta_t *= tm.timewt;
      ue = tm.alpha_u * cu + beta_u;
      Ud[ib] = tm.alpha_t * cu + beta_t;
    }
    U[ib] = ue;
  }
  // nodal values -> values at the integration points
  apply1d<DIM, M, 0, true>(U, tab.phi);
  apply1d<DIM, M, 1, true>(U, tab.phi);
  if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, true>(U, tab.phi);
  if constexpr (TR) {
    apply1d<DIM, M, 0, true>(Ud, tab.phi);
    apply1d<DIM, M, 1, true>(Ud, tab.phi);
    if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, true>(Ud, tab.phi);
  }
  // cached geometry of the (affine) element
  const double *g = geo + (size_t)e * kGeoRec;
  double G[DIM][DIM], J[DIM][DIM], xc[DIM];
  {
    int k = 0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
      for (int c = a; c < DIM; ++c) { G[a][c] = g[k]; G[c][a] = G[a][c]; ++k; }
  }
  const double det = g[kGeoDet];
#pragma unroll
  for (int r = 0; r < DIM; ++r) {
    xc[r] = g[kGeoXc + r];
#pragma unroll
    for (int c = 0; c < DIM; ++c) J[r][c] = g[kGeoJ + r * DIM + c];
  }
  const double kap = ph.diff.amp, rc = ph.rho.amp * ph.cp.amp;  // element-wise constants on this path

  // point loop: W accumulates what multiplies the basis VALUES at each point (the flux terms enter through D^T)
  double W[N];
#pragma unroll
  for (int i = 0; i < N; ++i) W[i] = 0.0;
#pragma unroll
  for (int pt = 0; pt < N; ++pt) {
    const int q0 = pt % M, q1 = (pt / M) % M, q2 = pt / (M * M);
    const int qd[3] = {q0, q1, q2};
    double gh[DIM], x[3] = {0.0, 0.0, 0.0}, wq = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      wq *= tab.gw[qd[d]];
      double s = 0.0;
#pragma unroll
      for (int v = 0; v < M; ++v) {
        const int src = d == 0 ? with_digit<M, 0>(pt, v) : (d == 1 ? with_digit<M, 1>(pt, v) : with_digit<M, 2>(pt, v));
        s += tab.dcol[qd[d] * M + v] * U[src];
      }
      gh[d] = s;
    }
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double s = xc[r];
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += J[r][c] * tab.gp[qd[c]];
      x[r] = s;
    }
    const double f = eval_func<DIM, EXPR>(ph.source, e, pt, N, x);
    const double tt = TR ? Ud[TR ? pt : 0] : 0.0;
    W[pt] += (rc * tt - f) * det * wq;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {  // F_a = w_q kappa detJ sum_c (J^-1 J^-T)_ac d_c T
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += G[a][c] * gh[c];
      const double Fa = wq * kap * s;
#pragma unroll
      for (int v = 0; v < M; ++v) {
        const int dst = a == 0 ? with_digit<M, 0>(pt, v) : (a == 1 ? with_digit<M, 1>(pt, v) : with_digit<M, 2>(pt, v));
        W[dst] += tab.dcol[qd[a] * M + v] * Fa;
      }
    }
  }
  // point weights -> residual rows
  apply1d<DIM, M, 0, false>(W, tab.phi);
  apply1d<DIM, M, 1, false>(W, tab.phi);
  if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, false>(W, tab.phi);
  // the global vector receives -res.val(); fixed rows are skipped (assemblyManager.cpp:4075, 4094)
#pragma unroll
  for (int ib = 0; ib < N; ++ib) {
    const int row = L[b.offsets[ib]];
    if (!(b.fixed && b.fixed[row])) atomicAdd(res + row, -W[ib]);
  }
}

template <int DIM, int P>
void launch_t(const BlockDev &b, const ThermalDev &ph, const double *geo, const AffineTables1D &tab, double *res,
              hipStream_t stream) {
  if (b.e_count <= 0) return;
  const int grid = (b.e_count + kK1tThreads - 1) / kK1tThreads;
  const bool tr = ph.time.transient != 0;
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(kK1tThreads), 0, stream, b, ph, geo, tab, res); };
  if (has_expression(ph.source)) {
    if (tr) go(thermal_affine_residual_kernel<DIM, P, true, true>);
    else go(thermal_affine_residual_kernel<DIM, P, false, true>);
  } else {
    if (tr) go(thermal_affine_residual_kernel<DIM, P, true, false>);
    else go(thermal_affine_residual_kernel<DIM, P, false, false>);
  }
  MHA_HIP(hipGetLastError());
}

}  // namespace

bool thermal_affine_residual_supported(int dim, int order, int nq1) {
  return nq1 == order + 1 && ((dim == 2 && (order == 1 || order == 2 || order == 4)) || (dim == 3 && (order == 1 || order == 2)));
}

void launch_thermal_affine_residual(int dim, int order, const BlockDev &b, const ThermalDev &ph, const double *geo,
                                    const AffineTables1D &tab, double *res, hipStream_t stream) {
  if (dim == 2 && order == 1) return launch_t<2, 1>(b, ph, geo, tab, res, stream);
  if (dim == 2 && order == 2) return launch_t<2, 2>(b, ph, geo, tab, res, stream);
  if (dim == 2 && order == 4) return launch_t<2, 4>(b, ph, geo, tab, res, stream);
  if (dim == 3 && order == 1) return launch_t<3, 1>(b, ph, geo, tab, res, stream);
  if (dim == 3 && order == 2) return launch_t<3, 2>(b, ph, geo, tab, res, stream);
  MHA_REQUIRE(false, MHA_ERR_INVALID, "thread-per-element residual kernel: unsupported (dim, order)");
}

}  // namespace mha
