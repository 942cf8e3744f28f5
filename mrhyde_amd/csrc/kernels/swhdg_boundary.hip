// swhdg_boundary.hip -- shallowwaterHybridized::boundaryResidual on a group of (element, side) entries.
//
// reference: src/physics/shallowwaterHybridized.cpp:190-263 (all boundary contributions are (F(Shat).n + Stab, v_i) on
// interface sides, the boundary term (B, v_i) on Far-field / Slip sides), with the side fields of the interior state
// (evaluateSideSolutionField, src/tools/workset.cpp:1069-1176) and the trace ("aux") state at the side points.  The
// trace state is data here (per-point arrays or constants registered as "aux H <side>", ...): the HFACE trace basis
// that produces it in the reference belongs to the subgrid solver.  d res / d u of the interior unknowns is assembled
// (d flux / d S by forward AD on Dual numbers, swhdg_side.hpp); d res / d trace is the caller's (mha_swhdg_side_terms).
// One wavefront per entry, as kernels/thermal_boundary.hip.
#include <hip/hip_runtime.h>

#include "../../../include/mrhyde_amd.h"
#include "device_math.hpp"
#include "launch.hpp"
#include "side_geometry.hpp"
#include "swhdg_side.hpp"

namespace mha {
namespace {

constexpr int kSwMaxN = 16, kSwMaxQ = 16, kSwWaves = 4;

__global__ __launch_bounds__(64 * kSwWaves) void swhdg_boundary_kernel(BlockDev b, SideTablesDev st, BoundaryDev bd,
                                                                       SwhBoundaryDev sw, TimeDev tm, ElemOut out) {
  constexpr int DIM = 2, NN = 4;
  __shared__ double s_u[kSwWaves][3 * kSwMaxN], s_fw[kSwWaves][kSwMaxQ * 3], s_Mw[kSwWaves][kSwMaxQ * 9];
  __shared__ int s_row[kSwWaves][3 * kSwMaxN];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * kSwWaves + wv;
  const bool active = k < bd.num;
  const int nt = b.n, n = nt / 3, nqs = st.nqs;
  const int e = active ? bd.elem[k] : 0, s = active ? bd.side[k] : 0;
  const int32_t *L = b.lids + (size_t)e * nt;
  // gather + seeding values (lane = flattened (variable, dof))
  if (active && lane < nt) {
    const int row = L[b.offsets[lane]];
    const double cu = tm.u[row];
    double ue = cu;
    if (tm.transient) {
      const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int st_ = 0; st_ < tm.stage; ++st_) beta_u += tm.stage_ratio[st_] * (cs[st_] - cp[0]);
      ue = tm.alpha_u * cu + beta_u;
    }
    s_u[wv][lane] = ue;
    s_row[wv][lane] = row;
  }
  __syncthreads();
  // side fields, trace state, flux and its derivative at the side points (lane = point)
  if (active && lane < nqs) {
    double Ji[DIM * DIM], nrm[DIM], w, x[DIM];
    side_point<DIM>(b.nodes + (size_t)e * NN * DIM, st, s, lane, Ji, nrm, w, x);
    double S[3] = {0, 0, 0}, Sh[3], Sinf[3];
    for (int dof = 0; dof < n; ++dof) {
      const double N = st.basis[(s * n + dof) * nqs + lane];
#pragma unroll
      for (int i = 0; i < 3; ++i) S[i] += s_u[wv][i * n + dof] * N;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Sh[i] = eval_func<DIM, true>(sw.aux[i], k, lane, nqs, x, nrm);
      Sinf[i] = eval_func<DIM, true>(sw.farfield[i], k, lane, nqs, x, nrm);
    }
    const bool roe = sw.roe != 0;
    double f[3];
    swh_interface_flux(sw.side_type, roe, S, Sh, Sinf, nrm[0], nrm[1], sw.g, f);
#pragma unroll
    for (int i = 0; i < 3; ++i) s_fw[wv][lane * 3 + i] = f[i] * w;
    for (int dir = 0; dir < 3; ++dir) {
      Dual dS[3], dSh[3], df[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { dS[i] = mk(S[i], dir == i ? 1.0 : 0.0); dSh[i] = mk(Sh[i]); }
      swh_interface_flux(sw.side_type, roe, dS, dSh, Sinf, nrm[0], nrm[1], sw.g, df);
#pragma unroll
      for (int i = 0; i < 3; ++i) s_Mw[wv][lane * 9 + i * 3 + dir] = df[i].d * w * tm.alpha_u;
    }
  }
  __syncthreads();
  if (!active) return;
  // residual rows (lane = flattened (variable, dof))
  if (lane < nt) {
    const int i = lane / n, a = lane - i * n;
    double r = 0.0;
    for (int q = 0; q < nqs; ++q) r += s_fw[wv][q * 3 + i] * st.basis[(s * n + a) * nqs + q];
    const int row = s_row[wv][lane];
    if (out.res && !(b.fixed && b.fixed[row])) unsafeAtomicAdd(out.res + row, -r);
  }
  // Jacobian entries of the interior unknowns
  if (out.compute_jacobian && out.crs_vals) {
    for (int idx = lane; idx < nt * nt; idx += 64) {
      const int r = idx / nt, c = idx - r * nt;
      const int i = r / n, a = r - i * n, kk = c / n, bb = c - kk * n;
      const int row = s_row[wv][r];
      if (b.fixed && b.fixed[row]) continue;
      double v = 0.0;
      for (int q = 0; q < nqs; ++q)
        v += s_Mw[wv][q * 9 + i * 3 + kk] * st.basis[(s * n + a) * nqs + q] * st.basis[(s * n + bb) * nqs + q];
      const int p = find_col(b.colind, b.rowptr[row], b.rowptr[row + 1], s_row[wv][c]);
      if (p >= 0) unsafeAtomicAdd(out.crs_vals + p, v);
    }
  }
}

}  // namespace

void launch_swhdg_boundary(const BlockDev &b, const SideTablesDev &st, const BoundaryDev &bd, const SwhBoundaryDev &sw,
                           const TimeDev &tm, const ElemOut &out, hipStream_t stream) {
  if (bd.num <= 0) return;
  MHA_REQUIRE(b.dim == 2 && b.n % 3 == 0 && b.n / 3 <= kSwMaxN && st.nqs <= kSwMaxQ, MHA_ERR_INVALID,
              "shallowwaterHybridized boundary kernel: 2-D, at most " << kSwMaxN << " dofs per variable and " << kSwMaxQ
                                                                     << " side points");
  const int grid = (bd.num + kSwWaves - 1) / kSwWaves;
  hipLaunchKernelGGL(swhdg_boundary_kernel, dim3(grid), dim3(64 * kSwWaves), 0, stream, b, st, bd, sw, tm, out);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
