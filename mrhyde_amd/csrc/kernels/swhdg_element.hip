// swhdg_element.hip -- the HDG element of shallowwaterHybridized, side part: residual and derivative blocks of the 12
// interior and 24 trace unknowns of every element.
//
// interior rows: shallowwaterHybridized::boundaryResidual (src/physics/shallowwaterHybridized.cpp:190-263) on the four
//   sides, res_(i,a) += flux_i wts N_a;
// trace rows:    computeFlux (:270-368) integrated against the trace basis as SubGridDtN_Solver::updateFlux does
//   (src/subgrid/subgridDtN_solver.cpp:1583-1601), res_(i,(edge,k)) += mu_k flux_i wts;
// trace basis:   Basis_HFACE_QUAD_In_FEM of degree 1, vendored by the reference
//   (src/tools/Intrepid2_HFACE_QUAD_In_FEMdef.hpp:84-196): per edge the two linear Lagrange functions of the edge's
//   reference coordinate (y on left/right, x on bottom/top), edges ordered left, bottom, right, top, zero elsewhere.
// The reference gets the four blocks d(res_u, res_lambda)/d(u, lambda) from SFad arithmetic inside the subgrid solver;
// here one wavefront per element evaluates the flux and its derivative columns at the side points with Dual numbers
// (one (point, direction) per lane) and contracts them with the 4 + 8 side functions.  The volume part of the
// interior block is the module's volumeResidual (point engine).
#include <hip/hip_runtime.h>

#include "../../../include/mrhyde_amd.h"
#include "device_math.hpp"
#include "launch.hpp"
#include "side_geometry.hpp"
#include "swhdg_side.hpp"

namespace mha {
namespace {

constexpr int kHdgWaves = 4, kHdgMaxPts = 16, kHdgN = 4, kHdgRows = 36;

__global__ __launch_bounds__(64 * kHdgWaves) void swhdg_element_kernel(BlockDev b, SideTablesDev st, SwhElementDev a,
                                                                       TimeDev tm) {
  constexpr int DIM = 2, NN = 4;
  __shared__ double s_u[kHdgWaves][12], s_l[kHdgWaves][24];
  __shared__ double s_T[kHdgWaves][kHdgMaxPts][12];                 // side functions at the points: 4 N_a, 8 mu
  __shared__ double s_f[kHdgWaves][kHdgMaxPts][3];                  // flux * w
  __shared__ double s_D[kHdgWaves][kHdgMaxPts][2][9];               // d flux / d S, d flux / d Shat (* w), [i][k]
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int el = blockIdx.x * kHdgWaves + wv;
  const bool active = el < b.e_count;
  const int e = b.e_begin + (active ? el : 0), nqs = st.nqs, npts = 4 * nqs;
  const int32_t *L = b.lids + (size_t)e * 12;
  if (active && lane < 12) {
    const int row = L[b.offsets[lane]];
    const double cu = tm.u[row];
    double ue = cu;
    if (tm.transient) {
      const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
      ue = tm.alpha_u * cu + beta_u;
    }
    s_u[wv][lane] = ue;
  }
  if (active && lane >= 32 && lane < 56) s_l[wv][lane - 32] = a.lambda[(size_t)e * 24 + lane - 32];
  __syncthreads();
  // one (side point, direction) per lane: direction 0 = value, 1..3 = d/dS_k, 4..6 = d/dShat_k
  if (active) {
    for (int idx = lane; idx < npts * 7; idx += 64) {
      const int p = idx / 7, dir = idx - p * 7, s = p / nqs, q = p - s * nqs;
      const int edge = (s + 1) & 3;  // shards side 0,1,2,3 (bottom, right, top, left) -> HFACE edge 1,2,3,0
      double Ji[DIM * DIM], nrm[DIM], w, x[DIM];
      side_point<DIM>(b.nodes + (size_t)e * NN * DIM, st, s, q, Ji, nrm, w, x);
      const double tc = (edge & 1) ? st.ip[(s * nqs + q) * DIM] : st.ip[(s * nqs + q) * DIM + 1];
      const double mu0 = 0.5 * (1.0 - tc), mu1 = 0.5 * (1.0 + tc);
      double S[3] = {0, 0, 0}, Sh[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int dof = 0; dof < kHdgN; ++dof) S[i] += s_u[wv][i * 4 + dof] * st.basis[(s * kHdgN + dof) * nqs + q];
        Sh[i] = s_l[wv][i * 8 + edge * 2] * mu0 + s_l[wv][i * 8 + edge * 2 + 1] * mu1;
      }
      const int stype = a.side_types ? a.side_types[(size_t)e * 4 + s] : 0;
      Dual dS[3], dSh[3], f[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { dS[i] = mk(S[i], dir == 1 + i ? 1.0 : 0.0); dSh[i] = mk(Sh[i], dir == 4 + i ? 1.0 : 0.0); }
      swh_interface_flux(stype, a.roe != 0, dS, dSh, a.farfield, nrm[0], nrm[1], a.g, f);
      if (dir == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s_f[wv][p][i] = f[i].v * w;
#pragma unroll
        for (int dof = 0; dof < kHdgN; ++dof) s_T[wv][p][dof] = st.basis[(s * kHdgN + dof) * nqs + q];
#pragma unroll
        for (int k = 0; k < 8; ++k) s_T[wv][p][4 + k] = (k >> 1) == edge ? ((k & 1) ? mu1 : mu0) : 0.0;
      } else {
        const int which = dir > 3, kk = (dir - 1) % 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) s_D[wv][p][which][i * 3 + kk] = f[i].d * w;
      }
    }
  }
  __syncthreads();
  if (!active) return;
  // rows r = (equation i, side function r'): interior i*4 + a, traces 12 + i*8 + edge*2 + k
  auto split = [](int r, int &i, int &rp) {
    if (r < 12) { i = r >> 2; rp = r & 3; }
    else { const int t = r - 12; i = t >> 3; rp = 4 + (t & 7); }
  };
  if (lane < kHdgRows && a.res) {
    int i, rp;
    split(lane, i, rp);
    double r = 0.0;
    for (int p = 0; p < npts; ++p) r += s_f[wv][p][i] * s_T[wv][p][rp];
    a.res[(size_t)(e - b.e_begin) * kHdgRows + lane] = -r;
  }
  if (a.blocks) {
    double *out = a.blocks + (size_t)(e - b.e_begin) * kHdgRows * kHdgRows;
    for (int idx = lane; idx < kHdgRows * kHdgRows; idx += 64) {
      const int r = idx / kHdgRows, c = idx - r * kHdgRows;
      int i, rp, k, cp;
      split(r, i, rp);
      split(c, k, cp);
      const int which = c >= 12;  // interior column: through S; trace column: through Shat
      double v = 0.0;
      for (int p = 0; p < npts; ++p) v += s_T[wv][p][rp] * s_D[wv][p][which][i * 3 + k] * s_T[wv][p][cp];
      out[idx] = which ? v : v * tm.alpha_u;
    }
  }
}

}  // namespace

void launch_swhdg_element(const BlockDev &b, const SideTablesDev &st, const SwhElementDev &a, const TimeDev &tm,
                          hipStream_t stream) {
  if (b.e_count <= 0) return;
  MHA_REQUIRE(b.dim == 2 && b.n == 12 && 4 * st.nqs <= kHdgMaxPts, MHA_ERR_INVALID,
              "HDG element kernel: 2-D, three order-1 HGRAD variables, at most " << kHdgMaxPts / 4 << " points per side");
  const int grid = (b.e_count + kHdgWaves - 1) / kHdgWaves;
  hipLaunchKernelGGL(swhdg_element_kernel, dim3(grid), dim3(64 * kHdgWaves), 0, stream, b, st, a, tm);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
