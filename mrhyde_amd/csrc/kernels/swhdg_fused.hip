// swhdg_fused.hip -- the whole element step of the HDG subgrid solve of shallowwaterHybridized in ONE kernel: side terms
// (boundaryResidual on the four sides + computeFlux against the trace basis), volume terms (volumeResidual), their
// derivative blocks with respect to the 12 interior and 24 trace unknowns, and the static condensation -- one wavefront
// per element, the [36 x 37] augmented block never leaves the chip.  Only S[24][24], g[24] and du[12] are written.
//
// reference: SubGridDtN_Solver::assembleJacobianResidual (src/subgrid/subgridDtN_solver.cpp:681-903: volumeResidual +
// boundaryResidual with SFad arrays, element-local solve), updateFlux (:1542-1616: computeFlux integrated against the
// trace basis, sensitivities), the loop bookkeeping of nonlinearSolver (:909-1041); physics
// src/physics/shallowwaterHybridized.cpp:113-184 (volume), :190-263 (boundary), :270-368 (flux).
// Unfused equivalents, kept as the independent implementation the tests compare with: swhdg_element.hip (sides),
// point_engine.hip with swhdg_point (volume), subgrid.hip (combine), condense.hip.
//
// Layout per wavefront: lane = (point, direction) during the point phase (Dual numbers: direction 0 the value, 1..3
// d/dS_k, 4..6 d/dShat_k), lane = column of the augmented block afterwards (0..11 interior, 12..35 traces, 36 the
// right-hand side).  HBM traffic per element: 64 B vertices + 48 B LIDs + 96 B u (+ history) + 192 B traces in,
// 4608 + 192 + 96 B out.
#include <hip/hip_runtime.h>

#include "../../../include/mrhyde_amd.h"
#include "condense_core.hpp"
#include "device_math.hpp"
#include "launch.hpp"
#include "side_geometry.hpp"
#include "swhdg_side.hpp"

namespace mha {
namespace {

constexpr int kFuWaves = 4, kFuMaxSidePts = 16, kFuMaxVolPts = 9, kFuRows = 36, kFuInt = 12, kFuTrace = 24;
constexpr int kFuSchurLds = 672;  // doubles per wavefront: A_lu [24][12] + X [12][32], then P [24][25] over them

template <int MAXP>  // side points of an element (4 sides x points per side) held in registers during the column assembly
__global__ __launch_bounds__(64 * kFuWaves, 3) void swhdg_fused_kernel(BlockDev b, SideTablesDev st, SwhElementDev a, TimeDev tm,
                                                                     PhysParamsDev pp, SwhFusedOut o) {
  constexpr int DIM = 2, NN = 4;
  __shared__ double s_u[kFuWaves][12], s_ud[kFuWaves][12], s_l[kFuWaves][24];
  __shared__ double s_T[kFuWaves][kFuMaxSidePts][12];    // side functions at the side points: 4 N_a, 8 mu
  __shared__ double s_f[kFuWaves][kFuMaxSidePts][3];     // interface flux * w
  __shared__ double s_D[kFuWaves][kFuMaxSidePts][2][9];  // d flux / d S, d flux / d Shat (* w), [i][k]
  __shared__ double s_vT[kFuWaves][kFuMaxVolPts][12];    // volume points: N_a, d N_a / dx, d N_a / dy
  __shared__ double s_vr[kFuWaves][kFuMaxVolPts][9];     // (Sdot_i - source_i) w, then -F_i^x w, -F_i^y w
  __shared__ double s_vD[kFuWaves][kFuMaxVolPts][18];    // d(F_i^d)/dS_k * w: [i][d][k]
  __shared__ double s_vw[kFuWaves][kFuMaxVolPts];        // w = reference weight * det J
  __shared__ double s_sch[kFuWaves][kFuSchurLds];        // staging of the Schur product on the matrix cores
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int el = blockIdx.x * kFuWaves + wv;
  const bool active = el < b.e_count;
  const int e = b.e_begin + (active ? el : 0), nqs = st.nqs, npts = 4 * nqs, nq = b.nq;
  const int32_t *L = b.lids + (size_t)e * 12;
  int myrow = 0;
  if (active && lane < 12) {
    const int row = L[b.offsets[lane]];
    myrow = row;
    const double cu = tm.u[row];
    double ue = cu, ud = 0.0;
    if (tm.transient) {  // Workset::computeSolnTransientSeeded (workset.cpp:589-623)
      const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
      double beta_t = 0.0;
      for (int s = 1; s < tm.nsteps + 1; ++s) beta_t += tm.bdf[s] * cp[s - 1];
      beta_t *= tm.timewt;
      ue = tm.alpha_u * cu + beta_u;
      ud = tm.alpha_t * cu + beta_t;
    }
    s_u[wv][lane] = ue;
    s_ud[wv][lane] = ud;
  }
  if (active && lane >= 32 && lane < 56) s_l[wv][lane - 32] = a.lambda[(size_t)e * 24 + lane - 32];
  __syncthreads();
  const double *xn = b.nodes + (size_t)e * NN * DIM;
  if (active) {
    // ---- side points: one (point, direction) per lane (as swhdg_element.hip) ----
    // (MAXP = 8: 56 tasks, one per lane -- no loop, which also keeps the compiler from overlapping two iterations' state)
    for (int idx = lane; idx < npts * 7; idx += (MAXP == 8 ? 1 << 20 : 64)) {
      const int p = idx / 7, dir = idx - p * 7, s = p / nqs, q = p - s * nqs;
      const int edge = (s + 1) & 3;  // shards side 0,1,2,3 (bottom, right, top, left) -> HFACE edge 1,2,3,0
      double Ji[DIM * DIM], nrm[DIM], w, x[DIM];
      side_point<DIM>(xn, st, s, q, Ji, nrm, w, x);
      const double tc = (edge & 1) ? st.ip[(s * nqs + q) * DIM] : st.ip[(s * nqs + q) * DIM + 1];
      const double mu0 = 0.5 * (1.0 - tc), mu1 = 0.5 * (1.0 + tc);
      double S[3] = {0, 0, 0}, Sh[3];
      int eo = edge * 2;
      asm volatile("" : "+v"(eo));  // (a real per-lane address: left to itself the compiler reads all 24 trace values
                                    //  into registers and selects -- 48 registers held through the flux evaluation)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int dof = 0; dof < 4; ++dof) S[i] += s_u[wv][i * 4 + dof] * st.basis[(s * 4 + dof) * nqs + q];
        Sh[i] = s_l[wv][i * 8 + eo] * mu0 + s_l[wv][i * 8 + eo + 1] * mu1;
      }
      const int stype = a.side_types ? a.side_types[(size_t)e * 4 + s] : 0;
      Dual dS[3], dSh[3], f[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { dS[i] = mk(S[i], dir == 1 + i ? 1.0 : 0.0); dSh[i] = mk(Sh[i], dir == 4 + i ? 1.0 : 0.0); }
      swh_interface_flux_lean(stype, a.roe != 0, dS, dSh, a.farfield, nrm[0], nrm[1], a.g, f);
      if (dir == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) s_f[wv][p][i] = f[i].v * w;
#pragma unroll
        for (int dof = 0; dof < 4; ++dof) s_T[wv][p][dof] = st.basis[(s * 4 + dof) * nqs + q];
#pragma unroll
        for (int k = 0; k < 8; ++k) s_T[wv][p][4 + k] = (k >> 1) == edge ? ((k & 1) ? mu1 : mu0) : 0.0;
      } else {
        const int which = dir > 3, kk = (dir - 1) % 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) s_D[wv][p][which][i * 3 + kk] = f[i].d * w;
      }
    }
    // ---- volume points: (point, direction 0..3) per lane; shallowwaterHybridized::volumeResidual ----
    for (int idx = lane; idx < nq * 4; idx += (MAXP == 8 ? 1 << 20 : 64)) {  // (at most 9 x 4 tasks)
      const int q = idx >> 2, dir = idx & 3;
      double J[DIM * DIM] = {0, 0, 0, 0}, Ji[DIM * DIM], det, x[DIM] = {0, 0}, xi[DIM] = {0, 0};
      const double vx[4] = {-1.0, 1.0, 1.0, -1.0}, vy[4] = {-1.0, -1.0, 1.0, 1.0};  // reference vertices (shards order)
#pragma unroll
      for (int k = 0; k < NN; ++k) {
        const double nv = b.nodeval[k * nq + q];
        xi[0] += nv * vx[k];
        xi[1] += nv * vy[k];
#pragma unroll
        for (int r = 0; r < DIM; ++r) x[r] += xn[k * DIM + r] * nv;
      }
      // the source functions first, while nothing else is live (their closed forms are the largest code of the phase)
      double src[3] = {0.0, 0.0, 0.0};
      if (dir == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) src[i] = eval_func<DIM, false>(pp.f[i], e, q, nq, x);
      }
      asm volatile("" : "+v"(src[0]), "+v"(src[1]), "+v"(src[2]), "+v"(xi[0]), "+v"(xi[1]));
#pragma unroll
      for (int k = 0; k < NN; ++k)
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
          for (int cc = 0; cc < DIM; ++cc) J[r * DIM + cc] += xn[k * DIM + r] * b.nodegrad[(k * nq + q) * DIM + cc];
      invert<DIM>(J, Ji, det);
      const double w = b.ref_wts[q] * det;
      // HGRAD order 1 in dof order (x fastest): N_a = (1 + sx xi)(1 + sy eta) / 4
      double N[4], Gx[4], Gy[4];
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) {
        const double sx = (aa & 1) ? 1.0 : -1.0, sy = (aa & 2) ? 1.0 : -1.0;
        N[aa] = 0.25 * (1.0 + sx * xi[0]) * (1.0 + sy * xi[1]);
        const double gxi = 0.25 * sx * (1.0 + sy * xi[1]), get = 0.25 * sy * (1.0 + sx * xi[0]);
        Gx[aa] = gxi * Ji[0] + get * Ji[2];  // J^-T grad_ref
        Gy[aa] = gxi * Ji[1] + get * Ji[3];
      }
      double S[3] = {0, 0, 0}, Sd[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) { S[i] += s_u[wv][i * 4 + aa] * N[aa]; Sd[i] += s_ud[wv][i * 4 + aa] * N[aa]; }
      Dual dS[3], F[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) dS[i] = mk(S[i], dir == 1 + i ? 1.0 : 0.0);
      swh_flux_vector(dS, a.g, F);
      if (dir == 0) {
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) { s_vT[wv][q][aa] = N[aa]; s_vT[wv][q][4 + aa] = Gx[aa]; s_vT[wv][q][8 + aa] = Gy[aa]; }
        s_vw[wv][q] = w;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          s_vr[wv][q][i] = (Sd[i] - src[i]) * w;  // (v, dS/dt) - (v, source)
          s_vr[wv][q][3 + i] = -F[i][0].v * w;                                           // -(dv/dx, F_x)
          s_vr[wv][q][6 + i] = -F[i][1].v * w;                                           // -(dv/dy, F_y)
        }
      } else {
        const int kk = dir - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          s_vD[wv][q][(i * 2 + 0) * 3 + kk] = F[i][0].d * w;
          s_vD[wv][q][(i * 2 + 1) * 3 + kk] = F[i][1].d * w;
        }
      }
    }
  }
  __syncthreads();
  if (!active) return;
  // ---- column `lane` of the augmented block: rows (equation i, side function r') ----
  auto split = [](int r, int &i, int &rp) {
    if (r < 12) { i = r >> 2; rp = r & 3; }
    else { const int t = r - 12; i = t >> 3; rp = 4 + (t & 7); }
  };
  const int c = lane;
  int ck = 0, cp = 0;
  if (c < kFuRows) split(c, ck, cp);
  const bool rhs = c == kFuRows, inner = c < 12, on = c <= kFuRows;
  double col[kFuInt], low[kFuTrace];
#pragma unroll
  for (int r = 0; r < kFuInt; ++r) col[r] = 0.0;
#pragma unroll
  for (int r = 0; r < kFuTrace; ++r) low[r] = 0.0;
  if (on && !rhs) {
    // entry (r, c) = sum_p T[p][r'] D[p][which][i][k_c] T[p][c'] (+ the volume terms for r, c < 12): the column's part
    // E[p][i] = D[p][which][i][k_c] T[p][c'] is formed once per lane, the row's factor T[p][r'] is wave-uniform
    const int which = inner ? 0 : 1;
    const double scale = inner ? tm.alpha_u : 1.0;
    double E[MAXP][3];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const double tc = p < npts ? s_T[wv][p][cp] * scale : 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) E[p][i] = p < npts ? s_D[wv][p][which][i * 3 + ck] * tc : 0.0;
    }
#pragma unroll
    for (int rp = 0; rp < 12; ++rp) {
      double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        if (p < npts) {  // uniform
          const double t = s_T[wv][p][rp];
#pragma unroll
          for (int i = 0; i < 3; ++i) v[i] += t * E[p][i];
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (rp < 4) col[i * 4 + rp] = v[i];
        else low[i * 8 + rp - 4] = v[i];
      }
    }
    if (inner) {  // volume: -alpha_u (dN_a/dx_d, dF_i^d/dS_k N_b) + alpha_t (N_a, N_b) on the diagonal variable blocks
#pragma unroll
      for (int q = 0; q < kFuMaxVolPts; ++q) {
        if (q < nq) {  // uniform
          const double nb = s_vT[wv][q][cp];
          const double mass = tm.alpha_t * s_vw[wv][q] * nb;
          double dx[3], dy[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            dx[i] = -tm.alpha_u * s_vD[wv][q][(i * 2 + 0) * 3 + ck] * nb;
            dy[i] = -tm.alpha_u * s_vD[wv][q][(i * 2 + 1) * 3 + ck] * nb;
          }
#pragma unroll
          for (int aa = 0; aa < 4; ++aa) {
            const double n_a = s_vT[wv][q][aa], gx = s_vT[wv][q][4 + aa], gy = s_vT[wv][q][8 + aa];
#pragma unroll
            for (int i = 0; i < 3; ++i) col[i * 4 + aa] += dx[i] * gx + dy[i] * gy + (i == ck ? mass * n_a : 0.0);
          }
        }
      }
    }
  }
  // right-hand side (column kFuRows, held by one lane): its 36 entries are formed one per LANE -- row c = (equation ck,
  // side function cp) -- and handed to that lane's registers with v_readlane; as a branch of the column code above it was
  // a second pass of the whole wavefront for the benefit of a single lane
  {
    double mine = 0.0;
    if (c < kFuRows) {
      for (int p = 0; p < npts; ++p) mine -= s_f[wv][p][ck] * s_T[wv][p][cp];
      if (cp < 4 && c < 12)
        for (int q = 0; q < nq; ++q)
          mine -= s_vr[wv][q][ck] * s_vT[wv][q][cp] + s_vr[wv][q][3 + ck] * s_vT[wv][q][4 + cp] + s_vr[wv][q][6 + ck] * s_vT[wv][q][8 + cp];
    }
#pragma unroll
    for (int r = 0; r < kFuInt; ++r) {
      const double v = readlane_f64(mine, r);
      if (rhs) col[r] = v;
    }
#pragma unroll
    for (int r = 0; r < kFuTrace; ++r) {
      const double v = readlane_f64(mine, kFuInt + r);
      if (rhs) low[r] = v;
    }
  }
  // ---- loop bookkeeping of nonlinearSolver (subgrid.hip: combine), on the interior residual ----
  if (rhs && o.pass >= 0 && o.rn0) {
    double nrm = 0.0;
#pragma unroll
    for (int r = 0; r < 12; ++r) nrm = fmax(nrm, fabs(col[r]));
    const int64_t eo = e - b.e_begin;
    if (o.pass == 0) {
      o.rn0[eo] = nrm;
      o.scaled[eo] = nrm > 0.0 ? 1.0 : 0.0;
      o.iters[eo] = 1;
      o.active[eo] = (nrm > 0.0 ? 1.0 : 0.0) > o.tol ? 1 : 0;
    } else if (o.active[eo]) {
      const double sc = nrm / o.rn0[eo];
      o.scaled[eo] = sc;
      o.iters[eo] += 1;
      o.active[eo] = sc > o.tol ? 1 : 0;
    }
  }
  // ---- static condensation in registers ----
  const int64_t eo = e - b.e_begin;
  if (!gauss_jordan_columns_static<kFuInt>(col)) { if (lane == 0 && o.singular) atomicAdd(o.singular, 1); return; }
  if (o.update_u) {  // sol += du for the elements still in their loop (subgrid.hip: update), fused
    const bool go = !o.active || o.active[eo];
    double dui = 0.0;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      const double v = readlane_f64(col[r], kFuRows);
      if (lane == r) dui = v;
    }
    if (go && lane < 12) o.update_u[myrow] += dui;
  }
  // ---- Schur complement S = A_ll - A_lu X_ul, g = r_l - A_lu x_r on the matrix cores ----
  // (schur_from_registers forms the same numbers with 576 v_readlane pairs and 288 FMAs per lane: a fifth of the kernel's
  // instructions.)  A_lu (rows of `low` in lanes < 12) and X = [X_ul | x_r] (`col` of lanes 12..36) go through the
  // wavefront's staging buffer into MFMA operand layout: P = A_lu X as 2 x 2 tiles of 16 x 16, K = 12 = 3 steps; P comes
  // back through the same buffer in column-per-lane order and is subtracted from the lane's column of [A_ll | r_l].
  {
    double *sb = s_sch[wv];
    auto wave_sync = [] {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (lane < kFuInt) {
#pragma unroll
      for (int a = 0; a < kFuTrace; ++a) sb[a * kFuInt + lane] = low[a];
    } else if (lane <= kFuRows) {
#pragma unroll
      for (int i = 0; i < kFuInt; ++i) sb[kFuTrace * kFuInt + i * 32 + (lane - kFuInt)] = col[i];
    }
    wave_sync();
    const int l15 = lane & 15, g4 = lane >> 4;
    double av[2][3], bv[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        av[t][ks] = sb[(t * 16 + l15) * kFuInt + 4 * ks + g4];  // rows >= 24 read past A_lu: they only reach P rows >= 24, never read
        bv[t][ks] = sb[kFuTrace * kFuInt + (4 * ks + g4) * 32 + t * 16 + l15];  // columns > 24 likewise
      }
    wave_sync();  // operands are in registers: P may overwrite them
    typedef double v4d __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        v4d d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) d = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt][ks], bv[ct][ks], d, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = rt * 16 + g4 + 4 * u, cc = ct * 16 + l15;
          if (row < kFuTrace && cc <= kFuTrace) sb[row * 25 + cc] = d[u];
        }
      }
    wave_sync();
    if (o.du && lane == kFuRows) {
#pragma unroll
      for (int i = 0; i < kFuInt; ++i) o.du[eo * kFuInt + i] = col[i];
    }
    if (lane >= kFuInt && lane <= kFuRows) {
      const int bcol = lane - kFuInt;
#pragma unroll
      for (int a = 0; a < kFuTrace; ++a) {
        const double sacc = low[a] - sb[a * 25 + bcol];
        if (lane < kFuRows) { if (o.schur) o.schur[(eo * kFuTrace + a) * kFuTrace + bcol] = sacc; }
        else if (o.gvec) o.gvec[eo * kFuTrace + a] = sacc;
      }
    }
  }
}

}  // namespace

void launch_swhdg_fused(const BlockDev &b, const SideTablesDev &st, const SwhElementDev &a, const TimeDev &tm,
                        const PhysParamsDev &pp, const SwhFusedOut &o, hipStream_t stream) {
  if (b.e_count <= 0) return;
  MHA_REQUIRE(b.dim == 2 && b.n == 12 && 4 * st.nqs <= kFuMaxSidePts && b.nq <= kFuMaxVolPts, MHA_ERR_INVALID,
              "fused HDG element kernel: 2-D, three order-1 HGRAD variables, at most " << kFuMaxSidePts / 4 << " points per side and "
                                                                                      << kFuMaxVolPts << " volume points");
  for (int i = 0; i < 3; ++i)
    MHA_REQUIRE(pp.f[i].kind != MHA_FUNC_EXPRESSION, MHA_ERR_INVALID, "fused HDG element kernel: deck-string sources go through the unfused path");
  const int grid = (b.e_count + kFuWaves - 1) / kFuWaves;
  if (4 * st.nqs <= 8) hipLaunchKernelGGL(swhdg_fused_kernel<8>, dim3(grid), dim3(64 * kFuWaves), 0, stream, b, st, a, tm, pp, o);
  else hipLaunchKernelGGL(swhdg_fused_kernel<kFuMaxSidePts>, dim3(grid), dim3(64 * kFuWaves), 0, stream, b, st, a, tm, pp, o);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
