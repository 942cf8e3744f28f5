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
// The sinprod source is separable on axis-aligned elements (checked per element): DIM x (order + 1) sines instead of
// DIM x (order + 1)^DIM.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int cpow(int b, int e) { return e == 0 ? 1 : b * cpow(b, e - 1); }
constexpr int kK1tThreads = 256;        // thread-per-element form
constexpr int kK1wThreads = kK1PlanThreads;  // workgroup-merged form

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

// 64 consecutive elements per wavefront.  Their LID lists (64 x N ints) and geometry records (64 x 20 doubles) are
// contiguous in memory: fetched with coalesced loads into a per-wave LDS buffer and read back one record per lane
// (odd strides: conflict-free), instead of 64 different cache lines per load instruction.
// SMALL: the launcher has checked |freq_d x_d| < 1e5 over the mesh (closed-form source): sines without the full-range fallback
template <int DIM, int P, bool TR, bool EXPR, bool SMALL>
#ifndef MHA_K1_WAVES
#define MHA_K1_WAVES 2
#endif
__global__ __launch_bounds__(kK1tThreads, MHA_K1_WAVES) void thermal_affine_residual_kernel(BlockDev b, ThermalDev ph,
                                                                              const double *__restrict__ geo,
                                                                              const AffineTables1D *__restrict__ tabp, double *res, int dbg) {
  // The 60 table entries are scalar operands of several hundred FMAs.  Passed by value they sat in 120 SGPRs for the
  // whole kernel (with the other arguments: > 1000 SGPR spill instructions); behind a pointer the compiler fetches them
  // through the scalar cache next to their uses.
  const AffineTables1D &tab = *tabp;
  constexpr int M = P + 1, N = cpow(M, DIM);
  constexpr int GS = kGeoRec + 1;  // padded record stride (doubles): odd -> one record per lane without bank conflicts
  __shared__ int s_lid[kK1tThreads / 64][64 * N];
  __shared__ double s_geo[kK1tThreads / 64][64 * GS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int first = (blockIdx.x * (kK1tThreads / 64) + wave) * 64;  // first element of this wavefront (inside the range)
  if (first >= b.e_count) return;
  const int nvalid = min(64, b.e_count - first);
  const bool active = lane < nvalid;
  const int e = b.e_begin + first + (active ? lane : 0);
  {
    const int32_t *src = b.lids + (size_t)(b.e_begin + first) * N;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int idx = k * 64 + lane;
      s_lid[wave][idx] = idx < nvalid * N ? src[idx] : 0;
    }
    const double *gsrc = geo + (size_t)(b.e_begin + first) * kGeoRec;
#pragma unroll
    for (int k = 0; k < kGeoRec; ++k) {
      const int idx = k * 64 + lane;
      s_geo[wave][(idx / kGeoRec) * GS + idx % kGeoRec] = idx < nvalid * kGeoRec ? gsrc[idx] : 0.0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int *L = &s_lid[wave][(active ? lane : 0) * N];
  const double *g = &s_geo[wave][(active ? lane : 0) * GS];
  const TimeDev &tm = ph.time;

  // performGather + computeSoln*Seeded values, basis (tensor) order
  double U[N], Ud[TR ? N : 1];
#pragma unroll
  for (int ib = 0; ib < N; ++ib) {
    const int row = L[b.offsets[ib]];
    const double cu = (dbg & 4) ? 1.0 : tm.u[row];
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

  // closed-form source amp prod_d sin(freq_d x_d) on an axis-aligned element: x_d at point q depends on q_d only, so
  // DIM x M sines serve all M^DIM points (the same values the general evaluation produces point by point)
  bool separable = ph.source.kind != MHA_FUNC_CONSTANT && ph.source.kind != MHA_FUNC_IP_ARRAY && ph.source.kind != MHA_FUNC_EXPRESSION;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM; ++c)
      if (r != c && J[r][c] != 0.0) separable = false;
  double s1d[DIM][M];
  if (dbg & 2) separable = false;
  if (__builtin_amdgcn_ballot_w64(!separable) == 0) {
#pragma unroll
    for (int d = 0; d < DIM; ++d)
#pragma unroll
      for (int q = 0; q < M; ++q) s1d[d][q] = SMALL ? sin_reduced(ph.source.freq[d] * (xc[d] + J[d][d] * tab.gp[q])) : sin_moderate(ph.source.freq[d] * (xc[d] + J[d][d] * tab.gp[q]));
  }

  // point loop: W accumulates what multiplies the basis VALUES at each point (the flux terms enter through D^T).  Two
  // copies, chosen per wavefront: every lane's element is axis-aligned (separable source) or the general evaluation
  auto point_loop = [&](auto sep_tag, double *W) {
    constexpr bool SEP = decltype(sep_tag)::value;
#pragma unroll
    for (int pt = 0; pt < N; ++pt) {
      const int q0 = pt % M, q1 = (pt / M) % M, q2 = pt / (M * M);
      const int qd[3] = {q0, q1, q2};
      double gh[DIM], wq = 1.0;
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
      double f;
      if constexpr (SEP) {
        f = ph.source.amp;
#pragma unroll
        for (int d = 0; d < DIM; ++d) f *= s1d[d][qd[d]];
      } else {
        double x[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          double s = xc[r];
#pragma unroll
          for (int c = 0; c < DIM; ++c) s += J[r][c] * tab.gp[qd[c]];
          x[r] = s;
        }
        f = eval_func<DIM, EXPR, SMALL>(ph.source, e, pt, N, x);
      }
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
      __builtin_amdgcn_sched_barrier(0);  // one point at a time: interleaved, the points' temporaries cost another 150 registers
    }
  };
  double W[N];
#pragma unroll
  for (int i = 0; i < N; ++i) W[i] = 0.0;
  if (__builtin_amdgcn_ballot_w64(!separable) == 0) point_loop(std::true_type(), W);
  else point_loop(std::false_type(), W);
  // point weights -> residual rows
  apply1d<DIM, M, 0, false>(W, tab.phi);
  apply1d<DIM, M, 1, false>(W, tab.phi);
  if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, false>(W, tab.phi);
  // the global vector receives -res.val(); fixed rows are skipped (assemblyManager.cpp:4075, 4094)
  // Consecutive lanes are consecutive elements; when those are x-neighbours (element numbering x fastest -- any mesh
  // where that holds for most of a wavefront) the left face nodes of lane l are the right face nodes of lane l-1: the
  // two contributions are added in registers (DPP shuffles) and leave as ONE atomic, a third fewer atomics for Q2 hexes.
  // Decided from the LIDs themselves, so a mesh without that structure simply merges nothing.
  int rowv[N];
#pragma unroll
  for (int ib = 0; ib < N; ++ib) rowv[ib] = active ? L[b.offsets[ib]] : -2 - lane;
  if (!(dbg & 8)) {
#pragma unroll
    for (int il = 0; il < N; il += M) {  // dof index x fastest: il = (0, iy, iz), ir = (M-1, iy, iz)
      const int ir = il + M - 1;
      const int prow = __shfl_up(rowv[ir], 1);
      const double pw = __shfl_up(W[ir], 1);
      const bool take = lane > 0 && prow == rowv[il];
      if (take) W[il] += pw;
      const int given = __shfl_down((int)take, 1);
      if (lane < 63 && given) rowv[ir] = -1;  // the next lane carries this node
    }
  }
  if (!(dbg & 1)) {
#pragma unroll
    for (int ib = 0; ib < N; ++ib) {
      const int row = rowv[ib];
      if (row >= 0 && !(b.fixed && b.fixed[row])) atomicAdd(res + row, -W[ib]);
    }
  }
}

// ---- workgroup-merged form (default) ------------------------------------------------------------------------------
// The same element arithmetic, one thread per element, around a host-made plan (K1PlanDev, AssemblyManager::
// prepareRowOwner): a workgroup takes 256 elements (consecutive ones, or neighbours along a Morton curve when the
// numbering scatters them); the DISTINCT rows their dofs touch are listed once, and every (element, dof) knows its
// position in that list (16 bits, stored [workgroup][dof in basis order][thread]: coalesced).
//   A  the seeded solution of the listed rows -> an LDS table (coalesced loads of an ascending row list, each row once)
//   B  a thread's 27 nodal values come out of the table
//   C  the element arithmetic in registers (sum factorisation through the point values, as above)
//   D  -r_i goes back into the (zeroed) table with ds_add_f64: contributions of the workgroup's elements to a shared
//      row meet in LDS
//   E  the table leaves with ONE atomic per listed row (9.9 per Q2 hex instead of 18-27)
// Against the form above (27 scattered loads and 18-27 atomics per lane, the LID lists and geometry records staged through
// 70 KB of LDS, 256 registers + spills, two wavefronts per SIMD): no LID traffic, 31 KB of LDS and a register budget of
// three wavefronts per SIMD (no spills).
// SEPK: the host has checked that every element is axis-aligned and the source is the closed form amp prod sin(freq_d
// x_d): the separable evaluation is the only one compiled in (the general copy of the point loop, with its 81 inlined
// sines, costs the register budget of the whole kernel)
template <int DIM, int P, bool TR, bool EXPR, bool SMALL, bool SEPK>
__global__ __launch_bounds__(kK1wThreads, (TR || EXPR || !SEPK) ? 2 : 3) void thermal_affine_residual_wg_kernel(
    BlockDev b, ThermalDev ph, const double *__restrict__ geo, const AffineTables1D *__restrict__ tabp, K1PlanDev pl,
    double *res, int dbg) {
  const AffineTables1D &tab = *tabp;
  constexpr int M = P + 1, N = cpow(M, DIM);
  extern __shared__ double s_tab[];  // [max_rows] seeded u, then the accumulated -r; TR: [max_rows] u_dot behind it;
                                     // then [max_rows] ints: the listed rows (~row = fixed), kept for E; then
                                     // [N][threads] 16-bit positions, kept for D
  int *s_row = reinterpret_cast<int *>(s_tab + (TR ? 2 : 1) * pl.max_rows);
  const int tid = threadIdx.x;
  const int g = blockIdx.x;
  const int r0 = pl.wg_row_ptr[g], nr = pl.wg_row_ptr[g + 1] - r0;
  const bool active = g * kK1wThreads + tid < b.e_count;
  const int e = pl.wg_elems[g * kK1wThreads + tid];
  const TimeDev &tm = ph.time;
  // the positions of this thread's dofs: requested first, used after the barrier (their latency runs under phase A)
  int kpos[N];
  {
    const uint16_t *loc = pl.loc + (size_t)g * N * kK1wThreads + tid;
#pragma unroll
    for (int ib = 0; ib < N; ++ib) kpos[ib] = loc[ib * kK1wThreads];
  }
  // ---- A. performGather + computeSoln*Seeded values (workset.cpp:589-623), once per listed row ----
  // (four rows per thread and pass, every load of a stage issued before the first use: the loop would otherwise run its
  //  dependent loads -- row id, then value -- one row at a time)
  constexpr int KB = 4;
  for (int base = tid; base < nr; base += KB * kK1wThreads) {
    int row[KB];
    double cu[KB];
    bool fx[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) row[k] = pl.wg_rows[r0 + min(base + k * kK1wThreads, nr - 1)];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      cu[k] = tm.u[row[k]];
      fx[k] = b.fixed && b.fixed[row[k]];
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int i = base + k * kK1wThreads;
      if (i >= nr) break;
      s_row[i] = fx[k] ? ~row[k] : row[k];
      double ue = cu[k];
      if constexpr (TR) {
        const double *cp = tm.u_prev + (size_t)row[k] * tm.nsteps;
        const double *cs = tm.u_stage + (size_t)row[k] * tm.nstages;
        double beta_u = (1.0 - tm.alpha_u) * cp[0];
        for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
        double beta_t = 0.0;
        for (int s = 1; s < tm.nsteps + 1; ++s) beta_t += tm.bdf[s] * cp[s - 1];
        beta_t *= tm.timewt;
        ue = tm.alpha_u * cu[k] + beta_u;
        s_tab[pl.max_rows + i] = tm.alpha_t * cu[k] + beta_t;
      }
      s_tab[i] = ue;
    }
  }
  // the element's geometry record: G (upper triangle), det, then J and the centroid (requested before the barrier)
  const double *gcen = geo + (size_t)e * kGeoRec;  // the element's own record: centroid
  const double *grec = pl.shape ? pl.shape + (size_t)pl.shape_idx[e] * 16 : gcen;  // shape part: through the database
  double G[DIM][DIM], xc[DIM], Jd[DIM];
  {
    int k = 0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
      for (int c = a; c < DIM; ++c) { G[a][c] = grec[k]; G[c][a] = G[a][c]; ++k; }
  }
  double det = grec[kGeoDet];
  bool separable = ph.source.kind != MHA_FUNC_CONSTANT && ph.source.kind != MHA_FUNC_IP_ARRAY && ph.source.kind != MHA_FUNC_EXPRESSION;
#pragma unroll
  for (int r = 0; r < DIM; ++r) {
    xc[r] = gcen[kGeoXc + r];
    Jd[r] = grec[kGeoJ + r * DIM + r];
#pragma unroll
    for (int c = 0; c < DIM; ++c)
      if (r != c && grec[kGeoJ + r * DIM + c] != 0.0) separable = false;
  }
  __syncthreads();
  // ---- B. nodal values of this thread's element, basis (tensor) order; the positions are parked in LDS for D ----
  uint16_t *s_loc = reinterpret_cast<uint16_t *>(s_row + pl.max_rows) + tid;  // [N][threads]
  double U[N], Ud[TR ? N : 1];
#pragma unroll
  for (int ib = 0; ib < N; ++ib) {
    const int k = kpos[ib];
    U[ib] = s_tab[k];
    if constexpr (TR) Ud[ib] = s_tab[pl.max_rows + k];
    s_loc[ib * kK1wThreads] = (uint16_t)k;
  }
  __syncthreads();
  for (int i = tid; i < nr; i += kK1wThreads) s_tab[i] = 0.0;  // (the barrier before D orders this against the adds)
  // ---- C. element arithmetic ----
  apply1d<DIM, M, 0, true>(U, tab.phi);
  apply1d<DIM, M, 1, true>(U, tab.phi);
  if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, true>(U, tab.phi);
  if constexpr (TR) {
    apply1d<DIM, M, 0, true>(Ud, tab.phi);
    apply1d<DIM, M, 1, true>(Ud, tab.phi);
    if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, true>(Ud, tab.phi);
  }
  const double kap = ph.diff.amp, rc = ph.rho.amp * ph.cp.amp;  // element-wise constants on this path
  if (dbg & 2) separable = false;
  const bool all_sep = SEPK || __builtin_amdgcn_ballot_w64(!separable) == 0;
  double s1d[DIM][M];
  if (all_sep) {
#pragma unroll
    for (int d = 0; d < DIM; ++d)
#pragma unroll
      for (int q = 0; q < M; ++q) {
        const double arg = ph.source.freq[d] * (xc[d] + Jd[d] * tab.gp[q]);
        s1d[d][q] = SMALL ? sin_reduced(arg) : sin_moderate(arg);
        __builtin_amdgcn_sched_barrier(0);  // one sine at a time
      }
  }
  double W[N];
#pragma unroll
  for (int i = 0; i < N; ++i) W[i] = 0.0;
  auto point_loop = [&](auto sep_tag) {
    constexpr bool SEP = decltype(sep_tag)::value;
#pragma unroll
    for (int pt = 0; pt < N; ++pt) {
      const int q0 = pt % M, q1 = (pt / M) % M, q2 = pt / (M * M);
      const int qd[3] = {q0, q1, q2};
      // One point at a time: the arithmetic is pure, so the optimiser is free to interleave all N points (and then needs
      // twice the register file; a sched_barrier does not stop it, the IR passes move the arithmetic across it).  Empty
      // volatile asms keep their order: a point starts behind this one and its updates of W end in the ones below.
      asm volatile("" : "+v"(U[pt]));
      double gh[DIM], wq = 1.0;
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
      double f;
      if constexpr (SEP) {
        f = ph.source.amp;
#pragma unroll
        for (int d = 0; d < DIM; ++d) f *= s1d[d][qd[d]];
      } else {
        double x[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          double s = xc[r];
#pragma unroll
          for (int c = 0; c < DIM; ++c) s += grec[kGeoJ + r * DIM + c] * tab.gp[qd[c]];
          x[r] = s;
        }
        f = eval_func<DIM, EXPR, SMALL>(ph.source, e, pt, N, x);
      }
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
          asm volatile("" : "+v"(W[dst]));
        }
      }
    }
  };
  if constexpr (SEPK) {
    point_loop(std::true_type());
  } else {
    if (all_sep) point_loop(std::true_type());
    else point_loop(std::false_type());
  }
  apply1d<DIM, M, 0, false>(W, tab.phi);
  apply1d<DIM, M, 1, false>(W, tab.phi);
  if constexpr (DIM == 3) apply1d<DIM, M, DIM - 1, false>(W, tab.phi);
  __syncthreads();
  // ---- D. the global vector receives -res.val() (assemblyManager.cpp:4075, 4094): met in LDS first ----
  if (active) {
#pragma unroll
    for (int ib = 0; ib < N; ++ib) atomicAdd(&s_tab[s_loc[ib * kK1wThreads]], -W[ib]);
  }
  __syncthreads();
  // ---- E. one atomic per listed row, fixed rows skipped; the row ids come from LDS (a load from memory here would be
  //      a latency nothing hides: 12 us of the kernel's 74 when the ids and flags were fetched again) ----
  if (!(dbg & 1)) {
    for (int i = tid; i < nr; i += kK1wThreads) {
      const int row = s_row[i];
      if (row >= 0) atomicAdd(res + row, s_tab[i]);
    }
  }
}

// device copy of a table, made once per (device, content) and kept for the life of the process
const AffineTables1D *device_copy(const AffineTables1D &tab) {
  struct Entry { int device; AffineTables1D host; AffineTables1D *dev; };
  static std::mutex mu;
  static std::vector<Entry> cache;
  int device = 0;
  MHA_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mu);
  for (const Entry &e : cache)
    if (e.device == device && std::memcmp(&e.host, &tab, sizeof(AffineTables1D)) == 0) return e.dev;
  AffineTables1D *d = nullptr;
  MHA_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(AffineTables1D)));
  MHA_HIP(hipMemcpy(d, &tab, sizeof(AffineTables1D), hipMemcpyHostToDevice));
  cache.push_back({device, tab, d});
  return d;
}

template <int DIM, int P>
void launch_t(const BlockDev &b, const ThermalDev &ph, const double *geo, const AffineTables1D &tab_host, const K1PlanDev *plan,
              double *res, bool small_args, hipStream_t stream) {
  if (b.e_count <= 0) return;
  const AffineTables1D *tab = device_copy(tab_host);
  static const int dbg_wg = [] { const char *m = std::getenv("MHA_K1_DBG"); return m ? std::atoi(m) : 0; }();
  if (plan && plan->loc) {  // workgroup-merged form (the plan covers the block's elements from e_begin = 0)
    MHA_REQUIRE(b.e_begin == 0 && b.e_count == plan->num_elems, MHA_ERR_INVALID, "K1 plan: element range mismatch");
    const int grid = (b.e_count + kK1wThreads - 1) / kK1wThreads;
    const bool tr = ph.time.transient != 0;
    const size_t lds = sizeof(double) * (size_t)plan->max_rows * (tr ? 2 : 1) + sizeof(int) * (size_t)plan->max_rows + sizeof(uint16_t) * cpow(P + 1, DIM) * kK1wThreads;
    auto go = [&](auto kern) {
      if (lds > 64 * 1024) MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(kK1wThreads), lds, stream, b, ph, geo, tab, *plan, res, dbg_wg);
    };
    const bool sinprod = ph.source.kind != MHA_FUNC_CONSTANT && ph.source.kind != MHA_FUNC_IP_ARRAY && ph.source.kind != MHA_FUNC_EXPRESSION;
    if (has_expression(ph.source)) {
      if (tr) go(thermal_affine_residual_wg_kernel<DIM, P, true, true, false, false>);
      else go(thermal_affine_residual_wg_kernel<DIM, P, false, true, false, false>);
    } else if (plan->axis_aligned && sinprod && !(dbg_wg & 2)) {
      if (small_args) {
        if (tr) go(thermal_affine_residual_wg_kernel<DIM, P, true, false, true, true>);
        else go(thermal_affine_residual_wg_kernel<DIM, P, false, false, true, true>);
      } else {
        if (tr) go(thermal_affine_residual_wg_kernel<DIM, P, true, false, false, true>);
        else go(thermal_affine_residual_wg_kernel<DIM, P, false, false, false, true>);
      }
    } else if (small_args) {
      if (tr) go(thermal_affine_residual_wg_kernel<DIM, P, true, false, true, false>);
      else go(thermal_affine_residual_wg_kernel<DIM, P, false, false, true, false>);
    } else {
      if (tr) go(thermal_affine_residual_wg_kernel<DIM, P, true, false, false, false>);
      else go(thermal_affine_residual_wg_kernel<DIM, P, false, false, false, false>);
    }
    MHA_HIP(hipGetLastError());
    return;
  }
  const int grid = (b.e_count + kK1tThreads - 1) / kK1tThreads;  // a wavefront takes 64 consecutive elements
  const bool tr = ph.time.transient != 0;
  static const int dbg = [] { const char *m = std::getenv("MHA_K1_DBG"); return m ? std::atoi(m) : 0; }();  // profiling aid: 1 no atomics, 2 general source evaluation, 4 no gather
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(kK1tThreads), 0, stream, b, ph, geo, tab, res, dbg); };
  if (has_expression(ph.source)) {
    if (tr) go(thermal_affine_residual_kernel<DIM, P, true, true, false>);
    else go(thermal_affine_residual_kernel<DIM, P, false, true, false>);
  } else if (small_args) {
    if (tr) go(thermal_affine_residual_kernel<DIM, P, true, false, true>);
    else go(thermal_affine_residual_kernel<DIM, P, false, false, true>);
  } else {
    if (tr) go(thermal_affine_residual_kernel<DIM, P, true, false, false>);
    else go(thermal_affine_residual_kernel<DIM, P, false, false, false>);
  }
  MHA_HIP(hipGetLastError());
}

}  // namespace

bool thermal_affine_residual_supported(int dim, int order, int nq1) {
  return nq1 == order + 1 && ((dim == 2 && (order == 1 || order == 2 || order == 4)) || (dim == 3 && (order == 1 || order == 2)));
}

void launch_thermal_affine_residual(int dim, int order, const BlockDev &b, const ThermalDev &ph, const double *geo,
                                    const AffineTables1D &tab, const K1PlanDev *plan, double *res, const double *max_abs_coord,
                                    hipStream_t stream) {
  // closed-form source amp prod sin(freq_d x_d): arguments bounded by |freq_d| max|x_d| over the mesh
  bool small_args = ph.source.kind != MHA_FUNC_CONSTANT && ph.source.kind != MHA_FUNC_IP_ARRAY && ph.source.kind != MHA_FUNC_EXPRESSION;
  for (int d = 0; d < dim; ++d) small_args = small_args && std::fabs(ph.source.freq[d]) * max_abs_coord[d] < 0.5e5;
  if (dim == 2 && order == 1) return launch_t<2, 1>(b, ph, geo, tab, plan, res, small_args, stream);
  if (dim == 2 && order == 2) return launch_t<2, 2>(b, ph, geo, tab, plan, res, small_args, stream);
  if (dim == 2 && order == 4) return launch_t<2, 4>(b, ph, geo, tab, plan, res, small_args, stream);
  if (dim == 3 && order == 1) return launch_t<3, 1>(b, ph, geo, tab, plan, res, small_args, stream);
  if (dim == 3 && order == 2) return launch_t<3, 2>(b, ph, geo, tab, plan, res, small_args, stream);
  MHA_REQUIRE(false, MHA_ERR_INVALID, "thread-per-element residual kernel: unsupported (dim, order)");
}

}  // namespace mha
