// porous_element.hip -- porousMixed volume terms, one THREAD per element.
//
// The lowest-order mixed element has 1 + 2*dim dofs (7 on a hex) and 2^dim integration points: far too small for a
// wavefront.  Here every thread owns one element and keeps the whole element matrix in registers; the module is linear,
// so the Sacado derivative array of the reference (porousMixed.cpp:158-338) is written down directly:
//   res_p      = sum_q (source - div u) w                      d res_p / d u_j   = -alpha_u div_j w
//   res_{u,i}  = sum_q (Kinv u . v_i / mobility - p div_i) w   d res_{u,i}/d p   = -alpha_u div_i w
//                                                              d res_{u,i}/d u_j = alpha_u (v_i . Kinv v_j / mobility) w
// with v_i = s_i J phihat_i / detJ, div_i = s_i divhat_i / detJ (HDIVtransformVALUE / DIV,
// discretizationInterface.cpp:1019,1053; s_i the orientation sign), phihat_{2c+h} = (1 -/+ x_c)/2 e_c.
// Output: dense local_J / local_res in LID-position order (updateJac / updateRes convention), from which
// kernels/row_gather.hip builds the CRS rows -- or (DIRECT, round 3, the default of the assembly) straight into the CRS:
// two elements of a conforming lowest-order mixed mesh share exactly one dof, so every matrix entry except the diagonal
// of a face row has ONE contributing element and is stored by that element's thread through the element-major slot
// map; its residual entry and diagonal part of each row go into that row's 4-double record (two incident elements),
// which a finishing pass (one thread per row, coalesced) turns into the residual entry and the diagonal entry.  No dense element matrices (784 B per element written and read back), no second pass over the matrix.  Same gather / seeding conventions as the point engine.
// A thread's 24 vertex coordinates and 56 results are contiguous per ELEMENT, i.e. 192 / 392 bytes apart between
// lanes.  Two forms: (DOF = false, the public updateJac / updateRes arrays) the workgroup's 128 elements are staged through
// LDS and move to and from memory as flat, fully coalesced arrays in LID-position order -- 448 B of LDS per thread, i.e.
// four wavefronts per CU; (DOF = true, the private scratch of the row-gather path) no staging and no barrier: every thread parks
// its vertices in its own 192 B of LDS (48 registers less) and writes its results from registers at compile-time offsets,
// rows and columns in dof order (the row gather maps positions to dofs); two wavefronts per SIMD instead of one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int kPorousThreads = 128;
#ifndef MHA_POROUS_RES_WAVES
#define MHA_POROUS_RES_WAVES 2
#endif

template <int DIM, bool EXPR, bool DOF, bool DIRECT = false, bool RESONLY = false>
__device__ __forceinline__ void porous_element_body(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp,
                                                    const TimeDev &tm, const ElemOut &out) {
  constexpr int NN = 1 << DIM, NU = 2 * DIM, N = 1 + NU;
  extern __shared__ double sm[];  // in: [128][NN*DIM] vertices; out: [128][N*N] element matrices + [128][N] residuals
  const int tid = threadIdx.x, e0 = blockIdx.x * kPorousThreads;
  const int cnt = min(kPorousThreads, b.e_count - e0);
  if constexpr (!DOF) {
    const double *src = b.nodes + (size_t)(b.e_begin + e0) * NN * DIM;
    for (int i = tid; i < cnt * NN * DIM; i += kPorousThreads) sm[i] = src[i];
    __syncthreads();
  }
  const bool active = tid < cnt;
  const int eidx = b.e_begin + e0 + (active ? tid : 0), NQ = vl.nq;
  const int e = (DIRECT && out.direct_elist) ? out.direct_elist[eidx] : eidx;
  const int32_t *L = b.lids + (size_t)e * N;
  // gather + seeding values; positions: p at offsets[0], u_i at offsets[1 + i]
  int pos[N];
  double u[N], sg[N];
#pragma unroll
  for (int f = 0; f < N; ++f) {
    pos[f] = b.offsets[f];
    const int row = L[pos[f]];
    sg[f] = (vl.orient && f > 0) ? (double)vl.orient[(size_t)e * N + f] : 1.0;
    const double cu = tm.u[row];
    double ue = cu;
    if (tm.transient) {  // Workset::computeSolnTransientSeeded (workset.cpp:589-623); the module has no time derivative
      const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
      ue = tm.alpha_u * cu + beta_u;
    }
    u[f] = ue;
  }
  double xn[DOF ? 1 : NN][DIM];
  double *sx = sm + (size_t)tid * NN * DIM;  // DOF: the thread's own vertices (read back by the same thread: no barrier)
  double x0[DIM];  // DIRECT: vertex 0; the other vertices are kept RELATIVE to it, so that elements of the same shape go
                   // through bit-identical geometry arithmetic wherever they sit (what the database mode relies on)
#pragma unroll
  for (int d = 0; d < DIM; ++d) x0[d] = 0.0;
  if constexpr (DOF) {
    const double *src = b.nodes + (size_t)e * NN * DIM;  // 16-byte aligned: NN * DIM is even
#pragma unroll
    for (int k = 0; k < NN * DIM; k += 2) *reinterpret_cast<double2 *>(sx + k) = *reinterpret_cast<const double2 *>(src + k);
    if constexpr (DIRECT) {
#pragma unroll
      for (int d = 0; d < DIM; ++d) x0[d] = sx[d];
#pragma unroll
      for (int k = 0; k < NN; ++k)
#pragma unroll
        for (int d = 0; d < DIM; ++d) sx[k * DIM + d] -= x0[d];
    }
  } else {
#pragma unroll
    for (int k = 0; k < NN; ++k)
#pragma unroll
      for (int d = 0; d < DIM; ++d) xn[k][d] = sm[((active ? tid : 0) * NN + k) * DIM + d];
    __syncthreads();  // vertices are in registers: the buffer is reused for the results
  }
  // Orientation signs: v_i and div_i carry s_i, so res_{u,i} ~ s_i, A_ij ~ s_i s_j, and u_i enters as s_i u_i.  The
  // point loop works on the unsigned basis with us_i = s_i u_i and the signs are applied once at the end.  The
  // divergence terms need no quadrature at all: div_i w = s_i (+-1/2) / detJ * (w_ref detJ) = s_i (+-1/2) w_ref, so
  //   Bv_i = -s_i (+-1/2) sum_q w_ref,  sum_q div u w = (sum_i us_i (+-1/2)) sum_q w_ref.
  double A[NU][NU], rp = 0.0, ru[NU], us[NU], wsum = 0.0;
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    ru[i] = 0.0;
    us[i] = u[1 + i] * sg[1 + i];
#pragma unroll
    for (int j = 0; j < NU; ++j) A[i][j] = 0.0;
  }
  // The matrix part (M, A: two thirds of the point's arithmetic, 42 registers) is compiled out of the loop for wavefronts
  // none of whose elements need it: residual-only assemblies, and in database mode every wavefront without an element
  // incident to a computed row (all but ~2 % of them at config 3).
  const bool need_matrix_lane = !RESONLY && out.compute_jacobian &&
      (DIRECT ? (active && out.direct_vals != nullptr && (!out.direct_jacflag || out.direct_jacflag[e])) : out.local_J != nullptr);
  const bool need_matrix = __builtin_amdgcn_ballot_w64(need_matrix_lane) != 0;
  // separable source on axis-aligned boxes with two points per direction: x_d at point q depends on bit d of q only
  // (tensor cubature, x fastest), so 2 * DIM sines serve all 2^DIM points
  const bool sep_src = DIRECT && !EXPR && out.direct_axis_aligned && pp.f[0].kind == MHA_FUNC_SINPROD && NQ == (1 << DIM);
  double sA[DIM], sB[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) { sA[d] = 0.0; sB[d] = 0.0; }
  if (sep_src) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double xa = x0[d], xb = x0[d];
#pragma unroll
      for (int k = 0; k < NN; ++k) {
        xa += sx[k * DIM + d] * b.nodeval[k * NQ];
        xb += sx[k * DIM + d] * b.nodeval[k * NQ + (1 << d)];
      }
      sA[d] = sin_moderate(pp.f[0].freq[d] * xa);
      sB[d] = sin_moderate(pp.f[0].freq[d] * xb);
    }
  }
  auto point_loop = [&](auto jm_tag) {
  constexpr bool JM = decltype(jm_tag)::value;
  for (int q = 0; q < NQ; ++q) {
    double J[DIM * DIM], Ji[DIM * DIM], det, x[DIM], xi[DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NN; ++k) s += (DOF ? sx[k * DIM + r] : xn[k][r]) * b.nodegrad[(k * NQ + q) * DIM + c];
        J[r * DIM + c] = s;
      }
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NN; ++k) s += (DOF ? sx[k * DIM + r] : xn[k][r]) * b.nodeval[k * NQ + q];
      x[r] = DIRECT ? x0[r] + s : s;  // (the geometry basis sums to one)
    }
    invert<DIM>(J, Ji, det);
    // reference point from the u table: phihat_{2c+1}(q) = (1 + x_c)/2
    const double *Tu = vl.tables + vl.table_off[1] + (size_t)q * (DIM + 1) * vl.cardpad[1];
#pragma unroll
    for (int c = 0; c < DIM; ++c) xi[c] = 2.0 * Tu[c * vl.cardpad[1] + 2 * c + 1] - 1.0;
    const double wr = b.ref_wts[q], w = wr * det, rdet = 1.0 / det;
    wsum += wr;
    double src;
    if (sep_src) {
      src = pp.f[0].amp;
#pragma unroll
      for (int d = 0; d < DIM; ++d) src *= ((q >> d) & 1) ? sB[d] : sA[d];
    } else {
      src = eval_func<DIM, EXPR>(pp.f[0], e, q, NQ, x);
    }
    const double mob = eval_func<DIM, EXPR>(pp.f[4], e, q, NQ, x);
    const double rmob = 1.0 / mob;  // one division per point instead of fifteen
    double kinv[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) kinv[d] = eval_func<DIM, EXPR>(pp.f[1 + d], e, q, NQ, x);
    rp += src * w;
    // unsigned basis: v_i = ph_i J[:,c_i], ph_i = (1 -/+ xi_c)/2 / det
    double ph[NU], uq[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) uq[d] = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int c = i >> 1;
      ph[i] = ((i & 1) ? 0.5 * (1.0 + xi[c]) : 0.5 * (1.0 - xi[c])) * rdet;
#pragma unroll
      for (int d = 0; d < DIM; ++d) uq[d] += us[i] * ph[i] * J[d * DIM + c];
    }
    // M[c][c'] = sum_d J[d][c] kinv_d J[d][c'] / mobility * w (symmetric), ku[c] = sum_d kinv_d uq_d J[d][c] / mobility * w
    const double wm = w * rmob;
    double M[DIM][DIM], ku[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double t = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) t += kinv[d] * uq[d] * J[d * DIM + c];
      ku[c] = t * wm;
      if constexpr (JM) {
#pragma unroll
        for (int c2 = c; c2 < DIM; ++c2) {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) s += J[d * DIM + c] * kinv[d] * J[d * DIM + c2];
          M[c][c2] = s * wm;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int c = i >> 1;
      ru[i] += ku[c] * ph[i];
      if constexpr (JM) {
#pragma unroll
        for (int j = i; j < NU; ++j) A[i][j] += ph[i] * ph[j] * M[c][j >> 1];  // symmetric: upper triangle only (c <= j >> 1)
      }
    }
  }
  };
  if constexpr (RESONLY) point_loop(std::false_type());
  else { if (need_matrix) point_loop(std::true_type()); else point_loop(std::false_type()); }
  // signs and the divergence terms
  double Bv[NU], divu = 0.0;
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const double hs = (i & 1) ? 0.5 : -0.5;
    divu += us[i] * hs;
    Bv[i] = -sg[1 + i] * hs * wsum;
    ru[i] = sg[1 + i] * (ru[i] - u[0] * hs * wsum);
#pragma unroll
    for (int j = i; j < NU; ++j) A[i][j] *= sg[1 + i] * sg[1 + j];
  }
  rp -= divu * wsum;
  const double au = tm.alpha_u;
  if constexpr (DIRECT) {
    // straight into the CRS: every entry this element owns alone (all but the diagonals of its face rows); residual
    // entries and diagonal parts go to the side array the finishing pass sums per row (ElemOut::direct_*)
    // (database mode: most wavefronts have no element whose entries are stored -- one scalar branch skips the 43 stores)
    const bool any_jac = !RESONLY && __builtin_amdgcn_ballot_w64(active && out.direct_vals && out.compute_jacobian &&
                                                                 (!out.direct_jacflag || out.direct_jacflag[e])) != 0;
    if (active) {
      const uint8_t *side = out.direct_side + (size_t)e * N;
      const bool jac = !RESONLY && out.direct_vals && out.compute_jacobian && (!out.direct_jacflag || out.direct_jacflag[e]);
      const uint8_t *sl = out.direct_slot + (size_t)e * N * N;
      double *vals = out.direct_vals;
      const bool ow = out.direct_overwrite != 0;
#pragma unroll
      for (int fi = 0; fi < N; ++fi) {
        const int row = L[pos[fi]];
        // the row's record: {residual part, diagonal part} of incident element 0, then of element 1
        double2 rec;
        rec.x = fi == 0 ? -rp : -ru[fi - 1];
        rec.y = (fi > 0 && jac) ? au * A[fi - 1][fi - 1] : 0.0;
        *reinterpret_cast<double2 *>(out.direct_part + (size_t)row * 4 + side[fi] * 2) = rec;
        if (!any_jac || !jac || (b.fixed && b.fixed[row])) continue;
        double *rowv = vals + b.rowptr[row];
        const uint8_t *srow = sl + pos[fi] * N;
#pragma unroll
        for (int fj = 0; fj < N; ++fj) {
          if (fi == fj && fi > 0) continue;  // the diagonal of a face row has two contributors: finishing pass
          double v;
          if (fi == 0) v = fj == 0 ? 0.0 : au * Bv[fj - 1];
          else if (fj == 0) v = au * Bv[fi - 1];
          else v = au * (fj > fi ? A[fi - 1][fj - 1] : A[fj - 1][fi - 1]);
          double *dst = rowv + srow[pos[fj]];
          if (ow) { if (out.direct_overwrite == 2) __builtin_nontemporal_store(v, dst); else *dst = v; }
          else *dst = *dst + v;
        }
      }
    }
    return;
  }
  if constexpr (DOF) {
    // dof order, stored (the row-gather scratch is never accumulated into): residual [p, u_0..], matrix rows [p | u_i]
    if (active) {
      if (out.local_res) {
        double *lr = out.local_res + (size_t)(e - out.local_base) * N;
        lr[0] = -rp;
#pragma unroll
        for (int i = 0; i < NU; ++i) lr[1 + i] = -ru[i];
      }
      if (out.local_J && out.compute_jacobian) {
        double *lj = out.local_J + (size_t)(e - out.local_base) * N * N;
        double row[N * N];
        row[0] = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          row[1 + i] = au * Bv[i];
          row[(1 + i) * N] = au * Bv[i];
#pragma unroll
          for (int j = 0; j < NU; ++j) row[(1 + i) * N + 1 + j] = au * (j >= i ? A[i][j] : A[j][i]);
        }
#pragma unroll
        for (int k = 0; k < N * N; ++k) lj[k] = row[k];
      }
    }
    return;
  }
  // dense output, LID-position order: into LDS per thread, then flat and coalesced to memory
  double *sJ = sm + (size_t)tid * N * N, *sR = sm + (size_t)kPorousThreads * N * N + (size_t)tid * N;
  if (active) {
    sR[pos[0]] = -rp;
#pragma unroll
    for (int i = 0; i < NU; ++i) sR[pos[1 + i]] = -ru[i];
    sJ[pos[0] * N + pos[0]] = 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      sJ[pos[0] * N + pos[1 + i]] = au * Bv[i];
      sJ[pos[1 + i] * N + pos[0]] = au * Bv[i];
#pragma unroll
      for (int j = 0; j < NU; ++j) sJ[pos[1 + i] * N + pos[1 + j]] = au * (j >= i ? A[i][j] : A[j][i]);
    }
  }
  __syncthreads();
  if (out.local_res) {
    double *lr = out.local_res + (size_t)(b.e_begin + e0 - out.local_base) * N;
    const double *srcr = sm + (size_t)kPorousThreads * N * N;
    for (int i = tid; i < cnt * N; i += kPorousThreads) lr[i] = out.local_store ? srcr[i] : lr[i] + srcr[i];
  }
  if (out.local_J && out.compute_jacobian) {
    double *lj = out.local_J + (size_t)(b.e_begin + e0 - out.local_base) * N * N;
    for (int i = tid; i < cnt * N * N; i += kPorousThreads) lj[i] = out.local_store ? sm[i] : lj[i] + sm[i];
  }
}

// The plain-coefficient instantiations are held to two wavefronts per SIMD (243 registers, no spills since the signs and
// the divergence terms left the point loop; three would need 168 and spill 240 B per lane -- tried with the constant
// coefficients compiled in, and with the point's stages fenced: no better); the deck-string ones call the interpreter
// and keep the default budget.
template <int DIM, bool DOF>
__global__ __launch_bounds__(kPorousThreads) __attribute__((amdgpu_waves_per_eu(2))) void porous_element_kernel(
    BlockDev b, VarLayoutDev vl, PhysParamsDev pp, TimeDev tm, ElemOut out) {
  porous_element_body<DIM, false, DOF>(b, vl, pp, tm, out);
}

template <int DIM, bool DOF>
__global__ __launch_bounds__(kPorousThreads) void porous_element_expr_kernel(BlockDev b, VarLayoutDev vl, PhysParamsDev pp,
                                                                            TimeDev tm, ElemOut out) {
  porous_element_body<DIM, true, DOF>(b, vl, pp, tm, out);
}
template <int DIM>
__global__ __launch_bounds__(kPorousThreads) __attribute__((amdgpu_waves_per_eu(2))) void porous_element_direct_kernel(
    BlockDev b, VarLayoutDev vl, PhysParamsDev pp, TimeDev tm, ElemOut out) {
  porous_element_body<DIM, false, true, true>(b, vl, pp, tm, out);
}
// the lean build of the direct form: residual parts only (database mode runs it over all elements, the full build over
// the few elements incident to computed rows)
template <int DIM>
__global__ __launch_bounds__(kPorousThreads) __attribute__((amdgpu_waves_per_eu(MHA_POROUS_RES_WAVES))) void porous_element_direct_res_kernel(
    BlockDev b, VarLayoutDev vl, PhysParamsDev pp, TimeDev tm, ElemOut out) {
  porous_element_body<DIM, false, true, true, true>(b, vl, pp, tm, out);
}
template <int DIM>
__global__ __launch_bounds__(kPorousThreads) void porous_element_direct_expr_kernel(BlockDev b, VarLayoutDev vl, PhysParamsDev pp,
                                                                                   TimeDev tm, ElemOut out) {
  porous_element_body<DIM, true, true, true>(b, vl, pp, tm, out);
}

// ---- database mode on a uniform block: the residual from the element matrix ---------------------------------------------
// porousMixed is linear: with constant permeability / mobility the volume residual of an element is
//   r = (1 / alpha_u) A^ u_e + (sum_q source(x_q) w_q) e_p,
// A^ the element matrix (the same for every element of a uniform block), u_e the seeded local values.
// Setup, whenever a coefficient or alpha_u has changed: the dense kernel above on element 0 leaves A^ in
// `uniform[0 .. n*n)` (dof order), porous_uniform_points_kernel
// the offsets x_q - x_vertex0 and the weights w_q = w_ref detJ of the common shape behind it.  Then one thread per element:
// 7 gathered values, 49 FMAs with wave-uniform operands, the source at 2^dim points (2 dim sines on an axis-aligned
// box), and the element's parts into its rows' records -- what the lean build of the direct kernel produces with the
// whole geometry and flux arithmetic, to round-off.
template <int DIM>
__global__ __launch_bounds__(64) void porous_uniform_points_kernel(BlockDev b, double *uniform) {
  constexpr int NN = 1 << DIM, N = 1 + 2 * DIM;
  const int q = threadIdx.x, NQ = b.nq;
  if (q >= NQ) return;
  const double *xn = b.nodes;  // element 0
  double J[DIM * DIM], Ji[DIM * DIM], det, c[DIM];
#pragma unroll
  for (int r = 0; r < DIM; ++r) {
    c[r] = 0.0;
#pragma unroll
    for (int cc = 0; cc < DIM; ++cc) J[r * DIM + cc] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < NN; ++k)
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      const double rel = xn[k * DIM + r] - xn[r];
      c[r] += rel * b.nodeval[k * NQ + q];
#pragma unroll
      for (int cc = 0; cc < DIM; ++cc) J[r * DIM + cc] += rel * b.nodegrad[(k * NQ + q) * DIM + cc];
    }
  invert<DIM>(J, Ji, det);
  double *pts = uniform + N * N;
#pragma unroll
  for (int r = 0; r < DIM; ++r) pts[q * DIM + r] = c[r];
  pts[NQ * DIM + q] = b.ref_wts[q] * det;
}

template <int DIM>
__global__ __launch_bounds__(256) void porous_uniform_residual_kernel(BlockDev b, VarLayoutDev vl, PhysParamsDev pp, TimeDev tm,
                                                                      ElemOut out) {
  constexpr int NU = 2 * DIM, N = 1 + NU, NN = 1 << DIM;
  const int e = blockIdx.x * 256 + threadIdx.x, NQ = vl.nq;
  if (e >= b.e_count) return;
  if (out.direct_jacflag && out.direct_jacflag[e]) return;  // the full build writes this element's records (it may run beside this kernel)
  const int32_t *L = b.lids + (size_t)e * N;
  const double *uni = out.direct_uniform, *pts = uni + N * N, *wq = pts + NQ * DIM;  // (wave-uniform reads)
  int row[N];
  double u[N];
#pragma unroll
  for (int f = 0; f < N; ++f) {
    row[f] = L[b.offsets[f]];
    const double cu = tm.u[row[f]];
    double ue = cu;
    if (tm.transient) {  // Workset::computeSolnTransientSeeded (workset.cpp:589-623), as in the kernel above
      const double *cp = tm.u_prev + (size_t)row[f] * tm.nsteps, *cs = tm.u_stage + (size_t)row[f] * tm.nstages;
      double beta_u = (1.0 - tm.alpha_u) * cp[0];
      for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);
      ue = tm.alpha_u * cu + beta_u;
    }
    u[f] = ue;
  }
  // source integral: x_q = first vertex + the common offset of point q
  const double *x0 = b.nodes + (size_t)e * NN * DIM;
  double srcint = 0.0;
  if (out.direct_axis_aligned && pp.f[0].kind == MHA_FUNC_SINPROD && NQ == (1 << DIM)) {
    double sA[DIM], sB[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      sA[d] = sin_moderate(pp.f[0].freq[d] * (x0[d] + pts[d]));
      sB[d] = sin_moderate(pp.f[0].freq[d] * (x0[d] + pts[(1 << d) * DIM + d]));
    }
    for (int q = 0; q < NQ; ++q) {
      double s = pp.f[0].amp;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s *= ((q >> d) & 1) ? sB[d] : sA[d];
      srcint += s * wq[q];
    }
  } else {
    for (int q = 0; q < NQ; ++q) {
      double x[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) x[d] = x0[d] + pts[q * DIM + d];
      srcint += eval_func<DIM, false>(pp.f[0], e, q, NQ, x) * wq[q];
    }
  }
  const double inv_au = 1.0 / tm.alpha_u;
  const uint8_t *side = out.direct_side + (size_t)e * N;
#pragma unroll
  for (int f = 0; f < N; ++f) {
    double r = 0.0;
#pragma unroll
    for (int g = 0; g < N; ++g) r += uni[f * N + g] * u[g];
    r = r * inv_au + (f == 0 ? srcint : 0.0);
    *reinterpret_cast<double2 *>(out.direct_part + (size_t)row[f] * 4 + side[f] * 2) = make_double2(-r, 0.0);
  }
}

// finishing pass of the direct form, one thread per row: coalesced reads of the row records, one residual entry and
// (face rows) one diagonal entry out
__global__ __launch_bounds__(256) void porous_direct_finish_kernel(BlockDev b, const int32_t *__restrict__ inc_ptr,
                                                                   const int32_t *__restrict__ diagpos,
                                                                   const double *__restrict__ part, double *res, double *vals,
                                                                   int overwrite) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= b.nrows) return;
  const int ni = inc_ptr[row + 1] - inc_ptr[row];
  const double4 rec = reinterpret_cast<const double4 *>(part)[row];
  const int dp = diagpos[row];  // < 0: a cell row (the element stored its zero diagonal itself) or no diagonal in the graph
  if (b.fixed && b.fixed[row]) {  // isFixedDOF rows are skipped by the scatter; overwriting leaves zeros
    if (overwrite) {
      if (vals) for (int k = b.rowptr[row]; k < b.rowptr[row + 1]; ++k) vals[k] = 0.0;
      if (res) res[row] = 0.0;
    }
    return;
  }
  const double r = (ni > 0 ? rec.x : 0.0) + (ni > 1 ? rec.z : 0.0);
  const double dg = (ni > 0 ? rec.y : 0.0) + (ni > 1 ? rec.w : 0.0);
  if (res) res[row] = overwrite ? r : res[row] + r;
  if (vals && dp >= 0) vals[dp] = overwrite ? dg : vals[dp] + dg;
}

}  // namespace

void launch_porous_element(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                           const ElemOut &out, hipStream_t stream) {
  if (b.e_count <= 0) return;
  MHA_REQUIRE(out.res == nullptr && out.crs_vals == nullptr, MHA_ERR_INVALID,
              "porous element kernel writes dense element arrays only");
  const int grid = (b.e_count + kPorousThreads - 1) / kPorousThreads;
  const int n = 1 + 2 * b.dim;
  if (out.direct_part && out.direct_uniform && out.direct_res_only && !has_expression(pp)) {
    launch_porous_uniform_residual(b, vl, pp, tm, out, out.direct_uniform, stream);
    return;
  }
  if (out.direct_part) {  // direct form: no dense arrays
    MHA_REQUIRE(out.direct_slot != nullptr && out.direct_side != nullptr && out.local_base == 0, MHA_ERR_INVALID, "porous direct form: slot map missing");
    const size_t ldsd = sizeof(double) * kPorousThreads * (size_t)(1 << b.dim) * b.dim;
    auto god = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(kPorousThreads), ldsd, stream, b, vl, pp, tm, out); };
    if (out.direct_res_only && !has_expression(pp)) { if (b.dim == 2) god(porous_element_direct_res_kernel<2>); else god(porous_element_direct_res_kernel<3>); }
    else if (has_expression(pp)) { if (b.dim == 2) god(porous_element_direct_expr_kernel<2>); else god(porous_element_direct_expr_kernel<3>); }
    else { if (b.dim == 2) god(porous_element_direct_kernel<2>); else god(porous_element_direct_kernel<3>); }
    MHA_HIP(hipGetLastError());
    return;
  }
  const bool dof = out.local_dof_order != 0;
  MHA_REQUIRE(!dof || out.local_store, MHA_ERR_INVALID, "dof-ordered element arrays are stored, never accumulated into");
  const size_t lds = dof ? sizeof(double) * kPorousThreads * (size_t)(1 << b.dim) * b.dim : sizeof(double) * kPorousThreads * std::max<size_t>((size_t)(1 << b.dim) * b.dim, (size_t)n * n + n);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(kPorousThreads), lds, stream, b, vl, pp, tm, out); };
  const bool expr = has_expression(pp);
  auto pick = [&](auto dim_c, auto dof_c) {
    constexpr int D = decltype(dim_c)::value;
    constexpr bool F = decltype(dof_c)::value;
    if (expr) go(porous_element_expr_kernel<D, F>); else go(porous_element_kernel<D, F>);
  };
  if (b.dim == 2) { if (dof) pick(std::integral_constant<int, 2>(), std::true_type()); else pick(std::integral_constant<int, 2>(), std::false_type()); }
  else { if (dof) pick(std::integral_constant<int, 3>(), std::true_type()); else pick(std::integral_constant<int, 3>(), std::false_type()); }
  MHA_HIP(hipGetLastError());
}

void launch_porous_uniform_residual(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                                    const ElemOut &out, double *uniform, hipStream_t stream) {
  if (b.e_count <= 0) return;
  MHA_REQUIRE(out.direct_part && out.direct_side && uniform && !has_expression(pp) && b.nq <= 64, MHA_ERR_INVALID,
              "porous uniform residual: missing tables");
  ElemOut o = out;
  o.direct_uniform = uniform;
  if (b.dim == 2) hipLaunchKernelGGL(porous_uniform_residual_kernel<2>, dim3((b.e_count + 255) / 256), dim3(256), 0, stream, b, vl, pp, tm, o);
  else hipLaunchKernelGGL(porous_uniform_residual_kernel<3>, dim3((b.e_count + 255) / 256), dim3(256), 0, stream, b, vl, pp, tm, o);
  MHA_HIP(hipGetLastError());
}

void launch_porous_uniform_points(const BlockDev &b, double *uniform, hipStream_t stream) {
  MHA_REQUIRE(b.nq <= 64, MHA_ERR_INVALID, "porous uniform tables: more than 64 integration points");
  if (b.dim == 2) hipLaunchKernelGGL(porous_uniform_points_kernel<2>, dim3(1), dim3(64), 0, stream, b, uniform);
  else hipLaunchKernelGGL(porous_uniform_points_kernel<3>, dim3(1), dim3(64), 0, stream, b, uniform);
  MHA_HIP(hipGetLastError());
}

void launch_porous_direct_finish(const BlockDev &b, const int32_t *inc_ptr, const int32_t *diagpos, const double *part,
                                 double *res, double *vals, int overwrite, hipStream_t stream) {
  if (b.nrows <= 0) return;
  hipLaunchKernelGGL(porous_direct_finish_kernel, dim3((b.nrows + 255) / 256), dim3(256), 0, stream, b, inc_ptr, diagpos, part,
                     res, vals, overwrite);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
