// point_engine.hip -- multi-variable element kernel: point-level forward AD + B^T C B contraction.
//
// Replaces, for any physics module written as a point function (physics_points.hpp), the reference's volume loop
//   performGather (assemblyManager.cpp:3598-3643) -> computeSoln*Seeded (workset.cpp:559-859) ->
//   evaluateSolutionField per field (workset.cpp:937-1062) -> <module>::volumeResidual -> scatter (:4031-4145)
// for blocks with several variables and HGRAD / HVOL / HDIV bases (getPhysicalVolumetricBasis,
// discretizationInterface.cpp:898-1127).
//
// With U_m(q) = sum_j u_j T_m(j,q) the fields at a point (T = basis "slots": value, gradient components, divergence)
// and res_i = sum_q w sum_k F_k(U(q)) T_k(i,q), the Sacado derivative array of the reference is
//   d res_i/d u_j = sum_q T(i,q)^T [ w dF/dU (q) ] T(j,q).
// Physical slots are reference slots times a per-point geometric block G (HGRAD: diag(1, J^-T); HDIV: J/detJ and
// 1/detJ), so the contraction runs on element-independent reference tables T^ kept in LDS:
//   res_i = sum_q T^(i,q) . F^(q),  J_ij = sum_q T^(i,q)^T C^(q) T^(j,q),  F^ = w G^T F,  C^ = w G^T (dF/dU) G.
// 64 threads (small elements, four per workgroup) or 256 threads (large ones) per element, persistent over elements:
//   1 gather + seeding values (x orientation sign), geometry per point
//   2 reference-slot fields U^(q) = sum_j u_j T^(j,q)
//   3 one thread per (point, direction): the module's point function on Dual numbers -> one column of C^(q)
//   4 residual rows
//   5 Jacobian = the dense B^T C^ B product on the matrix cores (v_mfma_f64_16x16x4_f64): panels of 16 rows of one
//     variable, P = T^(i,.) C^ with one MFMA per point, then 16x16 output tiles with K = points x slots of the column
//     variable.  fp64 MFMA runs at the vector rate on gfx950; the gain is 1024 FMAs per two 8-byte LDS operands.
//     The 89-dof navierstokes element runs these products streamed over the points instead ("5q" below): owner waves
//     with the accumulators of a column tile for all row panels, P blocks two points ahead, one barrier per point.
// Output: dense local_J / local_res (updateJac / updateRes convention; the row-gather kernel turns them into CRS rows
// without global atomics) or atomics into res / CRS through the element-major slot map.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_math.hpp"
#include "launch.hpp"
#include "physics_points.hpp"

namespace mha {
namespace {

// static variable layout of a module (must agree with the host's VarLayoutDev; checked in the launcher)
template <int PHYS, int DIM>
struct Layout;
template <int DIM>
struct Layout<MHA_PHYSICS_THERMAL, DIM> {
  static constexpr int nvars = 1, NS = 1 + DIM;
  __host__ __device__ static constexpr int type(int) { return MHA_BASIS_HGRAD; }
};
template <int DIM>
struct Layout<MHA_PHYSICS_POROUS_MIXED, DIM> {
  static constexpr int nvars = 2, NS = 2 + DIM;
  __host__ __device__ static constexpr int type(int v) { return v == 0 ? MHA_BASIS_HVOL : MHA_BASIS_HDIV; }
};
template <int DIM>
struct Layout<MHA_PHYSICS_NAVIERSTOKES, DIM> {
  static constexpr int nvars = 1 + DIM, NS = (1 + DIM) * (1 + DIM);
  __host__ __device__ static constexpr int type(int) { return MHA_BASIS_HGRAD; }
};

template <int DIM>
struct Layout<MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED, DIM> {
  static constexpr int nvars = 3, NS = 3 * (1 + DIM);
  __host__ __device__ static constexpr int type(int) { return MHA_BASIS_HGRAD; }
};

__host__ __device__ constexpr int slots_of(int type, int dim) { return type == MHA_BASIS_HVOL ? 1 : 1 + dim; }
// slots that carry a time derivative: every value, not gradients / divergence
__host__ __device__ constexpr bool value_like(int type, int s, int dim) {
  return type == MHA_BASIS_HDIV ? s < dim : s == 0;
}

template <class L, int DIM>
__host__ __device__ constexpr int slotptr_of(int v) {
  int p = 0;
  for (int k = 0; k < v; ++k) p += slots_of(L::type(k), DIM);
  return p;
}

// physical slot values from reference slot values (one variable): G ref
template <int DIM>
__device__ __forceinline__ void to_phys(int type, const double *ref, const double *J, const double *Ji, double det,
                                        double *phys) {
  if (type == MHA_BASIS_HVOL) {
    phys[0] = ref[0];
  } else if (type == MHA_BASIS_HGRAD) {
    phys[0] = ref[0];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += Ji[c * DIM + d] * ref[1 + c];
      phys[1 + d] = s;
    }
  } else {
    const double r = 1.0 / det;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += J[d * DIM + c] * ref[c];
      phys[d] = s * r;
    }
    phys[DIM] = ref[DIM] * r;
  }
}

// reference-slot coefficients from physical-slot coefficients: G^T phys
template <int DIM>
__device__ __forceinline__ void to_ref_T(int type, const double *phys, const double *J, const double *Ji, double det,
                                         double *ref) {
  if (type == MHA_BASIS_HVOL) {
    ref[0] = phys[0];
  } else if (type == MHA_BASIS_HGRAD) {
    ref[0] = phys[0];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += Ji[c * DIM + d] * phys[1 + d];
      ref[1 + c] = s;
    }
  } else {
    const double r = 1.0 / det;
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += J[d * DIM + c] * phys[d];
      ref[c] = s * r;
    }
    ref[DIM] = phys[DIM] * r;
  }
}

constexpr int kEngineThreads = 512, kPanelRows = 16;

// orders a wave's LDS writes before its later LDS reads (data private to the wave: no workgroup barrier needed)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int DIM>
constexpr int geo_size() { return 2 * DIM * DIM + 2 + DIM; }  // J, Ji, det, w, x

// per-element LDS (doubles): u, udot, sign | geometry | U^, Udot^, F^ | C^ | P panel | row, pos (ints)
__host__ __device__ inline size_t engine_group_doubles(const VarLayoutDev &vl, int geo) {
  const size_t n = vl.n_tot, NS = vl.ns_tot, NQ = vl.nq;
  return (3 * n + NQ * geo + 3 * NQ * NS + NQ * NS * NS + NQ * NS * kPanelRows + n + 1) & ~size_t(1);  // even: 16-B aligned groups
}

// TPE threads work on one element; a workgroup (512 threads) holds 512/TPE elements at a time (TPE = 64: one wave per
// element and only wave-level synchronisation inside the element loop; TPE = 512: the whole workgroup, block barriers).
// NQ1 = integration points per direction when it is 2 or 3 (the loops over points then have compile-time bounds and
// the compiler batches their LDS loads), 0 = taken from the layout at run time.
template <int DIM, int PHYS, int TPE, int NQ1, int EXPR>  // EXPR: 0 / 1 deck strings / 2 deck strings that read the solution fields (thermal)
__global__ __launch_bounds__(kEngineThreads) void point_engine_kernel(BlockDev b, VarLayoutDev vl, PhysParamsDev pp,
                                                                      TimeDev tm, ElemOut out_all,
                                                                      const uint8_t *slot8_all, const uint16_t *slot16_all) {
  using L = Layout<PHYS, DIM>;
  constexpr int NS = L::NS, NN = 1 << DIM, GEO = geo_size<DIM>(), NG = kEngineThreads / TPE;
  extern __shared__ double smem[];
  const int NQ = NQ1 ? (DIM == 2 ? NQ1 * NQ1 : NQ1 * NQ1 * NQ1) : vl.nq;
  const int n = vl.n_tot, tid = threadIdx.x, group = tid / TPE, gt = tid % TPE;
  // Barriers order LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for the element-matrix stores of the
  // panel just finished to be acknowledged by memory -- fourteen times per 89-dof element (gfx950 retires loads and
  // stores through one in-order counter); nothing in this kernel reads global memory another thread has written.
  auto sync = [&]() {
    if constexpr (TPE == 64) {
      wave_lds_sync();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
  };
  double *tab = smem;
  double *gbase = tab + vl.tables_size + group * (int)engine_group_doubles(vl, GEO);
  double *s_P = gbase;  // first: read as 16-byte vectors
  double *s_Ch = s_P + NQ * NS * kPanelRows;
  double *s_Uh = s_Ch + NQ * NS * NS, *s_Udh = s_Uh + NQ * NS, *s_Fh = s_Udh + NQ * NS;
  double *s_geo = s_Fh + NQ * NS;
  double *s_u = s_geo + NQ * GEO, *s_ud = s_u + n, *s_sgn = s_ud + n;
  int *s_row = reinterpret_cast<int *>(s_sgn + n), *s_pos = s_row + n;

  const bool mass_mode = pp.physics < 0;  // launch_point_engine_mass: p[v] = mass weight of variable v
  const int dbg_stop = mass_mode ? 0 : (int)pp.p[7];  // profiling aid (MHA_ENGINE_STOP): leave the element after phase k
  for (int k = tid; k < vl.tables_size; k += kEngineThreads) tab[k] = vl.tables[k];
  if (dbg_stop & 64) {
    // diagnostic (MHA_ENGINE_STOP=64): every per-element LDS array starts as NaN, so a value that is read before the
    // element loop has written it shows in the results (the hunt for the deck-string garbage of round 2: none found)
    const int total = NG * (int)engine_group_doubles(vl, GEO);
    for (int k = tid; k < total; k += kEngineThreads) tab[vl.tables_size + k] = __builtin_nan("");
  }
  __syncthreads();

  // blocked element ranges per group: groups running at the same time work far apart in the mesh
  const int ngroups = gridDim.x * NG, gid = blockIdx.x * NG + group;
  const int chunk = (b.e_count + ngroups - 1) / ngroups;
  const int el_begin = gid * chunk, el_end = min(b.e_count, el_begin + chunk);
  // TPE > 64 synchronises with block barriers, which every group of the workgroup must reach the same number of
  // times: all groups run `chunk` iterations; a group past its range recomputes its last element with outputs off
  // the q-streamed form of phase 5 (below): the 89-dof navierstokes shape -- every variable four slots, seven 16-dof
  // pieces, one per wave and a spare wave -- writing the row-gather scratch
  constexpr bool REGB = !EXPR && PHYS == MHA_PHYSICS_NAVIERSTOKES && DIM == 3 && TPE == 512;
  bool q_stream = false;
  if constexpr (REGB) {
    int ntile = 0;
    bool shape_ok = true;
    for (int vj = 0; vj < vl.nvars; ++vj) {
      if (vl.nslot[vj] != 4) shape_ok = false;
      ntile += (vl.card[vj] + 15) / 16;
      if (vl.card[vj] > 32) shape_ok = false;
    }
    q_stream = shape_ok && ntile == TPE / 64 - 1 && n <= 128 && NQ == 27 && n * n <= NQ * NS * (NS + kPanelRows) &&
               3 * ntile * 256 <= NQ * NS * kPanelRows && out_all.compute_jacobian && out_all.local_J &&
               out_all.local_store && !out_all.crs_vals && !mass_mode && !(dbg_stop & (8 | 32));
  }
  // layout of the variable that holds slot m / dof f.  The loops run over the module's compile-time variable count, so
  // the layout arrays are indexed statically (kernel arguments in scalar registers); indexed by a run-time variable
  // number they are memory loads, one dependent on the other, in every phase of every element
  struct VarAt { int sp, nsl, card, cp, vp, to; };
  auto var_at = [&](int key, bool by_slot) {
    VarAt r = {vl.slotptr[0], vl.nslot[0], vl.card[0], vl.cardpad[0], vl.varptr[0], vl.table_off[0]};
#pragma unroll
    for (int k = 1; k < L::nvars; ++k)
      if (key >= (by_slot ? vl.slotptr[k] : vl.varptr[k]))
        r = {vl.slotptr[k], vl.nslot[k], vl.card[k], vl.cardpad[k], vl.varptr[k], vl.table_off[k]};
    return r;
  };
  const int iters = (TPE == 64) ? max(el_end - el_begin, 0) : (((int)blockIdx.x * NG * chunk < b.e_count) ? chunk : 0);
  // q-streamed form: the thread's dof position is the same in every element, and the next element's row id and solution
  // value are requested while this element's products run (three dependent global loads open every element otherwise)
  int pos_k = 0, row_pf = 0;
  double cu_pf = 0.0;
  bool have_pf = false;
  if constexpr (REGB) {
    if (q_stream && gt < n) pos_k = b.offsets[gt];
  }
  for (int it = 0; it < iters; ++it) {
    const bool live = el_begin + it < el_end;
    const int el = live ? el_begin + it : max(min(el_end, b.e_count) - 1, 0);
    const int e = b.e_begin + el;
    ElemOut out = live ? out_all : ElemOut();
    out.compute_jacobian = out_all.compute_jacobian;
    const uint8_t *slot8 = slot8_all;
    const uint16_t *slot16 = slot16_all;
    sync();  // previous element done with the group's LDS
    // ---- 1. gather + seeding values, geometry ----
    for (int f = gt; f < n; f += TPE) {
      int pos, row;
      double cu;
      if (REGB && q_stream && have_pf) { pos = pos_k; row = row_pf; cu = cu_pf; }
      else { pos = b.offsets[f]; row = b.lids[(size_t)e * n + pos]; cu = tm.u[row]; }
      const double sg = vl.orient ? (double)vl.orient[(size_t)e * n + f] : 1.0;
      double ue = cu, ud = 0.0;
      if (tm.transient) {  // Workset::computeSolnTransientSeeded (workset.cpp:589-623)
        const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;
        double beta_u = (1.0 - tm.alpha_u) * cp[0];
        for (int st = 0; st < tm.stage; ++st) beta_u += tm.stage_ratio[st] * (cs[st] - cp[0]);
        double beta_t = 0.0;
        for (int st = 1; st < tm.nsteps + 1; ++st) beta_t += tm.bdf[st] * cp[st - 1];
        beta_t *= tm.timewt;
        ue = tm.alpha_u * cu + beta_u;
        ud = tm.alpha_t * cu + beta_t;
      }
      s_u[f] = ue * sg;
      s_ud[f] = ud * sg;
      s_sgn[f] = sg;
      s_row[f] = row;
      s_pos[f] = pos;
    }
    for (int q = gt; q < NQ; q += TPE) {
      const double *xn = b.nodes + (size_t)e * NN * DIM;
      double J[DIM * DIM], Ji[DIM * DIM], det, x[DIM];
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double sum = 0.0;
          for (int k = 0; k < NN; ++k) sum += xn[k * DIM + r] * b.nodegrad[(k * NQ + q) * DIM + c];
          J[r * DIM + c] = sum;
        }
        double sum = 0.0;
        for (int k = 0; k < NN; ++k) sum += xn[k * DIM + r] * b.nodeval[k * NQ + q];
        x[r] = sum;
      }
      invert<DIM>(J, Ji, det);
      double *g = s_geo + q * GEO;
#pragma unroll
      for (int k = 0; k < DIM * DIM; ++k) { g[k] = J[k]; g[DIM * DIM + k] = Ji[k]; }
      g[2 * DIM * DIM] = det;
      g[2 * DIM * DIM + 1] = b.ref_wts[q] * det;
#pragma unroll
      for (int d = 0; d < DIM; ++d) g[2 * DIM * DIM + 2 + d] = x[d];
    }
    sync();
    if (dbg_stop == 1) continue;
    // ---- 2. reference-slot fields: U^(q,m) = sum_dof u_dof T^[q][slot][dof] ----
    for (int idx = gt; idx < NQ * NS; idx += TPE) {
      const int q = idx / NS, m = idx - q * NS;
      const VarAt va = var_at(m, true);
      const int sl = m - va.sp, card = va.card;
      const double *T = tab + va.to + (q * va.nsl + sl) * va.cp;
      const double *uu = s_u + va.vp, *ud = s_ud + va.vp;
      double a = 0.0, ad = 0.0;
      if (tm.transient) {
#pragma unroll 4
        for (int dof = 0; dof < card; ++dof) {
          a += uu[dof] * T[dof];
          ad += ud[dof] * T[dof];
        }
      } else {  // steady: the time-derivative coefficients are zero
#pragma unroll 4
        for (int dof = 0; dof < card; ++dof) a += uu[dof] * T[dof];
      }
      s_Uh[idx] = a;
      s_Udh[idx] = ad;
    }
    sync();
    if (dbg_stop == 2) continue;
    // ---- 3. point function, one (point, direction) per thread ----
    double vol = 0.0;
    for (int q = 0; q < NQ; ++q) vol += s_geo[q * GEO + 2 * DIM * DIM + 1];
    const double h = (DIM == 2) ? sqrt(vol) : cbrt(vol);  // Workset::getElementSize (workset.cpp:2666-2679)
    for (int idx = gt; idx < NQ * (NS + 1); idx += TPE) {
      const int q = idx / (NS + 1), m = idx - q * (NS + 1);
      const double *g = s_geo + q * GEO;
      const double *J = g, *Ji = g + DIM * DIM;
      const double det = g[2 * DIM * DIM], w = g[2 * DIM * DIM + 1];
      Dual U[NS], Ud[NS], F[NS];
#pragma unroll
      for (int v = 0; v < L::nvars; ++v) {
        const int type = L::type(v), sp = slotptr_of<L, DIM>(v), ns = slots_of(type, DIM);
        double ref[1 + DIM], phys[1 + DIM], refd[1 + DIM], physd[1 + DIM], dir[1 + DIM], pdir[1 + DIM];
#pragma unroll
        for (int sl = 0; sl < ns; ++sl) {
          ref[sl] = s_Uh[q * NS + sp + sl];
          refd[sl] = s_Udh[q * NS + sp + sl];
          dir[sl] = (m == sp + sl) ? 1.0 : 0.0;
        }
        to_phys<DIM>(type, ref, J, Ji, det, phys);
        to_phys<DIM>(type, refd, J, Ji, det, physd);
        to_phys<DIM>(type, dir, J, Ji, det, pdir);
#pragma unroll
        for (int sl = 0; sl < ns; ++sl) {
          U[sp + sl] = mk(phys[sl], tm.alpha_u * pdir[sl]);
          Ud[sp + sl] = value_like(type, sl, DIM) ? mk(physd[sl], tm.alpha_t * pdir[sl]) : mk(0.0);
        }
      }
      PointArgs<DIM> pa;
      pa.U = U; pa.Ud = Ud; pa.x = g + 2 * DIM * DIM + 2; pa.h = h; pa.dt = tm.dt;
      pa.transient = tm.transient; pa.e = e; pa.q = q; pa.nq = NQ; pa.pp = &pp;
      if (mass_mode) {
        // getWeightedMass: F = masswts[var] * value on the value-like slots, nothing on gradients / divergence
#pragma unroll
        for (int v = 0; v < L::nvars; ++v) {
          const int type = L::type(v), sp = slotptr_of<L, DIM>(v), ns = slots_of(type, DIM);
#pragma unroll
          for (int sl = 0; sl < ns; ++sl) F[sp + sl] = value_like(type, sl, DIM) ? U[sp + sl] * pp.p[v] : mk(0.0);
        }
      } else if constexpr (PHYS == MHA_PHYSICS_THERMAL) thermal_point<DIM, EXPR>(pa, F);
      else if constexpr (PHYS == MHA_PHYSICS_POROUS_MIXED) porous_point<DIM, (EXPR != 0)>(pa, F);
      else if constexpr (PHYS == MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED) swhdg_point<DIM, (EXPR != 0)>(pa, F);
      else navierstokes_point<DIM, (EXPR != 0)>(pa, F);
#pragma unroll
      for (int v = 0; v < L::nvars; ++v) {
        const int type = L::type(v), sp = slotptr_of<L, DIM>(v), ns = slots_of(type, DIM);
        double phys[1 + DIM], ref[1 + DIM];
#pragma unroll
        for (int sl = 0; sl < ns; ++sl) phys[sl] = (m == NS) ? F[sp + sl].v : F[sp + sl].d;
        to_ref_T<DIM>(type, phys, J, Ji, det, ref);
#pragma unroll
        for (int sl = 0; sl < ns; ++sl) {
          if (m == NS) s_Fh[q * NS + sp + sl] = w * ref[sl];
          else s_Ch[(q * NS + sp + sl) * NS + m] = w * ref[sl];
        }
      }
    }
    sync();
    if (dbg_stop == 3) continue;
    // ---- 4. residual rows ----  (q-streamed form: by the spare wave inside phase 5)
    for (int f = gt; f < (q_stream ? 0 : n); f += TPE) {
      int v = 0;
      while (f >= vl.varptr[v + 1]) ++v;
      const int ns = vl.nslot[v], sp = vl.slotptr[v], cp = vl.cardpad[v];
      const double *T = tab + vl.table_off[v] + (f - vl.varptr[v]);
      double r = 0.0;
      if (ns == 1) {
#pragma unroll 4
        for (int q = 0; q < NQ; ++q) r += T[q * cp] * s_Fh[q * NS + sp];
      } else {
#pragma unroll 3
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
          for (int sl = 0; sl < 1 + DIM; ++sl) r += T[(q * (1 + DIM) + sl) * cp] * s_Fh[q * NS + sp + sl];
        }
      }
      r *= s_sgn[f];
      const int row = s_row[f];
      if (out.local_res) {
        double *lr = out.local_res + (size_t)(e - out.local_base) * n + s_pos[f];
        *lr = out.local_store ? -r : *lr - r;
      }
      if (out.res && !(b.fixed && b.fixed[row])) unsafeAtomicAdd(out.res + row, -r);
    }
    if (dbg_stop == 4) continue;
    // ---- 5. Jacobian = B^T C^ B on the matrix cores (v_mfma_f64_16x16x4_f64), in panels of 16 rows of ONE variable:
    //   P[(q,m)][r] = sum_s T^[q][s][i_r] C^(q)[sp_i+s][m]      one MFMA per point and 16 slots m   (K = the row variable's slots)
    //   J[i_r][j]   = sum_(q,s) P[(q,sp_j+s)][r] T^[q][s][j]     16x16 tiles over 16 dofs of one column variable, K = NQ*nslot_j
    // operand maps (CDNA4): A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15], D reg t: row = (lane>>4) + 4t, col = lane&15
    if (out.compute_jacobian) {
      typedef double v4d __attribute__((ext_vector_type(4)));
      double *lj_e = out.local_J ? out.local_J + (size_t)(e - out.local_base) * n * n : nullptr;
      const bool nt_store = (dbg_stop & 16) != 0;
      constexpr int NWV = TPE / 64;             // waves working on this element
      const int wv = gt >> 6, lane = gt & 63, l15 = lane & 15, l4 = lane >> 4;
      // When every wave has at most one column tile (the 89-dof navierstokes element: seven tiles on eight waves), the
      // tile's B operand -- table values, the same for all row panels -- is read once per element and kept in registers:
      // the products of a panel then read only P from the LDS, half the traffic of the 2 x 512 B per product that kept
      // the LDS pipe as busy as the matrix pipe (profiles/r2_ns_regb.log)
      // (only in the instantiation that has that shape; the offset passes through an empty asm so that the reads stay
      // inside the element loop -- hoisted, they would be live across the point functions: +56 registers, spills)
      constexpr int QB = 27;
      bool reg_b = REGB && NQ == QB && !(dbg_stop & 8);
      // ---- 5q. the same products streamed over the points (the 89-dof navierstokes element, row-gather scratch) ----
      // Row panels and column tiles are the same 16-dof pieces of the variables (seven of them).  Wave w < 7 owns column
      // tile w and keeps its accumulators for ALL row panels (7 x 4 doubles per lane).  Per point q a wave
      //   * produces the P block of its own panel two points ahead (1 MFMA, three buffers in LDS)
      //       P_q[p][m][r] = sum_s C^(q)[sp_i+s][m] T^[q][s][i_r]     (its B operand is the tile's B operand of that point)
      //   * consumes the seven blocks of the current point, requested from LDS during the previous point
      //       acc[p] += P_q[p][sp_j+s][r] T^[q][s][j]                (7 MFMAs, the B operand read once per point)
      // and meets the others at ONE LDS barrier: the MFMAs of a wave between two barriers are independent and their
      // operands are in registers when the barrier opens.  The panel-by-panel form below serialises 14 short phases per
      // element behind block barriers (P of a panel: 4 dependent LDS round trips; tiles: 27 MFMAs per wave) and ran the
      // matrix pipe at a third.  The eighth wave has no tile: it accumulates the residual rows point by point beside
      // the others (phase 4 is skipped).  The finished element matrix is put together in LDS (over C^ and P, both dead
      // by then) in the layout of the scratch and leaves as contiguous runs of 8-byte stores instead of 16 lanes
      // 24-32 B apart.
      if constexpr (REGB) {
        if (q_stream) {
          constexpr int MAXT = NWV - 1, PB = MAXT * 256;  // doubles of one buffer: [panel][m][r]
          // the seven pieces: (variable, first dof), in variable order; piece w is wave w's tile
          int pan_n[MAXT], pan_i0[MAXT];
          int my_c0 = 0, my_sp = 0, my_cp = 0, my_to = 0, my_card = 0, my_vp = 0;
          {
            int tile = 0;
#pragma unroll
            for (int vj = 0; vj < L::nvars; ++vj) {
#pragma unroll
              for (int c0 = 0; c0 < 32; c0 += 16) {
                if (c0 < vl.card[vj]) {
                  if (tile == wv) {
                    my_c0 = c0; my_sp = vl.slotptr[vj]; my_cp = vl.cardpad[vj]; my_to = vl.table_off[vj];
                    my_card = vl.card[vj]; my_vp = vl.varptr[vj];
                  }
#pragma unroll
                  for (int p = 0; p < MAXT; ++p)
                    if (p == tile) { pan_n[p] = min(16, vl.card[vj] - c0); pan_i0[p] = vl.varptr[vj] + c0; }
                  ++tile;
                }
              }
            }
          }
          const bool owner = wv < MAXT;
          const int spj = my_sp, cpj = my_cp;
          const double *Tj = tab + my_to;
          const bool cb = owner && my_c0 + l15 < my_card;
          const int colj = cb ? my_c0 + l15 : 0;
          auto tile_b = [&](int q) {  // T^[q][s = l4][j = the tile's column l15], zero beyond the variable
            const double t = Tj[(min(q, 26) * 4 + l4) * cpj + colj];
            return cb ? t : 0.0;
          };
          auto put = [&](int q, v4d d) {
            double *wb = s_P + (q % 3) * PB + (wv * 16 + l4) * 16 + l15;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) wb[4 * t4 * 16] = d[t4];
          };
          // residual rows of the spare wave: f = lane and lane + 64
          const double *rT[2];
          int rsp[2], rcp[2];
          double rr[2] = {0.0, 0.0};
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int f = min(lane + 64 * k, n - 1);
            const VarAt va = var_at(f, false);
            rsp[k] = va.sp; rcp[k] = va.cp;
            rT[k] = tab + va.to + (f - va.vp);
          }
          v4d acc[MAXT];
#pragma unroll
          for (int p = 0; p < MAXT; ++p) acc[p] = {0.0, 0.0, 0.0, 0.0};
          // profiling (MHA_ENGINE_STOP): 5 = no point loop, 6 = the point loop only, 7 = all but the global stores
          const int nq_run = (dbg_stop & 7) == 5 ? 0 : NQ;
          if (it + 1 < iters && gt < n) {  // next element's row id (its value is requested after the point loop)
            const int el_n = (el_begin + it + 1 < el_end) ? el_begin + it + 1 : max(min(el_end, b.e_count) - 1, 0);
            row_pf = b.lids[(size_t)(b.e_begin + el_n) * n + pos_k];
          }
          // operands of the own-panel product are requested a point before it is issued: it goes into the pipe FIRST
          // and its block is complete when the seven products behind it have been issued (written without a wait)
          auto load_ca = [&](int q) { return s_Ch[(min(q, 26) * NS + spj + l4) * NS + l15]; };  // A[row = m][k = s]
          auto own_block = [&](double ca, double bq) {
            v4d d = {0.0, 0.0, 0.0, 0.0};
            return __builtin_amdgcn_mfma_f64_16x16x4f64(ca, bq, d, 0, 0, 0);
          };
          double bb = tile_b(0), b1 = tile_b(1), b2 = tile_b(2), b3;
          double ca2 = load_ca(2), ca3;
          double av[MAXT], an[MAXT];
          if (owner) {
            put(0, own_block(load_ca(0), bb));
            put(1, own_block(load_ca(1), b1));
          }
          sync();
          if (owner) {
            const double *buf = s_P + (spj + l4) * 16 + l15;
#pragma unroll
            for (int p = 0; p < MAXT; ++p) av[p] = buf[p * 256];
          }
          // (27 points, unrolled: the buffer rotation, the per-point addresses and the register moves of the software
          // pipeline become constants and renaming)
          constexpr int QN = 27;
          if (nq_run)
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            if (owner) {
              const v4d d = own_block(ca2, b2);  // P of point q + 2
              __builtin_amdgcn_sched_barrier(0);
              const double *buf = s_P + ((q + 1) % 3) * PB + (spj + l4) * 16 + l15;
#pragma unroll
              for (int p = 0; p < MAXT; ++p) an[p] = buf[p * 256];  // next point's blocks (complete since the last barrier)
              b3 = tile_b(q + 3);
              ca3 = load_ca(q + 3);
#pragma unroll
              for (int p = 0; p < MAXT; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p], bb, acc[p], 0, 0, 0);
              put(q + 2, d);
#pragma unroll
              for (int p = 0; p < MAXT; ++p) av[p] = an[p];
              bb = b1; b1 = b2; b2 = b3; ca2 = ca3;
            } else {
              // (all sixteen operands requested before the first product: left to itself the compiler waits for each
              // pair, ten LDS round trips in a row -- longer than the owners' eight MFMAs, and they wait at the barrier)
              double tv[2][4], fv[2][4];
#pragma unroll
              for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) { tv[k][sl] = rT[k][(q * 4 + sl) * rcp[k]]; fv[k][sl] = s_Fh[q * NS + rsp[k] + sl]; }
              }
              asm volatile("" : "+v"(tv[0][0]), "+v"(tv[0][1]), "+v"(tv[0][2]), "+v"(tv[0][3]), "+v"(tv[1][0]), "+v"(tv[1][1]),
                                "+v"(tv[1][2]), "+v"(tv[1][3]), "+v"(fv[0][0]), "+v"(fv[0][1]), "+v"(fv[0][2]), "+v"(fv[0][3]),
                                "+v"(fv[1][0]), "+v"(fv[1][1]), "+v"(fv[1][2]), "+v"(fv[1][3]));
#pragma unroll
              for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) rr[k] += tv[k][sl] * fv[k][sl];
              }
            }
            sync();
          }
          if (it + 1 < iters && gt < n) cu_pf = tm.u[row_pf];
          have_pf = true;
          if (!owner && nq_run) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int f = lane + 64 * k;
              if (f < n) {
                const double r = rr[k] * s_sgn[f];
                const int row = s_row[f];
                if (out.local_res) {
                  double *lr = out.local_res + (size_t)(e - out.local_base) * n + s_pos[f];
                  *lr = out.local_store ? -r : *lr - r;
                }
                if (out.res && !(b.fixed && b.fixed[row])) unsafeAtomicAdd(out.res + row, -r);
              }
            }
          }
          if ((dbg_stop & 7) == 6) {
            double k = 0.0;
#pragma unroll
            for (int p = 0; p < MAXT; ++p) k += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
            if (k == 1.2345e300 && lj_e) lj_e[0] = k;
            continue;
          }
          // element matrix in LDS (scratch layout: [pos_i][pos_j]), signs applied
          double *img = s_P;
          if (cb) {
            const int j = my_vp + my_c0 + l15, pos_j = s_pos[j];
            const bool signs = vl.orient != nullptr;  // (HGRAD blocks carry none: every s_sgn is 1)
            const double sgj = signs ? s_sgn[j] : 1.0;
#pragma unroll
            for (int p = 0; p < MAXT; ++p) {
#pragma unroll
              for (int t4 = 0; t4 < 4; ++t4) {
                const int r = l4 + 4 * t4;
                if (r < pan_n[p]) {
                  const int i = pan_i0[p] + r;
                  img[s_pos[i] * n + pos_j] = signs ? acc[p][t4] * s_sgn[i] * sgj : acc[p][t4];
                }
              }
            }
          }
          sync();
          if (lj_e && (dbg_stop & 7) != 7) {
            if (nt_store) { for (int k = tid; k < n * n; k += TPE) __builtin_nontemporal_store(img[k], lj_e + k); }
            else { for (int k = tid; k < n * n; k += TPE) lj_e[k] = img[k]; }
          }
          continue;  // (the sync at the top of the element loop keeps the image until every wave has read its part)
        }
      }
      double breg[REGB ? QB : 1];
      if constexpr (REGB) {
        int tile = 0, my_vj = -1, my_c0 = 0;
        for (int vj = 0; vj < vl.nvars; ++vj) {
          if (vl.nslot[vj] != 4) reg_b = false;
          for (int c0 = 0; c0 < vl.card[vj]; c0 += 16, ++tile)
            if (tile == wv) { my_vj = vj; my_c0 = c0; }
        }
        if (tile > NWV) reg_b = false;
        if (reg_b) {
          const int vj = max(my_vj, 0);
          const double *Tj = tab + vl.table_off[vj];
          const int cpj = vl.cardpad[vj];
          const bool cb = my_vj >= 0 && my_c0 + l15 < vl.card[vj];
          int col = cb ? my_c0 + l15 : 0;
          asm volatile("" : "+v"(col));
#pragma unroll
          for (int q = 0; q < QB; ++q) {
            const double t = Tj[(q * 4 + l4) * cpj + col];
            breg[q] = cb ? t : 0.0;
          }
        }
      }
      for (int vi = 0; vi < vl.nvars; ++vi) {
        const int nsi = vl.nslot[vi], spi = vl.slotptr[vi], cpi = vl.cardpad[vi], cardi = vl.card[vi];
        const double *Ti = tab + vl.table_off[vi];
        for (int r0 = 0; r0 < cardi; r0 += kPanelRows) {
          // P panel, computed transposed (D[m][dof]) so that the 16 lanes of a row store 16 consecutive doubles:
          // conflict-free LDS writes; points q are dealt to the element's waves
          for (int q = wv; q < NQ; q += NWV) {
            const bool ka = l4 < nsi;
            const double tb = (ka && r0 + l15 < cardi) ? Ti[(q * nsi + l4) * cpi + r0 + l15] : 0.0;  // B[k = s][col = dof]
            for (int m0 = 0; m0 < NS; m0 += 16) {
              const double ca = (ka && m0 + l15 < NS) ? s_Ch[(q * NS + spi + l4) * NS + m0 + l15] : 0.0;  // A[row = m][k = s]
              v4d d = {0.0, 0.0, 0.0, 0.0};
              d = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, tb, d, 0, 0, 0);
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int m = m0 + l4 + 4 * t;
                if (m < NS) s_P[(q * NS + m) * kPanelRows + l15] = d[t];
              }
            }
          }
          sync();
          if (dbg_stop == 5) { sync(); continue; }
          // column tiles: (variable vj, 16 dofs), dealt to the waves
          int tile = 0;
          for (int vj = 0; vj < vl.nvars; ++vj) {
            const int nsj = vl.nslot[vj], spj = vl.slotptr[vj], cpj = vl.cardpad[vj], cardj = vl.card[vj];
            const double *Tj = tab + vl.table_off[vj];
            for (int c0 = 0; c0 < cardj; c0 += 16, ++tile) {
              if (tile % NWV != wv) continue;
              const bool cb = c0 + l15 < cardj;
              v4d d = {0.0, 0.0, 0.0, 0.0};
              if (!EXPR && (dbg_stop & 7) == 7) {  // profiling (plain-coefficient instantiations only: the deck-string ones are at their scratch limit): no products, the stores only
              } else if (REGB && reg_b) {
                v4d d1 = {0.0, 0.0, 0.0, 0.0};
                const double *pa = s_P + (spj + l4) * kPanelRows + l15;
#pragma unroll
                for (int q0 = 0; q0 < QB; q0 += 9) {
                  double av[9];
#pragma unroll
                  for (int u = 0; u < 9; ++u) av[u] = pa[(q0 + u) * NS * kPanelRows];
#pragma unroll
                  for (int u = 0; u < 9; ++u) {
                    if (u & 1) d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], breg[REGB ? q0 + u : 0], d1, 0, 0, 0);
                    else d = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], breg[REGB ? q0 + u : 0], d, 0, 0, 0);
                  }
                }
                d += d1;
              } else if (!EXPR && nsj == 4 && NQ % 9 == 0) {  // (the deck-string instantiations keep the plain loop: they are at their scratch limit)
                // one point per MFMA (k = slot).  Operands of nine points are requested before the first product and the
                // products alternate between two accumulators: the loop used to be a chain of 27 dependent MFMAs, each
                // waiting for its own two LDS reads (a wave has one tile per panel: nothing else hides that latency) --
                // 70 % of the kernel's time at 89 dofs per element
                v4d d1 = {0.0, 0.0, 0.0, 0.0};
                for (int q0 = 0; q0 < NQ; q0 += 9) {
                  double av[9], bv[9];
#pragma unroll
                  for (int u = 0; u < 9; ++u) {
                    av[u] = s_P[((q0 + u) * NS + spj + l4) * kPanelRows + l15];
                    const double t = Tj[((q0 + u) * 4 + l4) * cpj + (cb ? c0 + l15 : 0)];
                    bv[u] = cb ? t : 0.0;
                  }
#pragma unroll
                  for (int u = 0; u < 9; ++u) {
                    if (u & 1) d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], d1, 0, 0, 0);
                    else d = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], d, 0, 0, 0);
                  }
                }
                d += d1;
              } else if (nsj == 4) {  // one point per MFMA: k = slot
                for (int q = 0; q < NQ; ++q) {
                  const double a = s_P[(q * NS + spj + l4) * kPanelRows + l15];
                  const double bb = cb ? Tj[(q * 4 + l4) * cpj + c0 + l15] : 0.0;
                  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, d, 0, 0, 0);
                }
              } else {
                const int K = NQ * nsj;
                for (int k0 = 0; k0 < K; k0 += 4) {
                  const int k = k0 + l4, q = k / nsj, sl = k - q * nsj;
                  const bool kv = k < K;
                  const double a = kv ? s_P[(q * NS + spj + sl) * kPanelRows + l15] : 0.0;
                  const double bb = (kv && cb) ? Tj[(q * nsj + sl) * cpj + c0 + l15] : 0.0;
                  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, d, 0, 0, 0);
                }
              }
              if (!EXPR && (dbg_stop & 7) == 6) {  // profiling: products only (kept alive), no stores
                if (d[0] + d[1] + d[2] + d[3] == 1.2345e300 && lj_e) lj_e[0] = 1.0;
              } else if (cb) {
                const int j = vl.varptr[vj] + c0 + l15, pos_j = s_pos[j];
                const double sgj = s_sgn[j];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                  const int r = r0 + l4 + 4 * t;
                  if (r >= cardi) continue;
                  const int i = vl.varptr[vi] + r, pos_i = s_pos[i], row_i = s_row[i];
                  const double val = d[t] * s_sgn[i] * sgj;
                  if (lj_e) {
                    double *lj = lj_e + (pos_i * n + pos_j);
                    // (the row-gather scratch is written once and read once, by another kernel: nontemporal)
                    if (out.local_store) { if (nt_store) __builtin_nontemporal_store(val, lj); else *lj = val; }
                    else *lj = *lj + val;
                  }
                  if (out.crs_vals && !(b.fixed && b.fixed[row_i])) {
                    const size_t so = ((size_t)e * n + pos_i) * n + pos_j;
                    int p;
                    if (slot8) p = b.rowptr[row_i] + slot8[so];
                    else if (slot16) p = b.rowptr[row_i] + slot16[so];
                    else p = find_col(b.colind, b.rowptr[row_i], b.rowptr[row_i + 1], s_row[j]);
                    if (p >= 0) unsafeAtomicAdd(out.crs_vals + p, val);
                  }
                }
              }
            }
          }
          sync();
        }
      }
    }
  }
}

template <int DIM, int PHYS>
void launch_typed(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                  const ElemOut &out, const void *slot, int slot_bytes, hipStream_t stream) {
  using L = Layout<PHYS, DIM>;
  MHA_REQUIRE(vl.nvars == L::nvars && vl.ns_tot == L::NS, MHA_ERR_INVALID,
              "variable layout does not match the physics module (" << vl.nvars << " variables, " << vl.ns_tot
                                                                    << " slots)");
  for (int v = 0; v < L::nvars; ++v)
    MHA_REQUIRE(vl.type[v] == L::type(v), MHA_ERR_INVALID, "basis type of variable " << v << " does not match the module");
  const size_t per_group = engine_group_doubles(vl, geo_size<DIM>()) * sizeof(double);
  const size_t tables = vl.tables_size * sizeof(double);
  // small elements: one wave per element, four elements per workgroup; large ones: the whole workgroup per element
  // as many elements per workgroup as the LDS holds (8, 4, 2 or 1): TPE = 64, 128, 256 or 512 threads per element
  int groups = 1;
  for (int g = kEngineThreads / 64; g >= 1; g >>= 1)
    if (tables + g * per_group <= 160 * 1024 && (g == 1 || vl.n_tot <= 64)) { groups = g; break; }
  const size_t lds = tables + groups * per_group;
  MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "element needs " << lds << " B of LDS (limit 160 KB)");
  const int num_cu = current_device_num_cus();
  const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(size_t(2), (160 * 1024) / lds)));
  PhysParamsDev ppd = pp;
  if (const char *st = std::getenv("MHA_ENGINE_STOP")) { if (pp.physics > 0) ppd.p[7] = std::atof(st); }
  const int grid = std::max(1, std::min((b.e_count + groups - 1) / groups, num_cu * per_cu));
  const uint8_t *s8 = (slot && slot_bytes == 1) ? static_cast<const uint8_t *>(slot) : nullptr;
  const uint16_t *s16 = (slot && slot_bytes == 2) ? static_cast<const uint16_t *>(slot) : nullptr;
  auto go = [&](auto kern) {
    require_modest_scratch(kern, "point engine");
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kEngineThreads), lds, stream, b, vl, ppd, tm, out, s8, s16);
    MHA_HIP(hipGetLastError());
  };
  // compile-time point counts pay for small elements; for large ones the extra unrolling spills (measured: navierstokes
  // 32^3 9.1 ms with, 7.2 ms without)
  const int nq1 = groups < 4 ? 0 : (vl.nq == (DIM == 2 ? 4 : 8)) ? 2 : (vl.nq == (DIM == 2 ? 9 : 27)) ? 3 : 0;
  auto pick = [&](auto tpe) {
    constexpr int T = decltype(tpe)::value;
    if (nq1 == 2) go(point_engine_kernel<DIM, PHYS, T, 2, 0>);
    else if (nq1 == 3) go(point_engine_kernel<DIM, PHYS, T, 3, 0>);
    else go(point_engine_kernel<DIM, PHYS, T, 0, 0>);
  };
  if (pp.physics > 0 && uses_fields(pp)) {
    // deck strings that read the solution fields: the Dual interpreter; built for the thermal module
    if constexpr (PHYS == MHA_PHYSICS_THERMAL) {
      if (groups == 8) go(point_engine_kernel<DIM, PHYS, 64, 0, 2>);
      else if (groups == 4) go(point_engine_kernel<DIM, PHYS, 128, 0, 2>);
      else if (groups == 2) go(point_engine_kernel<DIM, PHYS, 256, 0, 2>);
      else go(point_engine_kernel<DIM, PHYS, 512, 0, 2>);
    } else {
      MHA_REQUIRE(false, MHA_ERR_INVALID, "functions of the solution fields are built for the thermal module");
    }
  } else if (pp.physics > 0 && has_expression(pp)) {
    // deck-string functions: the interpreter call costs registers and scratch, so only these instantiations carry it
    if (groups == 8) go(point_engine_kernel<DIM, PHYS, 64, 0, 1>);
    else if (groups == 4) go(point_engine_kernel<DIM, PHYS, 128, 0, 1>);
    else if (groups == 2) go(point_engine_kernel<DIM, PHYS, 256, 0, 1>);
    else go(point_engine_kernel<DIM, PHYS, 512, 0, 1>);
  } else if (groups == 8) pick(std::integral_constant<int, 64>());
  else if (groups == 4) pick(std::integral_constant<int, 128>());
  else if (groups == 2) go(point_engine_kernel<DIM, PHYS, 256, 0, 0>);
  else go(point_engine_kernel<DIM, PHYS, 512, 0, 0>);
}

}  // namespace

void launch_point_engine(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                         const ElemOut &out, const void *slot, int slot_bytes, hipStream_t stream) {
  if (b.e_count <= 0) return;
  const int key = b.dim * 10 + (pp.physics < 0 ? -pp.physics : pp.physics);
  switch (key) {
    case 20 + MHA_PHYSICS_THERMAL: launch_typed<2, MHA_PHYSICS_THERMAL>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    case 30 + MHA_PHYSICS_THERMAL: launch_typed<3, MHA_PHYSICS_THERMAL>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    case 20 + MHA_PHYSICS_POROUS_MIXED: launch_typed<2, MHA_PHYSICS_POROUS_MIXED>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    case 30 + MHA_PHYSICS_POROUS_MIXED: launch_typed<3, MHA_PHYSICS_POROUS_MIXED>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    case 20 + MHA_PHYSICS_NAVIERSTOKES: launch_typed<2, MHA_PHYSICS_NAVIERSTOKES>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    case 30 + MHA_PHYSICS_NAVIERSTOKES: launch_typed<3, MHA_PHYSICS_NAVIERSTOKES>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    case 20 + MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED: launch_typed<2, MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED>(b, vl, pp, tm, out, slot, slot_bytes, stream); break;
    default: MHA_REQUIRE(false, MHA_ERR_INVALID, "no point-engine kernel for physics " << pp.physics << " in " << b.dim << "-D");
  }
}

}  // namespace mha
