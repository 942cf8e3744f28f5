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
// One 256-thread workgroup per element, persistent over elements:
//   1 gather + seeding values (x orientation sign), geometry per point
//   2 reference-slot fields U^(q) = sum_j u_j T^(j,q)
//   3 one thread per (point, direction): the module's point function on Dual numbers -> one column of C^(q)
//   4 residual rows;  5 Jacobian rows, one row per wave at a time: P = T^(i,.) C^ then P . T^(j,.) across lanes.
// Scatter: atomics into res / CRS (column search) or dense local_J / local_res (updateJac / updateRes convention).
#include <hip/hip_runtime.h>

#include <algorithm>

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

constexpr int kEngineThreads = 256, kEngineWaves = kEngineThreads / 64;

template <int DIM>
constexpr int geo_size() { return 2 * DIM * DIM + 2 + DIM; }  // J, Ji, det, w, x

__host__ __device__ inline size_t engine_lds_doubles(const VarLayoutDev &vl, int geo) {
  const size_t n = vl.n_tot, NS = vl.ns_tot, NQ = vl.nq;
  return vl.tables_size + 3 * n + NQ * geo + 3 * NQ * NS + NQ * NS * NS + kEngineWaves * NQ * NS + n /*row,pos as int pairs*/;
}

template <int DIM, int PHYS>
__global__ __launch_bounds__(kEngineThreads) void point_engine_kernel(BlockDev b, VarLayoutDev vl, PhysParamsDev pp,
                                                                      TimeDev tm, ElemOut out) {
  using L = Layout<PHYS, DIM>;
  constexpr int NS = L::NS, NN = 1 << DIM, GEO = geo_size<DIM>();
  extern __shared__ double smem[];
  const int n = vl.n_tot, NQ = vl.nq, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *tab = smem;
  double *s_u = tab + vl.tables_size, *s_ud = s_u + n, *s_sgn = s_ud + n;
  double *s_geo = s_sgn + n;
  double *s_Uh = s_geo + NQ * GEO, *s_Udh = s_Uh + NQ * NS, *s_Fh = s_Udh + NQ * NS;
  double *s_Ch = s_Fh + NQ * NS;
  double *s_P = s_Ch + NQ * NS * NS;
  int *s_row = reinterpret_cast<int *>(s_P + kEngineWaves * NQ * NS), *s_pos = s_row + n;

  for (int k = tid; k < vl.tables_size; k += kEngineThreads) tab[k] = vl.tables[k];

  for (int el = blockIdx.x; el < b.e_count; el += gridDim.x) {
    const int e = b.e_begin + el;
    __syncthreads();  // tables loaded / previous element done with LDS
    // ---- 1. gather + seeding values, geometry ----
    if (tid < n) {
      const int pos = b.offsets[tid], row = b.lids[(size_t)e * n + pos];
      const double sg = vl.orient ? (double)vl.orient[(size_t)e * n + tid] : 1.0;
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
      s_u[tid] = ue * sg;
      s_ud[tid] = ud * sg;
      s_sgn[tid] = sg;
      s_row[tid] = row;
      s_pos[tid] = pos;
    }
    if (tid >= 64 && tid < 64 + NQ) {
      const int q = tid - 64;
      const double *xn = b.nodes + (size_t)e * NN * DIM;
      double J[DIM * DIM], Ji[DIM * DIM], det, x[DIM];
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double s = 0.0;
          for (int k = 0; k < NN; ++k) s += xn[k * DIM + r] * b.nodegrad[(k * NQ + q) * DIM + c];
          J[r * DIM + c] = s;
        }
        double s = 0.0;
        for (int k = 0; k < NN; ++k) s += xn[k * DIM + r] * b.nodeval[k * NQ + q];
        x[r] = s;
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
    __syncthreads();
    // ---- 2. reference-slot fields ----
    for (int idx = tid; idx < NQ * NS; idx += kEngineThreads) {
      const int q = idx / NS, m = idx - q * NS;
      int v = 0;
      while (m >= vl.slotptr[v + 1]) ++v;
      const int s = m - vl.slotptr[v], ns = vl.nslot[v], card = vl.card[v];
      const double *T = tab + vl.table_off[v] + (size_t)q * ns + s;
      const double *uu = s_u + vl.varptr[v], *ud = s_ud + vl.varptr[v];
      double a = 0.0, ad = 0.0;
      for (int dof = 0; dof < card; ++dof) {
        const double t = T[(size_t)dof * NQ * ns];
        a += uu[dof] * t;
        ad += ud[dof] * t;
      }
      s_Uh[idx] = a;
      s_Udh[idx] = ad;
    }
    __syncthreads();
    // ---- 3. point function, one (point, direction) per thread ----
    double vol = 0.0;
    for (int q = 0; q < NQ; ++q) vol += s_geo[q * GEO + 2 * DIM * DIM + 1];
    const double h = (DIM == 2) ? sqrt(vol) : cbrt(vol);  // Workset::getElementSize (workset.cpp:2666-2679)
    for (int idx = tid; idx < NQ * (NS + 1); idx += kEngineThreads) {
      const int q = idx / (NS + 1), m = idx - q * (NS + 1);
      const double *g = s_geo + q * GEO;
      const double *J = g, *Ji = g + DIM * DIM;
      const double det = g[2 * DIM * DIM], w = g[2 * DIM * DIM + 1];
      Dual U[NS], Ud[NS], F[NS];
#pragma unroll
      for (int v = 0; v < L::nvars; ++v) {
        constexpr int dummy = 0;
        (void)dummy;
        const int type = L::type(v), sp = slotptr_of<L, DIM>(v), ns = slots_of(type, DIM);
        double ref[1 + DIM], phys[1 + DIM], refd[1 + DIM], physd[1 + DIM], dir[1 + DIM], pdir[1 + DIM];
#pragma unroll
        for (int s = 0; s < ns; ++s) {
          ref[s] = s_Uh[q * NS + sp + s];
          refd[s] = s_Udh[q * NS + sp + s];
          dir[s] = (m == sp + s) ? 1.0 : 0.0;
        }
        to_phys<DIM>(type, ref, J, Ji, det, phys);
        to_phys<DIM>(type, refd, J, Ji, det, physd);
        to_phys<DIM>(type, dir, J, Ji, det, pdir);
#pragma unroll
        for (int s = 0; s < ns; ++s) {
          U[sp + s] = mk(phys[s], tm.alpha_u * pdir[s]);
          Ud[sp + s] = value_like(type, s, DIM) ? mk(physd[s], tm.alpha_t * pdir[s]) : mk(0.0);
        }
      }
      PointArgs<DIM> pa;
      pa.U = U; pa.Ud = Ud; pa.x = g + 2 * DIM * DIM + 2; pa.h = h; pa.dt = tm.dt;
      pa.transient = tm.transient; pa.e = e; pa.q = q; pa.nq = NQ; pa.pp = &pp;
      if constexpr (PHYS == MHA_PHYSICS_THERMAL) thermal_point<DIM>(pa, F);
      else if constexpr (PHYS == MHA_PHYSICS_POROUS_MIXED) porous_point<DIM>(pa, F);
      else navierstokes_point<DIM>(pa, F);
#pragma unroll
      for (int v = 0; v < L::nvars; ++v) {
        const int type = L::type(v), sp = slotptr_of<L, DIM>(v), ns = slots_of(type, DIM);
        double phys[1 + DIM], ref[1 + DIM];
#pragma unroll
        for (int s = 0; s < ns; ++s) phys[s] = (m == NS) ? F[sp + s].v : F[sp + s].d;
        to_ref_T<DIM>(type, phys, J, Ji, det, ref);
#pragma unroll
        for (int s = 0; s < ns; ++s) {
          if (m == NS) s_Fh[q * NS + sp + s] = w * ref[s];
          else s_Ch[(q * NS + sp + s) * NS + m] = w * ref[s];
        }
      }
    }
    __syncthreads();
    // ---- 4. residual rows ----
    if (tid < n) {
      int v = 0;
      while (tid >= vl.varptr[v + 1]) ++v;
      const int dof = tid - vl.varptr[v], ns = vl.nslot[v], sp = vl.slotptr[v];
      const double *T = tab + vl.table_off[v] + (size_t)dof * NQ * ns;
      double r = 0.0;
      for (int q = 0; q < NQ; ++q)
        for (int s = 0; s < ns; ++s) r += T[q * ns + s] * s_Fh[q * NS + sp + s];
      r *= s_sgn[tid];
      const int row = s_row[tid];
      if (out.local_res) out.local_res[(size_t)(e - out.local_base) * n + s_pos[tid]] -= r;
      if (out.res && !(b.fixed && b.fixed[row])) unsafeAtomicAdd(out.res + row, -r);
    }
    // ---- 5. Jacobian rows: wave `wave` takes rows wave, wave+4, ... ----
    if (out.compute_jacobian) {
      double *P = s_P + wave * NQ * NS;
      for (int i0 = 0; i0 < n; i0 += kEngineWaves) {
        const int i = i0 + wave;
        const bool active = i < n;
        int vi = 0;
        if (active) while (i >= vl.varptr[vi + 1]) ++vi;
        if (active) {
          const int dof = i - vl.varptr[vi], ns = vl.nslot[vi], sp = vl.slotptr[vi];
          const double *T = tab + vl.table_off[vi] + (size_t)dof * NQ * ns;
          for (int idx = lane; idx < NQ * NS; idx += 64) {
            const int q = idx / NS, m = idx - q * NS;
            double a = 0.0;
            for (int s = 0; s < ns; ++s) a += T[q * ns + s] * s_Ch[(q * NS + sp + s) * NS + m];
            P[idx] = a;
          }
        }
        __syncthreads();
        if (active) {
          const int row_i = s_row[i];
          const bool skip = out.crs_vals == nullptr || (b.fixed && b.fixed[row_i]);
          const double sgi = s_sgn[i];
          for (int j = lane; j < n; j += 64) {
            int vj = 0;
            while (j >= vl.varptr[vj + 1]) ++vj;
            const int dofj = j - vl.varptr[vj], nsj = vl.nslot[vj], spj = vl.slotptr[vj];
            const double *T = tab + vl.table_off[vj] + (size_t)dofj * NQ * nsj;
            double a = 0.0;
            for (int q = 0; q < NQ; ++q)
              for (int s = 0; s < nsj; ++s) a += P[q * NS + spj + s] * T[q * nsj + s];
            a *= sgi * s_sgn[j];
            if (out.local_J) out.local_J[((size_t)(e - out.local_base) * n + s_pos[i]) * n + s_pos[j]] += a;
            if (!skip) {
              const int p = find_col(b.colind, b.rowptr[row_i], b.rowptr[row_i + 1], s_row[j]);
              if (p >= 0) unsafeAtomicAdd(out.crs_vals + p, a);
            }
          }
        }
        __syncthreads();
      }
    }
  }
}

template <int DIM, int PHYS>
void launch_typed(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                  const ElemOut &out, hipStream_t stream) {
  using L = Layout<PHYS, DIM>;
  MHA_REQUIRE(vl.nvars == L::nvars && vl.ns_tot == L::NS, MHA_ERR_INVALID,
              "variable layout does not match the physics module (" << vl.nvars << " variables, " << vl.ns_tot
                                                                    << " slots)");
  for (int v = 0; v < L::nvars; ++v)
    MHA_REQUIRE(vl.type[v] == L::type(v), MHA_ERR_INVALID, "basis type of variable " << v << " does not match the module");
  MHA_REQUIRE(vl.n_tot <= kEngineThreads && vl.nq <= kEngineThreads - 64, MHA_ERR_INVALID,
              "point engine supports at most " << kEngineThreads << " dofs and " << kEngineThreads - 64
                                               << " integration points per element");
  const size_t lds = engine_lds_doubles(vl, geo_size<DIM>()) * sizeof(double);
  MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "element needs " << lds << " B of LDS (limit 160 KB)");
  auto kern = point_engine_kernel<DIM, PHYS>;
  static bool attr_set = false;
  if (!attr_set) {
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    MHA_HIP(hipGetDevice(&dev));
    MHA_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds)));
  const int grid = std::min(b.e_count, num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kEngineThreads), lds, stream, b, vl, pp, tm, out);
  MHA_HIP(hipGetLastError());
}

}  // namespace

void launch_point_engine(const BlockDev &b, const VarLayoutDev &vl, const PhysParamsDev &pp, const TimeDev &tm,
                         const ElemOut &out, hipStream_t stream) {
  if (b.e_count <= 0) return;
  const int key = b.dim * 10 + pp.physics;
  switch (key) {
    case 20 + MHA_PHYSICS_THERMAL: launch_typed<2, MHA_PHYSICS_THERMAL>(b, vl, pp, tm, out, stream); break;
    case 30 + MHA_PHYSICS_THERMAL: launch_typed<3, MHA_PHYSICS_THERMAL>(b, vl, pp, tm, out, stream); break;
    case 20 + MHA_PHYSICS_POROUS_MIXED: launch_typed<2, MHA_PHYSICS_POROUS_MIXED>(b, vl, pp, tm, out, stream); break;
    case 30 + MHA_PHYSICS_POROUS_MIXED: launch_typed<3, MHA_PHYSICS_POROUS_MIXED>(b, vl, pp, tm, out, stream); break;
    case 20 + MHA_PHYSICS_NAVIERSTOKES: launch_typed<2, MHA_PHYSICS_NAVIERSTOKES>(b, vl, pp, tm, out, stream); break;
    case 30 + MHA_PHYSICS_NAVIERSTOKES: launch_typed<3, MHA_PHYSICS_NAVIERSTOKES>(b, vl, pp, tm, out, stream); break;
    default: MHA_REQUIRE(false, MHA_ERR_INVALID, "no point-engine kernel for physics " << pp.physics << " in " << b.dim << "-D");
  }
}

}  // namespace mha
