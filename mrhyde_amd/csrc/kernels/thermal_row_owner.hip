// thermal_row_owner.hip -- thermal volume assembly for affine elements on gfx950, without global
// atomics on the Jacobian and without dense element matrices in HBM.
//
// Two kernels per assembly:
//
//  K1  thermal_affine_element_kernel   (element-wise, 32 lanes per element)
//      gather + seeding values, solution fields at the integration points, the residual rows by
//      quadrature, scattered as -res.val() with one f64 atomic per dof (n values per element -- 1/n of
//      the Jacobian volume).  Element geometry {detJ*J^{-1}J^{-T}, detJ, J, centroid} comes from a
//      160-byte per-element cache filled once per mesh (affine_geometry_kernel).
//
//  K2  row_owner_jacobian_kernel       (row-owner, one workgroup per row block, row_blocks.hpp)
//      every CRS row is produced by exactly one workgroup: it walks the elements incident to its rows,
//      accumulates their contributions in LDS (ds_add_f64) and streams whole rows to HBM with plain
//      coalesced stores.  Lane (si,sj) keeps "its" entries of the reference tables in registers:
//        res(e,si).dx(sj) = alpha_u * sum_{a<=b} Gs_ab * Khat_ab[si][sj] + alpha_t * rho*cp*detJ * Mhat[si][sj]
//      with Gs = kappa*detJ*J^{-1}J^{-T} (constant cell Jacobian; Khat/Mhat integrated once at setup
//      with the block's cubature).
//      Element data are wave-uniform scalar loads; a wave skips elements none of whose owned rows
//      fall into its (si) range; finished rows leave in contiguous runs.
//
// Together they cover the reference routines listed in thermal_element.hip (performGather, seeding,
// getPhysicalVolumetricBasis, evaluateSolutionField, thermal::volumeResidual, scatter with fixed-row
// skip: src/managers/assemblyManager.cpp:3598-3643, 4031-4145; src/tools/workset.cpp:559-623, 823-859,
// 937-1062; src/interfaces/discretizationInterface.cpp:732-776, 898-981; src/physics/thermal.cpp:71-165).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include <algorithm>
#include <cstdlib>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int cpow(int b, int e) { return e == 0 ? 1 : b * cpow(b, e - 1); }

// ---------------------------------------------------------------------------------------------
// setup kernels
// ---------------------------------------------------------------------------------------------

// flags[e] = 1 when the (multi)linear map of element e is affine: all mixed coefficients of the
// trilinear/bilinear geometry vanish relative to the size of the Jacobian columns.
template <int DIM>
__global__ __launch_bounds__(256) void classify_affine_kernel(BlockDev b, uint8_t *flags, double tol) {
  constexpr int NN = 1 << DIM;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < b.nelem; e += gridDim.x * blockDim.x) {
    const double *xn = b.nodes + (size_t)e * NN * DIM;
    double lin = 0.0, mix = 0.0;
    for (int r = 0; r < DIM; ++r) {
      for (int mask = 1; mask < NN; ++mask) {  // subset of directions entering the monomial
        double c = 0.0;
        for (int v = 0; v < NN; ++v) {  // signs of vertex v in shards order
          const int q = v & 3;
          const int sx = (q == 1 || q == 2) ? 1 : -1, sy = (q >= 2) ? 1 : -1, sz = (v >= 4) ? 1 : -1;
          int s = 1;
          if (mask & 1) s *= sx;
          if (mask & 2) s *= sy;
          if (mask & 4) s *= sz;
          c += s * xn[v * DIM + r];
        }
        c = fabs(c);
        if ((mask & (mask - 1)) == 0) lin = fmax(lin, c); else mix = fmax(mix, c);
      }
    }
    flags[e] = (mix <= tol * lin) ? 1 : 0;
  }
}

// Block-major slot table: bslot[pair][sj] = position of column LIDs[e][sj] inside the CRS row of the
// pair (one workgroup per row block; read once, fully coalesced, by K2).
template <typename SlotT>
__global__ __launch_bounds__(256) void build_block_slots_kernel(BlockDev b, RowBlocksDev rb, SlotT *bslot) {
  const int k = blockIdx.x, n = b.n;
  const int p0 = rb.pair_ptr[k], np = rb.pair_ptr[k + 1] - p0;
  const int r0 = rb.row_ptr[k], t0 = rb.elem_ptr[k];
  for (int item = threadIdx.x; item < np * n; item += blockDim.x) {
    const int p = item / n, sj = item - p * n;
    const uint32_t pk = rb.pairs[p0 + p];
    const int o = pk >> 16, t = (pk >> 8) & 0xff;
    const int row = rb.rows[r0 + o];
    const int e = rb.elems[t0 + t];
    const int lo = b.rowptr[row];
    const int c = find_col(b.colind, lo, b.rowptr[row + 1], b.lids[(size_t)e * n + sj]);
    bslot[rb.slot_ptr[k] / (int64_t)sizeof(SlotT) + item] = (SlotT)(c < 0 ? 0 : c - lo);
  }
}

// Cached geometry of every element, evaluated at the element centre (exact for affine elements):
// geo[e] = { detJ*(J^{-1}J^{-T})_sym (6 slots), detJ, J (9 slots), centroid (3 slots), pad }.
// The reference stores basis/basis_grad/wts per element at setup (Group::computeBasis, 24 KB per Q2
// hex); for affine elements these 160 bytes carry the same information.
template <int DIM>
__global__ __launch_bounds__(256) void affine_geometry_kernel(BlockDev b, double *geo) {
  constexpr int NN = 1 << DIM;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < b.nelem; e += gridDim.x * blockDim.x) {
    const double *xn = b.nodes + (size_t)e * NN * DIM;
    double J[DIM * DIM], Ji[DIM * DIM], det, xc[DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double c = 0.0;
#pragma unroll
      for (int cdir = 0; cdir < DIM; ++cdir) J[r * DIM + cdir] = 0.0;
#pragma unroll
      for (int v = 0; v < NN; ++v) {
        const double x = xn[v * DIM + r];
        const int q = v & 3;
        c += x;
        J[r * DIM + 0] += ((q == 1 || q == 2) ? x : -x);
        J[r * DIM + 1] += ((q >= 2) ? x : -x);
        if constexpr (DIM == 3) J[r * DIM + DIM - 1] += ((v >= 4) ? x : -x);
      }
      xc[r] = c * (1.0 / NN);
#pragma unroll
      for (int cdir = 0; cdir < DIM; ++cdir) J[r * DIM + cdir] *= (1.0 / NN);
    }
    invert<DIM>(J, Ji, det);
    double *g = geo + (size_t)e * kGeoRec;
    int k = 0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
      for (int c = a; c < DIM; ++c) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) s += Ji[a * DIM + d] * Ji[c * DIM + d];
        g[k++] = det * s;
      }
    for (; k < kGeoDet; ++k) g[k] = 0.0;
    g[kGeoDet] = det;
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) g[kGeoJ + i] = J[i];
#pragma unroll
    for (int d = 0; d < DIM; ++d) g[kGeoXc + d] = xc[d];
  }
}

// ---------------------------------------------------------------------------------------------
// K1: element-wise residual
// ---------------------------------------------------------------------------------------------

template <int DIM, int P, int NQ1, bool TR>
struct EK {
  static constexpr int M = P + 1;
  static constexpr int N = cpow(M, DIM);
  static constexpr int NQ = cpow(NQ1, DIM);
  static constexpr int NN = 1 << DIM;
  static constexpr int NSYM = DIM * (DIM + 1) / 2;
  static constexpr int TAB = 2 * M * NQ1 + 2 * NQ1;  // phi, dphi, gauss wts, gauss pts
  static constexpr int O_UE = 0;                 // u_eval  [N]   (basis order)
  static constexpr int O_UD = O_UE + N;          // u_dot   [N]   (transient runs only)
  static constexpr int O_GEO = O_UD + (TR ? N : 0);  // cached geometry record [kGeoRec]
  static constexpr int O_G = O_GEO;              // detJ*J^{-1}J^{-T}  [NSYM]
  static constexpr int O_DET = O_GEO + kGeoDet;  // detJ
  static constexpr int O_J = O_GEO + kGeoJ;      // J       [DIM*DIM]
  static constexpr int O_XC = O_GEO + kGeoXc;    // centroid [DIM]
  static constexpr int O_F = O_GEO + kGeoRec;    // w_q * Gs * grad_ref T(q)   [NQ][DIM]
  static constexpr int O_RQ = O_F + NQ * DIM;    // (rho cp T_t - f) detJ w_q  [NQ]
  static constexpr int REC = O_RQ + NQ;
};

constexpr int kK1Threads = 256, kK1Lanes = 32, kK1Elems = kK1Threads / kK1Lanes;

template <int DIM, int P, int NQ1, bool TR, bool EXPR>
__global__ __launch_bounds__(kK1Threads) void thermal_affine_element_kernel(BlockDev b, ThermalDev ph, AffineDev af,
                                                                             double *res) {
  using S = EK<DIM, P, NQ1, TR>;
  constexpr int M = S::M, N = S::N, NQ = S::NQ, NN = S::NN, REC = S::REC;
  static_assert(N <= kK1Lanes && NQ <= kK1Lanes && kGeoRec <= kK1Lanes, "one 32-lane group per element");
  (void)NN;
  __shared__ double tab[S::TAB];
  __shared__ double rec_all[kK1Elems * REC];
  __shared__ int s_offs[N], s_inv[N];
  const int tid = threadIdx.x, l = tid & (kK1Lanes - 1), slot = tid / kK1Lanes;
  const int e = b.e_begin + blockIdx.x * kK1Elems + slot;
  const bool active = (int)(blockIdx.x * kK1Elems + slot) < b.e_count;
  double *E = rec_all + slot * REC;
  const double *phi = tab, *dphi = tab + M * NQ1, *gw = tab + 2 * M * NQ1, *gp = gw + NQ1;
  const TimeDev &tm = ph.time;

  for (int i = tid; i < N; i += kK1Threads) { const int s = b.offsets[i]; s_offs[i] = s; s_inv[s] = i; }
  for (int i = tid; i < M * NQ1; i += kK1Threads) { tab[i] = af.phi1d[i]; tab[M * NQ1 + i] = af.dphi1d[i]; }
  for (int i = tid; i < NQ1; i += kK1Threads) { tab[2 * M * NQ1 + i] = af.gw1d[i]; tab[2 * M * NQ1 + NQ1 + i] = af.gp1d[i]; }
  // A. cached geometry; performGather + computeSoln*Seeded values (l = basis dof)
  if (active && l < kGeoRec) E[S::O_GEO + l] = af.geo[(size_t)e * kGeoRec + l];
  __syncthreads();
  if (active && l < N) {
    const int row = b.lids[(size_t)e * N + s_offs[l]];
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
      E[S::O_UD + l] = tm.alpha_t * cu + beta_t;
    }
    E[S::O_UE + l] = ue;
  }
  __syncthreads();

  // B. fields at the integration points (e, e_t, grad(e)) and the point-wise residual data (l = q)
  double my_rq = 0.0, my_F[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) my_F[a] = 0.0;
  if (active && l < NQ) {
    const int q = l;
    double gh[DIM], tv, gd[DIM], tt;
    eval_ref<DIM, P, NQ1>(E + S::O_UE, phi, dphi, q, gh, tv);
    double wq = 1.0, x[3] = {0, 0, 0}, xi[DIM];
    {
      int qq = q;
#pragma unroll
      for (int d = 0; d < DIM; ++d) { wq *= gw[qq % NQ1]; xi[d] = gp[qq % NQ1]; qq /= NQ1; }
    }
    if constexpr (TR) eval_ref<DIM, P, NQ1>(E + S::O_UD, phi, dphi, q, gd, tt); else tt = 0.0;
    (void)gd; (void)tv;
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double s = E[S::O_XC + r];
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += E[S::O_J + r * DIM + c] * xi[c];
      x[r] = s;
    }
    double G[DIM][DIM];
    {
      int k = 0;
#pragma unroll
      for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int c = a; c < DIM; ++c) { G[a][c] = E[S::O_G + k]; G[c][a] = E[S::O_G + k]; ++k; }
    }
    const double kap = ph.diff.amp, rc = ph.rho.amp * ph.cp.amp;  // element-wise constants on this path
#pragma unroll
    for (int a = 0; a < DIM; ++a) {  // F_a = w_q * kappa * detJ * sum_b (J^{-1}J^{-T})_ab * d_b T
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += G[a][c] * gh[c];
      my_F[a] = wq * kap * s;
      E[S::O_F + q * DIM + a] = my_F[a];
    }
    const double f = eval_func<DIM, EXPR>(ph.source, e, q, NQ, x);
    my_rq = (rc * tt - f) * E[S::O_DET] * wq;
    E[S::O_RQ + q] = my_rq;
  }
  __syncthreads();

  // C. residual rows by quadrature (l = LID slot): r_i = sum_q rq N_i + F . grad_ref N_i
  if (active && l < N) {
    const int row = b.lids[(size_t)e * N + l];
    if (!(b.fixed && b.fixed[row])) {  // fixed rows are skipped (assemblyManager.cpp:4075)
      const int ib = s_inv[l];
      const int i0 = ib % M, i1 = (ib / M) % M, i2 = ib / (M * M);
      double r = 0.0;
      if constexpr (DIM == 2) {
#pragma unroll
        for (int q1 = 0; q1 < NQ1; ++q1)
#pragma unroll
          for (int q0 = 0; q0 < NQ1; ++q0) {
            const int q = q1 * NQ1 + q0;
            const double a0 = phi[i0 * NQ1 + q0], d0 = dphi[i0 * NQ1 + q0];
            const double a1 = phi[i1 * NQ1 + q1], d1 = dphi[i1 * NQ1 + q1];
            r += E[S::O_RQ + q] * a0 * a1 + E[S::O_F + q * DIM] * d0 * a1 + E[S::O_F + q * DIM + 1] * a0 * d1;
          }
        (void)i2;
      } else {  // (a sum-factorised form of this loop nest was measured 15 % slower: longer dependent chains)
#pragma unroll
        for (int q2 = 0; q2 < NQ1; ++q2)
#pragma unroll
          for (int q1 = 0; q1 < NQ1; ++q1)
#pragma unroll
            for (int q0 = 0; q0 < NQ1; ++q0) {
              const int q = (q2 * NQ1 + q1) * NQ1 + q0;
              const double a0 = phi[i0 * NQ1 + q0], d0 = dphi[i0 * NQ1 + q0];
              const double a1 = phi[i1 * NQ1 + q1], d1 = dphi[i1 * NQ1 + q1];
              const double a2 = phi[i2 * NQ1 + q2], d2 = dphi[i2 * NQ1 + q2];
              r += E[S::O_RQ + q] * a0 * a1 * a2 + E[S::O_F + q * DIM] * d0 * a1 * a2 +
                   E[S::O_F + q * DIM + 1] * a0 * d1 * a2 + E[S::O_F + q * DIM + DIM - 1] * a0 * a1 * d2;
            }
      }
      atomicAdd(res + row, -r);  // the global vector receives -res.val() (assemblyManager.cpp:4094)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K2: row-owner Jacobian
// ---------------------------------------------------------------------------------------------

// Per (row block, touched element) record, block-major, static per mesh: the element's geometric
// factors and its ownership data in one 64-byte line, so that K2 needs no dependent loads:
//   erec[0..NSYM-1] = detJ*(J^{-1}J^{-T})_sym, erec[NSYM] = detJ, erec[7] = {ownership mask, first pair}
constexpr int kERec = 8;

template <int DIM>
__global__ __launch_bounds__(256) void build_erec_kernel(RowBlocksDev rb, const double *geo, double *erec, int total) {
  constexpr int NSYM = DIM * (DIM + 1) / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const double *g = geo + (size_t)rb.elems[i] * kGeoRec;
    double *o = erec + (size_t)i * kERec;
#pragma unroll
    for (int k = 0; k < NSYM; ++k) o[k] = g[k];
    o[NSYM] = g[kGeoDet];
    for (int k = NSYM + 1; k < 7; ++k) o[k] = 0.0;
    o[7] = __hiloint2double(rb.epbase[i], rb.emask[i]);
  }
}

template <int DIM, int N, int NT, typename SlotT>
__global__ __launch_bounds__(NT, (NT == 384 ? 6 : 1)) void row_owner_jacobian_kernel(
    RowBlocksDev rb, const double *__restrict__ erec, const double *__restrict__ khat,
    const uint4 *__restrict__ slot16, const uint16_t *__restrict__ pair_off16, RowOut out, double su, double st) {
  constexpr int NSYM = DIM * (DIM + 1) / 2, NN2 = N * N, NITER = (NN2 + NT - 1) / NT;
  static_assert(N <= 32, "ownership masks are 32 bits wide");
  const int tid = threadIdx.x, wave = tid >> 6;
  const int blk = rb.block_list ? rb.block_list[blockIdx.x] : blockIdx.x;
  const int t0 = rb.elem_ptr[blk], T = rb.elem_ptr[blk + 1] - t0;
  const int p0 = rb.pair_ptr[blk], NP = rb.pair_ptr[blk + 1] - p0;
  const int A = rb.acc_size[blk];

  extern __shared__ double smem[];
  double *acc = smem;                                                    // [lds_acc], lds_acc even
  SlotT *s_slot = reinterpret_cast<SlotT *>(acc + rb.lds_acc);           // [lds_pairs*N -> 16 B]
  uint16_t *s_pairoff = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(s_slot) +
                                                     ((size_t)rb.lds_pairs * N * sizeof(SlotT) + 15) / 16 * 16);

  // this lane's reference table entries, kept in registers for the whole block; the LID slots (si)
  // this WAVE covers, as a bit mask, for the wave-level skip
  double kh[NITER][NSYM + 1];
  unsigned my_bit[NITER], my_low[NITER];
  int my_sj[NITER];
  unsigned wave_si = 0u;
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int idx = tid + it * NT;
    const int si = (idx < NN2) ? idx / N : 0;
    my_bit[it] = (idx < NN2) ? (1u << si) : 0u;
    my_low[it] = (1u << si) - 1u;
    my_sj[it] = idx - si * N;
#pragma unroll
    for (int k = 0; k <= NSYM; ++k)  // the run-time scale factors are folded into the table entries once
      kh[it][k] = (idx < NN2) ? (k < NSYM ? su : st) * khat[k * NN2 + idx] : 0.0;
    const int lo = wave * 64 + it * NT, hi = min(lo + 63, NN2 - 1);
    if (lo < NN2) {
      const int a = lo / N, c = hi / N;
      wave_si |= (c >= 31 ? 0xffffffffu : ((1u << (c + 1)) - 1u)) & ~((1u << a) - 1u);
    }
  }

  // ---- block tables into LDS, zero the accumulators ----
  for (int p = tid; p < NP; p += NT) s_pairoff[p] = pair_off16[p0 + p];
  {
    const uint4 *src = slot16 + rb.slot_ptr[blk] / 16;
    uint4 *dst = reinterpret_cast<uint4 *>(s_slot);
    const int n16 = (int)((rb.slot_ptr[blk + 1] - rb.slot_ptr[blk]) / 16);
    for (int i = tid; i < n16; i += NT) dst[i] = src[i];
  }
  {
    double2 *a2 = reinterpret_cast<double2 *>(acc);
    for (int i = tid; i < (A + 1) / 2; i += NT) a2[i] = make_double2(0.0, 0.0);
  }
  __syncthreads();

  // ---- contributions: lane (si,sj) walks the block's elements; the 64-byte element record is one
  //      wave-uniform (scalar) load from the block-major table; a wave skips elements none of whose
  //      owned rows fall into its si range ----
  for (int t = 0; t < T; ++t) {
    const double *E = erec + (size_t)(t0 + t) * kERec;
    const double mp = E[7];
    const unsigned mask = (unsigned)__double2loint(mp);
    if ((mask & wave_si) == 0u) continue;
    const int pb = __double2hiint(mp);
    double g[NSYM + 1];
#pragma unroll
    for (int k = 0; k <= NSYM; ++k) g[k] = E[k];
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      if (mask & my_bit[it]) {
        const int p = pb + __popc(mask & my_low[it]);
        double v = g[NSYM] * kh[it][NSYM];
#pragma unroll
        for (int k = 0; k < NSYM; ++k) v += g[k] * kh[it][k];
        atomicAdd(&acc[(int)s_pairoff[p] + (int)s_slot[p * N + my_sj[it]]], v);
      }
    }
  }
  __syncthreads();

  // ---- stream the finished rows to HBM, one contiguous run of rows at a time ----
  for (int s = rb.seg_ptr[blk]; s < rb.seg_ptr[blk + 1]; ++s) {
    int len = rb.seg_len[s];
    if (len < 0) {  // run of fixed rows: zeros when storing, untouched when accumulating
      if (!out.overwrite) continue;
      len = -len;
    }
    double *dst = out.vals + rb.seg_base[s];
    const double *src = acc + rb.seg_acc[s];
    if (out.overwrite) {
      for (int k = tid; k < len; k += NT) dst[k] = src[k];
    } else {
      for (int k = tid; k < len; k += NT) dst[k] += src[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K2, persistent + software-pipelined form
//   - workgroups walk the row blocks blockIdx.x, blockIdx.x + gridDim.x, ...; the reference-table
//     registers are set up once per workgroup;
//   - the tables of block i+1 are prefetched into registers while block i is accumulated;
//   - the stores of block i are only ISSUED before the workgroup moves on, so they drain behind block
//     i+1's work (a workgroup that ends sits on its LDS until its stores are acknowledged: s_endpgm
//     waits for outstanding memory operations);
//   - barriers wait for LDS traffic only: gfx950 counts loads and stores in one in-order vmcnt, so a
//     __syncthreads() (which implies vmcnt(0)) would stall on the stores in flight;
//   - element records live in LDS (a scalar load per element inside the loop costs a memory round
//     trip each: measured 1.6x slower).
// DBG != 0 only in profiling launches (env MHA_K2_ABLATE): 1 no contributions, 2 no stores, 3 neither.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxBlockElems = 27, kMaxBlockPairs = 256;  // caps of a row block (host: prepareRowOwner)


template <int DIM, int N, int NT, typename SlotT, int DBG>
__global__ __launch_bounds__(NT, (NT == 384 ? 5 : (NT == 256 ? 4 : 1))) void row_owner_jacobian_persistent_kernel(
    RowBlocksDev rb, const double *__restrict__ erec, const double *__restrict__ khat,
    const uint4 *__restrict__ slot16, const uint16_t *__restrict__ pair_off16, const int *__restrict__ slot_pair,
    RowOut out, double su, double st) {
  constexpr int NSYM = DIM * (DIM + 1) / 2, NN2 = N * N;
  constexpr int NW = NT / 64;
  constexpr int NR = (N + 1) / 2;             // slot ranges: two LID slots (si) per wave-instruction
  constexpr int NITER = (NR + NW - 1) / NW;   // ranges (register sets) per wave
  static_assert(N <= 32, "one LID slot per 32-lane half; ownership masks are 32 bits wide");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nwork = rb.block_list ? rb.list_len : rb.num_blocks;

  extern __shared__ double smem[];
  double *acc = smem;                                                    // [lds_acc], lds_acc even
  double *s_erec = acc + rb.lds_acc;                                     // [lds_elems][kERec]
  SlotT *s_slot = reinterpret_cast<SlotT *>(s_erec + (size_t)rb.lds_elems * kERec);  // [lds_pairs*N -> 16 B]
  uint16_t *s_pairoff = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(s_slot) +
                                                     ((size_t)rb.lds_pairs * N * sizeof(SlotT) + 15) / 16 * 16);

  // Lane layout: each 32-lane half of a wave works on ONE LID slot si at a time, its lanes are the
  // columns sj.  Range r = (si_a, si_b) puts two slots side by side in one wave-instruction; the host
  // pairs slots that are usually owned together (slot_pair), so a visit is either skipped by the whole
  // wave or keeps most lanes busy.  Wave w owns ranges w, w+NW, ...; the reference-table entries of
  // "its" (si,sj) -- pre-scaled by the run-time factors -- stay in registers for the life of the workgroup.
  double kh[NITER][NSYM + 1];
  unsigned my_bit[NITER], my_low[NITER];
  const int my_sj = min(lane & 31, N - 1);
  unsigned wave_si = 0u;
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int r = wave + it * NW;
    const int si = (r < NR) ? slot_pair[2 * r + (lane >> 5)] : -1;  // -1: no slot (odd N, or past the last range)
    const bool ok = si >= 0 && (lane & 31) < N;
    my_bit[it] = ok ? (1u << si) : 0u;
    my_low[it] = ok ? (1u << si) - 1u : 0u;
#pragma unroll
    for (int k = 0; k <= NSYM; ++k) kh[it][k] = (k < NSYM ? su : st) * khat[k * NN2 + max(si, 0) * N + my_sj];
    if (r < NR) {
      const int sa = slot_pair[2 * r], sb = slot_pair[2 * r + 1];
      wave_si |= (sa >= 0 ? 1u << sa : 0u) | (sb >= 0 ? 1u << sb : 0u);
    }
  }

  // per-thread pieces of one block's tables (sized for the caps the host enforces)
  constexpr int EREC_PT = (kMaxBlockElems * kERec + NT - 1) / NT;
  constexpr int PAIR_PT = (kMaxBlockPairs + NT - 1) / NT;
  constexpr int SLOT16_PT = ((kMaxBlockPairs * N * (int)sizeof(SlotT) + 15) / 16 + NT - 1) / NT;
  struct Prefetch {
    double erec[EREC_PT];
    uint4 slot[SLOT16_PT];
    uint16_t pairoff[PAIR_PT];
    int sg_acc, sg_base, sg_len;
  };
  auto prefetch = [&](int work, Prefetch &pf) {  // unconditional loads, clamped indices: all in flight together
    const int blk = rb.block_list ? rb.block_list[work] : work;
    const int t0 = rb.elem_ptr[blk], T = rb.elem_ptr[blk + 1] - t0;
    const int p0 = rb.pair_ptr[blk], NP = rb.pair_ptr[blk + 1] - p0;
    const int g0 = rb.seg_ptr[blk], NS = rb.seg_ptr[blk + 1] - g0;
#pragma unroll
    for (int j = 0; j < EREC_PT; ++j) pf.erec[j] = erec[(size_t)t0 * kERec + min(tid + j * NT, T * kERec - 1)];
#pragma unroll
    for (int j = 0; j < PAIR_PT; ++j) pf.pairoff[j] = pair_off16[p0 + min(tid + j * NT, max(NP - 1, 0))];
    const uint4 *src = slot16 + rb.slot_ptr[blk] / 16;
    const int n16 = (int)((rb.slot_ptr[blk + 1] - rb.slot_ptr[blk]) / 16);
#pragma unroll
    for (int j = 0; j < SLOT16_PT; ++j) pf.slot[j] = src[min(tid + j * NT, max(n16 - 1, 0))];
    const int sidx = min(wave + NW * lane, NS - 1);  // store runs of this wave, one per lane
    pf.sg_acc = rb.seg_acc[g0 + sidx];
    pf.sg_base = rb.seg_base[g0 + sidx];
    pf.sg_len = rb.seg_len[g0 + sidx];
  };
  auto tables_to_lds = [&](const Prefetch &pf) {
#pragma unroll
    for (int j = 0; j < EREC_PT; ++j)
      if (tid + j * NT < rb.lds_elems * kERec) s_erec[tid + j * NT] = pf.erec[j];
#pragma unroll
    for (int j = 0; j < PAIR_PT; ++j)
      if (tid + j * NT < rb.lds_pairs) s_pairoff[tid + j * NT] = pf.pairoff[j];
    uint4 *dst = reinterpret_cast<uint4 *>(s_slot);
    const int cap16 = (int)(((size_t)rb.lds_pairs * N * sizeof(SlotT) + 15) / 16);
#pragma unroll
    for (int j = 0; j < SLOT16_PT; ++j)
      if (tid + j * NT < cap16) dst[tid + j * NT] = pf.slot[j];
  };
  auto lds_barrier = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  auto zero_acc = [&]() {
    double2 *a2 = reinterpret_cast<double2 *>(acc);
    for (int i = tid; i < rb.lds_acc / 2; i += NT) a2[i] = make_double2(0.0, 0.0);
  };

  if ((int)blockIdx.x >= nwork) return;
  Prefetch cur;
  prefetch(blockIdx.x, cur);
  tables_to_lds(cur);
  zero_acc();
  lds_barrier();

  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int blk = rb.block_list ? rb.block_list[work] : work;
    const int T = rb.elem_ptr[blk + 1] - rb.elem_ptr[blk];
    const int NS = rb.seg_ptr[blk + 1] - rb.seg_ptr[blk];
    const int sg_acc = cur.sg_acc, sg_base = cur.sg_base, sg_len = cur.sg_len;
    const int next = work + gridDim.x;
    Prefetch nxt = cur;
    if (next < nwork) prefetch(next, nxt);  // lands while this block is accumulated

    // contributions: lane (si,sj) walks the block's elements; the 64-byte element record is read from LDS at a
    // wave-uniform address; a wave skips elements none of whose owned rows fall into its si range
    // The LDS array is the busiest unit of this kernel, and wave-uniform reads cost as much as any other:
    // ownership data of all elements are fetched once per wave (lane t <- element t) and consulted through
    // v_readlane; the geometric factors are read with 16-byte accesses, only for elements the wave works on.
    const double mp_l = s_erec[min(lane, T - 1) * kERec + 7];
    const int mk_l = __double2loint(mp_l), pb_l = __double2hiint(mp_l);
    for (int t = 0; t < ((DBG & 1) ? 0 : T); ++t) {
      const unsigned mask = (unsigned)__builtin_amdgcn_readlane(mk_l, t);
      if ((mask & wave_si) == 0u) continue;
      const int pb = __builtin_amdgcn_readlane(pb_l, t);
      // (fetching the factors through v_readlane instead was measured 20 % slower: VALU is as loaded as LDS)
      const double2 *E2 = reinterpret_cast<const double2 *>(s_erec + t * kERec);
      const double2 e0 = E2[0], e1 = E2[1], e2 = E2[2];
      const double g[7] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, s_erec[t * kERec + 6]};
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        if (mask & my_bit[it]) {
          const int p = pb + __popc(mask & my_low[it]);
          double v = g[NSYM] * kh[it][NSYM];
#pragma unroll
          for (int k = 0; k < NSYM; ++k) v += g[k] * kh[it][k];
          atomicAdd(&acc[(int)s_pairoff[p] + (int)s_slot[p * N + my_sj]], v);
        }
      }
    }
    lds_barrier();  // accumulators complete; tables no longer needed

    if (next < nwork) tables_to_lds(nxt);  // the prefetch has had the whole accumulation phase to land

    // stream the finished rows: each wave takes whole contiguous runs, reads four 512-byte pieces out of the
    // accumulator, issues their stores (left in flight) and zeroes what it has just read -- no other wave
    // touches these runs until the next block's accumulation, so one barrier per block suffices here
    for (int j = 0; wave + NW * j < ((DBG & 2) ? 0 : NS); ++j) {
      int len = __builtin_amdgcn_readlane(sg_len, j);
      const bool fixed_run = len < 0;  // run of fixed rows: zeros when storing, untouched when accumulating
      if (fixed_run) len = -len;
      double *dst = out.vals + __builtin_amdgcn_readlane(sg_base, j);
      double *src = acc + __builtin_amdgcn_readlane(sg_acc, j);
      if (fixed_run && !out.overwrite) continue;
      for (int k0 = 0; k0 < len; k0 += 256) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * 64 + lane;
          v[u] = (k < len) ? src[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * 64 + lane;
          if (k < len) {
            if (out.overwrite) dst[k] = v[u]; else dst[k] += v[u];
            src[k] = 0.0;
          }
        }
      }
    }
    lds_barrier();  // accumulator read out and zeroed, next block's tables in place
    cur = nxt;
  }
}

size_t k2_lds_bytes(const RowBlocksDev &rb, int n, int slot_bytes) {
  const size_t acc = ((size_t)rb.lds_acc + 1) / 2 * 2 * sizeof(double);
  const size_t slots = ((size_t)rb.lds_pairs * n * slot_bytes + 15) / 16 * 16;
  return acc + slots + ((size_t)rb.lds_pairs * 2 + 15) / 16 * 16;
}

template <int DIM, int N, int NT, typename SlotT>
void launch_k2_t(RowBlocksDev rb, const AffineDev &af, const RowOut &out, double su, double st, hipStream_t stream) {
  const int grid = rb.block_list ? rb.list_len : rb.num_blocks;
  if (grid <= 0) return;
  rb.lds_acc = (rb.lds_acc + 1) / 2 * 2;
  const size_t lds = k2_lds_bytes(rb, N, sizeof(SlotT));
  MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "row-owner kernel needs " << lds << " B of LDS (> 160 KiB)");
  MHA_REQUIRE(rb.lds_acc < 65536, MHA_ERR_INVALID, "row-owner kernel: accumulator offsets must fit 16 bits");
  int mode = 1, per_cu = 3, dbg = 0;
  if (const char *e = std::getenv("MHA_K2_MODE")) mode = std::atoi(e);            // tuning / profiling knobs
  if (const char *e = std::getenv("MHA_K2_WGS_PER_CU")) per_cu = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("MHA_K2_ABLATE")) dbg = std::atoi(e);
  if (mode == 1) {
    MHA_REQUIRE(rb.lds_elems <= kMaxBlockElems && rb.lds_pairs <= kMaxBlockPairs && rb.lds_segs <= 64 * (NT / 64),
                MHA_ERR_INVALID, "persistent row-owner kernel: row block exceeds its caps");
    const size_t lds_p = lds + (size_t)rb.lds_elems * kERec * sizeof(double);
    auto go = [&](auto kern) {
      MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
      hipLaunchKernelGGL(kern, dim3(std::min(grid, 256 * per_cu)), dim3(NT), lds_p, stream, rb, af.erec, af.khat,
                         static_cast<const uint4 *>(af.slot), af.pair_off16, af.slot_pair, out, su, st);
    };
    switch (dbg) {
      case 1: go(row_owner_jacobian_persistent_kernel<DIM, N, NT, SlotT, 1>); break;
      case 2: go(row_owner_jacobian_persistent_kernel<DIM, N, NT, SlotT, 2>); break;
      case 3: go(row_owner_jacobian_persistent_kernel<DIM, N, NT, SlotT, 3>); break;
      default: go(row_owner_jacobian_persistent_kernel<DIM, N, NT, SlotT, 0>); break;
    }
  } else {
    auto kern = row_owner_jacobian_kernel<DIM, N, NT, SlotT>;
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, rb, af.erec, af.khat,
                       static_cast<const uint4 *>(af.slot), af.pair_off16, out, su, st);
  }
  MHA_HIP(hipGetLastError());
}

template <int DIM, int N, int NT>
void launch_k2(const RowBlocksDev &rb, const AffineDev &af, const RowOut &out, double su, double st,
               hipStream_t stream) {
  if (af.slot_bytes == 1) launch_k2_t<DIM, N, NT, uint8_t>(rb, af, out, su, st, stream);
  else launch_k2_t<DIM, N, NT, uint16_t>(rb, af, out, su, st, stream);
}

template <int DIM, int P, int NQ1>
void launch_k1(const BlockDev &b, const ThermalDev &ph, const AffineDev &af, double *res, hipStream_t stream) {
  if (b.e_count <= 0) return;
  const bool tr = ph.time.transient != 0;
  const int grid = (b.e_count + kK1Elems - 1) / kK1Elems;
  // MHA_K1_LDS_PAD: unused dynamic LDS per workgroup -- throttles how many K1 workgroups share a CU with K2 (experiment)
  static const size_t pad = [] { const char *m = std::getenv("MHA_K1_LDS_PAD"); return m ? (size_t)std::atoi(m) : (size_t)0; }();
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(kK1Threads), pad, stream, b, ph, af, res); };
  if (has_expression(ph.source)) {  // the only named function K1 evaluates per point (coefficients are constants here)
    if (tr) go(thermal_affine_element_kernel<DIM, P, NQ1, true, true>);
    else go(thermal_affine_element_kernel<DIM, P, NQ1, false, true>);
  } else {
    if (tr) go(thermal_affine_element_kernel<DIM, P, NQ1, true, false>);
    else go(thermal_affine_element_kernel<DIM, P, NQ1, false, false>);
  }
  MHA_HIP(hipGetLastError());
}

inline int grid_for(size_t total, int block) {
  const size_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

}  // namespace

void launch_classify_affine(const BlockDev &b, uint8_t *flags, double tol, hipStream_t stream) {
  if (b.dim == 2) hipLaunchKernelGGL(classify_affine_kernel<2>, dim3(grid_for(b.nelem, 256)), dim3(256), 0, stream, b, flags, tol);
  else hipLaunchKernelGGL(classify_affine_kernel<3>, dim3(grid_for(b.nelem, 256)), dim3(256), 0, stream, b, flags, tol);
  MHA_HIP(hipGetLastError());
}

void launch_affine_geometry(const BlockDev &b, double *geo, hipStream_t stream) {
  if (b.dim == 2) hipLaunchKernelGGL(affine_geometry_kernel<2>, dim3(grid_for(b.nelem, 256)), dim3(256), 0, stream, b, geo);
  else hipLaunchKernelGGL(affine_geometry_kernel<3>, dim3(grid_for(b.nelem, 256)), dim3(256), 0, stream, b, geo);
  MHA_HIP(hipGetLastError());
}

void launch_build_erec(int dim, const RowBlocksDev &rb, const double *geo, double *erec, int total,
                       hipStream_t stream) {
  if (total <= 0) return;
  if (dim == 2) hipLaunchKernelGGL(build_erec_kernel<2>, dim3(grid_for(total, 256)), dim3(256), 0, stream, rb, geo, erec, total);
  else hipLaunchKernelGGL(build_erec_kernel<3>, dim3(grid_for(total, 256)), dim3(256), 0, stream, rb, geo, erec, total);
  MHA_HIP(hipGetLastError());
}

void launch_build_block_slots(const BlockDev &b, const RowBlocksDev &rb, void *bslot, int slot_bytes,
                              hipStream_t stream) {
  if (rb.num_blocks <= 0) return;
  if (slot_bytes == 1)
    hipLaunchKernelGGL(build_block_slots_kernel<uint8_t>, dim3(rb.num_blocks), dim3(256), 0, stream, b, rb,
                       static_cast<uint8_t *>(bslot));
  else
    hipLaunchKernelGGL(build_block_slots_kernel<uint16_t>, dim3(rb.num_blocks), dim3(256), 0, stream, b, rb,
                       static_cast<uint16_t *>(bslot));
  MHA_HIP(hipGetLastError());
}

bool thermal_row_owner_supported(int dim, int order, int nq1) {
  return (dim == 2 && ((order == 1 && nq1 == 2) || (order == 2 && nq1 == 3) || (order == 4 && nq1 == 5))) ||
         (dim == 3 && ((order == 1 && nq1 == 2) || (order == 2 && nq1 == 3)));
}

size_t row_owner_jacobian_lds(const RowBlocksDev &rb, int n, int slot_bytes) { return k2_lds_bytes(rb, n, slot_bytes); }

void launch_thermal_affine_element(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                                   const AffineDev &af, double *res, hipStream_t stream) {
  if (dim == 2 && order == 1 && nq1 == 2) return launch_k1<2, 1, 2>(b, ph, af, res, stream);
  if (dim == 2 && order == 2 && nq1 == 3) return launch_k1<2, 2, 3>(b, ph, af, res, stream);
  if (dim == 2 && order == 4 && nq1 == 5) return launch_k1<2, 4, 5>(b, ph, af, res, stream);
  if (dim == 3 && order == 1 && nq1 == 2) return launch_k1<3, 1, 2>(b, ph, af, res, stream);
  if (dim == 3 && order == 2 && nq1 == 3) return launch_k1<3, 2, 3>(b, ph, af, res, stream);
  MHA_REQUIRE(false, MHA_ERR_INVALID, "affine element kernel: unsupported (dim,order,points/dir)");
}

void launch_row_owner_jacobian(int dim, int n, const RowBlocksDev &rb, const AffineDev &af, const RowOut &out,
                               double scale_u, double scale_t, hipStream_t stream) {
  if (dim == 2 && n == 4) return launch_k2<2, 4, 64>(rb, af, out, scale_u, scale_t, stream);
  if (dim == 2 && n == 9) return launch_k2<2, 9, 128>(rb, af, out, scale_u, scale_t, stream);
  if (dim == 2 && n == 25) return launch_k2<2, 25, 320>(rb, af, out, scale_u, scale_t, stream);
  if (dim == 3 && n == 8) return launch_k2<3, 8, 64>(rb, af, out, scale_u, scale_t, stream);
  if (dim == 3 && n == 27) {
    const char *e = std::getenv("MHA_K2_NT");  // tuning knob
    if (e && std::atoi(e) == 384) return launch_k2<3, 27, 384>(rb, af, out, scale_u, scale_t, stream);
    if (e && std::atoi(e) == 256) return launch_k2<3, 27, 256>(rb, af, out, scale_u, scale_t, stream);
    return launch_k2<3, 27, 192>(rb, af, out, scale_u, scale_t, stream);
  }
  MHA_REQUIRE(false, MHA_ERR_INVALID, "row-owner Jacobian kernel: unsupported (dim, dofs/elem)");
}

}  // namespace mha
