// thermal_row_owner.hip -- fused, row-owner thermal volume assembly for gfx950 (affine elements).
//
// One workgroup owns a block of CRS rows (row_blocks.hpp) and produces them completely: it visits
// every element incident to its rows, forms that element's contributions on chip and accumulates
// them in LDS (ds_add_f64), then streams whole CRS rows and residual entries to HBM with plain
// coalesced stores.  No global atomics, no dense element matrices in HBM, no column search.
//
// Covers, for affine (parallelepiped) elements with element-wise constant coefficients, the same
// reference routines as thermal_element.hip (gather, seeding, basis/quadrature, field evaluation,
// thermal::volumeResidual, scatter with fixed-row skip; see the citations there).  For such elements
// the cell Jacobian is constant, so
//   res(e,i).dx(j) = alpha_u * sum_{a<=b} Gs_ab * Khat_ab[i][j] + alpha_t * rho*cp*detJ * Mhat[i][j]
// with Gs = kappa*detJ*J^{-1}J^{-T} and the reference tables Khat/Mhat integrated once at setup with
// the block's cubature.  Every lane keeps the table entries of "its" (i,j) pair in registers and
// walks the block's elements.  The residual is integrated by quadrature exactly as in the reference
// (it needs the source at the physical integration points).
#include <hip/hip_runtime.h>

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
    // signs of vertex v in shards order
    for (int r = 0; r < DIM; ++r) {
      for (int mask = 1; mask < NN; ++mask) {  // subset of directions entering the monomial
        double c = 0.0;
        for (int v = 0; v < NN; ++v) {
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

// ---------------------------------------------------------------------------------------------
// the fused kernel
// ---------------------------------------------------------------------------------------------

template <int DIM, int P, int NQ1, bool TR>
struct RO {
  static constexpr int M = P + 1;
  static constexpr int N = cpow(M, DIM);
  static constexpr int NQ = cpow(NQ1, DIM);
  static constexpr int NN = 1 << DIM;
  static constexpr int NSYM = DIM * (DIM + 1) / 2;
  static constexpr int TAB = 2 * M * NQ1 + 2 * NQ1;  // phi, dphi, gauss wts, gauss pts
  // per-element LDS record (doubles)
  static constexpr int O_UE = 0;                 // u_eval  [N]   (basis order)
  static constexpr int O_UD = O_UE + N;          // u_dot   [N]   (transient runs only)
  static constexpr int O_G = O_UD + (TR ? N : 0);  // Gs    [NSYM]
  static constexpr int O_M = O_G + NSYM;         // rho*cp*detJ
  static constexpr int O_DET = O_M + 1;          // detJ
  static constexpr int O_J = O_DET + 1;          // J       [DIM*DIM]
  static constexpr int O_XC = O_J + DIM * DIM;   // centroid [DIM]
  static constexpr int O_F = O_XC + DIM;         // w_q * Gs * grad_ref T(q)   [NQ][DIM]
  static constexpr int O_RQ = O_F + NQ * DIM;    // (rho cp T_t - f) detJ w_q  [NQ]
  static constexpr int EL = O_RQ + NQ;
};

// reference-space gradient and value of sum_j c[j] N_j at integration point q (tensor basis)
template <int DIM, int P, int NQ1>
__device__ __forceinline__ void eval_ref(const double *c, const double *phi, const double *dphi, int q,
                                         double *grad, double &val) {
  constexpr int M = P + 1;
  const int q0 = q % NQ1, q1 = (q / NQ1) % NQ1, q2 = q / (NQ1 * NQ1);
  if constexpr (DIM == 2) {
    double g0 = 0, g1 = 0, v = 0;
#pragma unroll
    for (int b1 = 0; b1 < M; ++b1) {
      double s = 0, sd = 0;
#pragma unroll
      for (int a = 0; a < M; ++a) {
        const double u = c[b1 * M + a];
        s += u * phi[a * NQ1 + q0];
        sd += u * dphi[a * NQ1 + q0];
      }
      g0 += sd * phi[b1 * NQ1 + q1];
      g1 += s * dphi[b1 * NQ1 + q1];
      v += s * phi[b1 * NQ1 + q1];
    }
    grad[0] = g0; grad[1] = g1; val = v;
  } else {
    double g0 = 0, g1 = 0, g2 = 0, v = 0;
#pragma unroll
    for (int c2 = 0; c2 < M; ++c2) {
      double t = 0, tx = 0, ty = 0;
#pragma unroll
      for (int b1 = 0; b1 < M; ++b1) {
        double s = 0, sd = 0;
#pragma unroll
        for (int a = 0; a < M; ++a) {
          const double u = c[(c2 * M + b1) * M + a];
          s += u * phi[a * NQ1 + q0];
          sd += u * dphi[a * NQ1 + q0];
        }
        t += s * phi[b1 * NQ1 + q1];
        tx += sd * phi[b1 * NQ1 + q1];
        ty += s * dphi[b1 * NQ1 + q1];
      }
      g0 += tx * phi[c2 * NQ1 + q2];
      g1 += ty * phi[c2 * NQ1 + q2];
      g2 += t * dphi[c2 * NQ1 + q2];
      v += t * phi[c2 * NQ1 + q2];
    }
    grad[0] = g0; grad[1] = g1; grad[DIM - 1] = g2; val = v;
  }
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Block-major slot table: bslot[pair][sj] = position of column LIDs[e][sj] inside the CRS row of the
// pair (one workgroup per row block; read once, fully coalesced, by the fused kernel).
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
    bslot[(size_t)p0 * n + item] = (SlotT)(c < 0 ? 0 : c - lo);
  }
}

template <int DIM, int P, int NQ1, int NT, typename SlotT, bool TR>
__global__ __launch_bounds__(NT) void thermal_row_owner_affine_kernel(BlockDev b, ThermalDev ph, RowBlocksDev rb,
                                                                       AffineDev af, RowOut out) {
  using S = RO<DIM, P, NQ1, TR>;
  constexpr int M = S::M, N = S::N, NQ = S::NQ, NN = S::NN, NSYM = S::NSYM, EL = S::EL;
  constexpr int NN2 = N * N, NITER = (NN2 + NT - 1) / NT;
  static_assert(N <= 32, "ownership masks are 32 bits wide");
  const int tid = threadIdx.x, lane = tid & 63;
  const int blk = rb.block_list ? rb.block_list[blockIdx.x] : blockIdx.x;
  const int r0 = rb.row_ptr[blk], R = rb.row_ptr[blk + 1] - r0;
  const int t0 = rb.elem_ptr[blk], T = rb.elem_ptr[blk + 1] - t0;
  const int p0 = rb.pair_ptr[blk], NP = rb.pair_ptr[blk + 1] - p0;
  const int A = rb.acc_size[blk];
  const TimeDev &tm = ph.time;

  extern __shared__ double smem[];
  double *acc = smem;                                  // [lds_acc]
  double *racc = acc + rb.lds_acc;                     // [lds_rows]
  double *tab = racc + rb.lds_rows;                    // 1-D tables
  double *el = tab + S::TAB;                           // [lds_elems][EL]
  int *s_rows = reinterpret_cast<int *>(el + (size_t)rb.lds_elems * EL);  // [lds_rows]
  int *s_off = s_rows + rb.lds_rows;                   // [lds_rows] accumulator offset
  int *s_base = s_off + rb.lds_rows;                   // [lds_rows] rowptr of the row
  int *s_len = s_base + rb.lds_rows;                   // [lds_rows] row length; < 0 marks a fixed row
  int *s_elem = s_len + rb.lds_rows;                   // [lds_elems]
  int *s_mask = s_elem + rb.lds_elems;                 // [lds_elems] owned slots of the element (bit si)
  int *s_pbase = s_mask + rb.lds_elems;                // [lds_elems] first pair of the element
  int *s_offs = s_pbase + rb.lds_elems;                // [N] offsets: basis dof -> LID slot
  int *s_inv = s_offs + N;                             // [N] LID slot -> basis dof
  int *s_pair = s_inv + N;                             // [lds_pairs] packed (row, elem, slot)
  int *s_pairoff = s_pair + rb.lds_pairs;              // [lds_pairs] accumulator offset of the pair's row
  SlotT *s_slot = reinterpret_cast<SlotT *>(s_pairoff + rb.lds_pairs);  // [lds_pairs][N]
  const double *phi = tab, *dphi = tab + M * NQ1, *gw = tab + 2 * M * NQ1, *gp = gw + NQ1;

  // this lane's reference table entries, kept in registers for the whole block
  double kh[NITER][NSYM + 1];
  int my_si[NITER];
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int idx = tid + it * NT;
    my_si[it] = (idx < NN2) ? idx / N : -1;
#pragma unroll
    for (int k = 0; k <= NSYM; ++k) kh[it][k] = (idx < NN2) ? af.khat[k * NN2 + idx] : 0.0;
  }

  // ---- P0a: block lists, slot table, 1-D tables, zero accumulators ----
  for (int o = tid; o < R; o += NT) {
    const int g = rb.rows[r0 + o];
    const int lo = b.rowptr[g];
    const bool fx = b.fixed && b.fixed[g];
    s_rows[o] = g;
    s_off[o] = rb.row_off[r0 + o];
    s_base[o] = lo;
    s_len[o] = fx ? -(b.rowptr[g + 1] - lo) - 1 : b.rowptr[g + 1] - lo;
    racc[o] = 0.0;
  }
  for (int t = tid; t < T; t += NT) { s_elem[t] = rb.elems[t0 + t]; s_mask[t] = 0; }
  for (int p = tid; p < NP; p += NT) s_pair[p] = (int)rb.pairs[p0 + p];
  {
    const SlotT *src = static_cast<const SlotT *>(af.slot) + (size_t)p0 * N;
    for (int i = tid; i < NP * N; i += NT) s_slot[i] = src[i];
  }
  for (int i = tid; i < N; i += NT) { const int s = b.offsets[i]; s_offs[i] = s; s_inv[s] = i; }
  for (int i = tid; i < M * NQ1; i += NT) { tab[i] = af.phi1d[i]; tab[M * NQ1 + i] = af.dphi1d[i]; }
  for (int i = tid; i < NQ1; i += NT) { tab[2 * M * NQ1 + i] = af.gw1d[i]; tab[2 * M * NQ1 + NQ1 + i] = af.gp1d[i]; }
  for (int i = tid; i < A; i += NT) acc[i] = 0.0;
  __syncthreads();

  // ---- P0b: pair bookkeeping, gather + seeding values, element geometry ----
  for (int p = tid; p < NP; p += NT) {
    const int pk = s_pair[p];
    s_pairoff[p] = s_off[(pk >> 16) & 0xffff];
    atomicOr(&s_mask[(pk >> 8) & 0xff], 1 << (pk & 0xff));
  }
  for (int item = tid; item < ((out.debug_skip & 16) ? 0 : T * N); item += NT) {
    const int t = item / N, k = item - t * N;
    const int e = s_elem[t];
    // performGather + computeSoln*Seeded values (k = basis dof)
    const int row = b.lids[(size_t)e * N + s_offs[k]];
    const double cu = tm.u[row];
    double ue = cu, ud = 0.0;
    if constexpr (TR) {
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
    el[(size_t)t * EL + S::O_UE + k] = ue;
    if constexpr (TR) el[(size_t)t * EL + S::O_UD + k] = ud;
    (void)ud;
  }
  for (int t = tid; t < T; t += NT) {
    const int e = s_elem[t];
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
    double *E = el + (size_t)t * EL;
    const double kap = ph.diff.amp, rc = ph.rho.amp * ph.cp.amp;  // element-wise constants on this path
    int k = 0;
#pragma unroll
    for (int a = 0; a < DIM; ++a)
#pragma unroll
      for (int c = a; c < DIM; ++c) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) s += Ji[a * DIM + d] * Ji[c * DIM + d];
        E[S::O_G + k++] = kap * det * s;
      }
    E[S::O_M] = rc * det;
    E[S::O_DET] = det;
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) E[S::O_J + i] = J[i];
#pragma unroll
    for (int d = 0; d < DIM; ++d) E[S::O_XC + d] = xc[d];
  }
  __syncthreads();

  // ---- P1: fields at the integration points (e, e_t, grad(e)) and the point-wise residual data ----
  for (int t = tid; t < T; t += NT) {  // first pair of every element (pairs are sorted by element)
    int pb = 0;
    for (int k = 0; k < t; ++k) pb += __popc((unsigned)s_mask[k]);
    s_pbase[t] = pb;
  }
  for (int item = tid; item < ((out.debug_skip & 1) ? 0 : T * NQ); item += NT) {
    const int t = item / NQ, q = item - t * NQ;
    double *E = el + (size_t)t * EL;
    double gh[DIM], tv, gd[DIM], tt;
    eval_ref<DIM, P, NQ1>(E + S::O_UE, phi, dphi, q, gh, tv);
    double wq = 1.0, x[3] = {0, 0, 0}, xi[DIM];
    {
      int qq = q;
#pragma unroll
      for (int d = 0; d < DIM; ++d) { wq *= gw[qq % NQ1]; xi[d] = gp[qq % NQ1]; qq /= NQ1; }
    }
    if constexpr (TR) eval_ref<DIM, P, NQ1>(E + S::O_UD, phi, dphi, q, gd, tt); else tt = 0.0;
    (void)gd;
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double s = E[S::O_XC + r];
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += E[S::O_J + r * DIM + c] * xi[c];
      x[r] = s;
    }
    // F_a = w_q * sum_b Gs_ab * gh_b
    double G[DIM][DIM];
    {
      int k = 0;
#pragma unroll
      for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int c = a; c < DIM; ++c) { G[a][c] = E[S::O_G + k]; G[c][a] = E[S::O_G + k]; ++k; }
    }
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s += G[a][c] * gh[c];
      E[S::O_F + q * DIM + a] = wq * s;
    }
    const double f = eval_func<DIM>(ph.source, s_elem[t], q, NQ, x);
    E[S::O_RQ + q] = (E[S::O_M] * tt - f * E[S::O_DET]) * wq;
    (void)tv;
  }
  __syncthreads();

  // ---- P3: Jacobian entries: lane (si,sj) walks the block's elements.  Element data (scaled Gs, mass
  //      factor, ownership mask, first pair) sit one element per lane and reach every lane through
  //      v_readlane, so an element nobody here owns rows of costs no LDS traffic at all. ----
  if (out.compute_jacobian && !(out.debug_skip & 2)) {
    double my_g[NSYM + 1];
    int my_mask = 0, my_pb = 0;
    {
      const bool have = lane < T;
      const double *E = el + (size_t)(have ? lane : 0) * EL;
#pragma unroll
      for (int k = 0; k < NSYM; ++k) my_g[k] = have ? tm.alpha_u * E[S::O_G + k] : 0.0;
      my_g[NSYM] = have ? tm.alpha_t * E[S::O_M] : 0.0;
      if (have) { my_mask = s_mask[lane]; my_pb = s_pbase[lane]; }
    }
    for (int t = 0; t < T; ++t) {
      const unsigned mask = (unsigned)__builtin_amdgcn_readlane(my_mask, t);
      if (mask == 0u) continue;
      const int pb = __builtin_amdgcn_readlane(my_pb, t);
      double g[NSYM + 1];
#pragma unroll
      for (int k = 0; k <= NSYM; ++k) g[k] = readlane_f64(my_g[k], t);
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int si = my_si[it];
        if (si >= 0 && ((mask >> si) & 1u)) {
          const int idx = tid + it * NT;
          const int p = pb + __popc(mask & ((1u << si) - 1u));
          double v = g[NSYM] * kh[it][NSYM];
#pragma unroll
          for (int k = 0; k < NSYM; ++k) v += g[k] * kh[it][k];
          atomicAdd(&acc[s_pairoff[p] + (int)s_slot[p * N + (idx - si * N)]], v);
        }
      }
    }
  }

  // ---- P4: residual rows by quadrature, one lane per contribution pair:
  //      r_i = sum_q rq N_i + F . grad_ref N_i ----
  for (int p = tid; p < ((out.debug_skip & 4) ? 0 : NP); p += NT) {
    const int pk = s_pair[p];
    const int o = (pk >> 16) & 0xffff, t = (pk >> 8) & 0xff, ib = s_inv[pk & 0xff];
    const double *E = el + (size_t)t * EL;
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
    } else {
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
    atomicAdd(&racc[o], r);
  }
  __syncthreads();

  // ---- P5: stream the finished rows to HBM ----
  if (!(out.debug_skip & 8)) {
    const int wave = tid >> 6;
    constexpr int NW = NT / 64;
    if (out.compute_jacobian) {
      for (int o = wave; o < R; o += NW) {
        int len = s_len[o];
        const bool fx = len < 0;
        if (fx) len = -len - 1;
        if (fx && !out.overwrite) continue;
        double *dst = out.vals + s_base[o];
        const double *src = acc + s_off[o];
        if (out.overwrite) {
          for (int k = lane; k < len; k += 64) dst[k] = src[k];
        } else {
          for (int k = lane; k < len; k += 64) dst[k] += src[k];
        }
      }
    }
    for (int o = tid; o < R; o += NT) {
      const bool fx = s_len[o] < 0;
      const int g = s_rows[o];
      if (out.overwrite) out.res[g] = fx ? 0.0 : -racc[o];
      else if (!fx) out.res[g] -= racc[o];
    }
  }
}

template <int DIM, int P, int NQ1>
size_t lds_bytes(const RowBlocksDev &rb, int slot_bytes, bool tr) {
  using S = RO<DIM, P, NQ1, false>;
  const size_t el = tr ? RO<DIM, P, NQ1, true>::EL : S::EL;
  const size_t dbl = (size_t)rb.lds_acc + rb.lds_rows + S::TAB + (size_t)rb.lds_elems * el;
  const size_t ints = 4 * (size_t)rb.lds_rows + 3 * (size_t)rb.lds_elems + 2 * S::N + 2 * (size_t)rb.lds_pairs;
  return dbl * sizeof(double) + ints * sizeof(int) + (size_t)rb.lds_pairs * S::N * slot_bytes;
}

template <int DIM, int P, int NQ1, int NT, typename SlotT, bool TR>
void launch_affine_t(const BlockDev &b, const ThermalDev &ph, const RowBlocksDev &rb, const AffineDev &af,
                     const RowOut &out, int grid, size_t lds, hipStream_t stream) {
  auto kern = thermal_row_owner_affine_kernel<DIM, P, NQ1, NT, SlotT, TR>;
  MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, b, ph, rb, af, out);
  MHA_HIP(hipGetLastError());
}

template <int DIM, int P, int NQ1, int NT>
void launch_affine(const BlockDev &b, const ThermalDev &ph, const RowBlocksDev &rb, const AffineDev &af,
                   const RowOut &out, hipStream_t stream) {
  const int grid = rb.block_list ? rb.list_len : rb.num_blocks;
  if (grid <= 0) return;
  const bool tr = ph.time.transient != 0;
  const size_t lds = lds_bytes<DIM, P, NQ1>(rb, af.slot_bytes, tr);
  MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "row-owner kernel needs " << lds << " B of LDS (> 160 KiB)");
  MHA_REQUIRE(rb.lds_elems <= 64, MHA_ERR_INVALID, "row-owner kernel: more than 64 elements per row block");
  if (af.slot_bytes == 1) {
    if (tr) launch_affine_t<DIM, P, NQ1, NT, uint8_t, true>(b, ph, rb, af, out, grid, lds, stream);
    else launch_affine_t<DIM, P, NQ1, NT, uint8_t, false>(b, ph, rb, af, out, grid, lds, stream);
  } else {
    if (tr) launch_affine_t<DIM, P, NQ1, NT, uint16_t, true>(b, ph, rb, af, out, grid, lds, stream);
    else launch_affine_t<DIM, P, NQ1, NT, uint16_t, false>(b, ph, rb, af, out, grid, lds, stream);
  }
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

size_t thermal_row_owner_affine_lds(int dim, int order, int nq1, const RowBlocksDev &rb, int slot_bytes,
                                    bool transient) {
  if (dim == 2 && order == 1 && nq1 == 2) return lds_bytes<2, 1, 2>(rb, slot_bytes, transient);
  if (dim == 2 && order == 2 && nq1 == 3) return lds_bytes<2, 2, 3>(rb, slot_bytes, transient);
  if (dim == 2 && order == 4 && nq1 == 5) return lds_bytes<2, 4, 5>(rb, slot_bytes, transient);
  if (dim == 3 && order == 1 && nq1 == 2) return lds_bytes<3, 1, 2>(rb, slot_bytes, transient);
  if (dim == 3 && order == 2 && nq1 == 3) return lds_bytes<3, 2, 3>(rb, slot_bytes, transient);
  return 0;
}

void launch_thermal_row_owner_affine(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                                     const RowBlocksDev &rb, const AffineDev &af, const RowOut &out,
                                     hipStream_t stream) {
  if (dim == 2 && order == 1 && nq1 == 2) return launch_affine<2, 1, 2, 64>(b, ph, rb, af, out, stream);
  if (dim == 2 && order == 2 && nq1 == 3) return launch_affine<2, 2, 3, 128>(b, ph, rb, af, out, stream);
  if (dim == 2 && order == 4 && nq1 == 5) return launch_affine<2, 4, 5, 640>(b, ph, rb, af, out, stream);
  if (dim == 3 && order == 1 && nq1 == 2) return launch_affine<3, 1, 2, 64>(b, ph, rb, af, out, stream);
  if (dim == 3 && order == 2 && nq1 == 3) return launch_affine<3, 2, 3, 384>(b, ph, rb, af, out, stream);
  MHA_REQUIRE(false, MHA_ERR_INVALID, "row-owner kernel: unsupported (dim,order,points/dir)");
}

}  // namespace mha
