// thermal_general.hip -- thermal volume residual + Jacobian for GENERAL elements (non-affine geometry,
// coefficients that vary from integration point to integration point) on gfx950.
//
// One wavefront per element, four elements per workgroup.  Per element:
//   1. geometry at every integration point from the cell vertices (CellTools::setJacobian/Det/Inv),
//      D_q = kappa(q) * w_q * detJ_q * J_q^{-1} J_q^{-T}  (symmetric dim x dim) and the mass weight;
//   2. solution fields at the integration points (sum-factorised), point-wise residual data;
//   3. residual rows by quadrature -> one f64 atomic per dof into the global vector (-res.val());
//   4. the element Jacobian as a small GEMM  K = P * Ghat^T  with
//        P[i][(q,a)] = sum_b D_q^{ab} dhat_b N_i(q),   Ghat[j][(q,a)] = dhat_a N_j(q)
//      (+ the mass columns sqrt-free: m_q N_i(q) against N_j(q)), the (q,a) dimension streamed through LDS in chunks.
//      Elements with more than 16 dofs (Q2 hexes, Q4 quads) run the product on the matrix cores
//      (v_mfma_f64_16x16x4_f64, 2 x 2 output tiles per wavefront): the same FMA rate as the vector unit on gfx950, but
//      1024 FMAs per two 8-byte LDS operands instead of 16 per four 16-byte ones -- the register-tile version of this
//      loop was bound by LDS bandwidth (4 SIMDs x 4 KB per k step against 128 B/clk).  Smaller elements keep 4x4
//      register tiles per lane;
//   5. scatter of the tile entries with f64 atomics into the CRS values; the position inside the row
//      comes from a one-byte-per-entry element-major slot map (no column search), fixed rows skipped.
// This is res(e,i).dx(j) = sum_q [ kappa w alpha_u grad N_j . grad N_i + rho cp alpha_t N_j N_i w ], the
// derivative array the reference's Sacado sweep produces (src/physics/thermal.cpp:125-163), with the same
// gather / seeding / scatter conventions as the other kernels (citations in thermal_element.hip).
//
// Roofline: compute-bound on the f64 VALU (about 1.0e5 flop per Q2-hex element against 6.6 KB of
// compulsory traffic); the scatter adds 8 n^2 bytes of HBM atomics per element.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int cpowg(int b, int e) { return e == 0 ? 1 : b * cpowg(b, e - 1); }

template <int DIM, int P, int NQ1, bool TR>
struct GK {
  static constexpr int M = P + 1;
  static constexpr int N = cpowg(M, DIM);
  static constexpr int NQ = cpowg(NQ1, DIM);
  static constexpr int NN = 1 << DIM;
  static constexpr int NSYM = DIM * (DIM + 1) / 2;
  static constexpr int TI = (N + 3) / 4;          // 4x4 tiles per direction
  static constexpr int NP = TI * 4;               // padded dof count
  static constexpr int QC = NQ < 9 ? NQ : 9;      // integration points per GEMM chunk
  static constexpr int NCH = (NQ + QC - 1) / QC;
  static constexpr int KC = QC * DIM + (TR ? QC : 0);  // GEMM depth of one chunk (+ mass columns)
#ifndef MHA_TG_MFMA
#define MHA_TG_MFMA 1
#endif
#ifndef MHA_TG_STOP
#define MHA_TG_STOP 9  // profiling aid (profiles/tg_ablate.sh): leave the kernel after phase 1, 2, 3 or the product (4)
#endif
  static constexpr bool MF = MHA_TG_MFMA && N > 16;  // Jacobian product on the matrix cores (2 x 2 tiles of 16 x 16)
#ifndef MHA_TG_QM
#define MHA_TG_QM 8
#endif
  static constexpr int QM = MHA_TG_QM;            // points per MFMA chunk: QM * DIM rows of P, a multiple of 4
  static constexpr int PR = QM * DIM;             // rows of the MFMA P panel
  static constexpr int PW = 32;                   // its width (dofs, zero padded)
#ifndef MHA_TG_EPB
#define MHA_TG_EPB 4
#endif
  static constexpr int EPB = MHA_TG_EPB;          // elements (waves) per workgroup
  static constexpr int NT = EPB * 64;             // threads per workgroup
  // shared tables (doubles)
  static constexpr int S_GT = 0;                  // Ghat^T  [NQ*DIM][NP]
  static constexpr int S_NT = S_GT + NQ * DIM * NP;   // Nhat^T  [NQ][NP]
  static constexpr int S_NG = S_NT + NQ * NP;     // vertex basis gradients [NN][NQ][DIM]
  static constexpr int S_NV = S_NG + NN * NQ * DIM;   // vertex basis values    [NN][NQ]
  static constexpr int S_W = S_NV + NN * NQ;      // reference weights [NQ]
  static constexpr int SHARED = (S_W + NQ + 1) / 2 * 2;  // even: records stay 16-byte aligned
  // per-element record (doubles)
  static constexpr int O_XN = 0;                  // vertices [NN][DIM]
  static constexpr int O_UE = O_XN + NN * DIM;    // u_eval [N]
  static constexpr int O_UD = O_UE + N;           // u_dot  [N]
  static constexpr int O_D = O_UD + (TR ? N : 0); // D_q    [NQ][NSYM]  (alpha_u folded in)
  static constexpr int O_MQ = O_D + NQ * NSYM;    // rho cp w det alpha_t [NQ]
  static constexpr int O_F = O_MQ + (TR ? NQ : 0);    // D_q grad_ref T (unscaled by alpha_u) [NQ][DIM]
  static constexpr int O_RQ = O_F + NQ * DIM;     // (rho cp T_t - f) w det [NQ]
  static constexpr int O_PT = (O_RQ + NQ + 1) / 2 * 2;  // P chunk, transposed [KC][NP] (16-byte aligned)
  static constexpr int REC = (O_PT + (MF ? PR * PW : KC * NP) + 1) / 2 * 2;
};

#ifndef MHA_TG_MINW
#define MHA_TG_MINW 2
#endif
template <int DIM, int P, int NQ1, bool TR, bool EXPR>
__global__ __launch_bounds__((GK<DIM, P, NQ1, TR>::NT), MHA_TG_MINW) void thermal_general_element_kernel(BlockDev b, ThermalDev ph, AffineDev af,
                                                                       const uint8_t *__restrict__ slot8,
                                                                       const uint16_t *__restrict__ slot16, ElemOut out) {
  using S = GK<DIM, P, NQ1, TR>;
  constexpr int M = S::M, N = S::N, NQ = S::NQ, NN = S::NN, NSYM = S::NSYM, NP = S::NP, TI = S::TI;
  constexpr int QC = S::QC, NCH = S::NCH;
  static_assert(TI * TI <= 64, "one wave holds all 4x4 tiles of the element matrix");
  extern __shared__ double smem[];
  double *sh = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *E = smem + S::SHARED + wave * S::REC;
  const TimeDev &tm = ph.time;
  __shared__ double tab[2 * M * NQ1];  // 1-D tables for the sum-factorised field evaluation
  __shared__ int s_offs[N];

  // ---- shared tables ----
  for (int i = tid; i < NQ * DIM * NP; i += S::NT) {
    const int k = i / NP, j = i - k * NP;  // k = q*DIM + a
    sh[S::S_GT + i] = (j < N) ? b.ref_grad[((size_t)j * NQ + k / DIM) * DIM + k % DIM] : 0.0;
  }
  for (int i = tid; i < NQ * NP; i += S::NT) {
    const int q = i / NP, j = i - q * NP;
    sh[S::S_NT + i] = (j < N) ? b.ref_basis[j * NQ + q] : 0.0;
  }
  for (int i = tid; i < NN * NQ * DIM; i += S::NT) sh[S::S_NG + i] = b.nodegrad[i];
  for (int i = tid; i < NN * NQ; i += S::NT) sh[S::S_NV + i] = b.nodeval[i];
  for (int i = tid; i < NQ; i += S::NT) sh[S::S_W + i] = b.ref_wts[i];
  for (int i = tid; i < M * NQ1; i += S::NT) { tab[i] = af.phi1d[i]; tab[M * NQ1 + i] = af.dphi1d[i]; }
  for (int i = tid; i < N; i += S::NT) s_offs[i] = b.offsets[i];
  __syncthreads();
  const double *phi = tab, *dphi = tab + M * NQ1;
  // persistent over elements: the shared tables are loaded once per workgroup (they were 0.7 ms of the 2.6 ms of the
  // one-shot version at 64^3 Q2 hexes); everything below is private to the wavefront, so no block barriers
  for (int el = blockIdx.x * S::EPB + wave; el < b.e_count; el += gridDim.x * S::EPB) {
  const int e = b.e_begin + el;
  constexpr bool active = true;
  wave_lds_sync();  // previous element consumed
  for (int i = lane; i < NN * DIM; i += 64) E[S::O_XN + i] = b.nodes[(size_t)e * NN * DIM + i];
  wave_lds_sync();
  const int32_t *L = b.lids + (size_t)e * N;

  // ---- 1. gather + seeding values (lane = basis dof), geometry + coefficients (lane = q) ----
  if (active) {
    for (int dof = lane; dof < N; dof += 64) {
      const int row = L[s_offs[dof]];
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
        E[S::O_UD + dof] = tm.alpha_t * cu + beta_t;
      }
      E[S::O_UE + dof] = ue;
    }
    for (int q = lane; q < NQ; q += 64) {
      double J[DIM * DIM], Ji[DIM * DIM], det, x[3] = {0, 0, 0};
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double s = 0.0;
#pragma unroll
          for (int v = 0; v < NN; ++v) s += E[S::O_XN + v * DIM + r] * sh[S::S_NG + (v * NQ + q) * DIM + c];
          J[r * DIM + c] = s;
        }
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < NN; ++v) s += E[S::O_XN + v * DIM + r] * sh[S::S_NV + v * NQ + q];
        x[r] = s;
      }
      invert<DIM>(J, Ji, det);
      const double w = sh[S::S_W + q] * det;
      const double kap = eval_func<DIM, EXPR>(ph.diff, e, q, NQ, x);
      const double rc = eval_func<DIM, EXPR>(ph.rho, e, q, NQ, x) * eval_func<DIM, EXPR>(ph.cp, e, q, NQ, x);
      const double f = eval_func<DIM, EXPR>(ph.source, e, q, NQ, x);
      int k = 0;
#pragma unroll
      for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int c = a; c < DIM; ++c) {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) s += Ji[a * DIM + d] * Ji[c * DIM + d];
          E[S::O_D + q * NSYM + k++] = kap * w * s;
        }
      if constexpr (TR) E[S::O_MQ + q] = rc * w;
      E[S::O_RQ + q] = -f * w;  // completed below
      E[S::O_F + q * DIM] = rc * w;  // parked until the fields are known
    }
  }
  wave_lds_sync();  // the record is private to this wavefront
  if (MHA_TG_STOP == 1) continue;

  // ---- 2. fields at the integration points, point-wise residual data (lane = q) ----
  if (active) {
    for (int q = lane; q < NQ; q += 64) {
      double gh[DIM], tv, gd[DIM], tt = 0.0;
      eval_ref<DIM, P, NQ1>(E + S::O_UE, phi, dphi, q, gh, tv);
      if constexpr (TR) eval_ref<DIM, P, NQ1>(E + S::O_UD, phi, dphi, q, gd, tt);
      (void)gd; (void)tv;
      const double rcw = E[S::O_F + q * DIM];
      double D[DIM][DIM];
      {
        int k = 0;
#pragma unroll
        for (int a = 0; a < DIM; ++a)
#pragma unroll
          for (int c = a; c < DIM; ++c) { D[a][c] = E[S::O_D + q * NSYM + k]; D[c][a] = D[a][c]; ++k; }
      }
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) s += D[a][c] * gh[c];
        E[S::O_F + q * DIM + a] = s;
      }
      E[S::O_RQ + q] += rcw * tt;
    }
  }
  wave_lds_sync();
  if (MHA_TG_STOP == 2) continue;

  // ---- 3. residual rows (lane = basis dof) ----
  if (active) {
    for (int i = lane; i < N; i += 64) {
      double r = 0.0;
      for (int q = 0; q < NQ; ++q) {
        r += E[S::O_RQ + q] * sh[S::S_NT + q * NP + i];
#pragma unroll
        for (int a = 0; a < DIM; ++a) r += E[S::O_F + q * DIM + a] * sh[S::S_GT + (q * DIM + a) * NP + i];
      }
      const int slot = s_offs[i];
      if (out.local_res) {
        double *lr = out.local_res + (size_t)(e - out.local_base) * N + slot;
        *lr = out.local_store ? -r : *lr - r;
      }
      if (out.res) {
        const int row = L[slot];
        if (!(b.fixed && b.fixed[row])) atomicAdd(out.res + row, -r);
      }
    }
  }
  if (out.compute_jacobian <= 0 || MHA_TG_STOP == 3) continue;  // uniform

  const double au = tm.alpha_u, at = tm.alpha_t;
  if constexpr (S::MF) {
    // ---- 4m. element Jacobian on the matrix cores.  K[i][j] = sum_k Pt[k][i] * Bt[k][j], k = (q,a) then (mass) q;
    //   Bt is the shared table Ghat^T followed by Nhat^T (contiguous), Pt is built in chunks of QM points per wave.
    //   Operand maps (CDNA4): A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15], D reg t: row = (lane>>4) + 4t.
    //   Rows / columns N..31 of the tiles are padding: they read finite table data and are never stored.
    static_assert(N <= 32 && S::PR % 4 == 0, "2 x 2 tiles of 16");
    typedef double v4d __attribute__((ext_vector_type(4)));
    constexpr int PW = S::PW, QM = S::QM;
    const int l15 = lane & 15, l4 = lane >> 4;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][c] = v4d{0.0, 0.0, 0.0, 0.0};
    double *Pt = E + S::O_PT;
    if (active) {
      auto product = [&](int krow0, int nrows) {  // nrows (multiple of 4) panel rows against table rows krow0..
        const double *Bt = sh + S::S_GT + (size_t)krow0 * NP;
        for (int k0 = 0; k0 < nrows; k0 += 4) {
          const double a0 = Pt[(k0 + l4) * PW + l15], a1 = Pt[(k0 + l4) * PW + 16 + l15];
          const double b0 = Bt[(k0 + l4) * NP + l15], b1 = Bt[(k0 + l4) * NP + 16 + l15];
          acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
      };
      for (int q0 = 0; q0 < NQ; q0 += QM) {
        wave_lds_sync();  // previous panel consumed (the record is private to this wavefront)
        // Pt[(qq*DIM + a)][i] = alpha_u * sum_b D_q^{ab} dhat_b N_i(q); rows of points past NQ are zero
        for (int item = lane; item < QM * PW; item += 64) {
          const int qq = item / PW, i = item - qq * PW;
          const int q = q0 + qq;
          const bool ok = q < NQ && i < NP;
          double D[DIM][DIM];
          {
            int k = 0;
#pragma unroll
            for (int a = 0; a < DIM; ++a)
#pragma unroll
              for (int c = a; c < DIM; ++c) { D[a][c] = ok ? E[S::O_D + q * NSYM + k] : 0.0; D[c][a] = D[a][c]; ++k; }
          }
          double gi[DIM];
#pragma unroll
          for (int c = 0; c < DIM; ++c) gi[c] = ok ? sh[S::S_GT + (q * DIM + c) * NP + i] : 0.0;
#pragma unroll
          for (int a = 0; a < DIM; ++a) {
            double sacc = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) sacc += D[a][c] * gi[c];
            Pt[(qq * DIM + a) * PW + i] = au * sacc;
          }
        }
        wave_lds_sync();
        const int nq = (q0 + QM <= NQ) ? QM : NQ - q0;
        product(q0 * DIM, (nq * DIM + 3) & ~3);
      }
      if constexpr (TR) {
        for (int q0 = 0; q0 < NQ; q0 += S::PR) {  // mass rows: Pt[r][i] = alpha_t m_q N_i(q)
          wave_lds_sync();
          for (int item = lane; item < S::PR * PW; item += 64) {
            const int r = item / PW, i = item - r * PW;
            const int q = q0 + r;
            Pt[r * PW + i] = (q < NQ && i < NP) ? at * E[S::O_MQ + q] * sh[S::S_NT + q * NP + i] : 0.0;
          }
          wave_lds_sync();
          const int nq = (q0 + S::PR <= NQ) ? S::PR : NQ - q0;
          product(NQ * DIM + q0, (nq + 3) & ~3);
        }
      }
      if (MHA_TG_STOP == 4) {  // keep the product alive without the stores
        if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.678) out.local_J[0] = 1.0;
        continue;
      }
      // ---- 5m. store / scatter the tiles: 16 consecutive columns per row and register ----
#pragma unroll
      for (int ti2 = 0; ti2 < 2; ++ti2)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int i = 16 * ti2 + l4 + 4 * t;
          if (i >= N) continue;
          const int si = s_offs[i];
          const int row = L[si];
          const bool fx = b.fixed && b.fixed[row];
          const int rbase = out.crs_vals ? b.rowptr[row] : 0;
#pragma unroll
          for (int tj2 = 0; tj2 < 2; ++tj2) {
            const int j = 16 * tj2 + l15;
            if (j >= N) continue;
            const int sj = s_offs[j];
            const double v = acc[ti2][tj2][t];
            if (out.local_J) {
              double *lj = out.local_J + ((size_t)(e - out.local_base) * N + si) * N + sj;
              *lj = out.local_store ? v : *lj + v;
            }
            if (out.crs_vals && !fx) {
              const size_t sidx = ((size_t)e * N + si) * N + sj;
              const int sl = slot8 ? (int)slot8[sidx] : (int)slot16[sidx];
              atomicAdd(out.crs_vals + rbase + sl, v);
            }
          }
        }
    }
  } else {
  // ---- 4. element Jacobian: K = P * Ghat^T (+ mass), 4x4 tiles per lane, chunks of QC points ----
  const int ti = lane / TI, tj = lane - ti * TI;
  const bool tile = lane < TI * TI;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
  for (int ch = 0; ch < NCH; ++ch) {
    const int q0 = ch * QC;
    wave_lds_sync();  // previous chunk consumed
    if (active) {
      // P chunk, transposed: Pt[(qq*DIM + a)][i] = alpha_u * sum_b D_q^{ab} dhat_b N_i(q); mass: alpha_t m_q N_i(q)
      for (int item = lane; item < QC * NP; item += 64) {
        const int qq = item / NP, i = item - qq * NP;
        const int q = q0 + qq;
        const bool ok = q < NQ;
        double D[DIM][DIM];
        {
          int k = 0;
#pragma unroll
          for (int a = 0; a < DIM; ++a)
#pragma unroll
            for (int c = a; c < DIM; ++c) { D[a][c] = ok ? E[S::O_D + q * NSYM + k] : 0.0; D[c][a] = D[a][c]; ++k; }
        }
        double gi[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) gi[c] = ok ? sh[S::S_GT + (q * DIM + c) * NP + i] : 0.0;
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < DIM; ++c) s += D[a][c] * gi[c];
          E[S::O_PT + (qq * DIM + a) * NP + i] = au * s;
        }
        if constexpr (TR) E[S::O_PT + (QC * DIM + qq) * NP + i] = ok ? at * E[S::O_MQ + q] * sh[S::S_NT + q * NP + i] : 0.0;
      }
    }
    wave_lds_sync();
    if (active && tile) {
      const int nq = (q0 + QC <= NQ) ? QC : NQ - q0;
      const double *A = E + S::O_PT + 4 * ti;
      const double *Bg = sh + S::S_GT + (size_t)q0 * DIM * NP + 4 * tj;
      for (int k = 0; k < nq * DIM; ++k) {
        const double2 a01 = *reinterpret_cast<const double2 *>(A + k * NP), a23 = *reinterpret_cast<const double2 *>(A + k * NP + 2);
        const double2 b01 = *reinterpret_cast<const double2 *>(Bg + k * NP), b23 = *reinterpret_cast<const double2 *>(Bg + k * NP + 2);
        const double av[4] = {a01.x, a01.y, a23.x, a23.y}, bv[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[a][c] += av[a] * bv[c];
      }
      if constexpr (TR) {
        const double *Am = E + S::O_PT + (size_t)QC * DIM * NP + 4 * ti;
        const double *Bn = sh + S::S_NT + (size_t)q0 * NP + 4 * tj;
        for (int k = 0; k < nq; ++k) {
          const double2 a01 = *reinterpret_cast<const double2 *>(Am + k * NP), a23 = *reinterpret_cast<const double2 *>(Am + k * NP + 2);
          const double2 b01 = *reinterpret_cast<const double2 *>(Bn + k * NP), b23 = *reinterpret_cast<const double2 *>(Bn + k * NP + 2);
          const double av[4] = {a01.x, a01.y, a23.x, a23.y}, bv[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] += av[a] * bv[c];
        }
      }
    }
  }

  // ---- 5. scatter the tile: dense local_J (updateJac convention) and/or atomics into the CRS values ----
  if (active && tile) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = 4 * ti + a;
      if (i >= N) continue;
      const int si = s_offs[i];
      const int row = L[si];
      const bool fx = b.fixed && b.fixed[row];
      const int rbase = out.crs_vals ? b.rowptr[row] : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = 4 * tj + c;
        if (j >= N) continue;
        const int sj = s_offs[j];
        if (out.local_J) {
          double *lj = out.local_J + ((size_t)(e - out.local_base) * N + si) * N + sj;
          *lj = out.local_store ? acc[a][c] : *lj + acc[a][c];
        }
        if (out.crs_vals && !fx) {
          const size_t sidx = ((size_t)e * N + si) * N + sj;
          const int sl = slot8 ? (int)slot8[sidx] : (int)slot16[sidx];
          atomicAdd(out.crs_vals + rbase + sl, acc[a][c]);
        }
      }
    }
  }
  }
  }  // elements
}

template <typename SlotT>
__global__ __launch_bounds__(256) void build_elem_slot_map_kernel(BlockDev b, SlotT *slot) {
  const int n = b.n;
  const size_t per = (size_t)n * n, total = (size_t)b.nelem * per;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx / per);
    const int rc = (int)(idx - (size_t)e * per);
    const int si = rc / n, sj = rc - si * n;
    const int32_t *L = b.lids + (size_t)e * n;
    const int row = L[si];
    const int lo = b.rowptr[row];
    const int p = find_col(b.colind, lo, b.rowptr[row + 1], L[sj]);
    slot[idx] = (SlotT)(p < 0 ? 0 : p - lo);
  }
}

template <int DIM, int P, int NQ1>
void launch_one(const BlockDev &b, const ThermalDev &ph, const AffineDev &af, const void *slot, int slot_bytes,
                const ElemOut &out, hipStream_t stream) {
  if (b.e_count <= 0) return;
  const bool tr = ph.time.transient != 0;
  using S0 = GK<DIM, P, NQ1, false>;
  using S1 = GK<DIM, P, NQ1, true>;
  const size_t lds = sizeof(double) * (tr ? S1::SHARED + (size_t)S1::EPB * S1::REC : S0::SHARED + (size_t)S0::EPB * S0::REC);
  MHA_REQUIRE(lds <= 150 * 1024, MHA_ERR_INVALID, "general element kernel needs " << lds << " B of LDS");
  const int num_cu = current_device_num_cus();
  // persistent workgroups: as many as are resident at once (LDS-limited), each wave strides over the elements
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 512)));
  const int grid = std::max(1, std::min((b.e_count + S0::EPB - 1) / S0::EPB, num_cu * per_cu));
  const uint8_t *s8 = slot_bytes == 1 ? static_cast<const uint8_t *>(slot) : nullptr;
  const uint16_t *s16 = slot_bytes == 2 ? static_cast<const uint16_t *>(slot) : nullptr;
  auto go = [&](auto kern) {
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S0::NT), lds, stream, b, ph, af, s8, s16, out);
    MHA_HIP(hipGetLastError());
  };
  if (has_expression(ph)) {
    if (tr) go(thermal_general_element_kernel<DIM, P, NQ1, true, true>);
    else go(thermal_general_element_kernel<DIM, P, NQ1, false, true>);
  } else {
    if (tr) go(thermal_general_element_kernel<DIM, P, NQ1, true, false>);
    else go(thermal_general_element_kernel<DIM, P, NQ1, false, false>);
  }
}

}  // namespace

void launch_build_elem_slot_map(const BlockDev &b, void *slot, int slot_bytes, hipStream_t stream) {
  const size_t total = (size_t)b.nelem * b.n * b.n;
  const size_t g = (total + 255) / 256;
  const int grid = (int)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
  if (slot_bytes == 1)
    hipLaunchKernelGGL(build_elem_slot_map_kernel<uint8_t>, dim3(grid), dim3(256), 0, stream, b, static_cast<uint8_t *>(slot));
  else
    hipLaunchKernelGGL(build_elem_slot_map_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream, b, static_cast<uint16_t *>(slot));
  MHA_HIP(hipGetLastError());
}

void launch_thermal_general(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph, const AffineDev &af,
                            const void *slot, int slot_bytes, const ElemOut &out, hipStream_t stream) {
  if (dim == 2 && order == 1 && nq1 == 2) return launch_one<2, 1, 2>(b, ph, af, slot, slot_bytes, out, stream);
  if (dim == 2 && order == 2 && nq1 == 3) return launch_one<2, 2, 3>(b, ph, af, slot, slot_bytes, out, stream);
  if (dim == 2 && order == 4 && nq1 == 5) return launch_one<2, 4, 5>(b, ph, af, slot, slot_bytes, out, stream);
  if (dim == 3 && order == 1 && nq1 == 2) return launch_one<3, 1, 2>(b, ph, af, slot, slot_bytes, out, stream);
  if (dim == 3 && order == 2 && nq1 == 3) return launch_one<3, 2, 3>(b, ph, af, slot, slot_bytes, out, stream);
  MHA_REQUIRE(false, MHA_ERR_INVALID, "general element kernel: unsupported (dim,order,points/dir)");
}

}  // namespace mha
