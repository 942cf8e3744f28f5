// condense.hip -- batched static condensation of element blocks: eliminate the interior unknowns of every element.
//
// The reference's subgrid (DtN / HDG) solver does this element by element with a sparse direct solve of the
// sub-problem and forward sensitivities d u / d lambda (SubGridDtN_Solver: local solve
// src/subgrid/subgridDtN_solver.cpp:681-903, flux sensitivities in updateFlux :1542-1616).  With the element blocks
//   [ A_uu A_ul ] [du]   [r_u]
//   [ A_lu A_ll ] [dl] = [r_l]          (blocks = d res / d (u, lambda), r = -res.val())
// the same information is  X = A_uu^{-1} [A_ul | r_u],  S = A_ll - A_lu X_ul,  g = r_l - A_lu x_r,  du0 = x_r.
// One wavefront per element: lane c owns column c of the augmented matrix [A_uu | A_ul | r_u] in registers and the
// wave runs Gauss-Jordan with partial pivoting on it (pivot column broadcast by shuffles); then lanes own columns of S.
// The trace rows [A_lu | A_ll | r_l] are requested before the elimination starts (one coalesced row per register, so
// their latency hides behind it) and the multipliers A_lu[a][i] of the Schur update come out of those registers by
// shuffle: one pass over the element's block, no wave-uniform reloads.
// n_int <= 32 and n_int + n_trace + 1 <= 64 (12 + 24 + 1 for the shallow-water HDG element).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "condense_core.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int kCondMaxInt = 32, kCondWaves = 4;

template <int MAXI>
__global__ __launch_bounds__(64 * kCondWaves) void condense_kernel(int ni, int nt, int64_t nelem,
                                                                   const double *__restrict__ blocks,
                                                                   const double *__restrict__ res, double *schur,
                                                                   double *gvec, double *du, int *singular) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * kCondWaves + wv;
  if (e >= nelem) return;
  const int n = ni + nt, ncol = n + 1;
  const double *B = blocks + e * n * n, *r = res + e * n;
  // column `lane` of [A_uu | A_ul | r_u], rows 0..ni-1
  double col[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    double v = 0.0;
    if (i < ni && lane < ncol) v = lane < n ? B[(size_t)i * n + lane] : r[i];
    col[i] = v;
  }
  // row ni + a of [A_lu | A_ll | r_l], entry `lane`
  const bool regs = nt <= kCondMaxTrace;
  double low[kCondMaxTrace];
#pragma unroll
  for (int a = 0; a < kCondMaxTrace; ++a) {
    double v = 0.0;
    if (regs && a < nt && lane < ncol) v = lane < n ? B[(size_t)(ni + a) * n + lane] : r[ni + a];
    low[a] = v;
  }
  if (!gauss_jordan_columns<MAXI>(ni, col)) { if (lane == 0 && singular) atomicAdd(singular, 1); return; }
  // lanes ni..n-1 now hold X_ul columns, lane n holds x_r = A_uu^{-1} r_u
  if (regs) {
    schur_from_registers<MAXI>(ni, nt, lane, e, col, low, schur, gvec, du);
  } else if (lane >= ni && lane < ncol) {
    if (du && lane == n) {
#pragma unroll
      for (int i = 0; i < MAXI; ++i)
        if (i < ni) du[e * ni + i] = col[i];
    }
    const int b = lane - ni;
    for (int a = 0; a < nt; ++a) {
      const double *Alu = B + (size_t)(ni + a) * n;  // row a of [A_lu | A_ll]
      double s = (lane < n) ? Alu[lane] : r[ni + a];
#pragma unroll
      for (int i = 0; i < MAXI; ++i)
        if (i < ni) s -= Alu[i] * col[i];
      if (lane < n) { if (schur) schur[(e * nt + a) * nt + b] = s; }
      else if (gvec) gvec[e * nt + a] = s;
    }
  }
}

}  // namespace

void launch_condense(int ni, int nt, int64_t nelem, const double *blocks, const double *res, double *schur, double *gvec,
                     double *du, int *singular, hipStream_t stream) {
  if (nelem <= 0) return;
  MHA_REQUIRE(ni >= 1 && ni <= kCondMaxInt && nt >= 1 && ni + nt + 1 <= 64, MHA_ERR_INVALID,
              "condensation: need 1 <= n_int <= " << kCondMaxInt << " and n_int + n_trace + 1 <= 64");
  const int grid = (int)((nelem + kCondWaves - 1) / kCondWaves);
  if (ni <= 8)
    hipLaunchKernelGGL(condense_kernel<8>, dim3(grid), dim3(64 * kCondWaves), 0, stream, ni, nt, nelem, blocks, res, schur,
                       gvec, du, singular);
  else if (ni <= 16)
    hipLaunchKernelGGL(condense_kernel<16>, dim3(grid), dim3(64 * kCondWaves), 0, stream, ni, nt, nelem, blocks, res, schur,
                       gvec, du, singular);
  else
    hipLaunchKernelGGL(condense_kernel<kCondMaxInt>, dim3(grid), dim3(64 * kCondWaves), 0, stream, ni, nt, nelem, blocks,
                       res, schur, gvec, du, singular);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
