// condense_core.hpp -- Gauss-Jordan elimination of [A_uu | A_ul | r_u] held one column per lane, then the Schur
// complement rows from the trace rows held one row per register (see condense.hip); shared by the stand-alone
// condensation kernel and the fused HDG element kernel (swhdg_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mha {

// Value of lane `src` (wave-uniform index) for every lane: v_readlane_b32 x2 into scalar registers.  The generic
// __shfl goes through the LDS crossbar (ds_bpermute); the elimination does ~1500 of these per element.
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

constexpr int kCondMaxTrace = 32;  // trace rows held in registers (more: read row by row in the Schur loop)

// col[i]: entry (i, lane) of [A_uu | A_ul | r_u] (rows i < ni, columns lane <= n = ni + nt).  Gauss-Jordan with partial
// pivoting; afterwards lanes ni .. n - 1 hold the columns of X_ul = A_uu^-1 A_ul and lane n holds x_r = A_uu^-1 r_u.
// Returns false when A_uu is singular.
template <int MAXI>
__device__ __forceinline__ bool gauss_jordan_columns(int ni, double (&col)[MAXI]) {
  bool bad = false;
  for (int k = 0; k < ni; ++k) {
    // partial pivoting on column k (held by lane k): every lane scans its OWN column (no cross-lane traffic), lane k's
    // answer is the one that counts -- the pivot row is the same for every lane
    double best = -1.0;
    int piv = k;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const double a = fabs(col[i]);
      if (i >= k && i < ni && a > best) { best = a; piv = i; }
    }
    piv = __builtin_amdgcn_readlane(piv, k);
    best = readlane_f64(best, k);
    if (!(best > 0.0)) { bad = true; break; }
    // swap rows k and piv of this lane's column (dynamic index -> select chain)
    double ck = 0.0, cp = 0.0;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) { if (i == k) ck = col[i]; if (i == piv) cp = col[i]; }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) { if (i == k) col[i] = cp; else if (i == piv) col[i] = ck; }
    // eliminate: row_i -= (a_ik / a_kk) row_k for all i != k, row_k /= a_kk
    double pk = 0.0;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) if (i == k) pk = col[i];
    const double akk = readlane_f64(pk, k);
    const double rk = pk / akk;  // this lane's entry of the normalised pivot row
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const double aik = readlane_f64(col[i], k);  // multiplier source: column k before the update
      if (i < ni) col[i] = (i == k) ? rk : col[i] - aik * rk;
    }
  }
  return !bad;
}

// The same elimination for a block size known at compile time (the fused HDG element kernel: 12 interior unknowns): the
// pivot loop is unrolled, so "row k" is a register name and only the pivot row found at run time is reached through a
// select chain -- a third of the instructions of the run-time form.
template <int NI>
__device__ __forceinline__ bool gauss_jordan_columns_static(double (&col)[NI]) {
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    double best = -1.0;
    int piv = k;
#pragma unroll
    for (int i = k; i < NI; ++i) {
      const double a = fabs(col[i]);
      if (a > best) { best = a; piv = i; }
    }
    piv = __builtin_amdgcn_readlane(piv, k);
    best = readlane_f64(best, k);
    if (!(best > 0.0)) bad = true;  // (keeps going on garbage: the caller reports the block as singular)
    // swap rows k and piv of this lane's column: col[k] is a register, col[piv] a select chain over rows > k
    // (piv is wave-uniform: no swap, no chain, when the diagonal entry is the pivot)
    double cp = col[k];
    if (piv != k) {
#pragma unroll
      for (int i = k + 1; i < NI; ++i) if (i == piv) cp = col[i];
      const double ck = col[k];
#pragma unroll
      for (int i = k + 1; i < NI; ++i) if (i == piv) col[i] = ck;
    }
    const double akk = readlane_f64(cp, k);
    const double rk = cp / akk;  // this lane's entry of the normalised pivot row
    col[k] = cp;                 // (column k of the multipliers below is read before the update)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i == k) continue;
      const double aik = readlane_f64(col[i], k);
      col[i] -= aik * rk;
    }
    col[k] = rk;
  }
  return !bad;
}

// low[a]: entry (ni + a, lane) of [A_lu | A_ll | r_l], col[] as gauss_jordan_columns leaves it: S = A_ll - A_lu X_ul
// (lanes ni .. n - 1, column lane - ni), g = r_l - A_lu x_r (lane n) and x_r (lane n) to element-major arrays (any may
// be null).
template <int MAXI, int MAXT = kCondMaxTrace>
__device__ __forceinline__ void schur_from_registers(int ni, int nt, int lane, int64_t e, const double (&col)[MAXI],
                                                     const double (&low)[MAXT], double *schur, double *gvec, double *du) {
  const int n = ni + nt;
  if (du && lane == n) {
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < ni) du[e * ni + i] = col[i];
  }
  const int b = lane - ni;
#pragma unroll
  for (int a = 0; a < MAXT; ++a) {
    if (a < nt) {  // uniform
      const double rowv = low[a];
      double sacc = rowv;
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const double m = readlane_f64(rowv, i);  // A_lu[a][i], executed by every lane
        if (i < ni) sacc -= m * col[i];
      }
      if (lane >= ni && lane < n) { if (schur) schur[(e * nt + a) * nt + b] = sacc; }
      else if (lane == n && gvec) gvec[e * nt + a] = sacc;
    }
  }
}

}  // namespace mha
