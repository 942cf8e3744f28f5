// swhdg_side.hpp -- shallowwaterHybridized at a side integration point (device functions, scalar type double or Dual).
//
// reference: src/physics/shallowwaterHybridized.cpp -- computeFluxVector :409-480, eigendecompFluxJacobian :765-823,
// computeStabilizationTerm :487-588, computeBoundaryTerm :595-758, computeFlux :270-368.  State order H, Hux, Huy
// (2-D); S = interior state, Sh = trace state ("aux" variables), Sinf = far-field state.
#pragma once
#include "dual.hpp"

namespace mha {

__device__ __forceinline__ double s_abs(double a) { return fabs(a); }
__device__ __forceinline__ Dual s_abs(Dual a) { return a.v < 0.0 ? -a : a; }
__device__ __forceinline__ double s_max(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ Dual s_max(Dual a, Dual b) { return a.v > b.v ? a : b; }
__device__ __forceinline__ double s_sqrt(double a) { return sqrt(a); }
__device__ __forceinline__ Dual s_sqrt(Dual a) { return dsqrt(a); }
__device__ __forceinline__ double s_const(double, double c) { return c; }
__device__ __forceinline__ Dual s_const(Dual, double c) { return mk(c); }

// F[eqn][dir] of a state (computeFluxVector, 2-D)
template <class T>
__device__ __forceinline__ void swh_flux_vector(const T *S, double g, T F[3][2]) {
  const T H = S[0], Hux = S[1], Huy = S[2];
  const T hh = H * H * (0.5 * g);
  F[0][0] = Hux; F[0][1] = Huy;
  F[1][0] = Hux * Hux / H + hh; F[1][1] = Hux * Huy / H;
  F[2][0] = Hux * Huy / H; F[2][1] = Huy * Huy / H + hh;
}

// A = R Lambda L of the normal flux Jacobian at the trace state (eigendecompFluxJacobian, 2-D)
template <class T>
__device__ __forceinline__ void swh_eigendecomp(const T *Sh, double nx, double ny, double g, T L[3][3], T lam[3], T R[3][3]) {
  const T H = Sh[0], ux = Sh[1] / H, uy = Sh[2] / H;
  const T vn = ux * nx + uy * ny, a = s_sqrt(H * g);
  const T one = s_const(H, 1.0), zero = s_const(H, 0.0);
  R[0][0] = one; R[1][0] = ux + a * nx; R[2][0] = uy + a * ny;
  R[0][1] = zero; R[1][1] = -(a * ny); R[2][1] = a * nx;
  R[0][2] = one; R[1][2] = ux - a * nx; R[2][2] = uy - a * ny;
  const T i2a = 0.5 / a, ia = 1.0 / a;
  L[0][0] = one * 0.5 - vn * i2a; L[0][1] = i2a * nx; L[0][2] = i2a * ny;
  L[1][0] = (ux * ny - uy * nx) * ia; L[1][1] = -(ia * ny); L[1][2] = ia * nx;
  L[2][0] = one * 0.5 + vn * i2a; L[2][1] = -(i2a * nx); L[2][2] = -(i2a * ny);
  lam[0] = vn + a; lam[1] = vn; lam[2] = vn - a;
}

template <class T>
__device__ __forceinline__ void swh_matvec(const T A[3][3], const T *x, T *y) {
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = A[i][0] * x[0] + A[i][1] * x[1] + A[i][2] * x[2];
}

// Stab (S - Sh): R |Lambda| L (Roe-like) or lambda_max I (computeStabilizationTerm)
template <class T>
__device__ __forceinline__ void swh_stab_term(const T *S, const T *Sh, double nx, double ny, double g, bool roe, T *out) {
  T dS[3] = {S[0] - Sh[0], S[1] - Sh[1], S[2] - Sh[2]};
  if (roe) {
    T L[3][3], lam[3], R[3][3], tmp[3];
    swh_eigendecomp(Sh, nx, ny, g, L, lam, R);
    swh_matvec(L, dS, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = tmp[i] * s_abs(lam[i]);
    swh_matvec(R, tmp, out);
  } else {
    const T vn = Sh[1] / Sh[0] * nx + Sh[2] / Sh[0] * ny, a = s_sqrt(Sh[0] * g);
    const T lmax = s_max(s_abs(vn + a), s_abs(vn - a));
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = dS[i] * lmax;
  }
}

// B: far-field (type 1) A+ (S - Sh) - A- (Sinf - Sh); slip (type 2)  (computeBoundaryTerm)
template <class T>
__device__ __forceinline__ void swh_boundary_term(int type, const T *S, const T *Sh, const double *Sinf, double nx, double ny,
                                                  double g, T *out) {
  if (type == MHA_SWH_FARFIELD) {
    T L[3][3], lam[3], R[3][3], tmp[3], neg[3];
    swh_eigendecomp(Sh, nx, ny, g, L, lam, R);
    T dS[3] = {S[0] - Sh[0], S[1] - Sh[1], S[2] - Sh[2]};
    swh_matvec(L, dS, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = tmp[i] * ((lam[i] + s_abs(lam[i])) * 0.5);
    swh_matvec(R, tmp, out);
#pragma unroll
    for (int i = 0; i < 3; ++i) dS[i] = s_const(Sh[0], Sinf[i]) - Sh[i];
    swh_matvec(L, dS, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = tmp[i] * ((lam[i] - s_abs(lam[i])) * 0.5);
    swh_matvec(R, tmp, neg);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = out[i] - neg[i];
  } else {
    const T vn = S[1] / S[0] * nx + S[2] / S[0] * ny;
    out[0] = S[0] - Sh[0];
    out[1] = (S[1] / S[0] - vn * nx) - Sh[1] / Sh[0];
    out[2] = (S[2] / S[0] - vn * ny) - Sh[2] / Sh[0];
  }
}

// out = R w(Lambda) L x at the trace state, without holding L and R (21 entries -- 84 registers as Dual numbers -- that
// the fused element kernel has no room for): the rows of L are applied and dropped one by one.  mode 0: w = |lambda|
// (Roe-like stabilisation), 1: (lambda + |lambda|) / 2 (A+), 2: (lambda - |lambda|) / 2 (A-).  Same arithmetic as
// swh_eigendecomp + swh_matvec, entry by entry.
template <class T>
__device__ __forceinline__ void swh_characteristic_apply(const T *Sh, double nx, double ny, double g, const T *x, int mode, T *out) {
  const T H = Sh[0], ux = Sh[1] / H, uy = Sh[2] / H;
  const T vn = ux * nx + uy * ny, a = s_sqrt(H * g);
  const T i2a = 0.5 / a, ia = 1.0 / a;
  T t0 = (s_const(H, 0.5) - vn * i2a) * x[0] + (i2a * nx) * x[1] + (i2a * ny) * x[2];
  T t1 = ((ux * ny - uy * nx) * ia) * x[0] + (-(ia * ny)) * x[1] + (ia * nx) * x[2];
  T t2 = (s_const(H, 0.5) + vn * i2a) * x[0] + (-(i2a * nx)) * x[1] + (-(i2a * ny)) * x[2];
  const T l0 = vn + a, l1 = vn, l2 = vn - a;
  auto wgt = [&](T l) { return mode == 0 ? s_abs(l) : (mode == 1 ? (l + s_abs(l)) * 0.5 : (l - s_abs(l)) * 0.5); };
  t0 = t0 * wgt(l0);
  t1 = t1 * wgt(l1);
  t2 = t2 * wgt(l2);
  out[0] = t0 + t2;  // (R's first row is 1 0 1)
  out[1] = (ux + a * nx) * t0 + (-(a * ny)) * t1 + (ux - a * nx) * t2;
  out[2] = (uy + a * ny) * t0 + (a * nx) * t1 + (uy - a * ny) * t2;
}

// swh_interface_flux with the characteristic products applied on the fly (fused element kernel)
template <class T>
__device__ __forceinline__ void swh_interface_flux_lean(int side_type, bool roe, const T *S, const T *Sh, const double *Sinf,
                                                        double nx, double ny, double g, T *out) {
  if (side_type == MHA_SWH_FARFIELD) {
    T dS[3] = {S[0] - Sh[0], S[1] - Sh[1], S[2] - Sh[2]}, neg[3];
    swh_characteristic_apply(Sh, nx, ny, g, dS, 1, out);
#pragma unroll
    for (int i = 0; i < 3; ++i) dS[i] = s_const(Sh[0], Sinf[i]) - Sh[i];
    swh_characteristic_apply(Sh, nx, ny, g, dS, 2, neg);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = out[i] - neg[i];
    return;
  }
  if (side_type != MHA_SWH_INTERFACE) {
    swh_boundary_term(side_type, S, Sh, Sinf, nx, ny, g, out);
    return;
  }
  T st[3];
  if (roe) {
    const T dS[3] = {S[0] - Sh[0], S[1] - Sh[1], S[2] - Sh[2]};
    swh_characteristic_apply(Sh, nx, ny, g, dS, 0, st);
  } else {
    swh_stab_term(S, Sh, nx, ny, g, false, st);
  }
  T F[3][2];
  swh_flux_vector(Sh, g, F);
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = F[i][0] * nx + F[i][1] * ny + st[i];
}

// what computeFlux leaves in wkset->flux(elem, eqn, pt)
template <class T>
__device__ __forceinline__ void swh_interface_flux(int side_type, bool roe, const T *S, const T *Sh, const double *Sinf,
                                                   double nx, double ny, double g, T *out) {
  if (side_type != MHA_SWH_INTERFACE) {
    swh_boundary_term(side_type, S, Sh, Sinf, nx, ny, g, out);
    return;
  }
  T F[3][2], st[3];
  swh_flux_vector(Sh, g, F);
  swh_stab_term(S, Sh, nx, ny, g, roe, st);
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = F[i][0] * nx + F[i][1] * ny + st[i];
}

}  // namespace mha
