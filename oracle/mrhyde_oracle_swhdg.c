/*
 * mrhyde_oracle_swhdg.c -- TEST INFRASTRUCTURE ONLY (see mrhyde_oracle.h).
 *
 * Point-level restatement of shallowwaterHybridized (reference: src/physics/shallowwaterHybridized.cpp):
 * flux vectors (:409-480), eigendecomposition of the normal flux Jacobian (:765-823), stabilisation term (:487-588),
 * boundary term (:595-758), matVec.  Pinned by the reference's own unit test values
 * (unit_tests/physics/shallowwaterHybridized.cpp:266-274, 304-315, 339-388, 441-496, 524-572, 609-618).
 * State order: H, Hux[, Huy]; matrices row-major.
 */
#include <math.h>
#include <string.h>

#include "mrhyde_oracle.h"

void orc_swh_matvec(int n, const double *A, const double *x, double *y) {
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += A[i * n + j] * x[j];
    y[i] = s;
  }
}

/* eigendecompFluxJacobian, 1-D (:771-790) and 2-D (:793-823); Shat = (H, Hux[, Huy]) */
void orc_swh_eigendecomp(int dim, const double *Shat, const double *nrm, double g, double *L, double *lam, double *R) {
  const double H = Shat[0], a = sqrt(H * g);
  if (dim == 1) {
    const double u = Shat[1] / H;
    R[0] = 1.0; R[2] = u - a;
    R[1] = 1.0; R[3] = u + a;
    L[0] = (u + a) / (2.0 * a); L[1] = -1.0 / (2.0 * a);
    L[2] = (a - u) / (2.0 * a); L[3] = 1.0 / (2.0 * a);
    lam[0] = u - a; lam[1] = u + a;
  } else {
    const double ux = Shat[1] / H, uy = Shat[2] / H, nx = nrm[0], ny = nrm[1];
    const double vn = ux * nx + uy * ny;
    R[0] = 1.0; R[3] = ux + a * nx; R[6] = uy + a * ny;
    R[1] = 0.0; R[4] = -a * ny;     R[7] = a * nx;
    R[2] = 1.0; R[5] = ux - a * nx; R[8] = uy - a * ny;
    L[0] = 0.5 - vn / (2.0 * a); L[1] = nx / (2.0 * a);  L[2] = ny / (2.0 * a);
    L[3] = (ny * ux - nx * uy) / a; L[4] = -ny / a;      L[5] = nx / a;
    L[6] = 0.5 + vn / (2.0 * a); L[7] = -nx / (2.0 * a); L[8] = -ny / (2.0 * a);
    lam[0] = vn + a; lam[1] = vn; lam[2] = vn - a;
  }
}

/* computeFluxVector (:409-480): F[eqn][dir] of state S */
void orc_swh_flux_vector(int dim, const double *S, double g, double *F) {
  const double H = S[0], Hux = S[1];
  if (dim == 1) {
    F[0] = Hux;
    F[1] = Hux * Hux / H + 0.5 * H * H * g;
  } else {
    const double Huy = S[2];
    F[0] = Hux; F[1] = Huy;
    F[2] = Hux * Hux / H + 0.5 * H * H * g; F[3] = Hux * Huy / H;
    F[4] = Hux * Huy / H; F[5] = Huy * Huy / H + 0.5 * H * H * g;
  }
}

/* computeStabilizationTerm (:487-588): Stab (S - Shat); roe != 0: R |Lambda| L, else lambda_max I */
void orc_swh_stab_term(int dim, const double *S, const double *Shat, const double *nrm, double g, int roe, double *out) {
  const int nv = dim + 1;
  double dS[3], L[9], lam[3], R[9], tmp[3];
  for (int i = 0; i < nv; ++i) dS[i] = S[i] - Shat[i];
  if (roe) {
    orc_swh_eigendecomp(dim, Shat, nrm, g, L, lam, R);
    orc_swh_matvec(nv, L, dS, tmp);
    for (int i = 0; i < nv; ++i) tmp[i] *= fabs(lam[i]);
    orc_swh_matvec(nv, R, tmp, out);
  } else {
    double vn = nrm[0] * Shat[1] / Shat[0];
    const double a = sqrt(Shat[0] * g);
    if (dim > 1) vn += nrm[1] * Shat[2] / Shat[0];
    const double lmax = fmax(fabs(vn + a), fabs(vn - a));
    for (int i = 0; i < nv; ++i) out[i] = dS[i] * lmax;
  }
}

/* computeBoundaryTerm (:595-758): type 1 = Far-field: A+ (S - Shat) - A- (Sinf - Shat); 2 = Slip */
void orc_swh_boundary_term(int dim, int type, const double *S, const double *Shat, const double *Sinf, const double *nrm,
                           double g, double *out) {
  const int nv = dim + 1;
  if (type == 1) {
    double dS[3], L[9], lam[3], R[9], tmp[3], neg[3];
    orc_swh_eigendecomp(dim, Shat, nrm, g, L, lam, R);
    for (int i = 0; i < nv; ++i) dS[i] = S[i] - Shat[i];
    orc_swh_matvec(nv, L, dS, tmp);
    for (int i = 0; i < nv; ++i) tmp[i] *= (lam[i] + fabs(lam[i])) / 2.0;
    orc_swh_matvec(nv, R, tmp, out);
    for (int i = 0; i < nv; ++i) dS[i] = Sinf[i] - Shat[i];
    orc_swh_matvec(nv, L, dS, tmp);
    for (int i = 0; i < nv; ++i) tmp[i] *= (lam[i] - fabs(lam[i])) / 2.0;
    orc_swh_matvec(nv, R, tmp, neg);
    for (int i = 0; i < nv; ++i) out[i] -= neg[i];
  } else {
    double vn = nrm[0] * S[1] / S[0];
    if (dim > 1) vn += nrm[1] * S[2] / S[0];
    out[0] = S[0] - Shat[0];
    out[1] = (S[1] / S[0] - vn * nrm[0]) - Shat[1] / Shat[0];
    if (dim > 1) out[2] = (S[2] / S[0] - vn * nrm[1]) - Shat[2] / Shat[0];
  }
}

/* computeFlux (:270-368): interface: F(Shat).n + Stab (S - Shat); Far-field / Slip: the boundary term */
void orc_swh_interface_flux(int dim, int side_type, int roe, const double *S, const double *Shat, const double *Sinf,
                            const double *nrm, double g, double *out) {
  const int nv = dim + 1;
  if (side_type != 0) {
    orc_swh_boundary_term(dim, side_type, S, Shat, Sinf, nrm, g, out);
    return;
  }
  double F[6], st[3];
  orc_swh_flux_vector(dim, Shat, g, F);
  orc_swh_stab_term(dim, S, Shat, nrm, g, roe, st);
  for (int i = 0; i < nv; ++i) {
    double fn = 0.0;
    for (int d = 0; d < dim; ++d) fn += F[i * dim + d] * nrm[d];
    out[i] = fn + st[i];
  }
}
