/*
 * mrhyde_oracle.c -- TEST INFRASTRUCTURE ONLY (see mrhyde_oracle.h).
 *
 * CPU restatement of the reference's element-local assembly, following the
 * reference's own data flow kernel by kernel:
 *   gather -> seed -> reset res -> field evaluation (one pass per field)
 *   -> source evaluation -> thermal volumeResidual with width-W forward-mode
 *   derivative arrays -> scatter (-res.val, +res.dx) with fixed-row skip.
 * Citations are file:line under /root/reference.
 */
#include "mrhyde_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "orc_internal.h"


/* ------------------------------------------------------------------------ */
/* reference tables                                                          */
/* ------------------------------------------------------------------------ */

int orc_gauss_npts(int degree) { return (degree + 2) / 2; /* ceil((degree+1)/2) */ }

/* Gauss-Legendre on [-1,1], points DESCENDING (pinned by HGRAD gold: point 0
 * of the 2-point rule is +0.57735, regression/discretization/HGRAD/mrhyde.gold:89) */
void orc_gauss_line(int n, double *pts, double *wts) {
  for (int i = 0; i < n; ++i) {
    double x = cos(M_PI * (i + 0.75) / (n + 0.5));
    double dp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p0 = 1.0, p1 = x;
      for (int k = 2; k <= n; ++k) {
        double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
        p0 = p1; p1 = pk;
      }
      if (n == 1) { p0 = 1.0; p1 = x; }
      dp = n * (x * p1 - p0) / (x * x - 1.0);
      double dx = p1 / dp;
      x -= dx;
      if (fabs(dx) < 1e-16) break;
    }
    /* recompute derivative at the converged root */
    {
      double p0 = 1.0, p1 = x;
      for (int k = 2; k <= n; ++k) {
        double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
        p0 = p1; p1 = pk;
      }
      dp = n * (x * p1 - p0) / (x * x - 1.0);
    }
    pts[i] = x;
    wts[i] = 2.0 / ((1.0 - x * x) * dp * dp);
  }
  /* symmetrise exactly */
  for (int i = 0; i < n / 2; ++i) {
    double a = 0.5 * (pts[i] - pts[n - 1 - i]);
    pts[i] = a; pts[n - 1 - i] = -a;
    double w = 0.5 * (wts[i] + wts[n - 1 - i]);
    wts[i] = w; wts[n - 1 - i] = w;
  }
  if (n % 2) pts[n / 2] = 0.0;
}

/* Lagrange basis of order p on equispaced nodes x_k = -1 + 2k/p
 * (Basis_HGRAD_LINE_Cn_FEM, POINTTYPE_EQUISPACED; discretizationInterface.cpp:356) */
void orc_lagrange_1d(int p, double x, double *val, double *der) {
  double xn[ORC_MAXP + 1];
  for (int k = 0; k <= p; ++k) xn[k] = -1.0 + 2.0 * k / p;
  for (int k = 0; k <= p; ++k) {
    double v = 1.0;
    for (int m = 0; m <= p; ++m) if (m != k) v *= (x - xn[m]) / (xn[k] - xn[m]);
    val[k] = v;
    double d = 0.0;
    for (int j = 0; j <= p; ++j) {
      if (j == k) continue;
      double t = 1.0 / (xn[k] - xn[j]);
      for (int m = 0; m <= p; ++m) if (m != k && m != j) t *= (x - xn[m]) / (xn[k] - xn[m]);
      d += t;
    }
    der[k] = d;
  }
}

int orc__ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }
#define ipow orc__ipow

/* shards vertex order (Quadrilateral_4 / Hexahedron_8) */
const double orc__quad_node[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
const double orc__hex_node[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                      {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};

int orc_ref_sizes(int dim, int order, int qdeg, int *nbasis, int *nq, int *nnodes) {
  if (dim < 2 || dim > 3 || order < 1 || order > ORC_MAXP) return -1;
  *nbasis = ipow(order + 1, dim);
  *nq = ipow(orc_gauss_npts(qdeg), dim);
  *nnodes = 1 << dim;
  return 0;
}

int orc_ref_tables(int dim, int order, int qdeg, double *ip, double *wts, double *basis,
                   double *grad, double *nodeval, double *nodegrad) {
  int nb, nq, nn;
  if (orc_ref_sizes(dim, order, qdeg, &nb, &nq, &nn)) return -1;
  int nq1 = orc_gauss_npts(qdeg), p1 = order + 1;
  double gp[64], gw[64];
  if (nq1 > 64) return -1;
  orc_gauss_line(nq1, gp, gw);
  for (int q = 0; q < nq; ++q) {
    int qi[3] = {q % nq1, (q / nq1) % nq1, q / (nq1 * nq1)};
    double x[3] = {0, 0, 0}, w = 1.0;
    for (int d = 0; d < dim; ++d) { x[d] = gp[qi[d]]; w *= gw[qi[d]]; }
    for (int d = 0; d < dim; ++d) ip[q * dim + d] = x[d];
    wts[q] = w;
    double v[3][ORC_MAXP + 1], dv[3][ORC_MAXP + 1];
    for (int d = 0; d < dim; ++d) orc_lagrange_1d(order, x[d], v[d], dv[d]);
    for (int f = 0; f < nb; ++f) {
      int fi[3] = {f % p1, (f / p1) % p1, f / (p1 * p1)};
      double val = 1.0;
      for (int d = 0; d < dim; ++d) val *= v[d][fi[d]];
      basis[f * nq + q] = val;
      for (int d = 0; d < dim; ++d) {
        double g = 1.0;
        for (int e = 0; e < dim; ++e) g *= (e == d) ? dv[e][fi[e]] : v[e][fi[e]];
        grad[(f * nq + q) * dim + d] = g;
      }
    }
    for (int n = 0; n < nn; ++n) {
      const double *s = (dim == 2) ? QUAD_NODE[n] : HEX_NODE[n];
      double val = 1.0;
      for (int d = 0; d < dim; ++d) val *= 0.5 * (1.0 + s[d] * x[d]);
      nodeval[n * nq + q] = val;
      for (int d = 0; d < dim; ++d) {
        double g = 1.0;
        for (int e = 0; e < dim; ++e) g *= (e == d) ? 0.5 * s[e] : 0.5 * (1.0 + s[e] * x[e]);
        nodegrad[(n * nq + q) * dim + d] = g;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* structured mesh + dof map                                                 */
/* ------------------------------------------------------------------------ */

int orc_mesh_sizes(int dim, int order, const int *nc, int *nverts, int *nelem, long long *ndof) {
  if (dim < 2 || dim > 3) return -1;
  long long nv = 1, ne = 1, nd = 1;
  for (int d = 0; d < dim; ++d) { nv *= nc[d] + 1; ne *= nc[d]; nd *= (long long)order * nc[d] + 1; }
  *nverts = (int)nv; *nelem = (int)ne; *ndof = nd;
  return 0;
}

int orc_mesh_structured(int dim, int order, const int *nc, const double *lo, const double *hi,
                        double *verts, int *cell2vert, int *lids, int *offsets,
                        unsigned char *boundary_dof) {
  if (dim < 2 || dim > 3) return -1;
  int p = order, p1 = order + 1, n = ipow(p1, dim), nn = 1 << dim;
  int nx = nc[0], ny = nc[1], nz = (dim == 3) ? nc[2] : 1;
  /* vertices: simplemeshmanager.hpp:639-657 (X0 + i*dx, dx = width/nx) */
  double dx = (hi[0] - lo[0]) / nx, dy = (hi[1] - lo[1]) / ny, dz = (dim == 3) ? (hi[2] - lo[2]) / nz : 0.0;
  int vct = 0;
  for (int k = 0; k <= ((dim == 3) ? nz : 0); ++k)
    for (int j = 0; j <= ny; ++j)
      for (int i = 0; i <= nx; ++i) {
        verts[vct * dim + 0] = lo[0] + i * dx;
        verts[vct * dim + 1] = lo[1] + j * dy;
        if (dim == 3) verts[vct * dim + 2] = lo[2] + k * dz;
        ++vct;
      }
  /* offsets: LID list = shards-ordered vertices, then remaining tensor dofs */
  int *pos = offsets;
  for (int t = 0; t < n; ++t) pos[t] = -1;
  for (int v = 0; v < nn; ++v) {
    const double *s = (dim == 2) ? QUAD_NODE[v] : HEX_NODE[v];
    int t = 0, mul = 1;
    for (int d = 0; d < dim; ++d) { t += mul * ((s[d] > 0) ? p : 0); mul *= p1; }
    pos[t] = v;
  }
  {
    int next = nn;
    for (int t = 0; t < n; ++t) if (pos[t] < 0) pos[t] = next++;
  }
  long long ndx = (long long)p * nx + 1, ndy = (long long)p * ny + 1;
  int ect = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        /* cell -> vertex: simplemeshmanager.hpp:659-675 (counter-clockwise) */
        for (int v = 0; v < nn; ++v) {
          const double *s = (dim == 2) ? QUAD_NODE[v] : HEX_NODE[v];
          int vi = i + (s[0] > 0), vj = j + (s[1] > 0), vk = (dim == 3) ? k + (s[2] > 0) : 0;
          cell2vert[ect * nn + v] = (vk * (ny + 1) + vj) * (nx + 1) + vi;
        }
        for (int t = 0; t < n; ++t) {
          int a = t % p1, b = (t / p1) % p1, c = t / (p1 * p1);
          long long gi = (long long)p * i + a, gj = (long long)p * j + b, gk = (dim == 3) ? (long long)p * k + c : 0;
          lids[ect * n + pos[t]] = (int)((gk * ndy + gj) * ndx + gi);
        }
        ++ect;
      }
  if (boundary_dof) {
    long long ndz = (dim == 3) ? (long long)p * nz + 1 : 1;
    for (long long gk = 0; gk < ndz; ++gk)
      for (long long gj = 0; gj < ndy; ++gj)
        for (long long gi = 0; gi < ndx; ++gi) {
          int b = (gi == 0 || gi == ndx - 1 || gj == 0 || gj == ndy - 1);
          if (dim == 3) b = b || gk == 0 || gk == ndz - 1;
          boundary_dof[(gk * ndy + gj) * ndx + gi] = (unsigned char)b;
        }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* physical basis (discretizationInterface.cpp:732-776, 898-981)             */
/* ------------------------------------------------------------------------ */

void orc__jac_inv_det(int dim, const double *J, double *Ji, double *det) {
  if (dim == 2) {
    double d = J[0] * J[3] - J[1] * J[2];
    *det = d;
    Ji[0] = J[3] / d; Ji[1] = -J[1] / d; Ji[2] = -J[2] / d; Ji[3] = J[0] / d;
  } else {
    double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    double d = J[0] * c00 + J[1] * c01 + J[2] * c02;
    *det = d;
    Ji[0] = c00 / d; Ji[1] = (J[2] * J[7] - J[1] * J[8]) / d; Ji[2] = (J[1] * J[5] - J[2] * J[4]) / d;
    Ji[3] = c01 / d; Ji[4] = (J[0] * J[8] - J[2] * J[6]) / d; Ji[5] = (J[2] * J[3] - J[0] * J[5]) / d;
    Ji[6] = c02 / d; Ji[7] = (J[1] * J[6] - J[0] * J[7]) / d; Ji[8] = (J[0] * J[4] - J[1] * J[3]) / d;
  }
}

int orc_physical_basis(int dim, int order, int qdeg, int nelem, const double *nodes,
                       double *basis, double *basis_grad, double *wts, double *ip) {
  int nb, nq, nn;
  if (orc_ref_sizes(dim, order, qdeg, &nb, &nq, &nn)) return -1;
  double *rip = malloc(sizeof(double) * nq * dim), *rw = malloc(sizeof(double) * nq);
  double *rb = malloc(sizeof(double) * nb * nq), *rg = malloc(sizeof(double) * nb * nq * dim);
  double *nv = malloc(sizeof(double) * nn * nq), *ng = malloc(sizeof(double) * nn * nq * dim);
  orc_ref_tables(dim, order, qdeg, rip, rw, rb, rg, nv, ng);
#pragma omp parallel for schedule(static)
  for (int e = 0; e < nelem; ++e) {
    const double *xn = nodes + (size_t)e * nn * dim;
    for (int q = 0; q < nq; ++q) {
      double J[9] = {0}, Ji[9] = {0}, det;
      /* CellTools::setJacobian: J(row,col) = sum_node x(node,row) * dN(node,col) */
      for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
          double s = 0.0;
          for (int n = 0; n < nn; ++n) s += xn[n * dim + r] * ng[(n * nq + q) * dim + c];
          J[r * dim + c] = s;
        }
      jac_inv_det(dim, J, Ji, &det);
      if (wts) wts[(size_t)e * nq + q] = rw[q] * det; /* computeCellMeasure */
      if (ip)
        for (int d = 0; d < dim; ++d) { /* mapToPhysicalFrame */
          double s = 0.0;
          for (int n = 0; n < nn; ++n) s += xn[n * dim + d] * nv[n * nq + q];
          ip[((size_t)e * nq + q) * dim + d] = s;
        }
      for (int f = 0; f < nb; ++f) {
        if (basis) basis[((size_t)e * nb + f) * nq + q] = rb[f * nq + q]; /* HGRADtransformVALUE */
        if (basis_grad)
          for (int d = 0; d < dim; ++d) { /* HGRADtransformGRAD: J^{-T} grad_ref */
            double s = 0.0;
            for (int k = 0; k < dim; ++k) s += Ji[k * dim + d] * rg[(f * nq + q) * dim + k];
            basis_grad[(((size_t)e * nb + f) * nq + q) * dim + d] = s;
          }
      }
    }
  }
  free(rip); free(rw); free(rb); free(rg); free(nv); free(ng);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* CRS graph (linearAlgebraInterface.cpp:218-229)                            */
/* ------------------------------------------------------------------------ */

static int cmp_int(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

int orc_build_graph(int nrows, int nelem, int n, const int *lids, int *rowptr, int *colind) {
  /* row -> incident elements */
  int *cnt = calloc((size_t)nrows + 1, sizeof(int));
  for (size_t k = 0; k < (size_t)nelem * n; ++k) cnt[lids[k] + 1]++;
  for (int r = 0; r < nrows; ++r) cnt[r + 1] += cnt[r];
  int *inc = malloc(sizeof(int) * (size_t)nelem * n);
  int *fill = calloc((size_t)nrows, sizeof(int));
  for (int e = 0; e < nelem; ++e)
    for (int i = 0; i < n; ++i) {
      int r = lids[(size_t)e * n + i];
      inc[cnt[r] + fill[r]++] = e;
    }
  int maxinc = 0;
  for (int r = 0; r < nrows; ++r) if (cnt[r + 1] - cnt[r] > maxinc) maxinc = cnt[r + 1] - cnt[r];
  int *buf = malloc(sizeof(int) * (size_t)(maxinc > 0 ? maxinc : 1) * n);
  rowptr[0] = 0;
  for (int r = 0; r < nrows; ++r) {
    int m = 0;
    for (int k = cnt[r]; k < cnt[r + 1]; ++k) {
      if (k > cnt[r] && inc[k] == inc[k - 1]) continue; /* element listing r twice */
      const int *l = lids + (size_t)inc[k] * n;
      for (int j = 0; j < n; ++j) buf[m++] = l[j];
    }
    qsort(buf, m, sizeof(int), cmp_int);
    int u = 0;
    for (int k = 0; k < m; ++k) if (k == 0 || buf[k] != buf[k - 1]) {
      if (colind) colind[rowptr[r] + u] = buf[k];
      ++u;
    }
    rowptr[r + 1] = rowptr[r] + u;
  }
  free(cnt); free(inc); free(fill); free(buf);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* thermal assembly                                                          */
/* ------------------------------------------------------------------------ */

/* AD width choice: assemblyManager.cpp:131-169 */
int orc_ad_width(int n) {
  static const int w[] = {2, 4, 8, 16, 18, 24, 32};
  for (int i = 0; i < 7; ++i) if (n <= w[i]) return w[i];
  return n <= 64 ? 64 : n; /* MAXDERIVS = 64 default; larger needs a rebuild of the reference */
}

static double eval_source(const orc_thermal_args *a, size_t e, int q, int nq, const double *x) {
  switch (a->source_kind) {
    case 1: return a->source_ip[e * nq + q];
    case 2: {
      double s = a->source_amp;
      for (int d = 0; d < a->dim; ++d) s *= sin(a->source_freq[d] * x[d]);
      return s;
    }
    default: return a->source_amp;
  }
}

int orc_assemble_thermal(const orc_thermal_args *a) {
  int n, nq, nn;
  if (orc_ref_sizes(a->dim, a->order, a->qdeg, &n, &nq, &nn)) return -1;
  const int dim = a->dim, W = orc_ad_width(n), W1 = W + 1;
  const int ws = a->workset_size > 0 ? a->workset_size : a->nelem; /* assemblyManager.cpp:326-332 */
  int nthreads = a->num_threads > 1 ? a->num_threads : 1;
#ifndef _OPENMP
  nthreads = 1;
#endif
  const int use_atomics = nthreads > 1; /* assemblyManager.cpp:4058-4061 */
  const int nfields = 2 + dim;          /* e, e_t, grad(e)[x..z] */

  double *usol = malloc(sizeof(double) * (size_t)ws * n);        /* gathered u      */
  double *uAD = malloc(sizeof(double) * (size_t)ws * n * W1);    /* seeded sol_vals */
  double *udotAD = malloc(sizeof(double) * (size_t)ws * n * W1);
  double *fld = malloc(sizeof(double) * (size_t)nfields * ws * nq * W1);
  double *src = malloc(sizeof(double) * (size_t)ws * nq);
  double *res = malloc(sizeof(double) * (size_t)ws * n * W1);
  if (!usol || !uAD || !udotAD || !fld || !src || !res) return -2;

  /* groups processed sequentially: assemblyManager.cpp:2355-2357, 514-529 */
  for (int e0 = 0; e0 < a->nelem; e0 += ws) {
    const int ne = (e0 + ws <= a->nelem) ? ws : a->nelem - e0;

    /* "assembly gather": assemblyManager.cpp:3633-3641 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int el = 0; el < ne; ++el) {
      const int *L = a->lids + (size_t)(e0 + el) * n;
      for (int dof = 0; dof < n; ++dof) usol[el * n + dof] = a->u[L[a->offsets[dof]]];
    }

    /* seeding: workset.cpp:836-847 (steady), 589-623 (transient, seedwhat 1) */
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int el = 0; el < ne; ++el) {
      const int *L = a->lids + (size_t)(e0 + el) * n;
      for (int dof = 0; dof < n; ++dof) {
        double *ua = uAD + ((size_t)el * n + dof) * W1, *ud = udotAD + ((size_t)el * n + dof) * W1;
        memset(ua, 0, sizeof(double) * W1);
        memset(ud, 0, sizeof(double) * W1);
        const int off = a->offsets[dof];
        const double cu = usol[el * n + dof];
        if (!a->transient) {
          if (a->compute_jacobian) { ua[0] = cu; ua[1 + off] = 1.0; }
          else ua[0] = cu;
        } else {
          const int st = a->stage, S = a->nstages, NS = a->nsteps;
          const int row = L[off];
          const double *cu_prev = a->u_prev + (size_t)row * NS;
          const double *cu_stage = a->u_stage + (size_t)row * S;
          double alpha_u = a->butcher_A[st * S + st] / a->butcher_b[st];
          double timewt = 1.0 / a->dt / a->butcher_b[st];
          double alpha_t = a->bdf[0] * timewt;
          double beta_u = (1.0 - alpha_u) * cu_prev[0];
          for (int s = 0; s < st; ++s) beta_u += a->butcher_A[st * S + s] / a->butcher_b[s] * (cu_stage[s] - cu_prev[0]);
          double beta_t = 0.0;
          for (int s = 1; s < NS + 1; ++s) beta_t += a->bdf[s] * cu_prev[s - 1];
          beta_t *= timewt;
          ua[0] = alpha_u * cu + beta_u;
          ud[0] = alpha_t * cu + beta_t;
          if (a->compute_jacobian) { ua[1 + off] = alpha_u * 1.0; ud[1 + off] = alpha_t * 1.0; }
        }
      }
    }

    /* "wkset reset res": workset.cpp:449-459 */
    memset(res, 0, sizeof(double) * (size_t)ne * n * W1);

    /* field evaluation, one pass per field ("wkset soln ip HGRAD",
     * workset.cpp:1044-1056): f(e,pt) = sum_dof u_AD(e,dof)*basis(e,dof,pt,comp).
     * field 0 = e, 1 = e_t (stays 0 in steady runs, workset.cpp:942-945),
     * 2.. = grad(e)[x],[y],[z] */
    for (int f = 0; f < nfields; ++f) {
      double *F = fld + (size_t)f * ws * nq * W1;
      if (f == 1 && !a->transient) { memset(F, 0, sizeof(double) * (size_t)ne * nq * W1); continue; }
      const double *sv = (f == 1) ? udotAD : uAD;
#pragma omp parallel for num_threads(nthreads) schedule(static)
      for (int el = 0; el < ne; ++el) {
        const size_t e = (size_t)e0 + el;
        for (int pt = 0; pt < nq; ++pt) {
          double *o = F + ((size_t)el * nq + pt) * W1;
          for (int dof = 0; dof < n; ++dof) {
            double b = (f < 2) ? a->basis[(e * n + dof) * nq + pt]
                               : a->basis_grad[((e * n + dof) * nq + pt) * dim + (f - 2)];
            const double *s = sv + ((size_t)el * n + dof) * W1;
            if (dof == 0) for (int k = 0; k < W1; ++k) o[k] = s[k] * b;
            else for (int k = 0; k < W1; ++k) o[k] += s[k] * b;
          }
        }
      }
    }

    /* functionManager->evaluate("thermal source","ip"): thermal.cpp:82 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int el = 0; el < ne; ++el)
      for (int pt = 0; pt < nq; ++pt)
        src[el * nq + pt] = eval_source(a, (size_t)e0 + el, pt, nq, a->ip + (((size_t)e0 + el) * nq + pt) * dim);

    /* "Thermal volume resid 3D part 1": thermal.cpp:125-163 */
    {
      const double *T_t = fld + (size_t)1 * ws * nq * W1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
      for (int el = 0; el < ne; ++el) {
        const size_t e = (size_t)e0 + el;
        for (int dof = 0; dof < n; ++dof) {
          double *r = res + ((size_t)el * n + a->offsets[dof]) * W1;
          for (int pt = 0; pt < nq; ++pt) {
            const double w = a->wts[e * nq + pt];
            const double kap = a->diff_ip ? a->diff_ip[e * nq + pt] : a->diff;
            const double bv = a->basis[(e * n + dof) * nq + pt];
            const double *dTdt = T_t + ((size_t)el * nq + pt) * W1;
            const double rc = a->rho * a->cp;
            r[0] += (rc * dTdt[0] - src[el * nq + pt]) * w * bv;
            for (int k = 1; k < W1; ++k) r[k] += (rc * dTdt[k]) * w * bv;
            for (int d = 0; d < dim; ++d) {
              const double *g = fld + (size_t)(2 + d) * ws * nq * W1 + ((size_t)el * nq + pt) * W1;
              const double bg = a->basis_grad[((e * n + dof) * nq + pt) * dim + d];
              for (int k = 0; k < W1; ++k) r[k] += kap * g[k] * w * bg;
            }
            if (a->have_advection) { /* thermal.cpp:150-160 */
              for (int d = 0; d < dim; ++d) {
                const double *g = fld + (size_t)(2 + d) * ws * nq * W1 + ((size_t)el * nq + pt) * W1;
                const double b = a->adv_ip ? a->adv_ip[(e * nq + pt) * dim + d] : a->adv[d];
                for (int k = 0; k < W1; ++k) r[k] += b * g[k] * w * bv;
              }
            }
          }
        }
      }
    }

    /* dense updateJac / updateRes (assemblyManager.cpp:7438-7451, 7140-7149) */
    if (a->local_J || a->local_res) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
      for (int el = 0; el < ne; ++el) {
        const size_t e = (size_t)e0 + el;
        for (int j = 0; j < n; ++j) {
          const int row = a->offsets[j];
          const double *r = res + ((size_t)el * n + row) * W1;
          if (a->local_res) a->local_res[e * n + row] -= r[0];
          if (a->local_J && a->compute_jacobian)
            for (int k = 0; k < n; ++k) {
              const int col = a->offsets[k];
              a->local_J[(e * n + row) * n + col] += r[1 + col];
            }
        }
      }
    }

    /* fused scatter "assembly insert Jac": assemblyManager.cpp:4063-4144 */
    if (a->res || a->crs_vals) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
      for (int el = 0; el < ne; ++el) {
        const int *L = a->lids + (size_t)(e0 + el) * n;
        for (int j = 0; j < n; ++j) {
          const int row = a->offsets[j];
          const int rowIndex = L[row];
          if (a->fixed && a->fixed[rowIndex]) continue;
          const double *r = res + ((size_t)el * n + row) * W1;
          if (a->res) {
            const double val = -r[0];
            if (use_atomics) {
#pragma omp atomic
              a->res[rowIndex] += val;
            } else a->res[rowIndex] += val;
          }
          if (a->crs_vals && a->compute_jacobian) {
            /* KokkosSparse sumIntoValues(row, cols, n, vals, isSorted=false, atomics) */
            const int rb = a->rowptr[rowIndex], re = a->rowptr[rowIndex + 1];
            int hint = rb;
            for (int k = 0; k < n; ++k) {
              const int col = a->offsets[k];
              const int gcol = L[col];
              const double val = r[1 + col];
              int p = hint;
              if (!(p < re && a->colind[p] == gcol)) {
                for (p = rb; p < re; ++p) if (a->colind[p] == gcol) break;
              }
              if (p < re) {
                if (use_atomics) {
#pragma omp atomic
                  a->crs_vals[p] += val;
                } else a->crs_vals[p] += val;
                hint = p + 1;
              }
            }
          }
        }
      }
    }
  }
  free(usol); free(uAD); free(udotAD); free(fld); free(src); free(res);
  return 0;
}

int orc_apply_dbc_diag(int nrows, const unsigned char *fixed, const int *rowptr, const int *colind,
                       double *crs_vals) {
  for (int r = 0; r < nrows; ++r) {
    if (!fixed[r]) continue;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p)
      if (colind[p] == r) crs_vals[p] = 1.0; /* replaceLocalValues(dof, 1, &one, &dof) */
  }
  return 0;
}

double orc_l2_error_sinprod(int dim, int order, int qdeg, int nelem, const int *lids,
                            const int *offsets, const double *basis, const double *wts,
                            const double *ip, const double *u, const double *freq) {
  int n, nq, nn;
  if (orc_ref_sizes(dim, order, qdeg, &n, &nq, &nn)) return -1.0;
  double tot = 0.0;
  for (size_t e = 0; e < (size_t)nelem; ++e)
    for (int pt = 0; pt < nq; ++pt) {
      double uh = 0.0;
      for (int dof = 0; dof < n; ++dof) uh += u[lids[e * n + offsets[dof]]] * basis[(e * n + dof) * nq + pt];
      double ut = 1.0;
      for (int d = 0; d < dim; ++d) ut *= sin(freq[d] * ip[(e * nq + pt) * dim + d]);
      double diff = uh - ut;
      tot += diff * diff * wts[e * nq + pt];
    }
  return sqrt(tot);
}

/* ------------------------------------------------------------------------ */
/* boundary (side) terms                                                     */
/* ------------------------------------------------------------------------ */

const int orc__quad_side[4][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}};
const int orc__hex_side[6][4] = {{0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {0, 4, 7, 3}, {0, 3, 2, 1}, {4, 5, 6, 7}};

int orc_side_sizes(int dim, int qdeg, int *nsides, int *nqs) {
  if (dim < 2 || dim > 3) return -1;
  *nsides = 2 * dim;
  *nqs = ipow(orc_gauss_npts(qdeg), dim - 1);
  return 0;
}

int orc_side_tables(int dim, int order, int qdeg, double *sip, double *swts, double *tanU, double *tanV,
                    double *sbasis, double *sgrad, double *snodeval, double *snodegrad) {
  int nb, nq, nn, ns, nqs;
  if (orc_ref_sizes(dim, order, qdeg, &nb, &nq, &nn) || orc_side_sizes(dim, qdeg, &ns, &nqs)) return -1;
  const int nq1 = orc_gauss_npts(qdeg), p1 = order + 1;
  double gp[64], gw[64];
  orc_gauss_line(nq1, gp, gw);
  for (int q = 0; q < nqs; ++q) swts[q] = (dim == 2) ? gw[q] : gw[q % nq1] * gw[q / nq1];
  for (int s = 0; s < ns; ++s) {
    double v[4][3] = {{0}};
    const int nsv = (dim == 2) ? 2 : 4;
    for (int k = 0; k < nsv; ++k) {
      const double *c = (dim == 2) ? QUAD_NODE[QUAD_SIDE[s][k]] : HEX_NODE[HEX_SIDE[s][k]];
      for (int d = 0; d < dim; ++d) v[k][d] = c[d];
    }
    for (int d = 0; d < dim; ++d) {
      if (dim == 2) {
        tanU[s * dim + d] = 0.5 * (v[1][d] - v[0][d]); /* getReferenceEdgeTangent */
        tanV[s * dim + d] = 0.0;
      } else {                                         /* getReferenceFaceTangents */
        tanU[s * dim + d] = 0.25 * (-v[0][d] + v[1][d] + v[2][d] - v[3][d]);
        tanV[s * dim + d] = 0.25 * (-v[0][d] - v[1][d] + v[2][d] + v[3][d]);
      }
    }
    for (int q = 0; q < nqs; ++q) {
      double x[3] = {0, 0, 0};
      if (dim == 2) {
        const double t = gp[q];
        for (int d = 0; d < dim; ++d) x[d] = 0.5 * (1 - t) * v[0][d] + 0.5 * (1 + t) * v[1][d];
      } else {
        const double a = gp[q % nq1], b = gp[q / nq1];
        for (int d = 0; d < dim; ++d)
          x[d] = 0.25 * ((1 - a) * (1 - b) * v[0][d] + (1 + a) * (1 - b) * v[1][d] + (1 + a) * (1 + b) * v[2][d] +
                         (1 - a) * (1 + b) * v[3][d]);
      }
      for (int d = 0; d < dim; ++d) sip[(s * nqs + q) * dim + d] = x[d];
      double bv[3][ORC_MAXP + 1], bd[3][ORC_MAXP + 1];
      for (int d = 0; d < dim; ++d) orc_lagrange_1d(order, x[d], bv[d], bd[d]);
      for (int f = 0; f < nb; ++f) {
        int fi[3] = {f % p1, (f / p1) % p1, f / (p1 * p1)};
        double val = 1.0;
        for (int d = 0; d < dim; ++d) val *= bv[d][fi[d]];
        sbasis[(s * nb + f) * nqs + q] = val;
        for (int d = 0; d < dim; ++d) {
          double g = 1.0;
          for (int e = 0; e < dim; ++e) g *= (e == d) ? bd[e][fi[e]] : bv[e][fi[e]];
          sgrad[((s * nb + f) * nqs + q) * dim + d] = g;
        }
      }
      for (int n = 0; n < nn; ++n) {
        const double *c = (dim == 2) ? QUAD_NODE[n] : HEX_NODE[n];
        double val = 1.0;
        for (int d = 0; d < dim; ++d) val *= 0.5 * (1.0 + c[d] * x[d]);
        snodeval[(s * nn + n) * nqs + q] = val;
        for (int d = 0; d < dim; ++d) {
          double g = 1.0;
          for (int e = 0; e < dim; ++e) g *= (e == d) ? 0.5 * c[e] : 0.5 * (1.0 + c[e] * x[e]);
          snodegrad[((s * nn + n) * nqs + q) * dim + d] = g;
        }
      }
    }
  }
  return 0;
}

typedef struct {
  int nb, nq, nn, ns, nqs;
  double *sip, *swts, *tanU, *tanV, *sbasis, *sgrad, *snv, *sng;
} side_tab;

static int side_tab_make(int dim, int order, int qdeg, side_tab *t) {
  if (orc_ref_sizes(dim, order, qdeg, &t->nb, &t->nq, &t->nn) || orc_side_sizes(dim, qdeg, &t->ns, &t->nqs)) return -1;
  t->sip = malloc(sizeof(double) * t->ns * t->nqs * dim);
  t->swts = malloc(sizeof(double) * t->nqs);
  t->tanU = malloc(sizeof(double) * t->ns * dim);
  t->tanV = malloc(sizeof(double) * t->ns * dim);
  t->sbasis = malloc(sizeof(double) * t->ns * t->nb * t->nqs);
  t->sgrad = malloc(sizeof(double) * t->ns * t->nb * t->nqs * dim);
  t->snv = malloc(sizeof(double) * t->ns * t->nn * t->nqs);
  t->sng = malloc(sizeof(double) * t->ns * t->nn * t->nqs * dim);
  return orc_side_tables(dim, order, qdeg, t->sip, t->swts, t->tanU, t->tanV, t->sbasis, t->sgrad, t->snv, t->sng);
}

static void side_tab_free(side_tab *t) {
  free(t->sip); free(t->swts); free(t->tanU); free(t->tanV); free(t->sbasis); free(t->sgrad); free(t->snv); free(t->sng);
}

int orc_physical_side_basis(int dim, int order, int qdeg, int nbnd, const double *nodes, const int *belem,
                            const int *bside, double *wts, double *normals, double *ip, double *basis,
                            double *basis_grad) {
  side_tab t;
  if (side_tab_make(dim, order, qdeg, &t)) return -1;
  const int n = t.nb, nn = t.nn, nqs = t.nqs;
  for (int k = 0; k < nbnd; ++k) {
    const double *xn = nodes + (size_t)belem[k] * nn * dim;
    const int s = bside[k];
    for (int q = 0; q < nqs; ++q) {
      double J[9] = {0}, Ji[9] = {0}, det;
      for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
          double sum = 0.0;
          for (int v = 0; v < nn; ++v) sum += xn[v * dim + r] * t.sng[((s * nn + v) * nqs + q) * dim + c];
          J[r * dim + c] = sum;
        }
      jac_inv_det(dim, J, Ji, &det);
      double nrm[3] = {0, 0, 0}, w;
      if (dim == 2) { /* t = J t_ref; n = R t, R = [[0,1],[-1,0]]; w = |t| w_ref (:1684-1697) */
        double tx = J[0] * t.tanU[s * 2] + J[1] * t.tanU[s * 2 + 1], ty = J[2] * t.tanU[s * 2] + J[3] * t.tanU[s * 2 + 1];
        nrm[0] = ty; nrm[1] = -tx;
        w = sqrt(tx * tx + ty * ty) * t.swts[q];
      } else {        /* n = (J tU) x (J tV); w = |n| w_ref (:1699-1710) */
        double a[3], b[3];
        for (int r = 0; r < 3; ++r) {
          a[r] = J[r * 3] * t.tanU[s * 3] + J[r * 3 + 1] * t.tanU[s * 3 + 1] + J[r * 3 + 2] * t.tanU[s * 3 + 2];
          b[r] = J[r * 3] * t.tanV[s * 3] + J[r * 3 + 1] * t.tanV[s * 3 + 1] + J[r * 3 + 2] * t.tanV[s * 3 + 2];
        }
        nrm[0] = a[1] * b[2] - a[2] * b[1]; nrm[1] = a[2] * b[0] - a[0] * b[2]; nrm[2] = a[0] * b[1] - a[1] * b[0];
        w = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]) * t.swts[q];
      }
      double len = 0.0;
      for (int d = 0; d < dim; ++d) len += nrm[d] * nrm[d];
      len = sqrt(len);
      if (wts) wts[(size_t)k * nqs + q] = w;
      for (int d = 0; d < dim; ++d) {
        if (normals) normals[((size_t)k * nqs + q) * dim + d] = nrm[d] * (1.0 / len); /* rescale (:1760-1786) */
        if (ip) {
          double sum = 0.0;
          for (int v = 0; v < nn; ++v) sum += xn[v * dim + d] * t.snv[(s * nn + v) * nqs + q];
          ip[((size_t)k * nqs + q) * dim + d] = sum;
        }
      }
      for (int f = 0; f < n; ++f) {
        if (basis) basis[((size_t)k * n + f) * nqs + q] = t.sbasis[(s * n + f) * nqs + q];
        if (basis_grad)
          for (int d = 0; d < dim; ++d) {
            double sum = 0.0;
            for (int c = 0; c < dim; ++c) sum += Ji[c * dim + d] * t.sgrad[((s * n + f) * nqs + q) * dim + c];
            basis_grad[(((size_t)k * n + f) * nqs + q) * dim + d] = sum;
          }
      }
    }
  }
  side_tab_free(&t);
  return 0;
}

int orc_assemble_thermal_boundary(const orc_thermal_bnd_args *a) {
  int n, nq, nn, ns, nqs;
  if (orc_ref_sizes(a->dim, a->order, a->qdeg, &n, &nq, &nn) || orc_side_sizes(a->dim, a->qdeg, &ns, &nqs)) return -1;
  const int dim = a->dim, W = orc_ad_width(n), W1 = W + 1, nb = a->nb;
  if (nb <= 0) return 0;
  double *wts = malloc(sizeof(double) * (size_t)nb * nqs), *nrm = malloc(sizeof(double) * (size_t)nb * nqs * dim);
  double *ip = malloc(sizeof(double) * (size_t)nb * nqs * dim);
  double *bas = malloc(sizeof(double) * (size_t)nb * n * nqs), *bgr = malloc(sizeof(double) * (size_t)nb * n * nqs * dim);
  orc_physical_side_basis(dim, a->order, a->qdeg, nb, a->nodes, a->belem, a->bside, wts, nrm, ip, bas, bgr);
  double *uAD = malloc(sizeof(double) * (size_t)n * W1), *res = malloc(sizeof(double) * (size_t)n * W1);
  double *fld = malloc(sizeof(double) * (size_t)(1 + dim) * nqs * W1);
  const double epen = 10.0, sf = a->form_param; /* thermal.cpp:236, 197 */
  for (int k = 0; k < nb; ++k) {
    const int *L = a->lids + (size_t)a->belem[k] * n;
    /* performBoundaryGather + seeding (same rules as the volume loop) */
    for (int dof = 0; dof < n; ++dof) {
      double *ua = uAD + (size_t)dof * W1;
      memset(ua, 0, sizeof(double) * W1);
      const int off = a->offsets[dof], row = L[off];
      const double cu = a->u[row];
      if (!a->transient) {
        ua[0] = cu;
        if (a->compute_jacobian) ua[1 + off] = 1.0;
      } else {
        const int st = a->stage, S = a->nstages, NS = a->nsteps;
        const double *cu_prev = a->u_prev + (size_t)row * NS, *cu_stage = a->u_stage + (size_t)row * S;
        const double alpha_u = a->butcher_A[st * S + st] / a->butcher_b[st];
        double beta_u = (1.0 - alpha_u) * cu_prev[0];
        for (int s = 0; s < st; ++s) beta_u += a->butcher_A[st * S + s] / a->butcher_b[s] * (cu_stage[s] - cu_prev[0]);
        ua[0] = alpha_u * cu + beta_u;
        if (a->compute_jacobian) ua[1 + off] = alpha_u;
      }
    }
    memset(res, 0, sizeof(double) * (size_t)n * W1);
    /* side fields: e, grad(e)[x..] (evaluateSideSolutionField, workset.cpp:1069-1176) */
    for (int f = 0; f <= dim; ++f)
      for (int pt = 0; pt < nqs; ++pt) {
        double *o = fld + ((size_t)f * nqs + pt) * W1;
        for (int dof = 0; dof < n; ++dof) {
          const double b = (f == 0) ? bas[((size_t)k * n + dof) * nqs + pt] : bgr[(((size_t)k * n + dof) * nqs + pt) * dim + (f - 1)];
          const double *s = uAD + (size_t)dof * W1;
          if (dof == 0) for (int j = 0; j < W1; ++j) o[j] = s[j] * b;
          else for (int j = 0; j < W1; ++j) o[j] += s[j] * b;
        }
      }
    /* getSideElementSize (workset.cpp:2682-2696) */
    double vol = 0.0;
    for (int pt = 0; pt < nqs; ++pt) vol += wts[(size_t)k * nqs + pt];
    const double h = pow(vol, 1.0 / ((double)dim - 1.0));
    for (int dof = 0; dof < n; ++dof) {
      double *r = res + (size_t)a->offsets[dof] * W1;
      for (int pt = 0; pt < nqs; ++pt) {
        const double w = wts[(size_t)k * nqs + pt], bv = bas[((size_t)k * n + dof) * nqs + pt];
        double g;
        switch (a->data_kind) {
          case 1: g = a->data_ip[(size_t)k * nqs + pt]; break;
          case 2: g = a->data_amp; for (int d = 0; d < dim; ++d) g *= sin(a->data_freq[d] * ip[((size_t)k * nqs + pt) * dim + d]); break;
          default: g = a->data_amp;
        }
        if (a->bc_type == ORC_BC_NEUMANN) { /* thermal.cpp:217-226 */
          r[0] += -g * w * bv;
        } else {                             /* weak Dirichlet, thermal.cpp:237-273 */
          const double *T = fld + (size_t)pt * W1;
          double bgn = 0.0;
          for (int d = 0; d < dim; ++d) bgn += bgr[(((size_t)k * n + dof) * nqs + pt) * dim + d] * nrm[((size_t)k * nqs + pt) * dim + d];
          for (int j = 0; j < W1; ++j) {
            const double Tj = T[j] - (j == 0 ? g : 0.0);
            double gTn = 0.0;
            for (int d = 0; d < dim; ++d) gTn += fld[((size_t)(1 + d) * nqs + pt) * W1 + j] * nrm[((size_t)k * nqs + pt) * dim + d];
            r[j] += epen / h * a->diff * Tj * w * bv;
            r[j] += -a->diff * gTn * w * bv;
            r[j] += -sf * a->diff * Tj * w * bgn;
          }
        }
      }
    }
    /* scatter (same conventions as the volume scatter) */
    for (int j = 0; j < n; ++j) {
      const int row = a->offsets[j], rowIndex = L[row];
      if (a->fixed && a->fixed[rowIndex]) continue;
      const double *r = res + (size_t)row * W1;
      if (a->res) a->res[rowIndex] += -r[0];
      if (a->crs_vals && a->compute_jacobian)
        for (int c = 0; c < n; ++c) {
          const int col = a->offsets[c], gcol = L[col];
          for (int p = a->rowptr[rowIndex]; p < a->rowptr[rowIndex + 1]; ++p)
            if (a->colind[p] == gcol) { a->crs_vals[p] += r[1 + col]; break; }
        }
    }
  }
  free(wts); free(nrm); free(ip); free(bas); free(bgr); free(uAD); free(res); free(fld);
  return 0;
}
