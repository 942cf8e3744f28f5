/*
 * mrhyde_oracle_multi.c -- TEST INFRASTRUCTURE ONLY (see mrhyde_oracle.h).
 *
 * Multi-variable part of the CPU oracle: HVOL / HDIV bases next to HGRAD, the subcell-major dof map of a
 * structured mesh, and the reference's data flow for porousMixed, navierstokes and thermal with width-n_tot
 * derivative arrays (one pass per solution field, one loop nest per residual block, exactly as the functors in
 * src/physics/{porousMixed,navierstokes,thermal}.cpp do it).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mrhyde_oracle.h"
#include "orc_internal.h"

/* ------------------------------------------------------------------------ */
/* bases                                                                     */
/* ------------------------------------------------------------------------ */

int orc_basis_card(int dim, int type, int order) {
  if (type == ORC_BASIS_HGRAD) return (order >= 1 && order <= ORC_MAXP) ? orc__ipow(order + 1, dim) : -1;
  if (type == ORC_BASIS_HVOL) return order == 0 ? 1 : -1;
  if (type == ORC_BASIS_HDIV) return (order == 1 && dim >= 2) ? 2 * dim : -1;
  return -1;
}

static int ncomp_of(int dim, int type) { return type == ORC_BASIS_HDIV ? dim : 1; }

int orc_ref_basis_var(int dim, int type, int order, int npts, const double *x, double *val, double *grad, double *div) {
  const int n = orc_basis_card(dim, type, order);
  if (n < 0) return -1;
  for (int pt = 0; pt < npts; ++pt) {
    const double *xp = x + (size_t)pt * dim;
    if (type == ORC_BASIS_HGRAD) {
      const int p1 = order + 1;
      double bv[3][ORC_MAXP + 1], bd[3][ORC_MAXP + 1];
      for (int d = 0; d < dim; ++d) orc_lagrange_1d(order, xp[d], bv[d], bd[d]);
      for (int f = 0; f < n; ++f) {
        const int fi[3] = {f % p1, (f / p1) % p1, f / (p1 * p1)};
        double v = 1.0;
        for (int d = 0; d < dim; ++d) v *= bv[d][fi[d]];
        val[(size_t)f * npts + pt] = v;
        if (grad)
          for (int d = 0; d < dim; ++d) {
            double g = 1.0;
            for (int e = 0; e < dim; ++e) g *= (e == d) ? bd[e][fi[e]] : bv[e][fi[e]];
            grad[((size_t)f * npts + pt) * dim + d] = g;
          }
      }
    } else if (type == ORC_BASIS_HVOL) {
      val[pt] = 1.0; /* Basis_HVOL_C0_FEM */
    } else {
      for (int c = 0; c < dim; ++c)
        for (int s = 0; s < 2; ++s) {
          const int f = 2 * c + s;
          for (int d = 0; d < dim; ++d) val[((size_t)f * npts + pt) * dim + d] = 0.0;
          val[((size_t)f * npts + pt) * dim + c] = s ? 0.5 * (1.0 + xp[c]) : 0.5 * (1.0 - xp[c]);
          if (div) div[(size_t)f * npts + pt] = s ? 0.5 : -0.5;
        }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* structured mesh + subcell-major multi-variable dof map                    */
/* ------------------------------------------------------------------------ */

static int max_hgrad_order(int nvars, const int *types, const int *orders) {
  int K = 0;
  for (int v = 0; v < nvars; ++v)
    if (types[v] == ORC_BASIS_HGRAD && orders[v] > K) K = orders[v];
  return K;
}

int orc_mesh_multi_sizes(int dim, const int *nc, int nvars, const int *types, const int *orders, int *nverts,
                         int *nelem, int *n_tot, long long *ndof) {
  if (dim < 2 || dim > 3 || nvars < 1 || nvars > ORC_MAX_VARS) return -1;
  const int K = max_hgrad_order(nvars, types, orders);
  long long nd = 0;
  int nt = 0, ne = 1, nv = 1;
  for (int d = 0; d < dim; ++d) { ne *= nc[d]; nv *= nc[d] + 1; }
  for (int v = 0; v < nvars; ++v) {
    const int card = orc_basis_card(dim, types[v], orders[v]);
    if (card < 0) return -1;
    nt += card;
    if (types[v] == ORC_BASIS_HGRAD) {
      if (K % orders[v]) return -1;
      long long m = 1;
      for (int d = 0; d < dim; ++d) m *= (long long)orders[v] * nc[d] + 1;
      nd += m;
    } else if (types[v] == ORC_BASIS_HVOL) {
      nd += ne;
    } else {
      for (int c = 0; c < dim; ++c) {
        long long m = 1;
        for (int d = 0; d < dim; ++d) m *= nc[d] + (d == c);
        nd += m;
      }
    }
  }
  *nverts = nv; *nelem = ne; *n_tot = nt; *ndof = nd;
  return 0;
}

int orc_mesh_multi(int dim, const int *nc, const double *lo, const double *hi, int nvars, const int *types,
                   const int *orders, double *verts, int *cell2vert, int *lids, int *offsets, signed char *orient,
                   unsigned char *side_mask, int *dof_var) {
  int nverts, nelem, n_tot;
  long long ndof;
  if (orc_mesh_multi_sizes(dim, nc, nvars, types, orders, &nverts, &nelem, &n_tot, &ndof)) return -1;
  const int K = max_hgrad_order(nvars, types, orders);
  const int nn = 1 << dim;
  int varptr[ORC_MAX_VARS + 1] = {0};
  for (int v = 0; v < nvars; ++v) varptr[v + 1] = varptr[v] + orc_basis_card(dim, types[v], orders[v]);
  { /* vertices + cell -> vertex map from the single-variable generator */
    int *tl = malloc(sizeof(int) * (size_t)nelem * nn), to[8];
    if (orc_mesh_structured(dim, 1, nc, lo, hi, verts, cell2vert, tl, to, NULL)) { free(tl); return -1; }
    free(tl);
  }
  /* ---- global ids ---- */
  int fd[3] = {1, 1, 1}; /* fine lattice extents */
  size_t nfine = 1;
  for (int d = 0; d < dim; ++d) { fd[d] = K * nc[d] + 1; nfine *= (size_t)fd[d]; }
  int *gnode = NULL; /* [nvars][nfine] */
  int next = 0;
  if (K > 0) {
    gnode = malloc(sizeof(int) * nvars * nfine);
    for (size_t s = 0; s < nfine; ++s) {
      const int a[3] = {(int)(s % fd[0]), (int)((s / fd[0]) % fd[1]), (int)(s / ((size_t)fd[0] * fd[1]))};
      for (int v = 0; v < nvars; ++v) {
        gnode[(size_t)v * nfine + s] = -1;
        if (types[v] != ORC_BASIS_HGRAD) continue;
        const int st = K / orders[v];
        int in = 1;
        for (int d = 0; d < dim; ++d) in &= (a[d] % st == 0);
        if (!in) continue;
        unsigned char m = 0;
        for (int d = 0; d < dim; ++d) {
          if (a[d] == 0) m |= 1u << (2 * d);
          if (a[d] == fd[d] - 1) m |= 1u << (2 * d + 1);
        }
        if (side_mask) side_mask[next] = m;
        if (dof_var) dof_var[next] = v;
        gnode[(size_t)v * nfine + s] = next++;
      }
    }
  }
  int gcell0[ORC_MAX_VARS], gface0[ORC_MAX_VARS][3];
  const int ncx = nc[0], ncy = nc[1], ncz = dim == 3 ? nc[2] : 1;
  /* cells: interleaved over HVOL variables per cell */
  int nhvol = 0, hvol_rank[ORC_MAX_VARS];
  for (int v = 0; v < nvars; ++v) { hvol_rank[v] = nhvol; if (types[v] == ORC_BASIS_HVOL) ++nhvol; }
  const int cell_base = next;
  for (int e = 0; e < nelem && nhvol; ++e)
    for (int v = 0; v < nvars; ++v)
      if (types[v] == ORC_BASIS_HVOL) {
        if (side_mask) side_mask[next] = 0;
        if (dof_var) dof_var[next] = v;
        ++next;
      }
  (void)gcell0;
  /* faces: direction by direction, interleaved over HDIV variables per face */
  int nhdiv = 0, hdiv_rank[ORC_MAX_VARS];
  for (int v = 0; v < nvars; ++v) { hdiv_rank[v] = nhdiv; if (types[v] == ORC_BASIS_HDIV) ++nhdiv; }
  int face_base[3] = {0, 0, 0};
  for (int c = 0; c < dim && nhdiv; ++c) {
    face_base[c] = next;
    const int ex[3] = {ncx + (c == 0), ncy + (c == 1), dim == 3 ? ncz + (c == 2) : 1};
    for (int k = 0; k < ex[2]; ++k)
      for (int j = 0; j < ex[1]; ++j)
        for (int i = 0; i < ex[0]; ++i)
          for (int v = 0; v < nvars; ++v)
            if (types[v] == ORC_BASIS_HDIV) {
              const int idx[3] = {i, j, k};
              unsigned char m = 0;
              if (idx[c] == 0) m |= 1u << (2 * c);
              if (idx[c] == ex[c] - 1) m |= 1u << (2 * c + 1);
              if (side_mask) side_mask[next] = m;
              if (dof_var) dof_var[next] = v;
              ++next;
            }
  }
  (void)gface0;
  if (next != (int)ndof) { free(gnode); return -2; }
  /* ---- element LID lists + offsets ---- */
  const int kp = K + 1;
  const int nsite = K > 0 ? orc__ipow(kp, dim) : 0;
  int *site_order = malloc(sizeof(int) * (nsite + 1));
  { /* vertices in shards order, then the remaining tensor sites */
    int cnt = 0;
    unsigned char *isv = calloc(nsite + 1, 1);
    for (int v = 0; v < nn && K > 0; ++v) {
      const double *cn = dim == 2 ? QUAD_NODE[v] : HEX_NODE[v];
      int t = 0, mul = 1;
      for (int d = 0; d < dim; ++d) { t += (cn[d] > 0 ? K : 0) * mul; mul *= kp; }
      site_order[cnt++] = t;
      isv[t] = 1;
    }
    for (int t = 0; t < nsite; ++t)
      if (!isv[t]) site_order[cnt++] = t;
    free(isv);
  }
  for (int e = 0; e < nelem; ++e) {
    const int ci[3] = {e % ncx, (e / ncx) % ncy, e / (ncx * ncy)};
    int *L = lids + (size_t)e * n_tot;
    int pos = 0;
    for (int so = 0; so < nsite; ++so) {
      const int t = site_order[so];
      const int a[3] = {t % kp, (t / kp) % kp, t / (kp * kp)};
      size_t s = 0, mul = 1;
      for (int d = 0; d < dim; ++d) { s += (size_t)(ci[d] * K + a[d]) * mul; mul *= (size_t)fd[d]; }
      for (int v = 0; v < nvars; ++v) {
        if (types[v] != ORC_BASIS_HGRAD) continue;
        const int st = K / orders[v], p1 = orders[v] + 1;
        int in = 1;
        for (int d = 0; d < dim; ++d) in &= (a[d] % st == 0);
        if (!in) continue;
        int dof = 0, m2 = 1;
        for (int d = 0; d < dim; ++d) { dof += (a[d] / st) * m2; m2 *= p1; }
        if (e == 0) offsets[varptr[v] + dof] = pos;
        L[pos++] = gnode[(size_t)v * nfine + s];
      }
    }
    for (int v = 0; v < nvars; ++v)
      if (types[v] == ORC_BASIS_HVOL) {
        if (e == 0) offsets[varptr[v]] = pos;
        L[pos++] = cell_base + e * nhvol + hvol_rank[v];
      }
    for (int f = 0; f < 2 * dim && nhdiv; ++f) {
      const int c = f / 2, s = f % 2;
      const int ex[3] = {ncx + (c == 0), ncy + (c == 1), dim == 3 ? ncz + (c == 2) : 1};
      int idx[3] = {ci[0], ci[1], ci[2]};
      idx[c] += s;
      const int lin = idx[0] + ex[0] * (idx[1] + ex[1] * idx[2]);
      for (int v = 0; v < nvars; ++v)
        if (types[v] == ORC_BASIS_HDIV) {
          if (e == 0) offsets[varptr[v] + f] = pos;
          L[pos++] = face_base[c] + lin * nhdiv + hdiv_rank[v];
        }
    }
    if (pos != n_tot) { free(gnode); free(site_order); return -3; }
    /* orientation signs */
    if (orient) {
      signed char *o = orient + (size_t)e * n_tot;
      for (int k = 0; k < n_tot; ++k) o[k] = 1;
      static const int face_of_dof3[6] = {3, 1, 0, 2, 4, 5}, edge_of_dof2[4] = {3, 1, 0, 2};
      const int *cv = cell2vert + (size_t)e * nn;
      for (int v = 0; v < nvars; ++v) {
        if (types[v] != ORC_BASIS_HDIV) continue;
        for (int f = 0; f < 2 * dim; ++f) {
          int flip;
          if (dim == 2) {
            const int *sn = QUAD_SIDE[edge_of_dof2[f]];
            flip = cv[sn[0]] > cv[sn[1]];
          } else {
            const int *sn = HEX_SIDE[face_of_dof3[f]];
            int rot = 0;
            for (int k = 1; k < 4; ++k)
              if (cv[sn[k]] < cv[sn[rot]]) rot = k;
            flip = cv[sn[(rot + 1) % 4]] > cv[sn[(rot + 3) % 4]];
          }
          const int sigma = (f % 2) ? 1 : -1; /* phi_raw . n_out on its own face */
          o[varptr[v] + f] = (signed char)(flip ? -sigma : sigma);
        }
      }
    }
  }
  free(gnode);
  free(site_order);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* physical basis of one variable                                            */
/* ------------------------------------------------------------------------ */

int orc_physical_basis_var(int dim, int type, int order, int qdeg, int nelem, const double *nodes,
                           const signed char *orient, int orient_stride, int orient_off, double *basis, double *grad,
                           double *div, double *wts, double *ip) {
  int n1, nq, nn;
  if (orc_ref_sizes(dim, 1, qdeg, &n1, &nq, &nn)) return -1;
  const int n = orc_basis_card(dim, type, order), nc = ncomp_of(dim, type);
  if (n < 0) return -1;
  double *rip = malloc(sizeof(double) * nq * dim), *rw = malloc(sizeof(double) * nq);
  double *rb1 = malloc(sizeof(double) * n1 * nq), *rg1 = malloc(sizeof(double) * n1 * nq * dim);
  double *nv = malloc(sizeof(double) * nn * nq), *ng = malloc(sizeof(double) * nn * nq * dim);
  orc_ref_tables(dim, 1, qdeg, rip, rw, rb1, rg1, nv, ng);
  double *rv = malloc(sizeof(double) * n * nq * nc), *rg = malloc(sizeof(double) * n * nq * dim);
  double *rd = malloc(sizeof(double) * n * nq);
  orc_ref_basis_var(dim, type, order, nq, rip, rv, rg, rd);
  for (int e = 0; e < nelem; ++e) {
    const double *xn = nodes + (size_t)e * nn * dim;
    for (int q = 0; q < nq; ++q) {
      double J[9] = {0}, Ji[9] = {0}, det;
      for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
          double s = 0.0;
          for (int v = 0; v < nn; ++v) s += xn[v * dim + r] * ng[(v * nq + q) * dim + c];
          J[r * dim + c] = s;
        }
      jac_inv_det(dim, J, Ji, &det);
      if (wts) wts[(size_t)e * nq + q] = rw[q] * det;
      if (ip)
        for (int d = 0; d < dim; ++d) {
          double s = 0.0;
          for (int v = 0; v < nn; ++v) s += xn[v * dim + d] * nv[v * nq + q];
          ip[((size_t)e * nq + q) * dim + d] = s;
        }
      for (int f = 0; f < n; ++f) {
        const double sg = orient ? (double)orient[(size_t)e * orient_stride + orient_off + f] : 1.0;
        const size_t o = ((size_t)e * n + f) * nq + q;
        if (type == ORC_BASIS_HDIV) {
          /* HDIVtransformVALUE: J phi / detJ; HDIVtransformDIV: div / detJ (discretizationInterface.cpp:1019,1053) */
          for (int d = 0; d < dim; ++d) {
            double s = 0.0;
            for (int c = 0; c < dim; ++c) s += J[d * dim + c] * rv[((size_t)f * nq + q) * dim + c];
            if (basis) basis[o * dim + d] = sg * s / det;
          }
          if (div) div[o] = sg * rd[(size_t)f * nq + q] / det;
        } else {
          if (basis) basis[o] = rv[(size_t)f * nq + q];
          if (grad && type == ORC_BASIS_HGRAD)
            for (int d = 0; d < dim; ++d) {
              double s = 0.0;
              for (int c = 0; c < dim; ++c) s += Ji[c * dim + d] * rg[((size_t)f * nq + q) * dim + c];
              grad[o * dim + d] = s;
            }
        }
      }
    }
  }
  free(rip); free(rw); free(rb1); free(rg1); free(nv); free(ng); free(rv); free(rg); free(rd);
  return 0;
}

int orc_physical_side_basis_hdiv(int dim, int qdeg, int nb, const double *nodes, const int *belem, const int *bside,
                                 const signed char *orient, int orient_stride, int orient_off, double *basis) {
  int n1, nq, nn, ns, nqs;
  if (orc_ref_sizes(dim, 1, qdeg, &n1, &nq, &nn) || orc_side_sizes(dim, qdeg, &ns, &nqs)) return -1;
  const int n = 2 * dim;
  double *sip = malloc(sizeof(double) * ns * nqs * dim), *sw = malloc(sizeof(double) * nqs);
  double *tu = malloc(sizeof(double) * ns * dim), *tv = malloc(sizeof(double) * ns * dim);
  double *sb = malloc(sizeof(double) * ns * n1 * nqs), *sg = malloc(sizeof(double) * ns * n1 * nqs * dim);
  double *snv = malloc(sizeof(double) * ns * nn * nqs), *sng = malloc(sizeof(double) * ns * nn * nqs * dim);
  orc_side_tables(dim, 1, qdeg, sip, sw, tu, tv, sb, sg, snv, sng);
  double *rv = malloc(sizeof(double) * n * nqs * dim);
  for (int k = 0; k < nb; ++k) {
    const int e = belem[k], s = bside[k];
    const double *xn = nodes + (size_t)e * nn * dim;
    orc_ref_basis_var(dim, ORC_BASIS_HDIV, 1, nqs, sip + (size_t)s * nqs * dim, rv, NULL, NULL);
    for (int q = 0; q < nqs; ++q) {
      double J[9] = {0}, Ji[9] = {0}, det;
      for (int r = 0; r < dim; ++r)
        for (int c = 0; c < dim; ++c) {
          double sum = 0.0;
          for (int v = 0; v < nn; ++v) sum += xn[v * dim + r] * sng[((s * nn + v) * nqs + q) * dim + c];
          J[r * dim + c] = sum;
        }
      jac_inv_det(dim, J, Ji, &det);
      for (int f = 0; f < n; ++f) {
        const double sgn = orient ? (double)orient[(size_t)e * orient_stride + orient_off + f] : 1.0;
        for (int d = 0; d < dim; ++d) {
          double sum = 0.0;
          for (int c = 0; c < dim; ++c) sum += J[d * dim + c] * rv[((size_t)f * nqs + q) * dim + c];
          basis[(((size_t)k * n + f) * nqs + q) * dim + d] = sgn * sum / det;
        }
      }
    }
  }
  free(sip); free(sw); free(tu); free(tv); free(sb); free(sg); free(snv); free(sng); free(rv);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* derivative-array arithmetic (Sacado SFad restated on plain arrays)         */
/* ------------------------------------------------------------------------ */

#define ADMAX 160
typedef struct { double v[ADMAX]; } ad_t;  /* v[0] value, v[1+k] d/d(slot k) */
static int g_w1;                           /* active width + 1 (set per call; the oracle is single-threaded here) */

static ad_t ad_c(double c) { ad_t r; memset(r.v, 0, sizeof(double) * g_w1); r.v[0] = c; return r; }
static ad_t ad_add(ad_t a, const ad_t *b) { for (int j = 0; j < g_w1; ++j) a.v[j] += b->v[j]; return a; }
static ad_t ad_sub(ad_t a, const ad_t *b) { for (int j = 0; j < g_w1; ++j) a.v[j] -= b->v[j]; return a; }
static ad_t ad_scale(ad_t a, double s) { for (int j = 0; j < g_w1; ++j) a.v[j] *= s; return a; }
static ad_t ad_mul(const ad_t *a, const ad_t *b) {
  ad_t r;
  r.v[0] = a->v[0] * b->v[0];
  for (int j = 1; j < g_w1; ++j) r.v[j] = a->v[j] * b->v[0] + a->v[0] * b->v[j];
  return r;
}
static ad_t ad_div(const ad_t *a, const ad_t *b) {
  ad_t r;
  r.v[0] = a->v[0] / b->v[0];
  for (int j = 1; j < g_w1; ++j) r.v[j] = (a->v[j] * b->v[0] - a->v[0] * b->v[j]) / (b->v[0] * b->v[0]);
  return r;
}
static ad_t ad_sqrt(const ad_t *a) {
  ad_t r;
  r.v[0] = sqrt(a->v[0]);
  for (int j = 1; j < g_w1; ++j) r.v[j] = a->v[j] / (2.0 * r.v[0]);
  return r;
}

/* ------------------------------------------------------------------------ */
/* block assembly                                                            */
/* ------------------------------------------------------------------------ */

typedef struct {
  const orc_block_args *a;
  int n_tot, nq, nn, varptr[ORC_MAX_VARS + 1];
  /* physical basis of the current element, per variable */
  double *basis[ORC_MAX_VARS], *grad[ORC_MAX_VARS], *div[ORC_MAX_VARS];
  double *wts, *ip;
  ad_t *uAD, *udAD; /* [n_tot] flattened (var,dof) */
} blk_ctx;

static double eval_func(const orc_func *f, int dim, size_t e, int q, int nq, const double *x) {
  if (f->kind == 0) return f->amp;
  if (f->kind == 1) return f->ip[e * nq + q];
  if (f->kind == 3) {
    double x3[3] = {x[0], x[1], dim > 2 ? x[2] : 0.0};
    return orc_eval_expression(f->expr, x3, f->t, NULL, 0.0, NULL);
  }
  double s = f->amp;
  for (int d = 0; d < dim; ++d) s *= sin(f->freq[d] * x[d]);
  return s;
}

enum { F_VAL = 0, F_GRAD = 1, F_DIV = 2, F_DOT = 3 };

/* Workset::evaluateSolutionField (workset.cpp:937-1062): f(pt) = sum_dof u_AD(dof) * basis(dof, pt, comp) */
static void eval_field(const blk_ctx *c, int var, int kind, int comp, ad_t *out /*[nq]*/) {
  const orc_block_args *a = c->a;
  const int n = c->varptr[var + 1] - c->varptr[var], nq = c->nq, dim = a->dim;
  const int nc = ncomp_of(dim, a->types[var]);
  for (int pt = 0; pt < nq; ++pt) {
    ad_t acc = ad_c(0.0);
    for (int dof = 0; dof < n; ++dof) {
      double b;
      const size_t o = (size_t)dof * nq + pt;
      if (kind == F_GRAD) b = c->grad[var][o * dim + comp];
      else if (kind == F_DIV) b = c->div[var][o];
      else b = c->basis[var][o * nc + comp];
      const ad_t *s = (kind == F_DOT ? c->udAD : c->uAD) + c->varptr[var] + dof;
      for (int j = 0; j < g_w1; ++j) acc.v[j] += s->v[j] * b;
    }
    out[pt] = acc;
  }
}

/* res(elem, off(dof)) += F * b */
static void res_add(ad_t *res, int pos, const ad_t *F, double b) {
  for (int j = 0; j < g_w1; ++j) res[pos].v[j] += F->v[j] * b;
}

static void porous_volume(const blk_ctx *c, size_t e, ad_t *res) {
  const orc_block_args *a = c->a;
  const int dim = a->dim, nq = c->nq, pnum = 0, unum = 1;
  const int nu = c->varptr[unum + 1] - c->varptr[unum], np = c->varptr[pnum + 1] - c->varptr[pnum];
  ad_t *psol = malloc(sizeof(ad_t) * nq), *udiv = malloc(sizeof(ad_t) * nq);
  ad_t *ucomp[3] = {NULL, NULL, NULL};
  eval_field(c, pnum, F_VAL, 0, psol);
  for (int d = 0; d < dim; ++d) { ucomp[d] = malloc(sizeof(ad_t) * nq); eval_field(c, unum, F_VAL, d, ucomp[d]); }
  eval_field(c, unum, F_DIV, 0, udiv);
  /* ((mobility K)^-1 u, v) - (p, div v)   (porousMixed.cpp:236-313) */
  for (int pt = 0; pt < nq; ++pt) {
    const double w = c->wts[pt], *x = c->ip + (size_t)pt * dim;
    const double mob = eval_func(&a->funcs[4], dim, e, pt, nq, x);
    ad_t p = ad_scale(psol[pt], w), Kiu[3];
    for (int d = 0; d < dim; ++d) {
      const double Kinv = eval_func(&a->funcs[1 + d], dim, e, pt, nq, x);
      Kiu[d] = ad_scale(ad_scale(ucomp[d][pt], Kinv), w);
      Kiu[d] = ad_scale(Kiu[d], 1.0 / mob);
    }
    for (int dof = 0; dof < nu; ++dof) {
      const int pos = a->offsets[c->varptr[unum] + dof];
      const size_t o = (size_t)dof * nq + pt;
      for (int d = 0; d < dim; ++d) res_add(res, pos, &Kiu[d], c->basis[unum][o * dim + d]);
      res_add(res, pos, &p, -c->div[unum][o]);
    }
  }
  /* -(div u, q) + (src, q)   (porousMixed.cpp:316-337) */
  for (int pt = 0; pt < nq; ++pt) {
    const double *x = c->ip + (size_t)pt * dim;
    ad_t F = ad_c(eval_func(&a->funcs[0], dim, e, pt, nq, x));
    F = ad_scale(ad_sub(F, &udiv[pt]), c->wts[pt]);
    for (int dof = 0; dof < np; ++dof)
      res_add(res, a->offsets[c->varptr[pnum] + dof], &F, c->basis[pnum][(size_t)dof * nq + pt]);
  }
  free(psol); free(udiv);
  for (int d = 0; d < dim; ++d) free(ucomp[d]);
}

/* navierstokes::computeTau (navierstokes.cpp:1054-1079) */
static ad_t ns_tau(double visc, const ad_t *vel, int dim, double h, double dt, int transient) {
  const double C1 = 4.0, C2 = 2.0, C3 = transient ? 2.0 : 0.0;
  ad_t nvel = ad_c(0.0);
  for (int d = 0; d < dim; ++d) { ad_t sq = ad_mul(&vel[d], &vel[d]); nvel = ad_add(nvel, &sq); }
  if (nvel.v[0] > 1e-12) nvel = ad_sqrt(&nvel);
  ad_t t2 = ad_scale(nvel, C2 / h);
  ad_t tau = ad_mul(&t2, &t2);
  tau.v[0] += (C1 * visc / h / h) * (C1 * visc / h / h) + (C3 / dt) * (C3 / dt);
  ad_t rt = ad_sqrt(&tau), one = ad_c(1.0);
  return ad_div(&one, &rt);
}

static void ns_volume(const blk_ctx *c, size_t e, ad_t *res) {
  const orc_block_args *a = c->a;
  const int dim = a->dim, nq = c->nq;
  /* myvars = ux, pr, uy[, uz] (navierstokes.cpp:27-34) */
  const int vnum[3] = {0, 2, 3}, prnum = 1;
  const int useSUPG = a->params[0] != 0.0, usePSPG = a->params[1] != 0.0, fix_uz = a->params[2] != 0.0;
  ad_t *U[3], *Ut[3], *dU[3][3], *pr = malloc(sizeof(ad_t) * nq), *dpr[3] = {NULL, NULL, NULL};
  for (int i = 0; i < dim; ++i) {
    U[i] = malloc(sizeof(ad_t) * nq); Ut[i] = malloc(sizeof(ad_t) * nq);
    eval_field(c, vnum[i], F_VAL, 0, U[i]);
    eval_field(c, vnum[i], F_DOT, 0, Ut[i]);
    for (int d = 0; d < dim; ++d) { dU[i][d] = malloc(sizeof(ad_t) * nq); eval_field(c, vnum[i], F_GRAD, d, dU[i][d]); }
  }
  eval_field(c, prnum, F_VAL, 0, pr);
  if (useSUPG || usePSPG)
    for (int d = 0; d < dim; ++d) { dpr[d] = malloc(sizeof(ad_t) * nq); eval_field(c, prnum, F_GRAD, d, dpr[d]); }
  double vol = 0.0; /* Workset::getElementSize (workset.cpp:2666-2679) */
  for (int pt = 0; pt < nq; ++pt) vol += c->wts[pt];
  const double h = pow(vol, 1.0 / (double)dim);
  for (int pt = 0; pt < nq; ++pt) {
    const double w = c->wts[pt], *x = c->ip + (size_t)pt * dim;
    const double src[3] = {eval_func(&a->funcs[0], dim, e, pt, nq, x), eval_func(&a->funcs[2], dim, e, pt, nq, x),
                           eval_func(&a->funcs[3], dim, e, pt, nq, x)};
    const double dens = eval_func(&a->funcs[4], dim, e, pt, nq, x), visc = eval_func(&a->funcs[5], dim, e, pt, nq, x);
    ad_t vel[3];
    for (int d = 0; d < dim; ++d) vel[d] = U[d][pt];
    ad_t tau = ad_c(0.0);
    if (useSUPG || usePSPG) tau = ns_tau(visc, vel, dim, h, a->dt, a->transient);
    ad_t stab[3]; /* strong momentum residual of component i */
    for (int i = 0; i < dim; ++i) {
      /* F_d = visc * d u_i/dx_d (- pr if d == i), * wts ; F = (u_i_t + u . grad u_i - source_i) dens wts */
      ad_t conv = ad_c(0.0);
      for (int d = 0; d < dim; ++d) { ad_t t = ad_mul(&vel[d], &dU[i][d][pt]); conv = ad_add(conv, &t); }
      ad_t F = ad_add(Ut[i][pt], &conv);
      F.v[0] -= src[i];
      F = ad_scale(F, dens * w);
      ad_t Fd[3];
      for (int d = 0; d < dim; ++d) {
        Fd[d] = ad_scale(dU[i][d][pt], visc);
        if (d == i) Fd[d] = ad_sub(Fd[d], &pr[pt]);
        Fd[d] = ad_scale(Fd[d], w);
      }
      /* the reference's 3-D uz block scatters through uy's offsets (navierstokes.cpp:688) */
      const int rowvar = (dim == 3 && i == 2 && !fix_uz) ? vnum[1] : vnum[i];
      const int v = vnum[i], n = c->varptr[v + 1] - c->varptr[v];
      for (int dof = 0; dof < n; ++dof) {
        const int pos = a->offsets[c->varptr[rowvar] + dof];
        const size_t o = (size_t)dof * nq + pt;
        for (int d = 0; d < dim; ++d) res_add(res, pos, &Fd[d], c->grad[v][o * dim + d]);
        res_add(res, pos, &F, c->basis[v][o]);
      }
      if (useSUPG || usePSPG) {
        stab[i] = ad_scale(ad_add(Ut[i][pt], &conv), dens);
        stab[i] = ad_add(stab[i], &dpr[i][pt]);
        stab[i].v[0] -= dens * src[i];
      }
      if (useSUPG) { /* S_d = tau * stabres * u_d * wts */
        ad_t ts = ad_mul(&tau, &stab[i]);
        for (int dof = 0; dof < n; ++dof) {
          const int pos = a->offsets[c->varptr[rowvar] + dof];
          for (int d = 0; d < dim; ++d) {
            ad_t S = ad_scale(ad_mul(&ts, &vel[d]), w);
            res_add(res, pos, &S, c->grad[v][((size_t)dof * nq + pt) * dim + d]);
          }
        }
      }
    }
    { /* pressure equation: div u * wts (+ PSPG) */
      ad_t divu = ad_c(0.0);
      for (int d = 0; d < dim; ++d) divu = ad_add(divu, &dU[d][d][pt]);
      divu = ad_scale(divu, w);
      const int n = c->varptr[prnum + 1] - c->varptr[prnum];
      for (int dof = 0; dof < n; ++dof) {
        const int pos = a->offsets[c->varptr[prnum] + dof];
        res_add(res, pos, &divu, c->basis[prnum][(size_t)dof * nq + pt]);
        if (usePSPG)
          for (int d = 0; d < dim; ++d) {
            ad_t S = ad_scale(ad_mul(&stab[d], &tau), w / dens);
            res_add(res, pos, &S, c->grad[prnum][((size_t)dof * nq + pt) * dim + d]);
          }
      }
    }
  }
  for (int i = 0; i < dim; ++i) {
    free(U[i]); free(Ut[i]);
    for (int d = 0; d < dim; ++d) free(dU[i][d]);
    free(dpr[i]);
  }
  free(pr);
}

static void thermal_volume(const blk_ctx *c, size_t e, ad_t *res) {
  const orc_block_args *a = c->a;
  const int dim = a->dim, nq = c->nq, n = c->varptr[1];
  ad_t *T = malloc(sizeof(ad_t) * nq), *Tt = malloc(sizeof(ad_t) * nq), *dT[3];
  eval_field(c, 0, F_VAL, 0, T);
  eval_field(c, 0, F_DOT, 0, Tt);
  for (int d = 0; d < dim; ++d) { dT[d] = malloc(sizeof(ad_t) * nq); eval_field(c, 0, F_GRAD, d, dT[d]); }
  for (int pt = 0; pt < nq; ++pt) {
    const double w = c->wts[pt], *x = c->ip + (size_t)pt * dim;
    const double f = eval_func(&a->funcs[0], dim, e, pt, nq, x), kap = eval_func(&a->funcs[1], dim, e, pt, nq, x);
    const double cp = eval_func(&a->funcs[2], dim, e, pt, nq, x), rho = eval_func(&a->funcs[3], dim, e, pt, nq, x);
    ad_t F = ad_scale(Tt[pt], rho * cp);
    F.v[0] -= f;
    F = ad_scale(F, w);
    for (int dof = 0; dof < n; ++dof) {
      const int pos = a->offsets[dof];
      const size_t o = (size_t)dof * nq + pt;
      res_add(res, pos, &F, c->basis[0][o]);
      for (int d = 0; d < dim; ++d) {
        ad_t G = ad_scale(dT[d][pt], kap * w);
        res_add(res, pos, &G, c->grad[0][o * dim + d]);
      }
    }
  }
  free(T); free(Tt);
  for (int d = 0; d < dim; ++d) free(dT[d]);
}

/* shallowwaterHybridized::volumeResidual (shallowwaterHybridized.cpp:113-184) with computeFluxVector(false) (:409-480):
 * res += S_t v w - (F_x dv/dx + F_y dv/dy + source v) w, per equation; params[0] = g */
static void swh_volume(const blk_ctx *c, size_t e, ad_t *res) {
  const orc_block_args *a = c->a;
  const int dim = a->dim, nq = c->nq, nv = dim + 1;
  const double g = a->params[0];
  ad_t *S[3], *St[3];
  for (int i = 0; i < nv; ++i) {
    S[i] = malloc(sizeof(ad_t) * nq); St[i] = malloc(sizeof(ad_t) * nq);
    eval_field(c, i, F_VAL, 0, S[i]);
    eval_field(c, i, F_DOT, 0, St[i]);
  }
  for (int pt = 0; pt < nq; ++pt) {
    const double w = c->wts[pt], *x = c->ip + (size_t)pt * dim;
    const ad_t *H = &S[0][pt], *Hux = &S[1][pt], *Huy = &S[2][pt];
    ad_t F[3][2];
    ad_t hh = ad_scale(ad_mul(H, H), 0.5 * g);
    ad_t uu = ad_mul(Hux, Hux), uv = ad_mul(Hux, Huy), vv = ad_mul(Huy, Huy);
    F[0][0] = *Hux; F[0][1] = *Huy;
    F[1][0] = ad_add(ad_div(&uu, H), &hh); F[1][1] = ad_div(&uv, H);
    F[2][0] = ad_div(&uv, H); F[2][1] = ad_add(ad_div(&vv, H), &hh);
    for (int i = 0; i < nv; ++i) {
      const double src = eval_func(&a->funcs[i], dim, e, pt, nq, x);
      const int n = c->varptr[i + 1] - c->varptr[i];
      for (int dof = 0; dof < n; ++dof) {
        const int pos = a->offsets[c->varptr[i] + dof];
        const size_t o = (size_t)dof * nq + pt;
        res_add(res, pos, &St[i][pt], c->basis[i][o] * w);
        for (int d = 0; d < dim; ++d) res_add(res, pos, &F[i][d], -c->grad[i][o * dim + d] * w);
        res[pos].v[0] += -src * c->basis[i][o] * w;
      }
    }
  }
  for (int i = 0; i < nv; ++i) { free(S[i]); free(St[i]); }
}

static int ctx_init(blk_ctx *c, const orc_block_args *a) {
  int n1;
  memset(c, 0, sizeof(*c));
  c->a = a;
  if (orc_ref_sizes(a->dim, 1, a->qdeg, &n1, &c->nq, &c->nn)) return -1;
  for (int v = 0; v < a->nvars; ++v) {
    const int card = orc_basis_card(a->dim, a->types[v], a->orders[v]);
    if (card < 0) return -1;
    c->varptr[v + 1] = c->varptr[v] + card;
  }
  c->n_tot = c->varptr[a->nvars];
  if (c->n_tot + 1 > ADMAX) return -1;
  for (int v = 0; v < a->nvars; ++v) {
    const int n = c->varptr[v + 1] - c->varptr[v];
    c->basis[v] = malloc(sizeof(double) * n * c->nq * ncomp_of(a->dim, a->types[v]));
    c->grad[v] = malloc(sizeof(double) * n * c->nq * a->dim);
    c->div[v] = malloc(sizeof(double) * n * c->nq);
  }
  c->wts = malloc(sizeof(double) * c->nq);
  c->ip = malloc(sizeof(double) * c->nq * a->dim);
  c->uAD = malloc(sizeof(ad_t) * c->n_tot);
  c->udAD = malloc(sizeof(ad_t) * c->n_tot);
  return 0;
}

static void ctx_free(blk_ctx *c) {
  for (int v = 0; v < c->a->nvars; ++v) { free(c->basis[v]); free(c->grad[v]); free(c->div[v]); }
  free(c->wts); free(c->ip); free(c->uAD); free(c->udAD);
}

/* performGather + computeSoln{Steady,Transient}Seeded (assemblyManager.cpp:3598-3643, workset.cpp:823-859, 559-623) */
static void gather_seed(blk_ctx *c, size_t e) {
  const orc_block_args *a = c->a;
  const int *L = a->lids + e * c->n_tot;
  for (int f = 0; f < c->n_tot; ++f) {
    const int off = a->offsets[f], row = L[off];
    const double cu = a->u[row];
    ad_t ua = ad_c(0.0), ud = ad_c(0.0);
    if (!a->transient) {
      ua.v[0] = cu;
      if (a->compute_jacobian) ua.v[1 + off] = 1.0;
    } else {
      const int st = a->stage, S = a->nstages, NS = a->nsteps;
      const double *cp = a->u_prev + (size_t)row * NS, *cs = a->u_stage + (size_t)row * S;
      const double alpha_u = a->butcher_A[st * S + st] / a->butcher_b[st];
      const double timewt = 1.0 / a->dt / a->butcher_b[st];
      const double alpha_t = a->bdf[0] * timewt;
      double beta_u = (1.0 - alpha_u) * cp[0];
      for (int s = 0; s < st; ++s) beta_u += a->butcher_A[st * S + s] / a->butcher_b[s] * (cs[s] - cp[0]);
      double beta_t = 0.0;
      for (int s = 1; s < NS + 1; ++s) beta_t += a->bdf[s] * cp[s - 1];
      beta_t *= timewt;
      ua.v[0] = alpha_u * cu + beta_u;
      ud.v[0] = alpha_t * cu + beta_t;
      if (a->compute_jacobian) { ua.v[1 + off] = alpha_u; ud.v[1 + off] = alpha_t; }
    }
    c->uAD[f] = ua;
    c->udAD[f] = ud;
  }
}

/* scatter (assemblyManager.cpp:4031-4145): residual gets -val, Jacobian +dx, fixed rows skipped */
static void scatter(const blk_ctx *c, size_t e, const ad_t *res) {
  const orc_block_args *a = c->a;
  const int n = c->n_tot;
  const int *L = a->lids + e * n;
  for (int row = 0; row < n; ++row) {
    const ad_t *r = &res[row];
    if (a->local_res) a->local_res[e * n + row] -= r->v[0];
    if (a->local_J && a->compute_jacobian)
      for (int col = 0; col < n; ++col) a->local_J[(e * n + row) * n + col] += r->v[1 + col];
    const int rowIndex = L[row];
    if (a->fixed && a->fixed[rowIndex]) continue;
    if (a->res) a->res[rowIndex] += -r->v[0];
    if (a->crs_vals && a->compute_jacobian)
      for (int col = 0; col < n; ++col) {
        const int gcol = L[col];
        for (int p = a->rowptr[rowIndex]; p < a->rowptr[rowIndex + 1]; ++p)
          if (a->colind[p] == gcol) { a->crs_vals[p] += r->v[1 + col]; break; }
      }
  }
}

int orc_assemble_block(const orc_block_args *a) {
  blk_ctx c;
  if (ctx_init(&c, a)) return -1;
  g_w1 = c.n_tot + 1;
  ad_t *res = malloc(sizeof(ad_t) * c.n_tot);
  for (size_t e = 0; e < (size_t)a->nelem; ++e) {
    const double *xn = a->nodes + e * c.nn * a->dim;
    for (int v = 0; v < a->nvars; ++v)
      orc_physical_basis_var(a->dim, a->types[v], a->orders[v], a->qdeg, 1, xn,
                             a->orient ? a->orient + e * c.n_tot : NULL, 0, c.varptr[v], c.basis[v], c.grad[v], c.div[v],
                             v == 0 ? c.wts : NULL, v == 0 ? c.ip : NULL);
    gather_seed(&c, e);
    for (int k = 0; k < c.n_tot; ++k) res[k] = ad_c(0.0);
    if (a->physics == ORC_PHYS_POROUS_MIXED) porous_volume(&c, e, res);
    else if (a->physics == ORC_PHYS_NAVIERSTOKES) ns_volume(&c, e, res);
    else if (a->physics == ORC_PHYS_THERMAL) thermal_volume(&c, e, res);
    else if (a->physics == ORC_PHYS_SHALLOWWATER_HYBRIDIZED && a->dim == 2) swh_volume(&c, e, res);
    else { free(res); ctx_free(&c); return -2; }
    scatter(&c, e, res);
  }
  free(res);
  ctx_free(&c);
  return 0;
}

int orc_get_mass(const orc_block_args *a, const double *masswts, double *mass) {
  blk_ctx c;
  if (ctx_init(&c, a)) return -1;
  const int n = c.n_tot, nq = c.nq, dim = a->dim;
  for (size_t e = 0; e < (size_t)a->nelem; ++e) {
    const double *xn = a->nodes + e * c.nn * dim;
    for (int v = 0; v < a->nvars; ++v)
      orc_physical_basis_var(dim, a->types[v], a->orders[v], a->qdeg, 1, xn, a->orient ? a->orient + e * n : NULL, 0,
                             c.varptr[v], c.basis[v], c.grad[v], c.div[v], v == 0 ? c.wts : NULL, NULL);
    for (int v = 0; v < a->nvars; ++v) {
      const int card = c.varptr[v + 1] - c.varptr[v], nc = ncomp_of(dim, a->types[v]);
      const double mwt = masswts ? masswts[v] : 1.0;
      for (int i = 0; i < card; ++i)
        for (int j = 0; j < card; ++j)
          for (int k = 0; k < nq; ++k)
            for (int d = 0; d < nc; ++d)
              mass[(e * n + a->offsets[c.varptr[v] + i]) * n + a->offsets[c.varptr[v] + j]] +=
                  c.basis[v][((size_t)i * nq + k) * nc + d] * c.basis[v][((size_t)j * nq + k) * nc + d] * c.wts[k] * mwt;
    }
  }
  ctx_free(&c);
  return 0;
}

/* shallowwaterHybridized::boundaryResidual (shallowwaterHybridized.cpp:190-263): res(off_i(dof)) += (F(Shat).n +
 * stab)_i wts basis_side(dof) on interface sides, the boundary term B on Far-field / Slip sides (computeFlux :270-368
 * stores the same quantity per point).  The eigendecomposition depends on the trace state only, so the flux is affine
 * in the interior state on interface / far-field sides: flux = c + M (S_AD - Shat); Slip is written out in AD ops. */
static int swh_boundary(const orc_block_args *a) {
  blk_ctx c;
  if (ctx_init(&c, a)) return -1;
  g_w1 = c.n_tot + 1;
  int ns, nqs;
  orc_side_sizes(a->dim, a->qdeg, &ns, &nqs);
  const int dim = 2, nv = 3, nb = a->nb, order = a->orders[0], n = c.varptr[1];
  const double g = a->params[0];
  const int roe = a->params[1] != 0.0, stype = a->bc_type - 10;
  double *wts = malloc(sizeof(double) * (size_t)nb * nqs), *nrm = malloc(sizeof(double) * (size_t)nb * nqs * dim);
  double *bas = malloc(sizeof(double) * (size_t)nb * n * nqs);
  orc_physical_side_basis(dim, order, a->qdeg, nb, a->nodes, a->belem, a->bside, wts, nrm, NULL, bas, NULL);
  ad_t *res = malloc(sizeof(ad_t) * c.n_tot);
  for (int k = 0; k < nb; ++k) {
    const size_t e = (size_t)a->belem[k];
    gather_seed(&c, e);
    for (int j = 0; j < c.n_tot; ++j) res[j] = ad_c(0.0);
    for (int pt = 0; pt < nqs; ++pt) {
      const double w = wts[(size_t)k * nqs + pt], *nq = nrm + ((size_t)k * nqs + pt) * dim;
      const double *Sh = a->aux_ip + ((size_t)k * nqs + pt) * nv;
      const double *Sinf = a->farfield_ip ? a->farfield_ip + ((size_t)k * nqs + pt) * nv : Sh;
      ad_t S[3], flux[3];
      double Sv[3];
      for (int i = 0; i < nv; ++i) { /* side solution fields (evaluateSideSolutionField, workset.cpp:1069-1176) */
        S[i] = ad_c(0.0);
        for (int dof = 0; dof < n; ++dof) {
          const double b = bas[((size_t)k * n + dof) * nqs + pt];
          const ad_t *s = &c.uAD[c.varptr[i] + dof];
          for (int j = 0; j < g_w1; ++j) S[i].v[j] += s->v[j] * b;
        }
        Sv[i] = S[i].v[0];
      }
      if (stype == 2) { /* Slip (:729-745) */
        ad_t ux = ad_div(&S[1], &S[0]), uy = ad_div(&S[2], &S[0]);
        ad_t vn = ad_scale(ux, nq[0]);
        ad_t t = ad_scale(uy, nq[1]);
        vn = ad_add(vn, &t);
        flux[0] = S[0]; flux[0].v[0] -= Sh[0];
        ad_t a1 = ad_scale(vn, nq[0]), a2 = ad_scale(vn, nq[1]);
        flux[1] = ad_sub(ux, &a1); flux[1].v[0] -= Sh[1] / Sh[0];
        flux[2] = ad_sub(uy, &a2); flux[2].v[0] -= Sh[2] / Sh[0];
      } else { /* value from the point functions, slope M = d flux / d S by columns (affine in S) */
        double f0[3], M[9];
        orc_swh_interface_flux(dim, stype, roe, Sv, Sh, Sinf, nq, g, f0);
        for (int j = 0; j < nv; ++j) {
          double Sp[3] = {Sv[0], Sv[1], Sv[2]}, f1[3];
          Sp[j] += 1.0;
          orc_swh_interface_flux(dim, stype, roe, Sp, Sh, Sinf, nq, g, f1);
          for (int i = 0; i < nv; ++i) M[i * 3 + j] = f1[i] - f0[i];
        }
        for (int i = 0; i < nv; ++i) {
          flux[i] = ad_c(f0[i]);
          for (int j = 0; j < nv; ++j)
            for (int q = 1; q < g_w1; ++q) flux[i].v[q] += M[i * 3 + j] * S[j].v[q];
        }
      }
      for (int i = 0; i < nv; ++i)
        for (int dof = 0; dof < n; ++dof)
          res_add(res, a->offsets[c.varptr[i] + dof], &flux[i], w * bas[((size_t)k * n + dof) * nqs + pt]);
    }
    scatter(&c, e, res);
  }
  free(wts); free(nrm); free(bas); free(res);
  ctx_free(&c);
  return 0;
}

int orc_assemble_block_boundary(const orc_block_args *a) {
  if (a->physics == ORC_PHYS_SHALLOWWATER_HYBRIDIZED) return (a->dim == 2 && a->bc_type >= 10 && a->bc_type <= 12 && a->aux_ip) ? swh_boundary(a) : -2;
  blk_ctx c;
  if (ctx_init(&c, a)) return -1;
  if (a->physics != ORC_PHYS_POROUS_MIXED || a->bc_type != 1) { ctx_free(&c); return -2; }
  g_w1 = c.n_tot + 1;
  int ns, nqs;
  orc_side_sizes(a->dim, a->qdeg, &ns, &nqs);
  const int dim = a->dim, unum = 1, nu = 2 * dim, nb = a->nb;
  double *wts = malloc(sizeof(double) * (size_t)nb * nqs), *nrm = malloc(sizeof(double) * (size_t)nb * nqs * dim);
  double *ip = malloc(sizeof(double) * (size_t)nb * nqs * dim), *sb = malloc(sizeof(double) * (size_t)nb * nu * nqs * dim);
  orc_physical_side_basis(dim, 1, a->qdeg, nb, a->nodes, a->belem, a->bside, wts, nrm, ip, NULL, NULL);
  orc_physical_side_basis_hdiv(dim, a->qdeg, nb, a->nodes, a->belem, a->bside, a->orient, c.n_tot, c.varptr[unum], sb);
  ad_t *res = malloc(sizeof(ad_t) * c.n_tot);
  for (int k = 0; k < nb; ++k) {
    const size_t e = (size_t)a->belem[k];
    for (int j = 0; j < c.n_tot; ++j) res[j] = ad_c(0.0);
    /* res(off_u(dof)) += bsource * wts * (v . n)   (porousMixed.cpp:400-418) */
    for (int pt = 0; pt < nqs; ++pt) {
      ad_t src = ad_c(eval_func(&a->bdata, dim, (size_t)k, pt, nqs, ip + ((size_t)k * nqs + pt) * dim) *
                      wts[(size_t)k * nqs + pt]);
      for (int dof = 0; dof < nu; ++dof) {
        double vdotn = 0.0;
        for (int d = 0; d < dim; ++d)
          vdotn += sb[(((size_t)k * nu + dof) * nqs + pt) * dim + d] * nrm[((size_t)k * nqs + pt) * dim + d];
        res_add(res, a->offsets[c.varptr[unum] + dof], &src, vdotn);
      }
    }
    scatter(&c, e, res);
  }
  free(wts); free(nrm); free(ip); free(sb); free(res);
  ctx_free(&c);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* shallowwaterHybridized: HDG element, side part, derivative arrays of width 36 */
/* ------------------------------------------------------------------------ */

static ad_t ad_abs(const ad_t *a) { return a->v[0] < 0.0 ? ad_scale(*a, -1.0) : *a; }
static ad_t ad_max(const ad_t *a, const ad_t *b) { return a->v[0] > b->v[0] ? *a : *b; }

static void adm_matvec(const ad_t A[3][3], const ad_t *x, ad_t *y) {
  for (int i = 0; i < 3; ++i) {
    y[i] = ad_c(0.0);
    for (int j = 0; j < 3; ++j) { ad_t t = ad_mul(&A[i][j], &x[j]); y[i] = ad_add(y[i], &t); }
  }
}

/* eigendecompFluxJacobian, 2-D (:793-823) on AD numbers */
static void adm_eigen(const ad_t *Sh, double nx, double ny, double g, ad_t L[3][3], ad_t *lam, ad_t R[3][3]) {
  ad_t ux = ad_div(&Sh[1], &Sh[0]), uy = ad_div(&Sh[2], &Sh[0]);
  ad_t t1 = ad_scale(ux, nx), t2 = ad_scale(uy, ny), vn = ad_add(t1, &t2);
  ad_t gh = ad_scale(Sh[0], g), a = ad_sqrt(&gh), one = ad_c(1.0), zero = ad_c(0.0);
  ad_t anx = ad_scale(a, nx), any = ad_scale(a, ny);
  R[0][0] = one; R[1][0] = ad_add(ux, &anx); R[2][0] = ad_add(uy, &any);
  R[0][1] = zero; R[1][1] = ad_scale(any, -1.0); R[2][1] = anx;
  R[0][2] = one; R[1][2] = ad_sub(ux, &anx); R[2][2] = ad_sub(uy, &any);
  ad_t two_a = ad_scale(a, 2.0), vn2a = ad_div(&vn, &two_a), half = ad_c(0.5);
  ad_t nx2a = ad_c(nx); nx2a = ad_div(&nx2a, &two_a);
  ad_t ny2a = ad_c(ny); ny2a = ad_div(&ny2a, &two_a);
  ad_t nxa = ad_c(nx); nxa = ad_div(&nxa, &a);
  ad_t nya = ad_c(ny); nya = ad_div(&nya, &a);
  L[0][0] = ad_sub(half, &vn2a); L[0][1] = nx2a; L[0][2] = ny2a;
  { ad_t p1 = ad_scale(ux, ny), p2 = ad_scale(uy, nx), d = ad_sub(p1, &p2); L[1][0] = ad_div(&d, &a); }
  L[1][1] = ad_scale(nya, -1.0); L[1][2] = nxa;
  L[2][0] = ad_add(half, &vn2a); L[2][1] = ad_scale(nx2a, -1.0); L[2][2] = ad_scale(ny2a, -1.0);
  lam[0] = ad_add(vn, &a); lam[1] = vn; lam[2] = ad_sub(vn, &a);
}

/* computeFlux (:270-368) with computeFluxVector(true), computeStabilizationTerm, computeBoundaryTerm on AD numbers */
static void adm_interface_flux(int stype, int roe, const ad_t *S, const ad_t *Sh, const double *Sinf, double nx, double ny,
                               double g, ad_t *out) {
  if (stype == 0) {
    ad_t hh = ad_scale(ad_mul(&Sh[0], &Sh[0]), 0.5 * g);
    ad_t uu = ad_mul(&Sh[1], &Sh[1]), uv = ad_mul(&Sh[1], &Sh[2]), vv = ad_mul(&Sh[2], &Sh[2]);
    ad_t F[3][2];
    F[0][0] = Sh[1]; F[0][1] = Sh[2];
    F[1][0] = ad_add(ad_div(&uu, &Sh[0]), &hh); F[1][1] = ad_div(&uv, &Sh[0]);
    F[2][0] = ad_div(&uv, &Sh[0]); F[2][1] = ad_add(ad_div(&vv, &Sh[0]), &hh);
    ad_t dS[3], st[3];
    for (int i = 0; i < 3; ++i) dS[i] = ad_sub(S[i], &Sh[i]);
    if (roe) {
      ad_t L[3][3], lam[3], R[3][3], tmp[3];
      adm_eigen(Sh, nx, ny, g, L, lam, R);
      adm_matvec(L, dS, tmp);
      for (int i = 0; i < 3; ++i) { ad_t al = ad_abs(&lam[i]); tmp[i] = ad_mul(&tmp[i], &al); }
      adm_matvec(R, tmp, st);
    } else {
      ad_t ux = ad_div(&Sh[1], &Sh[0]), uy = ad_div(&Sh[2], &Sh[0]);
      ad_t t1 = ad_scale(ux, nx), t2 = ad_scale(uy, ny), vn = ad_add(t1, &t2);
      ad_t gh = ad_scale(Sh[0], g), a = ad_sqrt(&gh);
      ad_t p = ad_add(vn, &a), m = ad_sub(vn, &a), ap = ad_abs(&p), am = ad_abs(&m), lmax = ad_max(&ap, &am);
      for (int i = 0; i < 3; ++i) st[i] = ad_mul(&dS[i], &lmax);
    }
    for (int i = 0; i < 3; ++i) {
      ad_t fx = ad_scale(F[i][0], nx), fy = ad_scale(F[i][1], ny);
      out[i] = ad_add(ad_add(fx, &fy), &st[i]);
    }
  } else if (stype == 1) {
    ad_t L[3][3], lam[3], R[3][3], tmp[3], dS[3], neg[3];
    adm_eigen(Sh, nx, ny, g, L, lam, R);
    for (int i = 0; i < 3; ++i) dS[i] = ad_sub(S[i], &Sh[i]);
    adm_matvec(L, dS, tmp);
    for (int i = 0; i < 3; ++i) { ad_t al = ad_abs(&lam[i]); ad_t f = ad_scale(ad_add(lam[i], &al), 0.5); tmp[i] = ad_mul(&tmp[i], &f); }
    adm_matvec(R, tmp, out);
    for (int i = 0; i < 3; ++i) { dS[i] = ad_scale(Sh[i], -1.0); dS[i].v[0] += Sinf[i]; }
    adm_matvec(L, dS, tmp);
    for (int i = 0; i < 3; ++i) { ad_t al = ad_abs(&lam[i]); ad_t f = ad_scale(ad_sub(lam[i], &al), 0.5); tmp[i] = ad_mul(&tmp[i], &f); }
    adm_matvec(R, tmp, neg);
    for (int i = 0; i < 3; ++i) out[i] = ad_sub(out[i], &neg[i]);
  } else {
    ad_t ux = ad_div(&S[1], &S[0]), uy = ad_div(&S[2], &S[0]);
    ad_t t1 = ad_scale(ux, nx), t2 = ad_scale(uy, ny), vn = ad_add(t1, &t2);
    ad_t hx = ad_div(&Sh[1], &Sh[0]), hy = ad_div(&Sh[2], &Sh[0]);
    ad_t a1 = ad_scale(vn, nx), a2 = ad_scale(vn, ny);
    out[0] = ad_sub(S[0], &Sh[0]);
    out[1] = ad_sub(ad_sub(ux, &a1), &hx);
    out[2] = ad_sub(ad_sub(uy, &a2), &hy);
  }
}

int orc_swh_hdg_element(const orc_block_args *a, const double *lambda, const unsigned char *side_types,
                        const double *farfield, double *blocks, double *res_out) {
  blk_ctx c;
  if (a->dim != 2 || a->nvars != 3 || ctx_init(&c, a)) return -1;
  const int dim = 2, n = c.varptr[1], ni = 3 * n, nl = 24, W = ni + nl;
  if (n != 4 || W + 1 > ADMAX) { ctx_free(&c); return -2; }
  g_w1 = W + 1;
  int ns, nqs;
  orc_side_sizes(dim, a->qdeg, &ns, &nqs);
  static const int hface_edge[4] = {1, 2, 3, 0}; /* shards side -> HFACE edge (left, bottom, right, top) */
  const double g = a->params[0];
  const int roe = a->params[1] != 0.0;
  double *sip = malloc(sizeof(double) * ns * nqs * dim), *sw = malloc(sizeof(double) * nqs);
  double *tu = malloc(sizeof(double) * ns * dim), *tv = malloc(sizeof(double) * ns * dim);
  double *sb = malloc(sizeof(double) * ns * n * nqs), *sg = malloc(sizeof(double) * ns * n * nqs * dim);
  double *snv = malloc(sizeof(double) * ns * 4 * nqs), *sng = malloc(sizeof(double) * ns * 4 * nqs * dim);
  orc_side_tables(dim, 1, a->qdeg, sip, sw, tu, tv, sb, sg, snv, sng);
  double *wts = malloc(sizeof(double) * nqs), *nrm = malloc(sizeof(double) * nqs * dim);
  ad_t *R = malloc(sizeof(ad_t) * W), *lamAD = malloc(sizeof(ad_t) * nl);
  for (size_t e = 0; e < (size_t)a->nelem; ++e) {
    gather_seed(&c, e); /* interior unknowns: value and derivative slot = position in the LID list */
    /* re-seed in the element-local numbering of this routine: interior (variable, dof) -> slot f, traces -> ni + t */
    ad_t uloc[12];
    for (int f = 0; f < ni; ++f) {
      uloc[f] = ad_c(c.uAD[f].v[0]);
      uloc[f].v[1 + f] = c.uAD[f].v[1 + a->offsets[f]];
    }
    for (int t = 0; t < nl; ++t) { lamAD[t] = ad_c(lambda[e * nl + t]); lamAD[t].v[1 + ni + t] = 1.0; }
    for (int r = 0; r < W; ++r) R[r] = ad_c(0.0);
    const int be = (int)e;
    for (int s = 0; s < 4; ++s) {
      orc_physical_side_basis(dim, 1, a->qdeg, 1, a->nodes, &be, &s, wts, nrm, NULL, NULL, NULL);
      const int edge = hface_edge[s];
      for (int pt = 0; pt < nqs; ++pt) {
        const double *xr = sip + (s * nqs + pt) * dim;
        const double tcoord = (edge == 0 || edge == 2) ? xr[1] : xr[0];
        const double mu[2] = {0.5 * (1.0 - tcoord), 0.5 * (1.0 + tcoord)};
        ad_t S[3], Sh[3], flux[3];
        for (int i = 0; i < 3; ++i) {
          S[i] = ad_c(0.0);
          for (int dof = 0; dof < n; ++dof) { ad_t t = ad_scale(uloc[i * n + dof], sb[(s * n + dof) * nqs + pt]); S[i] = ad_add(S[i], &t); }
          Sh[i] = ad_c(0.0);
          for (int k = 0; k < 2; ++k) { ad_t t = ad_scale(lamAD[i * 8 + edge * 2 + k], mu[k]); Sh[i] = ad_add(Sh[i], &t); }
        }
        adm_interface_flux(side_types[e * 4 + s], roe, S, Sh, farfield, nrm[pt * dim], nrm[pt * dim + 1], g, flux);
        for (int i = 0; i < 3; ++i) {
          for (int dof = 0; dof < n; ++dof) res_add(R, i * n + dof, &flux[i], wts[pt] * sb[(s * n + dof) * nqs + pt]);
          for (int k = 0; k < 2; ++k) res_add(R, ni + i * 8 + edge * 2 + k, &flux[i], wts[pt] * mu[k]);
        }
      }
    }
    for (int r = 0; r < W; ++r) {
      res_out[e * W + r] = -R[r].v[0];
      for (int col = 0; col < W; ++col) blocks[(e * W + r) * W + col] = R[r].v[1 + col];
    }
  }
  free(sip); free(sw); free(tu); free(tv); free(sb); free(sg); free(snv); free(sng); free(wts); free(nrm); free(R); free(lamAD);
  ctx_free(&c);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* PhysicsInterface::fluxConditions (physicsInterface.cpp:1702-1762)         */
/* ------------------------------------------------------------------------ */

/* One variable with bctype "Flux" on the sides of a boundary group:
 *   res(elem, off(dof)) += -fluxvals(elem,pt) * wts(elem,pt) * basis(elem,dof,pt,0)      (:1727-1733)
 * followed by scatterRes (assemblyManager.cpp:3943-3978): the global vector receives -res.val() at
 * LIDs(elem, off(dof)), fixed rows skipped.  flux[nb][nqs], wts[nb][nqs], basis[nb][card][nqs][ncomp] are the
 * group's stored side views (wts_side, getBasisSide(var)); off[card] = offsets of the variable.          */
int orc_flux_condition(int nb, int card, int nqs, int ncomp, const int *belem, const int *lids, int n_tot,
                       const int *off, const unsigned char *fixed, const double *flux, const double *wts,
                       const double *basis, double *res) {
  if (nb < 0 || card <= 0 || nqs <= 0 || ncomp <= 0) return -1;
  for (int k = 0; k < nb; ++k) {
    const int *L = lids + (size_t)belem[k] * n_tot;
    for (int dof = 0; dof < card; ++dof) {
      double r = 0.0; /* res(elem, off(dof)).val() */
      for (int pt = 0; pt < nqs; ++pt)
        r += -flux[(size_t)k * nqs + pt] * wts[(size_t)k * nqs + pt] * basis[(((size_t)k * card + dof) * nqs + pt) * ncomp];
      const int row = L[off[dof]];
      if (fixed && fixed[row]) continue;
      res[row] -= r;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* L2-projection systems: setInitial, setDirichlet                           */
/* ------------------------------------------------------------------------ */

/* KokkosSparse sumIntoValues / replaceValues on a sorted CRS row: columns the row does not hold are ignored */
static int prj_find(const int *rowptr, const int *colind, int row, int col) {
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
    if (colind[k] == col) return k;
  return -1;
}

/* getInitial(project = true) for one variable (assemblyManager.cpp:7643-7682) followed by the vector loop of setInitial
 * (:1243-1254): initialvals(e, off(dof)) += data(e,pt,c) basis(e,dof,pt,c) wts(e,pt); rhs[LIDs(e,row)] += initialvals.
 * data[E][nq][ncomp], basis[E][card][nq][ncomp], wts[E][nq]                                                       */
int orc_project_rhs(int nelem, int card, int nq, int ncomp, const int *lids, int n_tot, const int *off, const double *data,
                    const double *basis, const double *wts, double *rhs) {
  if (nelem < 0 || card <= 0 || nq <= 0 || ncomp <= 0) return -1;
  for (size_t e = 0; e < (size_t)nelem; ++e)
    for (int dof = 0; dof < card; ++dof) {
      double v = 0.0;
      for (int pt = 0; pt < nq; ++pt)
        for (int c = 0; c < ncomp; ++c)
          v += data[(e * nq + pt) * ncomp + c] * basis[((e * card + dof) * nq + pt) * ncomp + c] * wts[e * nq + pt];
      rhs[lids[e * n_tot + off[dof]]] += v;
    }
  return 0;
}

/* matrix loop of setInitial (:1256-1280) over dense element mass matrices mass[E][n][n] (getMass), then fix_zero_rows
 * (:1284-1302)                                                                                                   */
int orc_set_initial_mass(int nelem, int n_tot, const int *lids, const double *mass, int lump, int nrows, const int *rowptr,
                         const int *colind, double *vals) {
  for (size_t e = 0; e < (size_t)nelem; ++e) {
    const int *L = lids + e * n_tot;
    for (int row = 0; row < n_tot; ++row) {
      const int rowIndex = L[row];
      for (int col = 0; col < n_tot; ++col) {
        const int c = lump ? rowIndex : L[col];
        const int k = prj_find(rowptr, colind, rowIndex, c);
        if (k >= 0) vals[k] += mass[(e * n_tot + row) * n_tot + col];
      }
    }
  }
  for (int row = 0; row < nrows; ++row) {
    double abssum = 0.0;
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) abssum += fabs(vals[k]);
    if (abssum < 1.0e-14) {
      const int k = prj_find(rowptr, colind, row, row);
      if (k >= 0) vals[k] = 1.0;
    }
  }
  return 0;
}

/* setInitial(set, initial, useadjoint) (:1830-1850) with getInitial(project = false) (:7683-7727): nodal values of one
 * variable; vals[E][nnodes] = "initial <var>" at the element's vertices.  The reference's order-1 HGRAD dof k sits on
 * vertex k (Intrepid2); this oracle's tables are in tensor order, so the vertex of dof k is vert_of_dof[k] (NULL = k).  */
int orc_set_initial_nodal(int nelem, int nnodes, const int *lids, int n_tot, const int *off, const int *vert_of_dof,
                          const double *vals, double *initial) {
  for (size_t e = 0; e < (size_t)nelem; ++e)
    for (int dof = 0; dof < nnodes; ++dof)
      initial[lids[e * n_tot + off[dof]]] = vals[e * nnodes + (vert_of_dof ? vert_of_dof[dof] : dof)];
  return 0;
}

/* getDirichletBoundary (:6288-6350) and getMassBoundary (:6360-6425) for one variable of a boundary group:
 * dvals[nb][n_tot], mass[nb][n_tot][n_tot] accumulated; dip[nb][nqs], basis[nb][card][nqs][ncomp], wts[nb][nqs],
 * normals[nb][nqs][ncomp] (HDIV only)                                                                             */
int orc_dirichlet_boundary(int nb, int card, int nqs, int ncomp, int n_tot, const int *off, int hdiv, const double *dip,
                           const double *basis, const double *wts, const double *normals, double *dvals, double *mass) {
  if (nb < 0 || card <= 0 || nqs <= 0 || ncomp <= 0) return -1;
  for (size_t e = 0; e < (size_t)nb; ++e)
    for (int i = 0; i < card; ++i) {
      for (int j = 0; j < nqs; ++j) {
        const double *bi = basis + ((e * card + i) * nqs + j) * ncomp;
        if (!hdiv) dvals[e * n_tot + off[i]] += dip[e * nqs + j] * bi[0] * wts[e * nqs + j];
        else
          for (int c = 0; c < ncomp; ++c)
            dvals[e * n_tot + off[i]] += dip[e * nqs + j] * bi[c] * normals[(e * nqs + j) * ncomp + c] * wts[e * nqs + j];
      }
      for (int j = 0; j < card; ++j)
        for (int k = 0; k < nqs; ++k) {
          const double *bi = basis + ((e * card + i) * nqs + k) * ncomp, *bj = basis + ((e * card + j) * nqs + k) * ncomp;
          double *m = mass + (e * n_tot + off[i]) * n_tot + off[j];
          if (!hdiv) *m += bi[0] * bj[0] * wts[e * nqs + k];
          else
            for (int c = 0; c < ncomp; ++c) {
              const double nc = normals[(e * nqs + k) * ncomp + c];
              *m += bi[c] * nc * bj[c] * nc * wts[e * nqs + k];
            }
        }
    }
  return 0;
}

/* setDirichlet, the boundary-group loop (:1870-1917) for one group: fixed rows only; lumped: the row total goes to the
 * column left in cols[0] by the summing loop = LIDs(c, n_tot - 1)                                                    */
int orc_set_dirichlet_group(int nb, int n_tot, const int *belem, const int *lids, const unsigned char *fixed,
                            const double *dvals, const double *mass, int lump, const int *rowptr, const int *colind,
                            double *vals, double *rhs) {
  for (size_t c = 0; c < (size_t)nb; ++c) {
    const int *L = lids + (size_t)belem[c] * n_tot;
    for (int row = 0; row < n_tot; ++row) {
      const int rowIndex = L[row];
      if (!(fixed && fixed[rowIndex])) continue;
      rhs[rowIndex] += dvals[c * n_tot + row];
      if (lump) {
        int col0 = 0;
        double totalval = 0.0;
        for (int col = 0; col < n_tot; ++col) {
          col0 = L[col];
          totalval += mass[(c * n_tot + row) * n_tot + col];
        }
        const int k = prj_find(rowptr, colind, rowIndex, col0);
        if (k >= 0) vals[k] += totalval;
      } else {
        for (int col = 0; col < n_tot; ++col) {
          const int k = prj_find(rowptr, colind, rowIndex, L[col]);
          if (k >= 0) vals[k] += mass[(c * n_tot + row) * n_tot + col];
        }
      }
    }
  }
  return 0;
}

/* setDirichlet, the closing loop (:1920-1938): ones on the diagonal of the rows that are not fixed */
int orc_set_dirichlet_identity(int nelem, int n_tot, const int *lids, const unsigned char *fixed, const int *rowptr,
                               const int *colind, double *vals) {
  for (size_t c = 0; c < (size_t)nelem; ++c)
    for (int row = 0; row < n_tot; ++row) {
      const int rowIndex = lids[c * n_tot + row];
      if (fixed && fixed[rowIndex]) continue;
      const int k = prj_find(rowptr, colind, rowIndex, rowIndex);
      if (k >= 0) vals[k] = 1.0;
    }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* applyMassMatrixFree, Sparse3DView, the basis database                     */
/* ------------------------------------------------------------------------ */

/* AssemblyManager::applyMassMatrixFree, the !storeMass branch (assemblyManager.cpp:1607-1672): per element and variable
 * massval(i,j) = sum_k basis(e,i,k,:) . basis(e,j,k,:) wts(e,k) mwt, y(LIDs(e,off(i))) += massval x(LIDs(e,off(j))) --
 * the basis recomputed per element (getPhysicalVolumetricBasis), sequential (no atomics).  y is accumulated into.     */
int orc_apply_mass_matrix_free(const orc_block_args *a, const double *masswts, const double *x, double *y) {
  blk_ctx c;
  if (ctx_init(&c, a)) return -1;
  const int n = c.n_tot, nq = c.nq, dim = a->dim;
  for (size_t e = 0; e < (size_t)a->nelem; ++e) {
    const double *xn = a->nodes + e * c.nn * dim;
    const int *L = a->lids + e * n;
    for (int v = 0; v < a->nvars; ++v)
      orc_physical_basis_var(dim, a->types[v], a->orders[v], a->qdeg, 1, xn, a->orient ? a->orient + e * n : NULL, 0,
                             c.varptr[v], c.basis[v], c.grad[v], c.div[v], v == 0 ? c.wts : NULL, NULL);
    for (int v = 0; v < a->nvars; ++v) {
      const int card = c.varptr[v + 1] - c.varptr[v], nc = ncomp_of(dim, a->types[v]);
      const double mwt = masswts ? masswts[v] : 1.0;
      const int *off = a->offsets + c.varptr[v];
      for (int i = 0; i < card; ++i)
        for (int j = 0; j < card; ++j) {
          double massval = 0.0;
          for (int k = 0; k < nq; ++k)
            for (int d = 0; d < nc; ++d)
              massval += c.basis[v][((size_t)i * nq + k) * nc + d] * c.basis[v][((size_t)j * nq + k) * nc + d] * c.wts[k] * mwt;
          y[L[off[i]]] += massval * x[L[off[j]]];
        }
    }
  }
  ctx_free(&c);
  return 0;
}

/* The stored-mass branches (assemblyManager.cpp:1674-1772): dense per-element mass (index == NULL) or the database mass
 * of the element's representative (index[e]), only the (var, var) blocks:
 *   y(LIDs(e,off(var,i))) += mass(eindex, off(var,i), off(var,j)) x(LIDs(e,off(var,j))).                            */
int orc_apply_mass_stored(int nelem, int n_tot, int nvars, const int *varptr, const int *offsets, const int *lids,
                          const int *index, const double *mass, const double *x, double *y) {
  for (size_t e = 0; e < (size_t)nelem; ++e) {
    const int *L = lids + e * n_tot;
    const double *M = mass + (size_t)(index ? index[e] : (int)e) * n_tot * n_tot;
    for (int v = 0; v < nvars; ++v)
      for (int i = varptr[v]; i < varptr[v + 1]; ++i)
        for (int j = varptr[v]; j < varptr[v + 1]; ++j)
          y[L[offsets[i]]] += M[(size_t)offsets[i] * n_tot + offsets[j]] * x[L[offsets[j]]];
  }
  return 0;
}

/* Sparse3DView(denseview, tol) (src/tools/sparse3DView.hpp:32-92): entries with |a|/max|a| > tol are kept, row by row,
 * in column order.  First call with values == NULL returns maxent; nnz_row[E][n], values / columns [E][n][maxent].  */
int orc_sparse3d(int nelem, int n, const double *dense, double tol, int *maxent, int *nnz_row, double *values, int *columns) {
  double maxval = 0.0;
  const size_t tot = (size_t)nelem * n * n;
  for (size_t k = 0; k < tot; ++k)
    if (fabs(dense[k]) > maxval) maxval = fabs(dense[k]);
  int me = 0;
  for (size_t r = 0; r < (size_t)nelem * n; ++r) {
    int nnz = 0;
    for (int j = 0; j < n; ++j)
      if (fabs(dense[r * n + j]) / maxval > tol) ++nnz;
    if (nnz_row) nnz_row[r] = nnz;
    if (nnz > me) me = nnz;
  }
  *maxent = me;
  if (!values || !columns) return 0;
  for (size_t r = 0; r < (size_t)nelem * n; ++r) {
    int prog = 0;
    for (int j = 0; j < n; ++j)
      if (fabs(dense[r * n + j]) / maxval > tol) {
        columns[r * me + prog] = j;
        values[r * me + prog] = dense[r * n + j];
        ++prog;
      }
  }
  return 0;
}

/* Sparse3DView::setLocalColumns + the sparse branch of applyMassMatrixFree (sparse3DView.hpp:128-146,
 * assemblyManager.cpp:1690-1726): local_columns(e,row,k) = j with offsets(var,j) == columns(e,row,k);
 * y(LIDs(elem,localrow)) += values(eindex,localrow,k) x(LIDs(elem, offsets(var, local_columns(eindex,localrow,k)))). */
int orc_apply_mass_sparse(int nelem, int n_tot, int nvars, const int *varptr, const int *offsets, const int *lids,
                          const int *index, int maxent, const int *nnz_row, const double *values, const int *columns,
                          const double *x, double *y) {
  for (size_t e = 0; e < (size_t)nelem; ++e) {
    const int *L = lids + e * n_tot;
    const size_t ei = (size_t)(index ? index[e] : (int)e);
    for (int v = 0; v < nvars; ++v)
      for (int i = varptr[v]; i < varptr[v + 1]; ++i) {
        const int localrow = offsets[i];
        for (int k = 0; k < nnz_row[ei * n_tot + localrow]; ++k) {
          const int col = columns[(ei * n_tot + localrow) * maxent + k];
          int lc = -1; /* setLocalColumns: only columns of the same variable are found */
          for (int j = varptr[v]; j < varptr[v + 1]; ++j)
            if (offsets[j] == col) lc = j;
          if (lc < 0) continue; /* the reference leaves local_columns 0 there; with block-diagonal mass it never happens */
          y[L[localrow]] += values[(ei * n_tot + localrow) * maxent + k] * x[L[offsets[lc]]];
        }
      }
  }
  return 0;
}

/* AssemblyManager::identifyVolumetricDatabase (assemblyManager.cpp:4314-4467): elements in order; an element joins the
 * first earlier representative with (1) the same orientation, (2) |measure - ref| / ref < tol, (3) at every
 * integration point ||J - J_ref||_F / ||J||_F <= tol; otherwise it becomes a representative.  index[E]; returns the
 * number of representatives (first_users[] receives their element ids, capacity nelem).                            */
int orc_identify_database(const orc_block_args *a, double tol, int *index, int *first_users) {
  int n1, nq, nn;
  if (orc_ref_sizes(a->dim, 1, a->qdeg, &n1, &nq, &nn)) return -1;
  const int dim = a->dim;
  int n_tot = 0;
  for (int v = 0; v < a->nvars; ++v) n_tot += orc_basis_card(dim, a->types[v], a->orders[v]);
  double *rip = malloc(sizeof(double) * nq * dim), *rw = malloc(sizeof(double) * nq);
  double *rb1 = malloc(sizeof(double) * n1 * nq), *rg1 = malloc(sizeof(double) * n1 * nq * dim);
  double *nv = malloc(sizeof(double) * nn * nq), *ng = malloc(sizeof(double) * nn * nq * dim);
  orc_ref_tables(dim, 1, a->qdeg, rip, rw, rb1, rg1, nv, ng);
  const size_t js = (size_t)nq * dim * dim;
  double *jac = malloc(sizeof(double) * (size_t)a->nelem * js), *meas = malloc(sizeof(double) * a->nelem);
  for (size_t e = 0; e < (size_t)a->nelem; ++e) {
    const double *xn = a->nodes + e * nn * dim;
    meas[e] = 0.0;
    for (int q = 0; q < nq; ++q) {
      double *J = jac + e * js + (size_t)q * dim * dim, Ji[9], det;
      for (int r = 0; r < dim; ++r)
        for (int c2 = 0; c2 < dim; ++c2) {
          double s = 0.0;
          for (int v = 0; v < nn; ++v) s += xn[v * dim + r] * ng[(v * nq + q) * dim + c2];
          J[r * dim + c2] = s;
        }
      jac_inv_det(dim, J, Ji, &det);
      meas[e] += rw[q] * det;
    }
  }
  int nu = 0;
  for (int e = 0; e < a->nelem; ++e) {
    int found = -1;
    for (int p = 0; p < nu && found < 0; ++p) {
      const int r = first_users[p];
      if (a->orient && memcmp(a->orient + (size_t)e * n_tot, a->orient + (size_t)r * n_tot, n_tot)) continue;
      if (!(fabs((meas[e] - meas[r]) / meas[r]) < tol)) continue;
      int ruled_out = 0;
      for (int q = 0; q < nq && !ruled_out; ++q) {
        double fronorm = 0.0, frodiff = 0.0;
        for (int k = 0; k < dim * dim; ++k) {
          const double je = jac[(size_t)e * js + (size_t)q * dim * dim + k], d = je - jac[(size_t)r * js + (size_t)q * dim * dim + k];
          frodiff += d * d;
          fronorm += je * je;
        }
        if (sqrt(frodiff) / sqrt(fronorm) > tol) ruled_out = 1;
      }
      if (!ruled_out) found = p;
    }
    if (found < 0) { found = nu; first_users[nu++] = e; }
    index[e] = found;
  }
  free(rip); free(rw); free(rb1); free(rg1); free(nv); free(ng); free(jac); free(meas);
  return nu;
}
