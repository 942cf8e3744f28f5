#include "ref_tables.hpp"

#include <cmath>

#include "common.hpp"

namespace mha {

int gauss_points_for_degree(int degree) { return degree / 2 + 1; }  // ceil((degree+1)/2)

namespace {
// Legendre P_n and P_n' by the three-term recurrence.
void legendre(int n, double x, double &pn, double &dpn) {
  double pm = 1.0, p = x;
  if (n == 0) { pn = 1.0; dpn = 0.0; return; }
  for (int k = 1; k < n; ++k) {
    const double pk = ((2 * k + 1) * x * p - k * pm) / (k + 1);
    pm = p;
    p = pk;
  }
  pn = p;
  dpn = n * (pm - x * p) / (1.0 - x * x);
}
}  // namespace

void gauss_legendre_descending(int n, std::vector<double> &pts, std::vector<double> &wts) {
  pts.assign(n, 0.0);
  wts.assign(n, 0.0);
  const double pi = std::acos(-1.0);
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double x = std::cos(pi * (4 * i + 3) / (4.0 * n + 2.0));  // Chebyshev-like start, largest root first
    double pn, dpn;
    for (int it = 0; it < 64; ++it) {
      legendre(n, x, pn, dpn);
      const double step = pn / dpn;
      x -= step;
      if (std::fabs(step) <= 1e-17 * std::fabs(x) + 1e-300) break;
    }
    legendre(n, x, pn, dpn);
    const double w = 2.0 / ((1.0 - x * x) * dpn * dpn);
    pts[i] = x;
    pts[n - 1 - i] = -x;
    wts[i] = wts[n - 1 - i] = w;
  }
  if (n & 1) pts[n / 2] = 0.0;
}

void lagrange_equispaced(int order, double x, double *val, double *der) {
  const int m = order + 1;
  std::vector<double> node(m);
  for (int k = 0; k < m; ++k) node[k] = -1.0 + 2.0 * k / order;
  for (int k = 0; k < m; ++k) {
    double v = 1.0, d = 0.0;
    for (int a = 0; a < m; ++a) {
      if (a == k) continue;
      v *= (x - node[a]) / (node[k] - node[a]);
      double t = 1.0 / (node[k] - node[a]);
      for (int b = 0; b < m; ++b)
        if (b != k && b != a) t *= (x - node[b]) / (node[k] - node[b]);
      d += t;
    }
    val[k] = v;
    der[k] = d;
  }
}

double ref_vertex_sign(int dim, int v, int d) {
  // Quadrilateral_4: (-,-) (+,-) (+,+) (-,+); Hexahedron_8: that loop at z=-1 then at z=+1
  if (d == 2) return (v >= 4) ? 1.0 : -1.0;
  const int c = v & 3;
  if (d == 0) return (c == 1 || c == 2) ? 1.0 : -1.0;
  (void)dim;
  return (c >= 2) ? 1.0 : -1.0;
}

RefTables make_ref_tables(int dim, int order, int quad_degree) {
  MHA_REQUIRE(dim == 2 || dim == 3, MHA_ERR_INVALID, "dimension must be 2 or 3");
  MHA_REQUIRE(order >= 1 && order <= 8, MHA_ERR_INVALID, "HGRAD order must be in [1,8]");
  RefTables t;
  t.dim = dim;
  t.order = order;
  t.nq1 = gauss_points_for_degree(quad_degree);
  t.nbasis = ipow(order + 1, dim);
  t.nq = ipow(t.nq1, dim);
  t.nnodes = 1 << dim;
  gauss_legendre_descending(t.nq1, t.gauss_pts, t.gauss_wts);
  const int m = order + 1;
  t.phi1d.assign(m * t.nq1, 0.0);
  t.dphi1d.assign(m * t.nq1, 0.0);
  std::vector<double> v(m), dv(m);
  for (int q = 0; q < t.nq1; ++q) {
    lagrange_equispaced(order, t.gauss_pts[q], v.data(), dv.data());
    for (int k = 0; k < m; ++k) { t.phi1d[k * t.nq1 + q] = v[k]; t.dphi1d[k * t.nq1 + q] = dv[k]; }
  }
  t.ip.assign(t.nq * dim, 0.0);
  t.wts.assign(t.nq, 0.0);
  t.basis.assign(t.nbasis * t.nq, 0.0);
  t.grad.assign(t.nbasis * t.nq * dim, 0.0);
  t.nodeval.assign(t.nnodes * t.nq, 0.0);
  t.nodegrad.assign(t.nnodes * t.nq * dim, 0.0);
  for (int q = 0; q < t.nq; ++q) {
    int qi[3] = {q % t.nq1, (q / t.nq1) % t.nq1, q / (t.nq1 * t.nq1)};
    double w = 1.0;
    for (int d = 0; d < dim; ++d) { t.ip[q * dim + d] = t.gauss_pts[qi[d]]; w *= t.gauss_wts[qi[d]]; }
    t.wts[q] = w;
    for (int f = 0; f < t.nbasis; ++f) {
      int fi[3] = {f % m, (f / m) % m, f / (m * m)};
      double val = 1.0;
      for (int d = 0; d < dim; ++d) val *= t.phi1d[fi[d] * t.nq1 + qi[d]];
      t.basis[f * t.nq + q] = val;
      for (int d = 0; d < dim; ++d) {
        double g = 1.0;
        for (int e = 0; e < dim; ++e)
          g *= (e == d) ? t.dphi1d[fi[e] * t.nq1 + qi[e]] : t.phi1d[fi[e] * t.nq1 + qi[e]];
        t.grad[(f * t.nq + q) * dim + d] = g;
      }
    }
    for (int v_ = 0; v_ < t.nnodes; ++v_) {
      double val = 1.0;
      for (int d = 0; d < dim; ++d) val *= 0.5 * (1.0 + ref_vertex_sign(dim, v_, d) * t.ip[q * dim + d]);
      t.nodeval[v_ * t.nq + q] = val;
      for (int d = 0; d < dim; ++d) {
        double g = 1.0;
        for (int e = 0; e < dim; ++e) {
          const double s = ref_vertex_sign(dim, v_, e);
          g *= (e == d) ? 0.5 * s : 0.5 * (1.0 + s * t.ip[q * dim + e]);
        }
        t.nodegrad[(v_ * t.nq + q) * dim + d] = g;
      }
    }
  }
  return t;
}

SideTables make_side_tables(const RefTables &ref) {
  static const int quad_side[4][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}};
  static const int hex_side[6][4] = {{0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {0, 4, 7, 3}, {0, 3, 2, 1}, {4, 5, 6, 7}};
  const int dim = ref.dim, nq1 = ref.nq1, m = ref.order + 1, n = ref.nbasis, nn = ref.nnodes;
  SideTables t;
  t.nsides = 2 * dim;
  t.nqs = (dim == 2) ? nq1 : nq1 * nq1;
  const int ns = t.nsides, nqs = t.nqs, nsv = (dim == 2) ? 2 : 4;
  t.ip.assign(ns * nqs * dim, 0.0);
  t.wts.assign(nqs, 0.0);
  t.tanU.assign(ns * dim, 0.0);
  t.tanV.assign(ns * dim, 0.0);
  t.basis.assign(ns * n * nqs, 0.0);
  t.grad.assign(ns * n * nqs * dim, 0.0);
  t.nodeval.assign(ns * nn * nqs, 0.0);
  t.nodegrad.assign(ns * nn * nqs * dim, 0.0);
  for (int q = 0; q < nqs; ++q)
    t.wts[q] = (dim == 2) ? ref.gauss_wts[q] : ref.gauss_wts[q % nq1] * ref.gauss_wts[q / nq1];
  std::vector<double> bv(3 * m), bd(3 * m);
  for (int s = 0; s < ns; ++s) {
    double v[4][3] = {{0}};
    for (int k = 0; k < nsv; ++k)
      for (int d = 0; d < dim; ++d) v[k][d] = ref_vertex_sign(dim, dim == 2 ? quad_side[s][k] : hex_side[s][k], d);
    for (int d = 0; d < dim; ++d) {
      if (dim == 2) {
        t.tanU[s * dim + d] = 0.5 * (v[1][d] - v[0][d]);
      } else {
        t.tanU[s * dim + d] = 0.25 * (-v[0][d] + v[1][d] + v[2][d] - v[3][d]);
        t.tanV[s * dim + d] = 0.25 * (-v[0][d] - v[1][d] + v[2][d] + v[3][d]);
      }
    }
    for (int q = 0; q < nqs; ++q) {
      double x[3] = {0, 0, 0};
      if (dim == 2) {
        const double a = ref.gauss_pts[q];
        for (int d = 0; d < dim; ++d) x[d] = 0.5 * (1 - a) * v[0][d] + 0.5 * (1 + a) * v[1][d];
      } else {
        const double a = ref.gauss_pts[q % nq1], b = ref.gauss_pts[q / nq1];
        for (int d = 0; d < dim; ++d)
          x[d] = 0.25 * ((1 - a) * (1 - b) * v[0][d] + (1 + a) * (1 - b) * v[1][d] + (1 + a) * (1 + b) * v[2][d] +
                         (1 - a) * (1 + b) * v[3][d]);
      }
      for (int d = 0; d < dim; ++d) {
        t.ip[(s * nqs + q) * dim + d] = x[d];
        lagrange_equispaced(ref.order, x[d], &bv[d * m], &bd[d * m]);
      }
      for (int f = 0; f < n; ++f) {
        const int fi[3] = {f % m, (f / m) % m, f / (m * m)};
        double val = 1.0;
        for (int d = 0; d < dim; ++d) val *= bv[d * m + fi[d]];
        t.basis[(s * n + f) * nqs + q] = val;
        for (int d = 0; d < dim; ++d) {
          double g = 1.0;
          for (int e = 0; e < dim; ++e) g *= (e == d) ? bd[e * m + fi[e]] : bv[e * m + fi[e]];
          t.grad[((s * n + f) * nqs + q) * dim + d] = g;
        }
      }
      for (int k = 0; k < nn; ++k) {
        double val = 1.0;
        for (int d = 0; d < dim; ++d) val *= 0.5 * (1.0 + ref_vertex_sign(dim, k, d) * x[d]);
        t.nodeval[(s * nn + k) * nqs + q] = val;
        for (int d = 0; d < dim; ++d) {
          double g = 1.0;
          for (int e = 0; e < dim; ++e) {
            const double sg = ref_vertex_sign(dim, k, e);
            g *= (e == d) ? 0.5 * sg : 0.5 * (1.0 + sg * x[e]);
          }
          t.nodegrad[((s * nn + k) * nqs + q) * dim + d] = g;
        }
      }
    }
  }
  return t;
}

int ref_basis_var(int dim, int type, int order, int npts, const double *x, std::vector<double> &val,
                  std::vector<double> &grad, std::vector<double> &div) {
  val.clear();
  grad.clear();
  div.clear();
  if (type == MHA_BASIS_HVOL) {
    val.assign(npts, 1.0);
    return 1;
  }
  if (type == MHA_BASIS_HDIV) {
    const int card = 2 * dim;
    val.assign(static_cast<size_t>(card) * npts * dim, 0.0);
    div.assign(static_cast<size_t>(card) * npts, 0.0);
    for (int c = 0; c < dim; ++c)
      for (int sd = 0; sd < 2; ++sd)
        for (int q = 0; q < npts; ++q) {
          const double xc = x[q * dim + c];
          val[(static_cast<size_t>(2 * c + sd) * npts + q) * dim + c] = sd ? 0.5 * (1.0 + xc) : 0.5 * (1.0 - xc);
          div[static_cast<size_t>(2 * c + sd) * npts + q] = sd ? 0.5 : -0.5;
        }
    return card;
  }
  MHA_REQUIRE(type == MHA_BASIS_HGRAD && order >= 1 && order <= 8, MHA_ERR_INVALID, "unsupported basis (type " << type << ", order " << order << ")");
  const int m = order + 1, card = ipow(m, dim);
  val.assign(static_cast<size_t>(card) * npts, 0.0);
  grad.assign(static_cast<size_t>(card) * npts * dim, 0.0);
  std::vector<double> bv(3 * m), bd(3 * m);
  for (int q = 0; q < npts; ++q) {
    for (int d = 0; d < dim; ++d) lagrange_equispaced(order, x[q * dim + d], &bv[d * m], &bd[d * m]);
    for (int f = 0; f < card; ++f) {
      const int fi[3] = {f % m, (f / m) % m, f / (m * m)};
      double v = 1.0;
      for (int d = 0; d < dim; ++d) v *= bv[d * m + fi[d]];
      val[static_cast<size_t>(f) * npts + q] = v;
      for (int d = 0; d < dim; ++d) {
        double g = 1.0;
        for (int e = 0; e < dim; ++e) g *= (e == d) ? bd[e * m + fi[e]] : bv[e * m + fi[e]];
        grad[(static_cast<size_t>(f) * npts + q) * dim + d] = g;
      }
    }
  }
  return card;
}

}  // namespace mha
