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

}  // namespace mha
