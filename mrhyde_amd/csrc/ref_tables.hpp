// ref_tables.hpp -- reference-element data of one block (host side).
// Mirrors what DiscretizationInterface::getBasis/getQuadrature/setReferenceData hand to the
// groups (reference: src/interfaces/discretizationInterface.cpp:346-478, 483-665):
// tensor Lagrange HGRAD on equispaced nodes (dof index x-fastest), tensor Gauss-Legendre
// cubature (point index x-fastest, each direction descending -- pinned by
// regression/discretization/HGRAD/mrhyde.gold), and the C1 nodal geometry basis in shards order.
#pragma once
#include <vector>

namespace mha {

struct RefTables {
  int dim = 0, order = 0, nq1 = 0;
  int nbasis = 0, nq = 0, nnodes = 0;
  std::vector<double> gauss_pts, gauss_wts;     // 1-D rule [nq1]
  std::vector<double> phi1d, dphi1d;            // 1-D Lagrange at 1-D points: [order+1][nq1]
  std::vector<double> ip;                       // [nq][dim]
  std::vector<double> wts;                      // [nq]
  std::vector<double> basis;                    // [nbasis][nq]
  std::vector<double> grad;                     // [nbasis][nq][dim]
  std::vector<double> nodeval;                  // [nnodes][nq]
  std::vector<double> nodegrad;                 // [nnodes][nq][dim]
};

int gauss_points_for_degree(int degree);
void gauss_legendre_descending(int n, std::vector<double> &pts, std::vector<double> &wts);
void lagrange_equispaced(int order, double x, double *val, double *der);
RefTables make_ref_tables(int dim, int order, int quad_degree);
// sign (+1/-1) of reference vertex v of the cell topology in direction d (shards order)
double ref_vertex_sign(int dim, int v, int d);

}  // namespace mha
