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

// Reference data of the cell's sides (setReferenceData, reference: src/interfaces/discretizationInterface.cpp:523-548):
// the (dim-1)-cubature mapped onto each side by CellTools::mapToReferenceSubcell, the reference edge tangent (2-D) or
// face tangents (3-D), and the cell basis / geometry basis evaluated at those points.  Sides in shards order:
// quad edges {0,1},{1,2},{2,3},{3,0}; hex faces {0,1,5,4},{1,2,6,5},{2,3,7,6},{0,4,7,3},{0,3,2,1},{4,5,6,7}.
struct SideTables {
  int nsides = 0, nqs = 0;
  std::vector<double> ip;         // [ns][nqs][dim] in cell reference coordinates
  std::vector<double> wts;        // [nqs]
  std::vector<double> tanU, tanV; // [ns][dim]
  std::vector<double> basis;      // [ns][nbasis][nqs]
  std::vector<double> grad;       // [ns][nbasis][nqs][dim]
  std::vector<double> nodeval;    // [ns][nnodes][nqs]
  std::vector<double> nodegrad;   // [ns][nnodes][nqs][dim]
};

int gauss_points_for_degree(int degree);
void gauss_legendre_descending(int n, std::vector<double> &pts, std::vector<double> &wts);
void lagrange_equispaced(int order, double x, double *val, double *der);
RefTables make_ref_tables(int dim, int order, int quad_degree);
SideTables make_side_tables(const RefTables &ref);
// Reference values of one variable's basis at npts points x[npts][dim] of the reference cell (what Basis::getValues
// returns for OPERATOR_VALUE / GRAD / DIV): type MHA_BASIS_HGRAD (tensor Lagrange of `order`), _HVOL (constant 1),
// _HDIV (lowest order, raw In_FEM functions: dof 2c = (1-x_c)/2 e_c, dof 2c+1 = (1+x_c)/2 e_c).
// val[card][npts][ncomp] (ncomp = dim for HDIV, else 1), grad[card][npts][dim] (HGRAD), div[card][npts] (HDIV);
// arrays a type does not define are left empty.  Returns the cardinality.
int ref_basis_var(int dim, int type, int order, int npts, const double *x, std::vector<double> &val,
                  std::vector<double> &grad, std::vector<double> &div);
// sign (+1/-1) of reference vertex v of the cell topology in direction d (shards order)
double ref_vertex_sign(int dim, int v, int d);

}  // namespace mha
