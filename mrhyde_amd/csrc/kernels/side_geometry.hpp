// side_geometry.hpp -- geometry of a cell side at its integration points (shared by the boundary kernels).
#pragma once
#include "device_math.hpp"

namespace mha {

// cell Jacobian and physical point at side point q of local side s
template <int DIM>
__device__ __forceinline__ void side_point_J(const double *xn, const SideTablesDev &st, int s, int q, double *J, double *x) {
  constexpr int NN = 1 << DIM;
  const int nqs = st.nqs;
#pragma unroll
  for (int r = 0; r < DIM; ++r) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double sum = 0.0;
      for (int k = 0; k < NN; ++k) sum += xn[k * DIM + r] * st.nodegrad[((s * NN + k) * nqs + q) * DIM + c];
      J[r * DIM + c] = sum;
    }
    double sum = 0.0;
    for (int k = 0; k < NN; ++k) sum += xn[k * DIM + r] * st.nodeval[(s * NN + k) * nqs + q];
    x[r] = sum;
  }
}

// Side geometry at side point q of local side s: J^{-1}, unit normal, measure-weighted cubature weight and the
// physical point.  2-D: t = J t_ref, n = R t with R = [[0,1],[-1,0]], w = |t| w_ref; 3-D: n = (J tU) x (J tV),
// w = |n| w_ref (discretizationInterface.cpp:1684-1710); normals rescaled to unit length (:1760-1786).
template <int DIM>
__device__ __forceinline__ void side_point(const double *xn, const SideTablesDev &st, int s, int q, double *Ji,
                                           double *nrm, double &w, double *x) {
  double J[DIM * DIM], det;
  side_point_J<DIM>(xn, st, s, q, J, x);
  invert<DIM>(J, Ji, det);
  double len;
  if constexpr (DIM == 2) {
    const double tx = J[0] * st.tanU[s * 2] + J[1] * st.tanU[s * 2 + 1];
    const double ty = J[2] * st.tanU[s * 2] + J[3] * st.tanU[s * 2 + 1];
    nrm[0] = ty;
    nrm[1] = -tx;
    len = sqrt(tx * tx + ty * ty);
  } else {
    double a[3], b[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      a[r] = J[r * 3] * st.tanU[s * 3] + J[r * 3 + 1] * st.tanU[s * 3 + 1] + J[r * 3 + 2] * st.tanU[s * 3 + 2];
      b[r] = J[r * 3] * st.tanV[s * 3] + J[r * 3 + 1] * st.tanV[s * 3 + 1] + J[r * 3 + 2] * st.tanV[s * 3 + 2];
    }
    nrm[0] = a[1] * b[2] - a[2] * b[1];
    nrm[1] = a[2] * b[0] - a[0] * b[2];
    nrm[2] = a[0] * b[1] - a[1] * b[0];
    len = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
  }
  w = len * st.wts[q];
  const double r = 1.0 / len;
#pragma unroll
  for (int d = 0; d < DIM; ++d) nrm[d] *= r;
}

}  // namespace mha
