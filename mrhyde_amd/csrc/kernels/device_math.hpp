// device_math.hpp -- small device helpers shared by the assembly kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/mrhyde_amd.h"
#include "device_types.hpp"

namespace mha {

// Inverse and determinant of the cell Jacobian (CellTools::setJacobianInv/Det,
// reference: src/interfaces/discretizationInterface.cpp:923-929).
template <int DIM>
__device__ __forceinline__ void invert(const double *J, double *Ji, double &det);

template <>
__device__ __forceinline__ void invert<2>(const double *J, double *Ji, double &det) {
  det = J[0] * J[3] - J[1] * J[2];
  const double r = 1.0 / det;
  Ji[0] = J[3] * r; Ji[1] = -J[1] * r; Ji[2] = -J[2] * r; Ji[3] = J[0] * r;
}

template <>
__device__ __forceinline__ void invert<3>(const double *J, double *Ji, double &det) {
  const double c0 = J[4] * J[8] - J[5] * J[7], c1 = J[5] * J[6] - J[3] * J[8], c2 = J[3] * J[7] - J[4] * J[6];
  det = J[0] * c0 + J[1] * c1 + J[2] * c2;
  const double r = 1.0 / det;
  Ji[0] = c0 * r; Ji[1] = (J[2] * J[7] - J[1] * J[8]) * r; Ji[2] = (J[1] * J[5] - J[2] * J[4]) * r;
  Ji[3] = c1 * r; Ji[4] = (J[0] * J[8] - J[2] * J[6]) * r; Ji[5] = (J[2] * J[3] - J[0] * J[5]) * r;
  Ji[6] = c2 * r; Ji[7] = (J[1] * J[6] - J[0] * J[7]) * r; Ji[8] = (J[0] * J[4] - J[1] * J[3]) * r;
}

// Value of a named function at integration point (e,q) with physical coordinates x.
template <int DIM>
__device__ __forceinline__ double eval_func(const FuncDesc &f, int e, int q, int nq, const double *x) {
  if (f.kind == MHA_FUNC_CONSTANT) return f.amp;
  if (f.kind == MHA_FUNC_IP_ARRAY) return f.ip[(size_t)e * nq + q];
  double s = f.amp;
#pragma unroll
  for (int d = 0; d < DIM; ++d) s *= sin(f.freq[d] * x[d]);
  return s;
}

// Position of column `col` in CRS row [lo,hi) (ascending colind), or -1.
// Plays the role of the column search inside KokkosSparse sumIntoValues
// (reference call site: src/managers/assemblyManager.cpp:4138).
__device__ __forceinline__ int find_col(const int32_t *colind, int lo, int hi, int col) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int c = colind[mid];
    if (c == col) return mid;
    if (c < col) lo = mid + 1; else hi = mid;
  }
  return -1;
}

}  // namespace mha
