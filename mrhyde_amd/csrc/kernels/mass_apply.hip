// mass_apply.hip -- y += M x without assembling M, and the Sparse3DView storage of element mass matrices.
//
// Replaces AssemblyManager::applyMassMatrixFree (src/managers/assemblyManager.cpp:1582-1778) -- the explicit time
// integrators' mass solve applies the block mass matrix element by element -- in its four forms:
//   on the fly (!storeMass, :1607-1672): basis and weights recomputed per element, here never materialised: per element
//       z_q = sum_j x_j phi_j(q), y_i += sum_q phi_i(q) . z_q w_q mwt  (two passes over the reference tables instead of
//       the reference's n^2 nq loop), HGRAD / HVOL values and HDIV Piola values with the caller's orientation signs
//   stored dense element mass (:1760-1772), dense database mass of the element's representative (:1730-1755),
//   database mass in Sparse3DView storage (:1690-1726)
// and Sparse3DView's constructor (src/tools/sparse3DView.hpp:32-92: keep |a| / max|a| > tol, row by row in column
// order) and setLocalColumns (:128-146) as device kernels.
// M is block diagonal by variable (only (var, var) couplings are applied, as in the reference).  The scatter uses f64
// atomics like the reference's device build (use_atomics_); not on any timed path of bench.py.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int kMaWaves = 4, kMaMaxQ = 128, kMaMaxN = 128;

template <int DIM>
__global__ __launch_bounds__(64 * kMaWaves) void mass_apply_free_kernel(BlockDev b, VarLayoutDev vl, double mw0, double mw1,
                                                                        double mw2, double mw3, double mw4, double mw5,
                                                                        double mw6, double mw7, const double *__restrict__ x,
                                                                        double *y) {
  constexpr int NN = 1 << DIM;
  __shared__ double s_x[kMaWaves][kMaMaxN], s_J[kMaWaves][kMaMaxQ][DIM * DIM + 2], s_z[kMaWaves][kMaMaxQ][DIM];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, n = vl.n_tot, nq = vl.nq;
  const double mw[8] = {mw0, mw1, mw2, mw3, mw4, mw5, mw6, mw7};
  for (int el = blockIdx.x * kMaWaves + wv; el < b.e_count; el += gridDim.x * kMaWaves) {
    const int e = b.e_begin + el;
    const int32_t *L = b.lids + (size_t)e * n;
    const double *xn = b.nodes + (size_t)e * NN * DIM;
    for (int f = lane; f < n; f += 64) {
      const double sg = vl.orient ? (double)vl.orient[(size_t)e * n + f] : 1.0;
      s_x[wv][f] = sg * x[L[b.offsets[f]]];
    }
    for (int q = lane; q < nq; q += 64) {
      double J[DIM * DIM], Ji[DIM * DIM], det;
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double s = 0.0;
          for (int v = 0; v < NN; ++v) s += xn[v * DIM + r] * b.nodegrad[(v * nq + q) * DIM + c];
          J[r * DIM + c] = s;
        }
      invert<DIM>(J, Ji, det);
#pragma unroll
      for (int i = 0; i < DIM * DIM; ++i) s_J[wv][q][i] = J[i];
      s_J[wv][q][DIM * DIM] = det;
      s_J[wv][q][DIM * DIM + 1] = b.ref_wts[q] * det;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int v = 0; v < vl.nvars; ++v) {
      const int card = vl.card[v], ns = vl.nslot[v], cp = vl.cardpad[v], vp = vl.varptr[v];
      const bool hdiv = vl.type[v] == MHA_BASIS_HDIV;
      const int nc = hdiv ? DIM : 1;
      const double *T = vl.tables + vl.table_off[v];
      for (int q = lane; q < nq; q += 64) {  // z at the point, folded with J^T J / det^2 (HDIV), the weight and mwt
        double zr[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) zr[c] = 0.0;
        for (int j = 0; j < card; ++j) {
          const double xj = s_x[wv][vp + j];
          for (int c = 0; c < nc; ++c) zr[c] += xj * T[(q * ns + c) * cp + j];
        }
        const double wm = s_J[wv][q][DIM * DIM + 1] * mw[v];
        if (hdiv) {
          const double *J = s_J[wv][q];
          const double rd = 1.0 / J[DIM * DIM];
          double zp[DIM];
#pragma unroll
          for (int r = 0; r < DIM; ++r) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) s += J[r * DIM + c] * zr[c];
            zp[r] = s * rd;
          }
#pragma unroll
          for (int c = 0; c < DIM; ++c) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) s += J[r * DIM + c] * zp[r];
            s_z[wv][q][c] = s * rd * wm;
          }
        } else {
          s_z[wv][q][0] = zr[0] * wm;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int i = lane; i < card; i += 64) {
        double yi = 0.0;
        for (int q = 0; q < nq; ++q)
          for (int c = 0; c < nc; ++c) yi += T[(q * ns + c) * cp + i] * s_z[wv][q][c];
        const double sg = vl.orient ? (double)vl.orient[(size_t)e * n + vp + i] : 1.0;
        unsafeAtomicAdd(y + L[b.offsets[vp + i]], sg * yi);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// stored mass: one thread per (element, flattened variable dof); dense [*][n][n] or Sparse3DView storage
__global__ __launch_bounds__(256) void mass_apply_stored_kernel(BlockDev b, VarLayoutDev vl, const int32_t *index,
                                                                const double *mass, int maxent, const int32_t *nnz_row,
                                                                const double *values, const int32_t *columns,
                                                                const int32_t *pos_var, const double *__restrict__ x,
                                                                double *y) {
  const int n = vl.n_tot;
  const size_t total = (size_t)b.e_count * n;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int e = b.e_begin + (int)(t / n), i = (int)(t % n);
    int v = 0;
    while (i >= vl.varptr[v + 1]) ++v;
    const int32_t *L = b.lids + (size_t)e * n;
    const size_t ei = index ? (size_t)index[e] : (size_t)e;
    const int localrow = b.offsets[i];
    double s = 0.0;
    if (mass) {
      const double *Mr = mass + (ei * n + localrow) * n;
      for (int j = vl.varptr[v]; j < vl.varptr[v + 1]; ++j) s += Mr[b.offsets[j]] * x[L[b.offsets[j]]];
    } else {
      const size_t r = ei * n + localrow;
      for (int k = 0; k < nnz_row[r]; ++k) {
        const int col = columns[r * maxent + k];
        if (pos_var[col] == v) s += values[r * maxent + k] * x[L[col]];  // setLocalColumns finds same-variable columns only
      }
    }
    unsafeAtomicAdd(y + L[localrow], s);
  }
}

// ---- Sparse3DView ----
__global__ __launch_bounds__(256) void s3d_max_kernel(const double *dense, size_t total, unsigned long long *maxbits) {
  double m = 0.0;
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) m = fmax(m, fabs(dense[k]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(maxbits, (unsigned long long)__double_as_longlong(m));  // non-negative doubles order as integers
}

__global__ __launch_bounds__(256) void s3d_count_kernel(const double *dense, size_t rows, int n, double tol,
                                                        const unsigned long long *maxbits, int32_t *nnz_row, int *maxent) {
  const double maxval = __longlong_as_double((long long)*maxbits);
  int me = 0;
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
    int nnz = 0;
    for (int j = 0; j < n; ++j) nnz += (fabs(dense[r * n + j]) / maxval > tol) ? 1 : 0;
    nnz_row[r] = nnz;
    me = max(me, nnz);
  }
  for (int o = 32; o > 0; o >>= 1) me = max(me, __shfl_xor(me, o));
  if ((threadIdx.x & 63) == 0) atomicMax(maxent, me);
}

__global__ __launch_bounds__(256) void s3d_fill_kernel(const double *dense, size_t rows, int n, double tol,
                                                       const unsigned long long *maxbits, int maxent, double *values,
                                                       int32_t *columns) {
  const double maxval = __longlong_as_double((long long)*maxbits);
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
    int prog = 0;
    for (int j = 0; j < n; ++j)
      if (fabs(dense[r * n + j]) / maxval > tol) {
        columns[r * maxent + prog] = j;
        values[r * maxent + prog] = dense[r * n + j];
        ++prog;
      }
  }
}

inline int grid_of(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 65536); }

}  // namespace

void launch_mass_apply_free(const BlockDev &b, const VarLayoutDev &vl, const double *masswts, const double *x, double *y,
                            hipStream_t stream) {
  if (b.e_count <= 0) return;
  MHA_REQUIRE(vl.nq <= kMaMaxQ && vl.n_tot <= kMaMaxN, MHA_ERR_INVALID,
              "matrix-free mass apply: element with " << vl.n_tot << " dofs / " << vl.nq << " points exceeds the kernel's LDS tables");
  double mw[8];
  for (int v = 0; v < 8; ++v) mw[v] = (masswts && v < vl.nvars) ? masswts[v] : 1.0;
  const int grid = std::min((b.e_count + kMaWaves - 1) / kMaWaves, 256 * 8);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kMaWaves), 0, stream, b, vl, mw[0], mw[1], mw[2], mw[3], mw[4], mw[5], mw[6],
                       mw[7], x, y);
  };
  if (b.dim == 2) go(mass_apply_free_kernel<2>); else go(mass_apply_free_kernel<3>);
  MHA_HIP(hipGetLastError());
}

void launch_mass_apply_stored(const BlockDev &b, const VarLayoutDev &vl, const int32_t *index, const double *mass, int maxent,
                              const int32_t *nnz_row, const double *values, const int32_t *columns, const int32_t *pos_var,
                              const double *x, double *y, hipStream_t stream) {
  if (b.e_count <= 0) return;
  hipLaunchKernelGGL(mass_apply_stored_kernel, dim3(grid_of((size_t)b.e_count * vl.n_tot)), dim3(256), 0, stream, b, vl, index,
                     mass, maxent, nnz_row, values, columns, pos_var, x, y);
  MHA_HIP(hipGetLastError());
}

void launch_sparse3d_max(const double *dense, size_t total, unsigned long long *maxbits, hipStream_t stream) {
  hipLaunchKernelGGL(s3d_max_kernel, dim3(grid_of(total)), dim3(256), 0, stream, dense, total, maxbits);
  MHA_HIP(hipGetLastError());
}
void launch_sparse3d_count(const double *dense, size_t rows, int n, double tol, const unsigned long long *maxbits,
                           int32_t *nnz_row, int *maxent, hipStream_t stream) {
  hipLaunchKernelGGL(s3d_count_kernel, dim3(grid_of(rows)), dim3(256), 0, stream, dense, rows, n, tol, maxbits, nnz_row, maxent);
  MHA_HIP(hipGetLastError());
}
void launch_sparse3d_fill(const double *dense, size_t rows, int n, double tol, const unsigned long long *maxbits, int maxent,
                          double *values, int32_t *columns, hipStream_t stream) {
  hipLaunchKernelGGL(s3d_fill_kernel, dim3(grid_of(rows)), dim3(256), 0, stream, dense, rows, n, tol, maxbits, maxent, values,
                     columns);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
