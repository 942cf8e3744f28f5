// projection.hip -- the L2-projection systems of initial and strong-Dirichlet data (mass matrix + right-hand side).
//
// Replaces, for one variable of a block on a list of elements (a workset's range) or of (element, side) entries:
//   AssemblyManager::getInitial (project / nodal)       src/managers/assemblyManager.cpp:7632-7728
//   PhysicsInterface::getInitial                         src/interfaces/physicsInterface.cpp:898-1024
//   AssemblyManager::setInitial (rhs + mass, zero rows)  src/managers/assemblyManager.cpp:1208-1305
//   AssemblyManager::setInitial (nodal values)           src/managers/assemblyManager.cpp:1830-1850
//   AssemblyManager::getDirichletBoundary                src/managers/assemblyManager.cpp:6288-6350
//   AssemblyManager::getMassBoundary                     src/managers/assemblyManager.cpp:6360-6425
//   AssemblyManager::setDirichlet                        src/managers/assemblyManager.cpp:1855-1943
// The element matrices never exist in memory: every (entry, row dof) thread forms its row of the variable's mass block
// from the stored basis views and adds it to the CRS row.  Setup path, not on the timed path.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

__device__ __forceinline__ int crs_find(const BlockDev &b, int row, int col) {
  int lo = b.rowptr[row], hi = b.rowptr[row + 1] - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1, c = b.colind[mid];
    if (c == col) return mid;
    if (c < col) lo = mid + 1; else hi = mid - 1;
  }
  return -1;  // sumIntoValues ignores columns the row does not have
}

// rhs[LIDs(e, off(dof))] += sum_pt sum_c data_c(pt) basis(e,dof,pt,c) wts(e,pt)            (getInitial, project)
//                        += sum_pt data(pt) sum_c basis(e,dof,pt,c) n_c(pt) wts(e,pt)       (getDirichletBoundary, HDIV)
template <int DIM, bool EXPR>
__global__ __launch_bounds__(256) void project_rhs_kernel(BlockDev b, ProjectDev p, FuncDesc f0, FuncDesc f1, FuncDesc f2,
                                                          double *rhs) {
  const int total = p.num * p.card;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / p.card, f = idx - k * p.card;
    const int e = p.elem ? p.elem[k] : p.e0 + k, fe = p.elem ? k : e;  // per-point data arrays: [E][np] or [num_sides][np]
    const int row = b.lids[(size_t)e * b.n + b.offsets[p.var_off + f]];
    if (p.fixed_only && !(b.fixed && b.fixed[row])) continue;
    double r = 0.0;
    for (int q = 0; q < p.np; ++q) {
      const size_t pt = (size_t)k * p.np + q;
      double x[3] = {p.xyz[0][pt], p.xyz[1][pt], DIM == 3 ? p.xyz[2][pt] : 0.0};
      double nrm[3] = {0.0, 0.0, 0.0};
      if (p.nrm[0]) { nrm[0] = p.nrm[0][pt]; nrm[1] = p.nrm[1][pt]; if (DIM == 3) nrm[2] = p.nrm[2][pt]; }
      const double w = p.wts[pt];
      const double *bs = p.basis + (((size_t)k * p.card + f) * p.np + q) * p.ncomp;
      if (p.normal_trace) {
        const double d = eval_func<DIM, EXPR>(f0, fe, q, p.np, x, nrm);
        for (int c = 0; c < p.ncomp; ++c) r += d * bs[c] * nrm[c] * w;
      } else {
        r += eval_func<DIM, EXPR>(f0, fe, q, p.np, x, nrm) * bs[0] * w;
        if (p.ncomp > 1) r += eval_func<DIM, EXPR>(f1, fe, q, p.np, x, nrm) * bs[1] * w;
        if (p.ncomp > 2) r += eval_func<DIM, EXPR>(f2, fe, q, p.np, x, nrm) * bs[2] * w;
      }
    }
    unsafeAtomicAdd(rhs + row, r);
  }
}

// one row of the variable's mass block per thread, summed into the CRS row (setInitial :1256-1280, setDirichlet :1883-1912)
__global__ __launch_bounds__(256) void project_mass_kernel(BlockDev b, ProjectDev p, int lump, double *vals) {
  const int total = p.num * p.card;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx / p.card, i = idx - k * p.card;
    const int e = p.elem ? p.elem[k] : p.e0 + k;
    const int32_t *L = b.lids + (size_t)e * b.n;
    const int row = L[b.offsets[p.var_off + i]];
    if (p.fixed_only && !(b.fixed && b.fixed[row])) continue;
    double total_row = 0.0;
    for (int j = 0; j < p.card; ++j) {
      double m = 0.0;
      for (int q = 0; q < p.np; ++q) {
        const size_t pt = (size_t)k * p.np + q;
        const double w = p.wts[pt];
        const double *bi = p.basis + (((size_t)k * p.card + i) * p.np + q) * p.ncomp;
        const double *bj = p.basis + (((size_t)k * p.card + j) * p.np + q) * p.ncomp;
        for (int c = 0; c < p.ncomp; ++c) {
          // getMassBoundary, HDIV (:6400-6408): component by component with the squared normal component
          if (p.normal_trace) m += bi[c] * p.nrm[c][pt] * bj[c] * p.nrm[c][pt] * w;
          else m += bi[c] * bj[c] * w;
        }
      }
      if (lump) { total_row += m; continue; }
      const int pos = crs_find(b, row, L[b.offsets[p.var_off + j]]);
      if (pos >= 0) unsafeAtomicAdd(vals + pos, m);
    }
    if (lump) {
      // setInitial lumps onto the diagonal (:1268-1270); setDirichlet adds the row total to the column of the LAST entry
      // of the element's LID list (the `cols[0] = LIDs(c,col)` left by its summing loop, :1888-1897) -- kept as it is
      const int col = p.fixed_only ? L[b.n - 1] : row;
      const int pos = crs_find(b, row, col);
      if (pos >= 0) unsafeAtomicAdd(vals + pos, total_row);
    }
  }
}

// setInitial, fix_zero_rows (:1284-1302): a row whose absolute sum is below 1e-14 gets a one on its diagonal
__global__ __launch_bounds__(256) void fix_zero_rows_kernel(BlockDev b, double *vals) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < b.nrows; row += gridDim.x * blockDim.x) {
    double abssum = 0.0;
    for (int k = b.rowptr[row]; k < b.rowptr[row + 1]; ++k) abssum += fabs(vals[k]);
    if (abssum < 1.0e-14) {
      const int pos = crs_find(b, row, row);
      if (pos >= 0) vals[pos] = 1.0;
    }
  }
}

// setDirichlet (:1920-1938): ones on the diagonal of the rows that are not fixed (replaceValues, every element that touches them)
__global__ __launch_bounds__(256) void free_row_identity_kernel(BlockDev b, double *vals) {
  const size_t total = (size_t)b.nelem * b.n;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int row = b.lids[idx];
    if (b.fixed && b.fixed[row]) continue;
    const int pos = crs_find(b, row, row);
    if (pos >= 0) vals[pos] = 1.0;
  }
}

// setInitial(set, initial, useadjoint) with getInitial(project = false): the value of "initial <var>" at the element's
// vertices replaces the vector entry of the vertex dof (HGRAD order 1; vert.v[k] = the vertex basis function k sits on --
// k itself in the reference's Intrepid2 ordering, a permutation in the tensor ordering of our tables)
struct VertOfDof { int v[8]; };
template <int DIM, bool EXPR>
__global__ __launch_bounds__(256) void interpolate_nodes_kernel(BlockDev b, FuncDesc f, int var_off, VertOfDof vert,
                                                                double *initial) {
  constexpr int NN = 1 << DIM;
  const size_t total = (size_t)b.nelem * NN;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx / NN), k = (int)(idx - (size_t)e * NN), v = vert.v[k];
    double x[3] = {0.0, 0.0, 0.0};
    for (int d = 0; d < DIM; ++d) x[d] = b.nodes[((size_t)e * NN + v) * DIM + d];
    initial[b.lids[(size_t)e * b.n + b.offsets[var_off + k]]] = eval_func<DIM, EXPR>(f, e, v, NN, x);
  }
}

}  // namespace

void launch_project_rhs(const BlockDev &b, const ProjectDev &p, const FuncDesc f[3], double *rhs, hipStream_t stream) {
  if (p.num <= 0) return;
  const int grid = (p.num * p.card + 255) / 256;
  const bool expr = has_expression(f[0]) || has_expression(f[1]) || has_expression(f[2]);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, b, p, f[0], f[1], f[2], rhs); };
  if (b.dim == 2) { if (expr) go(project_rhs_kernel<2, true>); else go(project_rhs_kernel<2, false>); }
  else { if (expr) go(project_rhs_kernel<3, true>); else go(project_rhs_kernel<3, false>); }
  MHA_HIP(hipGetLastError());
}

void launch_project_mass(const BlockDev &b, const ProjectDev &p, int lump, double *vals, hipStream_t stream) {
  if (p.num <= 0) return;
  const int grid = (p.num * p.card + 255) / 256;
  hipLaunchKernelGGL(project_mass_kernel, dim3(grid), dim3(256), 0, stream, b, p, lump, vals);
  MHA_HIP(hipGetLastError());
}

void launch_fix_zero_rows(const BlockDev &b, double *vals, hipStream_t stream) {
  if (b.nrows <= 0) return;
  hipLaunchKernelGGL(fix_zero_rows_kernel, dim3((b.nrows + 255) / 256), dim3(256), 0, stream, b, vals);
  MHA_HIP(hipGetLastError());
}

void launch_free_row_identity(const BlockDev &b, double *vals, hipStream_t stream) {
  const size_t total = (size_t)b.nelem * b.n;
  if (total == 0) return;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
  hipLaunchKernelGGL(free_row_identity_kernel, dim3(grid), dim3(256), 0, stream, b, vals);
  MHA_HIP(hipGetLastError());
}

void launch_interpolate_nodes(const BlockDev &b, const FuncDesc &f, int var_off, const int *vert_of_dof, double *initial,
                              hipStream_t stream) {
  const size_t total = (size_t)b.nelem * (1 << b.dim);
  if (total == 0) return;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
  const bool expr = has_expression(f);
  VertOfDof vert;
  for (int k = 0; k < 8; ++k) vert.v[k] = k < (1 << b.dim) ? vert_of_dof[k] : 0;
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, b, f, var_off, vert, initial); };
  if (b.dim == 2) { if (expr) go(interpolate_nodes_kernel<2, true>); else go(interpolate_nodes_kernel<2, false>); }
  else { if (expr) go(interpolate_nodes_kernel<3, true>); else go(interpolate_nodes_kernel<3, false>); }
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
