// row_gather.hip -- scatter without global atomics: every CRS row is summed by one wavefront from the dense element
// matrices of the elements incident to it.
//
// Replaces scatterJac / scatterRes (reference: src/managers/assemblyManager.cpp:3882-3935, 3943-3978) -- and, with the
// element kernels writing local_J/local_res first, the fused scatter (:4031-4145) -- for any physics module and any
// LID map.  The reference adds element by element into the CRS with a column search (and atomics on a parallel
// device); here a row's owner walks its (element, LID position) incidences, reads row `position` of each element
// matrix (n contiguous doubles, coalesced), adds the entries into an LDS image of the CRS row through the
// element-major slot map (ds_add_f64: two elements of the row may hit the same column), then stores the row once,
// coalesced.  No global atomics, results independent of scheduling up to the order of the LDS adds within one row.
// HBM traffic: local_J once, the slot map once, the CRS values once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LPR lanes per row: 64 (one row per wavefront), 16 or 8 (four / eight short rows per wavefront: mixed lowest-order elements have
// CRS rows of ~10 entries)
// DOF: the element arrays are in (variable, dof) order (RowGatherDev::inc_dof); a template flag so that the position-order
// instantiations carry none of it (as a run-time branch it cost the 89-dof gather 0.94 -> 1.5 ms at 32^3)
template <typename SlotT, int LPR, bool DOF>
__global__ __launch_bounds__(256) void row_gather_kernel(BlockDev b, RowGatherDev g, const double *__restrict__ local_J,
                                                         const double *__restrict__ local_res, double *res, double *vals,
                                                         int overwrite) {
  extern __shared__ double acc_all[];
  constexpr int RPW = 64 / LPR;  // rows per wavefront
  const int wave = threadIdx.x >> 6, sub = (threadIdx.x & 63) / LPR, lane = threadIdx.x & (LPR - 1), n = b.n;
  double *acc = acc_all + (size_t)(wave * RPW + sub) * g.max_row;
  const SlotT *slot = static_cast<const SlotT *>(g.slot);
  const int nwaves = gridDim.x * 4 * RPW;  // row streams
  for (int k = lane; k < g.max_row; k += LPR) acc[k] = 0.0;
  wave_lds_sync();
  // Two-deep software pipeline over the wave's rows: a row needs rowptr/inc_ptr (level 1), then its incidences
  // (level 2), then the element-matrix rows (level 3) -- three dependent global latencies.  While row r is summed, the
  // incidences of row r+1 and the pointers of row r+2 are already in flight.
  struct Meta { int lo, len, i0, ni, fixed; };
  auto load_meta = [&](int row) {
    Meta m = {0, 0, 0, 0, 1};
    if (row < b.nrows) {
      m.lo = b.rowptr[row];
      m.len = b.rowptr[row + 1] - m.lo;
      m.i0 = g.inc_ptr[row];
      m.ni = g.inc_ptr[row + 1] - m.i0;
      m.fixed = (b.fixed && b.fixed[row]) ? 1 : 0;
    }
    return m;
  };
  const int row0 = (blockIdx.x * 4 + wave) * RPW + sub;
  Meta m_cur = load_meta(row0), m_nxt = load_meta(row0 + nwaves);
  int e_cur = (lane < m_cur.ni) ? g.inc_elem[m_cur.i0 + lane] : 0, p_cur = (lane < m_cur.ni) ? g.inc_pos[m_cur.i0 + lane] : 0;
  int d_cur = (DOF && lane < m_cur.ni) ? g.inc_dof[m_cur.i0 + lane] : 0;  // dof-ordered element arrays only
  for (int row = row0; row < b.nrows; row += nwaves) {
    // prefetch: incidences of the next row (its pointers were requested one iteration ago), pointers of the one after
    const int e_nxt = (lane < m_nxt.ni) ? g.inc_elem[m_nxt.i0 + lane] : 0;
    const int p_nxt = (lane < m_nxt.ni) ? g.inc_pos[m_nxt.i0 + lane] : 0;
    const int d_nxt = (DOF && lane < m_nxt.ni) ? g.inc_dof[m_nxt.i0 + lane] : 0;
    const Meta m_nn = load_meta(row + 2 * nwaves);
    const int lo = m_cur.lo, len = m_cur.len, i0 = m_cur.i0, ni = m_cur.ni;
    if (m_cur.fixed) {  // isFixedDOF rows are skipped by the scatter (assemblyManager.cpp:4075,4120)
      if (overwrite) {
        if (vals) for (int k = lane; k < len; k += LPR) vals[lo + k] = 0.0;
        if (res && lane == 0) res[row] = 0.0;
      }
    } else {
      if (vals) {
        const int total = ni * n;
        for (int t0 = 0; t0 < total; t0 += LPR) {  // trip count uniform in the row group: every lane takes part in the shuffles
          const int t = t0 + lane;
          const bool valid = t < total;
          const int k = valid ? t / n : 0, sj = t - k * n;
          // incidences 0..LPR-1 sit in the row group's registers; longer rows (never for quads / hexes) re-read them
          int e = __shfl(e_cur, k & (LPR - 1), LPR), pos = __shfl(p_cur, k & (LPR - 1), LPR);
          int dof = DOF ? __shfl(d_cur, k & (LPR - 1), LPR) : 0;
          if (valid) {
            if (k >= LPR) { e = g.inc_elem[i0 + k]; pos = g.inc_pos[i0 + k]; if constexpr (DOF) dof = g.inc_dof[i0 + k]; }
            const size_t off = ((size_t)e * n + pos) * n + sj, offd = ((size_t)e * n + pos) * n + pos;
            // isAdjoint_: vals[col] = res(elem,row).fastAccessDx(row) for every col; lump_mass_: cols[col] = rowIndex
            if constexpr (DOF) {  // element arrays in dof order (never with those two options): column sj = dof sj at position offsets[sj]
              const size_t so = ((size_t)e * n + pos) * n + b.offsets[sj];
              unsafeAtomicAdd(acc + slot[so], local_J[((size_t)e * n + dof) * n + sj]);
            } else {
              unsafeAtomicAdd(acc + slot[g.lump_mass ? offd : off], local_J[g.adjoint ? offd : off]);
            }
          }
        }
        wave_lds_sync();
        for (int k = lane; k < len; k += LPR) {
          const double a = acc[k];
          acc[k] = 0.0;
          vals[lo + k] = overwrite ? a : vals[lo + k] + a;
        }
        wave_lds_sync();
      }
      if (res) {
        double r = 0.0;
        for (int k = lane; k < ni; k += LPR) {
          const int e = k < LPR ? e_cur : g.inc_elem[i0 + k], pos = k < LPR ? p_cur : g.inc_pos[i0 + k];
          const int dof = !DOF ? 0 : k < LPR ? d_cur : g.inc_dof[i0 + k];
          r += local_res[(size_t)e * n + (DOF ? dof : pos)];
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) r += __shfl_xor(r, o, LPR);
        if (lane == 0) res[row] = overwrite ? r : res[row] + r;
      }
    }
    m_cur = m_nxt;
    m_nxt = m_nn;
    e_cur = e_nxt;
    p_cur = p_nxt;
    d_cur = d_nxt;
  }
}

}  // namespace

void launch_row_gather(const BlockDev &b, const RowGatherDev &g, const double *local_J, const double *local_res,
                       double *res, double *vals, int overwrite, hipStream_t stream) {
  if (b.nrows <= 0) return;
  const bool small_rows = g.max_row <= 32 && b.n <= 16;
  // eight rows per wavefront for the lowest-order mixed element (CRS rows of <= 13 entries from <= 2 elements): twice the
  // rows in flight per wavefront; MHA_GATHER_LPR=16 keeps four
  static const int lpr_env = [] { const char *m = std::getenv("MHA_GATHER_LPR"); return m ? std::atoi(m) : 0; }();
  const bool tiny_ok = small_rows && g.max_row <= 16 && b.n <= 8 && g.slot_bytes == 1;
  // medium rows (the 42-entry trace rows of the HDG scatter: two incident elements of 24 dofs): four rows per wavefront
  // (config 5's scatter: 64 lanes per row 336 us, 32: 230, 16: 165; MHA_GATHER_LPR=64|32|8 select the others)
  const bool medium = !small_rows && g.max_row <= 64 && b.n <= 32 && g.slot_bytes == 1 && g.inc_dof == nullptr && lpr_env != 64;
  const int lpr = medium ? (lpr_env == 32 ? 32 : lpr_env == 8 ? 8 : 16) : !small_rows ? 64 : (tiny_ok && lpr_env != 16) ? 8 : 16;  // (4 lanes per row: 1.65 against 1.53 ms at config 3)
  const int rpw = 64 / lpr;
  const size_t lds = sizeof(double) * 4 * rpw * (size_t)g.max_row;
  MHA_REQUIRE(lds <= 64 * 1024, MHA_ERR_INVALID, "CRS rows of " << g.max_row << " entries do not fit the row-gather kernel");
  const int grid = std::min((b.nrows + 4 * rpw - 1) / (4 * rpw), 256 * 8);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, b, g, local_J, local_res, res, vals, overwrite);
  };
  const bool dof = g.inc_dof != nullptr;
  MHA_REQUIRE(!dof || (small_rows && g.slot_bytes == 1 && !g.adjoint && !g.lump_mass), MHA_ERR_INVALID,
              "dof-ordered element arrays are gathered by the short-row kernels only");
  if (lpr == 8) { if (dof) go(row_gather_kernel<uint8_t, 8, true>); else go(row_gather_kernel<uint8_t, 8, false>); }
  else if (lpr == 32) go(row_gather_kernel<uint8_t, 32, false>);
  else if (medium) go(row_gather_kernel<uint8_t, 16, false>);
  else if (g.slot_bytes == 1) {
    if (!small_rows) go(row_gather_kernel<uint8_t, 64, false>);
    else if (dof) go(row_gather_kernel<uint8_t, 16, true>);
    else go(row_gather_kernel<uint8_t, 16, false>);
  } else { if (small_rows) go(row_gather_kernel<uint16_t, 16, false>); else go(row_gather_kernel<uint16_t, 64, false>); }
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
