// row_pattern.hpp -- CRS rows grouped by assembly pattern, for the matrix-core form of the row-owner Jacobian.
//
// On an affine element the thermal element matrix is a combination of a few constant reference matrices,
//   K_e = sum_c g_c(e) Khat_c     (c: the symmetric components of detJ J^-1 J^-T, and detJ for the mass term),
// so a CRS row is   vals[slot] = sum_{(e,c)} g_c(e) * W[(e,c)][slot]   with a matrix W that depends only on HOW the row
// is assembled: which local dof of each incident element it is and where that element's columns sit in the row.  Rows
// with the same pattern (all interior vertex rows of a structured mesh, all x-edge rows, ...) share W, and 16 of them
// at a time are one small GEMM  [16 rows x K] * [K x row length]  on v_mfma_f64_16x16x4_f64
// (kernels/row_pattern.hip).  This replaces the reference's per-entry sumIntoValues
// (src/managers/assemblyManager.cpp:4031-4145) for affine elements.
//
// Nothing here assumes a structured mesh: patterns are found by hashing the slot lists of every row.  A mesh whose rows
// share few patterns (too many distinct W) reports !usable and the caller keeps the row-block kernel.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mha {

struct RowPatterns {
  bool usable = false;
  std::string why;               // reason when !usable
  int ke = 0;                    // GEMM depth per incident element: components padded to a multiple of 4
  int num_patterns = 0;
  int max_w_doubles = 0;         // LDS doubles of the largest W
  std::vector<int32_t> pat_ni;       // incident elements per row
  std::vector<int32_t> pat_len;      // row length
  std::vector<int32_t> pat_cols;     // row length padded to a multiple of 16
  std::vector<int32_t> pat_stride;   // doubles between consecutive k rows of W (pat_cols, + 16 when that avoids bank aliasing)
  std::vector<int64_t> pat_woff;     // offset of the pattern's W inside `w`
  std::vector<double> w;             // all W, [pattern][ni*ke][stride]
  // super tiles: kRowsPerSuperTile rows of one pattern (one workgroup pass: 16 rows per wavefront)
  std::vector<int32_t> st_pat;
  std::vector<int64_t> st_off;       // offset of the super tile's record inside st_rec
  // record of a super tile, per wavefront tile of 16 rows: [2 + ni][16] ints = CRS offset of the row, its length
  // (bit 30: fixed row; 0: no row), the incident elements in pattern order -- everything the kernel needs about a row
  // in one coalesced read, so that the gather of the geometry factors is only one dependent load away
  std::vector<int32_t> st_rec;
  // descriptor of a super tile, 8 ints: pattern * 16 + shift, ni | shift << 8 | row length << 16, columns of the shifted
  // row padded to whole lines, LDS stride | memory stride << 16, record offset (lo, hi), W offset (lo, hi)
  // -- one 32-byte read tells the kernel everything about the tile (no dependent lookups in the pattern tables)
  std::vector<int32_t> st_desc;
  std::vector<int32_t> wg_ptr;       // [num_wgs + 1] super-tile ranges of the persistent workgroups
};

// slot: element-major map [e][si][sj] -> position of column lids[e][sj] inside CRS row lids[e][si] (slot_bytes 1 or 2);
// khat: [nsym + 1][n*n] reference matrices in LID-slot space (last = mass).
constexpr int kRowsPerSuperTile = 128;  // 8 wavefronts x 16 rows per workgroup pass

RowPatterns build_row_patterns(int nrows, int n, int nsym, const int32_t *rowptr, const uint8_t *fixed,
                               const std::vector<int32_t> &inc_ptr, const std::vector<int32_t> &inc_elem,
                               const std::vector<int32_t> &inc_pos, const void *slot, int slot_bytes,
                               const double *khat, int num_wgs, int chunk, int max_patterns, size_t max_w_bytes,
                               int max_lds_bytes);

}  // namespace mha
