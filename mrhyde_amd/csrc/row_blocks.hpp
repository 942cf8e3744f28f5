// row_blocks.hpp -- host-side partition of the CRS rows into workgroup-sized "row blocks".
//
// The fused assembly kernel is row-owner: every CRS row (and residual entry) is produced by
// exactly one workgroup, which visits all elements incident to its rows, so the global scatter
// needs no atomics and no second pass.  This replaces the per-element sumIntoValues scatter of the
// reference (src/managers/assemblyManager.cpp:4063-4144) and its column search; the result is the
// same sum of element contributions per (row, col).
//
// Blocks are spatially compact irrespective of the caller's element numbering: elements are
// ordered along a Morton curve of their centroids and cut into chunks; a row belongs to the chunk
// of its first (lowest Morton rank) incident element; a chunk's rows are split further until the
// block fits the LDS budget (accumulator entries, rows, touched elements).
#pragma once
#include <cstdint>
#include <vector>

namespace mha {

struct RowBlockCaps {
  int chunk_elems = 8;   // elements per Morton chunk
  int max_acc = 4608;    // CRS entries accumulated in LDS per block
  int max_rows = 128;    // owned rows per block
  int max_elems = 40;    // touched elements per block
  int max_pairs = 1024;  // (element, slot) contribution pairs per block
};

struct RowBlocks {
  int num_blocks = 0;
  std::vector<int32_t> row_ptr;   // [nb+1] -> rows / row_off
  std::vector<int32_t> rows;      // owned rows of each block, ascending
  std::vector<int32_t> row_off;   // offset (in entries) of the row's accumulator inside its block
  std::vector<int32_t> acc_size;  // [nb] accumulator entries of the block
  std::vector<int32_t> elem_ptr;  // [nb+1] -> elems
  std::vector<int32_t> elems;     // touched elements of each block, ascending
  // (element, LID slot) pairs whose row the block owns and that receive contributions (fixed rows
  // excluded), sorted by (element, slot): pair = local_row << 16 | local_elem << 8 | slot
  std::vector<int32_t> pair_ptr;  // [nb+1] -> pairs / pair_off
  std::vector<uint32_t> pairs;
  std::vector<int32_t> pair_off;  // accumulator offset of the pair's row (= row_off of its row)
  // per owned row (parallel to rows): CRS offset and length; length stored as -(len)-1 for fixed rows
  std::vector<int32_t> row_base, row_len;
  // per touched element (parallel to elems): bit si set when (elem, si) is a pair; index of its first pair
  std::vector<int32_t> emask, epbase;
  // byte offset of the block's slot table inside the block-major slot array (16-byte aligned), [nb+1]
  std::vector<int64_t> slot_ptr;
  // store segments: maximal runs of owned rows that are contiguous both in the accumulator and in the
  // CRS value array.  seg_len < 0 marks a run of fixed rows (-len entries; written as zeros in
  // overwrite mode, untouched otherwise).
  std::vector<int32_t> seg_ptr;   // [nb+1]
  std::vector<int32_t> seg_acc, seg_base, seg_len;
  int max_segs = 0;
  int max_rows = 0, max_elems = 0, max_acc = 0, max_pairs = 0;
};

RowBlockCaps default_caps(int dim, int n);

RowBlocks build_row_blocks(int dim, int nnodes, int nelem, int n, int nrows, const double *nodes,
                           const int32_t *lids, const int32_t *rowptr, const RowBlockCaps &caps,
                           const uint8_t *fixed = nullptr, int slot_bytes = 1);

}  // namespace mha
