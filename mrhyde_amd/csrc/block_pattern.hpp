// block_pattern.hpp -- row blocks keyed by their assembly pattern, for the matrix-core row-owner Jacobian
// (kernels/block_pattern.hip).
//
// On an affine element with element-wise constant coefficients the thermal element matrix is a combination of a few
// constant reference matrices,  K_e = sum_m g_m(e) Khat_m  (m: the symmetric components of detJ J^-1 J^-T and detJ for
// the mass term), so a CRS row is   vals[slot] = sum_{(k,m)} g_m(e_k) * W[(k,m)][slot]   where e_k are the elements
// around the row's dof and W depends only on HOW the row is assembled (which local dof of each incident element it is
// and where that element's columns sit in the row).  This replaces the per-entry sumIntoValues of the reference's
// scatter (src/managers/assemblyManager.cpp:4031-4145, res(e,i).dx(j) from src/physics/thermal.cpp:125-163).
//
// The unit of work is a ROW BLOCK (row_blocks.hpp: the rows owned by a Morton chunk of elements).  Two blocks have the
// same PATTERN when their owned rows, touched elements and every column slot agree position by position -- on a
// structured mesh all interior chunks share one pattern whatever the element sizes are (the geometric factors g_m(e)
// stay per element and are read every assembly).  Per pattern the host derives, once per mesh:
//   * row CLASSES: rows of the block with the same W (vertex / edge / face / cell dofs of a Q2 hex block), their W
//     images in the layout the kernel keeps in LDS;
//   * TILES of 16 rows of one class: one v_mfma_f64_16x16x4_f64 row panel; per lane the offsets of its A operands
//     (geometric factors) inside the block's element records and the block-local index of its result rows;
//   * PARTS: a tile restricted to every nphase-th block, dealt to the 16 wavefronts of a persistent workgroup so that
//     every wavefront carries the same matrix-core load;
//   * ROLES: the classes a workgroup keeps resident in LDS (one role per pattern unless the W images exceed the LDS).
// All per-block data the kernel reads are block-major with a stride fixed per role (element records, CRS offsets of
// the rows): every address is computed from the block's ordinal, there are no dependent loads.
//
// IMAGE ROLES (round 3).  The CRS rows of a block are odd-length neighbours in memory, so row pieces stored straight
// from the MFMA registers share their first and last cache line with another wavefront's piece: measured at 3.4 TB/s
// against 5.3-6.2 TB/s for whole aligned lines (profiles/micro/write_shape.hip).  Where the LDS has room, a role
// therefore assembles every block as an IMAGE -- the block's runs of consecutive rows, each at its global offset mod
// 16 entries inside a slot of its own -- and streams it out in aligned 1 KB chunks (one 16-byte store instruction each);
// only the first and last line of a run are partial.  The LDS for the image comes from keeping the largest W (the
// rows around a vertex dof) in REGISTERS: its class is cut into units of two column tiles whose 4 x 16 blocks of W sit in
// at most 56 registers for the life of the segment (REGW units).  Wave 0 of such a role only loads element records.
//
// Nothing here assumes a structured mesh: patterns are found by hashing.  A mesh whose blocks share too few patterns
// reports !usable and the caller keeps the LDS-accumulator row-block kernel.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "row_blocks.hpp"

namespace mha {

#ifndef MHA_BP_WAVES
#define MHA_BP_WAVES 12
#endif
constexpr int kBpWaves = MHA_BP_WAVES;  // wavefronts of a persistent workgroup (768 threads, 168 registers each: one workgroup per CU)
constexpr int kBpMaxKSteps = 16;   // GEMM depth held in registers: 64 = 9 hexes x 7 or 16 quads x 4
constexpr int kBpLaneRows = 24;    // per part and lane: [0..15] A offsets (doubles), [16..19] result rows of the lane's registers, [20] tile row lane & 15: run << 20 | CRS offset inside the run, [21] the same row's entry offset inside the block's LDS image (image roles)
constexpr int kBpStreamWaves = kBpWaves - 1;  // image roles: wave 0 loads the element records, the others stream the image out
constexpr int kBpChunkInts = 4;    // per 128-entry chunk of an image: run, LDS entry of lane 0, entry of lane 0 relative to the run's aligned start, entries of the run
constexpr int kBpHdrInts = 12;     // per unit: LDS offset of W, k-steps, column tiles, row length, first tile, tiles, flags, class, 3 words: bit s * 5 + q set = W block (k-step s, tile q) is not zero
constexpr int kBpRoleInts = 24;    // see block_pattern.cpp
constexpr int kBpSegInts = 2048;    // (blocks of one segment) x (runs of consecutive rows per block): their CRS offsets sit in LDS
constexpr int kBpRecDoubles = 8;   // element record: g_0..g_{nsym-1}, detJ, zero padding

// Columns of a UNIT (up to four 16-column tiles of a 16-row class tile + an optional tail tile) as the kernel's
// accumulation chains see them: the tiles come in PAIRS whose columns interleave -- lane c of chain 2p holds column
// 32p + 2c, of chain 2p + 1 column 32p + 2c + 1 -- so that a lane owns two neighbouring entries of a row and stores
// them with one 16-byte instruction (the store path of a CU takes one instruction per ~40 cycles whatever its width);
// an odd last tile and the tail tile keep the plain order.  Offsets are relative to the row, in entries.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int bp_unit_col(int ct0, int ntile, int q, int c) {
  return q < 2 * (ntile / 2) ? 16 * ct0 + 32 * (q >> 1) + 2 * c + (q & 1) : 16 * (ct0 + q) + c;
}

struct BlockPatternPlan {
  bool usable = false;
  std::string why;
  int ke = 0, nsym = 0;
  int num_patterns = 0, num_roles = 0, num_wgs = 0, num_parts = 0, num_classes = 0;
  int max_w_doubles = 0;               // LDS doubles of the largest role
  int max_rec_doubles = 0;             // doubles of the largest block's element records
  std::vector<int32_t> role;           // [num_roles][kBpRoleInts]
  std::vector<int32_t> seg;            // [num_segs][4]: role, first block (inside the role), blocks, 0
  std::vector<int32_t> wg_seg_ptr;     // [num_wgs + 1] -> segments of a workgroup (roles that store from the registers)
  std::vector<int32_t> wg_seg_ptr_img; // the same for the image roles (a kernel of their own)
  std::vector<int32_t> part_ptr;       // [num_roles][kBpWaves + 1] -> parts of (role, wave)
  std::vector<int32_t> part_hdr;       // [num_parts][kBpHdrInts]
  std::vector<int32_t> part_lane;      // [num_parts][kBpLaneRows][64]
  std::vector<double> w;               // LDS images of the roles: per role [stiffness rows | mass rows], each w_doubles long
  std::vector<int32_t> erec_elem;      // role-major, block-major [T + 1] element ids; -1 = the block's zero record
  std::vector<int32_t> rowbase;        // role-major, block-major [runs]: CRS offset of the first row of every run of consecutive owned rows
  std::vector<int32_t> runlen;         // role-major [runs]: CRS entries of each run (the same for every block of a role)
  std::vector<int64_t> role_runlen_off;// [num_roles] -> first entry of the role in runlen
  std::vector<int32_t> chunk_tab;      // image roles: [chunks][kBpChunkInts], role-major
  int num_image_roles = 0;
  int64_t mfma_per_assembly = 0;       // diagnostics
};

// seg_blocks > 0: a workgroup's wavefronts meet at a barrier every seg_blocks blocks (their stores stay close in time).
// rb: row-block partition (any caps); elem_slot: element-major map [e][si][sj] -> position of column lids[e][sj] inside
// CRS row lids[e][si] (slot_bytes 1 or 2); khat: [nsym + 1][n*n] reference matrices in LID-slot space (last = mass).
BlockPatternPlan build_block_patterns(const RowBlocks &rb, int n, int nsym, const int32_t *rowptr, const uint8_t *fixed,
                                      const void *elem_slot, int slot_bytes, const double *khat, int num_cus,
                                      size_t lds_budget_bytes, int max_patterns, int seg_blocks = 0);

// Walks the plan exactly as the kernel does (workgroup -> wavefront -> part -> block -> MFMA panel) on the host:
// vals[...] = the CRS values; every entry of an owned row is written exactly once.  factors: [E][ke].
void block_patterns_host_apply(const BlockPatternPlan &plan, const double *factors, double scale_u, double scale_t,
                               bool overwrite, double *vals);

}  // namespace mha
