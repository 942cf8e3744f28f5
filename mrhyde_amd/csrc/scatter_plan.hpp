// scatter_plan.hpp -- scatter of dense element blocks into a CRS matrix / global vector for an arbitrary LID map.
//
// The hot-path blocks scatter through AssemblyManager (their own LIDs).  The HDG caller needs one more scatter with a
// DIFFERENT map: the condensed trace blocks [E][24][24] and flux vectors [E][24] of the subgrid elements go into the
// macro trace system (reference: SubGridDtN_Solver::updateFlux + the macro assembly's sumIntoValues,
// src/subgrid/subgridDtN_solver.cpp:1542-1616, src/managers/assemblyManager.cpp:4031-4145).  A plan holds what that
// needs on the device -- row incidences and the element-major slot map -- and applies kernels/row_gather.hip: one
// wavefront per CRS row, no global atomics.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"
#include "kernels/device_types.hpp"

namespace mha {

class ScatterPlan {
 public:
  // lids [num_elems][n] (row of unknown t of element e), CRS graph of the target; colind ascending inside a row;
  // rowptr / colind null: the graph every-dof-of-an-element-couples (linearAlgebraInterface.cpp:218-229) is built
  ScatterPlan(int n, int num_elems, int num_rows, const int32_t *lids, const int32_t *rowptr, const int32_t *colind,
              const uint8_t *fixed);
  // vals[rowptr[r] + slot] (+)= sum of blocks[e][t][s], res[r] (+)= sum of vec[e][t]; fixed rows: zeros when storing,
  // untouched when accumulating.  blocks / vec / res / vals on the device; blocks+vals or vec+res may be null.
  void apply(const double *blocks, const double *vec, double *res, double *vals, bool overwrite, hipStream_t stream) const;
  int64_t nnz() const { return nnz_; }
  void graph(int32_t *rowptr, int32_t *colind) const;

 private:
  int n_ = 0, nelem_ = 0, nrows_ = 0, max_row_ = 0, slot_bytes_ = 1;
  int64_t nnz_ = 0;
  std::vector<int32_t> h_rowptr_, h_colind_;
  DeviceBuffer<int32_t> lids_, rowptr_, colind_, inc_ptr_, inc_elem_, inc_pos_;
  DeviceBuffer<uint8_t> fixed_, slot_;
};

}  // namespace mha
