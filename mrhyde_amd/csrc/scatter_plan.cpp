// scatter_plan.cpp -- see scatter_plan.hpp.
#include "scatter_plan.hpp"

#include <algorithm>
#include <vector>

#include "kernels/launch.hpp"
#include "mesh.hpp"

namespace mha {

ScatterPlan::ScatterPlan(int n, int num_elems, int num_rows, const int32_t *lids, const int32_t *rowptr,
                         const int32_t *colind, const uint8_t *fixed)
    : n_(n), nelem_(num_elems), nrows_(num_rows) {
  MHA_REQUIRE(n > 0 && num_elems > 0 && num_rows > 0 && lids, MHA_ERR_INVALID, "scatter plan: bad sizes");
  const size_t nl = static_cast<size_t>(num_elems) * n;
  for (size_t k = 0; k < nl; ++k)
    MHA_REQUIRE(lids[k] >= 0 && lids[k] < num_rows, MHA_ERR_INVALID, "scatter plan: LID " << lids[k] << " out of range");
  if (rowptr && colind) {
    MHA_REQUIRE(rowptr[0] == 0, MHA_ERR_INVALID, "rowptr[0] must be 0");
    for (int r = 0; r < num_rows; ++r)
      MHA_REQUIRE(rowptr[r + 1] >= rowptr[r], MHA_ERR_INVALID, "rowptr must be non-decreasing");
    validate_crs_graph(num_rows, num_elems, n, lids, rowptr, colind);  // sorted rows, every element coupling present
    h_rowptr_.assign(rowptr, rowptr + num_rows + 1);
    h_colind_.assign(colind, colind + h_rowptr_[num_rows]);
  } else {
    build_crs_graph(num_rows, num_elems, n, lids, h_rowptr_, h_colind_);
  }
  nnz_ = h_rowptr_[num_rows];
  for (int r = 0; r < num_rows; ++r) max_row_ = std::max(max_row_, h_rowptr_[r + 1] - h_rowptr_[r]);
  MHA_REQUIRE(max_row_ <= 65536, MHA_ERR_INVALID, "CRS rows longer than 65536 entries are not supported");
  slot_bytes_ = max_row_ <= 256 ? 1 : 2;
  std::vector<int32_t> ptr, elem, lpos;
  build_row_incidence(num_rows, num_elems, n, lids, ptr, elem, lpos);
  lids_.upload(lids, nl);
  rowptr_.upload(h_rowptr_);
  colind_.upload(h_colind_);
  inc_ptr_.upload(ptr);
  inc_elem_.upload(elem);
  inc_pos_.upload(lpos);
  if (fixed) fixed_.upload(fixed, num_rows);
  slot_.resize(nl * n * slot_bytes_);
  BlockDev b;
  b.nelem = num_elems;
  b.nrows = num_rows;
  b.n = n;
  b.lids = lids_.data();
  b.rowptr = rowptr_.data();
  b.colind = colind_.data();
  launch_build_elem_slot_map(b, slot_.data(), slot_bytes_, nullptr);
  MHA_HIP(hipStreamSynchronize(nullptr));
}

void ScatterPlan::graph(int32_t *rowptr, int32_t *colind) const {
  std::copy(h_rowptr_.begin(), h_rowptr_.end(), rowptr);
  std::copy(h_colind_.begin(), h_colind_.end(), colind);
}

void ScatterPlan::apply(const double *blocks, const double *vec, double *res, double *vals, bool overwrite,
                        hipStream_t stream) const {
  MHA_REQUIRE((blocks != nullptr) == (vals != nullptr) && (vec != nullptr) == (res != nullptr), MHA_ERR_INVALID,
              "scatter plan: blocks go with vals, vec with res");
  BlockDev b;
  b.nelem = nelem_;
  b.nrows = nrows_;
  b.n = n_;
  b.rowptr = rowptr_.data();
  b.fixed = fixed_.size() ? fixed_.data() : nullptr;
  RowGatherDev g;
  g.inc_ptr = inc_ptr_.data();
  g.inc_elem = inc_elem_.data();
  g.inc_pos = inc_pos_.data();
  g.slot = slot_.data();
  g.slot_bytes = slot_bytes_;
  g.max_row = max_row_;
  launch_row_gather(b, g, blocks, vec, res, vals, overwrite ? 1 : 0, stream);
}

}  // namespace mha
