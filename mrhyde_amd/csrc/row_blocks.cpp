#include "row_blocks.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.hpp"
#include "mesh.hpp"

namespace mha {

RowBlockCaps default_caps(int dim, int n) {
  RowBlockCaps c;
  c.chunk_elems = (dim == 3) ? 8 : 16;
  (void)n;
  return c;
}

namespace {

// spread the low 21 bits of v so that two zero bits separate consecutive bits
uint64_t spread3(uint64_t v) {
  v &= 0x1fffff;
  v = (v | v << 32) & 0x1f00000000ffffULL;
  v = (v | v << 16) & 0x1f0000ff0000ffULL;
  v = (v | v << 8) & 0x100f00f00f00f00fULL;
  v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}

uint64_t spread2(uint64_t v) {
  v &= 0xffffffffULL;
  v = (v | v << 16) & 0x0000ffff0000ffffULL;
  v = (v | v << 8) & 0x00ff00ff00ff00ffULL;
  v = (v | v << 4) & 0x0f0f0f0f0f0f0f0fULL;
  v = (v | v << 2) & 0x3333333333333333ULL;
  v = (v | v << 1) & 0x5555555555555555ULL;
  return v;
}

}  // namespace

RowBlocks build_row_blocks(int dim, int nnodes, int nelem, int n, int nrows, const double *nodes,
                           const int32_t *lids, const int32_t *rowptr, const RowBlockCaps &caps,
                           const uint8_t *fixed, int slot_bytes) {
  MHA_REQUIRE(caps.max_rows <= 65535 && caps.max_elems <= 255 && n <= 255, MHA_ERR_INVALID,
              "row-block caps exceed the pair encoding (rows <= 65535, elements <= 255, dofs <= 255)");
  MHA_REQUIRE(caps.chunk_elems >= 1 && caps.max_acc >= 1 && caps.max_rows >= 1 && caps.max_elems >= 1,
              MHA_ERR_INVALID, "bad row-block caps");
  // --- Morton rank of every element (centroid of its vertices) ---
  std::vector<double> cen(static_cast<size_t>(nelem) * dim);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int e = 0; e < nelem; ++e)
    for (int d = 0; d < dim; ++d) {
      double s = 0.0;
      for (int v = 0; v < nnodes; ++v) s += nodes[(static_cast<size_t>(e) * nnodes + v) * dim + d];
      s /= nnodes;
      cen[static_cast<size_t>(e) * dim + d] = s;
      lo[d] = std::min(lo[d], s);
      hi[d] = std::max(hi[d], s);
    }
  // Integer cell coordinates of the centroids.  Tensor-product meshes: exact index along each axis
  // (coordinate compression), so chunks are aligned 2x2x2 (4x4) element cubes.  Otherwise: a uniform
  // grid with one cell per mean element spacing, same scale in all directions.
  std::vector<std::vector<uint32_t>> q(dim, std::vector<uint32_t>(nelem, 0));
  const double per_axis = std::pow(static_cast<double>(nelem), 1.0 / dim);
  bool tensor_like = true;
  {
    std::vector<std::pair<double, int32_t>> srt(nelem);
    for (int d = 0; d < dim && tensor_like; ++d) {
      for (int e = 0; e < nelem; ++e) srt[e] = {cen[static_cast<size_t>(e) * dim + d], e};
      std::sort(srt.begin(), srt.end());
      const double tol = 1e-9 * std::max(hi[d] - lo[d], 1e-300);
      uint32_t r = 0;
      for (int k = 0; k < nelem; ++k) {
        if (k > 0 && srt[k].first - srt[k - 1].first > tol) ++r;
        q[d][srt[k].second] = r;
      }
      if (r + 1 > 4.0 * per_axis + 16.0) tensor_like = false;
    }
  }
  if (!tensor_like) {
    double span = 0.0;
    for (int d = 0; d < dim; ++d) span = std::max(span, hi[d] - lo[d]);
    if (!(span > 0.0)) span = 1.0;
    const double qmax = (dim == 3) ? 2097151.0 : 4294967295.0;
    // one cell per mean element spacing: a perturbed structured mesh still lands on its index grid
    const double scale = std::min(qmax, std::max(1.0, std::round(per_axis) - 1.0)) / span;
    for (int e = 0; e < nelem; ++e)
      for (int d = 0; d < dim; ++d)
        q[d][e] = static_cast<uint32_t>(std::floor((cen[static_cast<size_t>(e) * dim + d] - lo[d]) * scale + 0.5));
  }
  std::vector<std::pair<uint64_t, int32_t>> key(nelem);
  for (int e = 0; e < nelem; ++e) {
    const uint64_t code = (dim == 3) ? (spread3(q[0][e]) | spread3(q[1][e]) << 1 | spread3(q[2][e]) << 2)
                                     : (spread2(q[0][e]) | spread2(q[1][e]) << 1);
    key[e] = {code, e};
  }
  std::sort(key.begin(), key.end());
  std::vector<int32_t> rank(nelem);
  for (int k = 0; k < nelem; ++k) rank[key[k].second] = k;

  // --- row incidence and owner chunk ---
  std::vector<int32_t> inc_ptr, inc_elem, inc_lpos;
  build_row_incidence(nrows, nelem, n, lids, inc_ptr, inc_elem, inc_lpos);
  const int nchunks = (nelem + caps.chunk_elems - 1) / caps.chunk_elems;
  std::vector<int32_t> row_chunk(nrows, -1), chunk_cnt(static_cast<size_t>(nchunks) + 1, 0);
  for (int r = 0; r < nrows; ++r) {
    int best = -1;
    for (int k = inc_ptr[r]; k < inc_ptr[r + 1]; ++k)
      if (best < 0 || rank[inc_elem[k]] < best) best = rank[inc_elem[k]];
    if (best < 0) continue;  // row without elements: nothing to assemble, stays untouched
    row_chunk[r] = best / caps.chunk_elems;
    chunk_cnt[row_chunk[r] + 1]++;
  }
  for (int c = 0; c < nchunks; ++c) chunk_cnt[c + 1] += chunk_cnt[c];
  std::vector<int32_t> chunk_rows(chunk_cnt[nchunks]);
  {
    std::vector<int32_t> fill(chunk_cnt.begin(), chunk_cnt.end() - 1);
    for (int r = 0; r < nrows; ++r)
      if (row_chunk[r] >= 0) chunk_rows[fill[row_chunk[r]]++] = r;  // ascending inside a chunk
  }

  // --- greedy split of every chunk's rows into blocks that fit the caps ---
  RowBlocks rb;
  rb.row_ptr.push_back(0);
  rb.elem_ptr.push_back(0);
  std::vector<int32_t> mark(nelem, -1);
  std::vector<int32_t> cur_elems;
  int cur_rows = 0, cur_acc = 0, cur_pairs = 0;
  auto close_block = [&]() {
    if (cur_rows == 0) return;
    std::sort(cur_elems.begin(), cur_elems.end());
    rb.elems.insert(rb.elems.end(), cur_elems.begin(), cur_elems.end());
    rb.elem_ptr.push_back(static_cast<int32_t>(rb.elems.size()));
    rb.row_ptr.push_back(static_cast<int32_t>(rb.rows.size()));
    rb.acc_size.push_back(cur_acc);
    rb.max_rows = std::max(rb.max_rows, cur_rows);
    rb.max_elems = std::max(rb.max_elems, static_cast<int>(cur_elems.size()));
    rb.max_acc = std::max(rb.max_acc, cur_acc);
    rb.num_blocks++;
    cur_elems.clear();
    cur_rows = 0;
    cur_acc = 0;
    cur_pairs = 0;
  };
  for (int c = 0; c < nchunks; ++c) {
    for (int k = chunk_cnt[c]; k < chunk_cnt[c + 1]; ++k) {
      const int r = chunk_rows[k];
      const int nnz = rowptr[r + 1] - rowptr[r];
      int fresh = 0;
      for (int p = inc_ptr[r]; p < inc_ptr[r + 1]; ++p)
        if (mark[inc_elem[p]] != rb.num_blocks) ++fresh;  // may double count a repeated element: conservative
      MHA_REQUIRE(nnz <= caps.max_acc && inc_ptr[r + 1] - inc_ptr[r] <= caps.max_elems, MHA_ERR_INVALID,
                  "row " << r << " alone exceeds the row-block caps (nnz " << nnz << ", incident elements "
                         << inc_ptr[r + 1] - inc_ptr[r] << ")");
      const int npair = inc_ptr[r + 1] - inc_ptr[r];
      MHA_REQUIRE(npair <= caps.max_pairs, MHA_ERR_INVALID, "row " << r << " alone exceeds the pair cap");
      if (cur_rows > 0 && (cur_rows + 1 > caps.max_rows || cur_acc + nnz > caps.max_acc ||
                           cur_pairs + npair > caps.max_pairs ||
                           static_cast<int>(cur_elems.size()) + fresh > caps.max_elems))
        close_block();
      for (int p = inc_ptr[r]; p < inc_ptr[r + 1]; ++p)
        if (mark[inc_elem[p]] != rb.num_blocks) {
          mark[inc_elem[p]] = rb.num_blocks;
          cur_elems.push_back(inc_elem[p]);
        }
      rb.rows.push_back(r);
      rb.row_off.push_back(cur_acc);
      cur_acc += nnz;
      cur_pairs += npair;
      cur_rows++;
    }
    close_block();
  }

  // --- CRS placement of the owned rows ---
  rb.row_base.resize(rb.rows.size());
  rb.row_len.resize(rb.rows.size());
  for (size_t k = 0; k < rb.rows.size(); ++k) {
    const int r = rb.rows[k];
    const int len = rowptr[r + 1] - rowptr[r];
    rb.row_base[k] = rowptr[r];
    rb.row_len[k] = (fixed && fixed[r]) ? -len - 1 : len;
  }
  // --- contribution pairs of every block ---
  rb.emask.assign(rb.elems.size(), 0);
  rb.epbase.assign(rb.elems.size(), 0);
  rb.pair_ptr.assign(1, 0);
  std::vector<uint32_t> tmp;
  for (int k = 0; k < rb.num_blocks; ++k) {
    tmp.clear();
    const int32_t *be = rb.elems.data() + rb.elem_ptr[k];
    const int ne = rb.elem_ptr[k + 1] - rb.elem_ptr[k];
    for (int o = 0; o < rb.row_ptr[k + 1] - rb.row_ptr[k]; ++o) {
      const int r = rb.rows[rb.row_ptr[k] + o];
      if (fixed && fixed[r]) continue;
      for (int p = inc_ptr[r]; p < inc_ptr[r + 1]; ++p) {
        const int t = static_cast<int>(std::lower_bound(be, be + ne, inc_elem[p]) - be);
        tmp.push_back(static_cast<uint32_t>(o) << 16 | static_cast<uint32_t>(t) << 8 | static_cast<uint32_t>(inc_lpos[p]));
      }
    }
    std::sort(tmp.begin(), tmp.end(), [](uint32_t a, uint32_t b) { return (a & 0xffffu) < (b & 0xffffu); });
    for (size_t i = 0; i < tmp.size(); ++i) {
      const int o = tmp[i] >> 16, t = (tmp[i] >> 8) & 0xff, si = tmp[i] & 0xff;
      rb.pair_off.push_back(rb.row_off[rb.row_ptr[k] + o]);
      int32_t &m = rb.emask[rb.elem_ptr[k] + t];
      if (m == 0) rb.epbase[rb.elem_ptr[k] + t] = static_cast<int32_t>(i);
      if (si < 32) m |= (1 << si);
    }
    rb.pairs.insert(rb.pairs.end(), tmp.begin(), tmp.end());
    rb.pair_ptr.push_back(static_cast<int32_t>(rb.pairs.size()));
    rb.max_pairs = std::max(rb.max_pairs, static_cast<int>(tmp.size()));
  }
  // --- aligned slot-table offsets ---
  rb.slot_ptr.assign(1, 0);
  for (int k = 0; k < rb.num_blocks; ++k) {
    const int64_t bytes = static_cast<int64_t>(rb.pair_ptr[k + 1] - rb.pair_ptr[k]) * n * slot_bytes;
    rb.slot_ptr.push_back(rb.slot_ptr.back() + (bytes + 15) / 16 * 16);
  }
  // --- store segments ---
  rb.seg_ptr.assign(1, 0);
  for (int k = 0; k < rb.num_blocks; ++k) {
    int nseg = 0;
    for (int i = rb.row_ptr[k]; i < rb.row_ptr[k + 1]; ++i) {
      const bool fx = rb.row_len[i] < 0;
      const int len = fx ? -rb.row_len[i] - 1 : rb.row_len[i];
      bool extend = false;
      if (nseg > 0) {
        const size_t s = rb.seg_acc.size() - 1;
        const bool sfx = rb.seg_len[s] < 0;
        const int slen = sfx ? -rb.seg_len[s] : rb.seg_len[s];
        extend = sfx == fx && rb.seg_acc[s] + slen == rb.row_off[i] && rb.seg_base[s] + slen == rb.row_base[i];
        if (extend) rb.seg_len[s] = fx ? -(slen + len) : slen + len;
      }
      if (!extend) {
        rb.seg_acc.push_back(rb.row_off[i]);
        rb.seg_base.push_back(rb.row_base[i]);
        rb.seg_len.push_back(fx ? -len : len);
        ++nseg;
      }
    }
    rb.seg_ptr.push_back(static_cast<int32_t>(rb.seg_acc.size()));
    rb.max_segs = std::max(rb.max_segs, nseg);
  }
  return rb;
}

}  // namespace mha
