// row_pattern.cpp -- see row_pattern.hpp.
#include "row_pattern.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <unordered_map>

namespace mha {
namespace {

constexpr int kMaxDepth = 64;  // GEMM depth the kernel keeps in registers (16 k-steps of 4)

// The assembly pattern of a row as bytes: length, incident elements, and per incidence the local dof and the slot of
// every column of that element.
void pattern_bytes(int r, int n, const int32_t *rowptr, const std::vector<int32_t> &inc_ptr,
                   const std::vector<int32_t> &inc_elem, const std::vector<int32_t> &inc_pos, const uint8_t *slot,
                   int slot_bytes, std::vector<uint8_t> &out) {
  out.clear();
  const int len = rowptr[r + 1] - rowptr[r], ni = inc_ptr[r + 1] - inc_ptr[r];
  out.push_back(static_cast<uint8_t>(len & 0xff));
  out.push_back(static_cast<uint8_t>((len >> 8) & 0xff));
  out.push_back(static_cast<uint8_t>(len >> 16));
  out.push_back(static_cast<uint8_t>(ni & 0xff));
  out.push_back(static_cast<uint8_t>(ni >> 8));
  for (int k = inc_ptr[r]; k < inc_ptr[r + 1]; ++k) {
    const int si = inc_pos[k];
    out.push_back(static_cast<uint8_t>(si));
    const uint8_t *s = slot + (static_cast<size_t>(inc_elem[k]) * n + si) * n * slot_bytes;
    out.insert(out.end(), s, s + static_cast<size_t>(n) * slot_bytes);
  }
}

uint64_t fnv1a(const std::vector<uint8_t> &b) {
  uint64_t h = 1469598103934665603ull;
  for (uint8_t c : b) { h ^= c; h *= 1099511628211ull; }
  return h;
}

}  // namespace

RowPatterns build_row_patterns(int nrows, int n, int nsym, const int32_t *rowptr, const uint8_t *fixed,
                               const std::vector<int32_t> &inc_ptr, const std::vector<int32_t> &inc_elem,
                               const std::vector<int32_t> &inc_pos, const void *slot_v, int slot_bytes,
                               const double *khat, int num_wgs, int chunk, int max_patterns, size_t max_w_bytes,
                               int max_lds_bytes) {
  RowPatterns rp;
  rp.ke = (nsym + 1 + 3) / 4 * 4;
  const uint8_t *slot = static_cast<const uint8_t *>(slot_v);
  auto fail = [&](const std::string &m) { rp.usable = false; rp.why = m; return rp; };

  // ---- 1. pattern of every row ----
  std::unordered_map<uint64_t, std::vector<int32_t>> by_hash;  // hash -> pattern ids (collisions are resolved by comparing bytes)
  std::vector<int32_t> rep;                                    // representative row of each pattern
  std::vector<int32_t> row_pat(nrows, -1);
  std::vector<uint8_t> cur, other;
  for (int r = 0; r < nrows; ++r) {
    const int ni = inc_ptr[r + 1] - inc_ptr[r];
    if (ni == 0) continue;  // row without elements: nothing to assemble
    if (ni * rp.ke > kMaxDepth) return fail("a dof is shared by more elements than the pattern kernel holds");
    pattern_bytes(r, n, rowptr, inc_ptr, inc_elem, inc_pos, slot, slot_bytes, cur);
    std::vector<int32_t> &ids = by_hash[fnv1a(cur)];
    int found = -1;
    for (int id : ids) {
      pattern_bytes(rep[id], n, rowptr, inc_ptr, inc_elem, inc_pos, slot, slot_bytes, other);
      if (other == cur) { found = id; break; }
    }
    if (found < 0) {
      found = static_cast<int>(rep.size());
      if (found >= max_patterns) return fail("rows share too few assembly patterns (unstructured numbering)");
      rep.push_back(r);
      ids.push_back(found);
    }
    row_pat[r] = found;
  }
  rp.num_patterns = static_cast<int>(rep.size());
  if (rp.num_patterns == 0) return fail("no rows");

  // ---- 2. W of every pattern ----
  const size_t nn = static_cast<size_t>(n) * n;
  size_t total = 0;
  for (int p = 0; p < rp.num_patterns; ++p) {
    const int r = rep[p];
    const int len = rowptr[r + 1] - rowptr[r], ni = inc_ptr[r + 1] - inc_ptr[r];
    const int cols = (len + 15) / 16 * 16;
    const int stride = ((cols / 16) % 2 == 0) ? cols + 16 : cols;  // k rows 32 banks apart: the four k of an MFMA step do not alias pairwise
    rp.pat_ni.push_back(ni);
    rp.pat_len.push_back(len);
    rp.pat_cols.push_back(cols);
    rp.pat_stride.push_back(stride);
    rp.pat_woff.push_back(static_cast<int64_t>(total));
    const size_t sz = static_cast<size_t>(ni) * rp.ke * stride;
    rp.max_w_doubles = std::max<int>(rp.max_w_doubles, static_cast<int>(sz));
    total += sz;
    if (total * sizeof(double) > max_w_bytes) return fail("pattern matrices exceed the memory budget");
  }
  if (static_cast<size_t>(rp.max_w_doubles) * sizeof(double) > static_cast<size_t>(max_lds_bytes))
    return fail("a pattern matrix does not fit the LDS");
  rp.w.assign(total, 0.0);
  for (int p = 0; p < rp.num_patterns; ++p) {
    const int r = rep[p];
    double *W = rp.w.data() + rp.pat_woff[p];
    const int stride = rp.pat_stride[p];
    int kk = 0;
    for (int k = inc_ptr[r]; k < inc_ptr[r + 1]; ++k, ++kk) {
      const int si = inc_pos[k];
      const uint8_t *s = slot + (static_cast<size_t>(inc_elem[k]) * n + si) * n * slot_bytes;
      for (int sj = 0; sj < n; ++sj) {
        const int sl = slot_bytes == 1 ? s[sj] : reinterpret_cast<const uint16_t *>(s)[sj];
        for (int c = 0; c <= nsym; ++c)
          W[static_cast<size_t>(kk * rp.ke + c) * stride + sl] += khat[c * nn + static_cast<size_t>(si) * n + sj];
      }
    }
  }

  // ---- 3. super tiles, by (pattern, alignment of the row inside its 128-byte line) -- heaviest patterns first, rows
  //         ascending.  Rows of one tile share shift = rowptr % 16: the kernel places W's columns `shift` entries to
  //         the right in LDS, so that every 16-column tile of the product is one whole, aligned cache line of the CRS
  //         values (partial-line stores were the largest cost of the first version of this kernel).
  std::vector<std::vector<int32_t>> rows_of(static_cast<size_t>(rp.num_patterns) * 16);
  for (int r = 0; r < nrows; ++r)
    if (row_pat[r] >= 0) rows_of[static_cast<size_t>(row_pat[r]) * 16 + (rowptr[r] & 15)].push_back(r);
  auto tile_cost = [&](int p) {
    const int ks = rp.pat_ni[p] * rp.ke / 4, ct = rp.pat_cols[p] / 16;
    return (static_cast<int64_t>(ks) * ct + 3 * ct + ks + 8) * (kRowsPerSuperTile / 64);
  };
  std::vector<int32_t> order(rp.num_patterns);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return tile_cost(a) > tile_cost(c); });
  std::vector<int64_t> cost;
  std::vector<int32_t> st_shift;
  for (int p : order)
    for (int shift = 0; shift < 16; ++shift) {
      const std::vector<int32_t> &rows = rows_of[static_cast<size_t>(p) * 16 + shift];
      const int ni = rp.pat_ni[p];
      for (size_t i = 0; i < rows.size(); i += kRowsPerSuperTile) {
        rp.st_pat.push_back(p);
        st_shift.push_back(shift);
        rp.st_off.push_back(static_cast<int64_t>(rp.st_rec.size()));
        for (int wv = 0; wv < kRowsPerSuperTile / 16; ++wv)
          for (int f = 0; f < 2 + ni; ++f)
            for (int l = 0; l < 16; ++l) {
              const size_t j = i + static_cast<size_t>(wv) * 16 + l;
              int32_t v = 0;
              if (j < rows.size()) {
                const int r = rows[j];
                if (f == 0) v = rowptr[r];
                else if (f == 1) v = (rowptr[r + 1] - rowptr[r]) | ((fixed && fixed[r]) ? (1 << 30) : 0);
                else v = inc_elem[inc_ptr[r] + f - 2];
              }
              rp.st_rec.push_back(v);
            }
        // a W reload costs about as much as a few tiles: charge it to the first tile of a pattern
        cost.push_back(tile_cost(p) + (i == 0 ? rp.pat_ni[p] * rp.ke * rp.pat_stride[p] / 256 : 0));
      }
    }
  // ---- 4. deal chunks of `chunk` consecutive super tiles to the persistent workgroups round-robin and lay the tile
  //         list out workgroup by workgroup.  The list is sorted by pattern, so every workgroup receives the same mix
  //         of heavy (vertex rows) and light (cell-interior rows) tiles whatever their real cost ratio is; a
  //         contiguous cost-model split left the workgroups with the light tiles (latency bound, many more of them)
  //         running long after the others had finished.
  const int nst = static_cast<int>(rp.st_pat.size());
  bool too_wide = false;
  auto finish = [&]() {
    rp.st_desc.resize(static_cast<size_t>(nst) * 8);
    rp.max_w_doubles = 0;
    for (int s = 0; s < nst; ++s) {
      const int p = rp.st_pat[s], shift = st_shift[s];
      const int len = rp.pat_len[p];
      const int cols = (len + shift + 15) / 16 * 16;                     // shifted row, whole lines
      if (cols > 16 * 9) too_wide = true;                                // kMaxColTiles of the kernel
      const int stride = ((cols / 16) % 2 == 0) ? cols + 16 : cols;      // LDS stride (bank rule as for pat_stride)
      rp.max_w_doubles = std::max(rp.max_w_doubles, rp.pat_ni[p] * rp.ke * stride);
      int32_t *d = &rp.st_desc[static_cast<size_t>(s) * 8];
      d[0] = p * 16 + shift;                       // identity of the LDS image
      d[1] = rp.pat_ni[p] | (shift << 8) | (len << 16);
      d[2] = cols;
      d[3] = stride | (rp.pat_stride[p] << 16);    // LDS stride, stride of W in memory
      d[4] = static_cast<int32_t>(rp.st_off[s] & 0xffffffffll);
      d[5] = static_cast<int32_t>(rp.st_off[s] >> 32);
      d[6] = static_cast<int32_t>(rp.pat_woff[p] & 0xffffffffll);
      d[7] = static_cast<int32_t>(rp.pat_woff[p] >> 32);
    }
    rp.usable = static_cast<size_t>(rp.max_w_doubles) * sizeof(double) <= static_cast<size_t>(max_lds_bytes);
    if (!rp.usable) rp.why = "a pattern matrix does not fit the LDS";
    if (too_wide) { rp.usable = false; rp.why = "CRS rows longer than the pattern kernel's register tile"; }
  };
  if (chunk <= 0) {  // contiguous ranges balanced by the cost model
    const int g = std::max(1, std::min(num_wgs, nst));
    int64_t sum = 0;
    for (int64_t c : cost) sum += c;
    rp.wg_ptr.assign(static_cast<size_t>(g) + 1, nst);
    rp.wg_ptr[0] = 0;
    int64_t acc = 0;
    int w = 1;
    for (int s = 0; s < nst && w < g; ++s) {
      acc += cost[s];
      while (w < g && acc * g >= sum * w) rp.wg_ptr[w++] = s + 1;
    }
    finish();
    return rp;
  }
  const int g = std::max(1, std::min(num_wgs, (nst + chunk - 1) / chunk));
  const int nchunks = (nst + chunk - 1) / chunk;
  std::vector<int32_t> new_pat, new_shift;
  std::vector<int64_t> new_off;
  new_pat.reserve(nst);
  new_off.reserve(nst);
  rp.wg_ptr.assign(static_cast<size_t>(g) + 1, 0);
  for (int w = 0; w < g; ++w) {
    for (int c = w; c < nchunks; c += g)
      for (int s = c * chunk; s < std::min(nst, (c + 1) * chunk); ++s) {
        new_pat.push_back(rp.st_pat[s]);
        new_shift.push_back(st_shift[s]);
        new_off.push_back(rp.st_off[s]);
      }
    rp.wg_ptr[w + 1] = static_cast<int32_t>(new_pat.size());
  }
  rp.st_pat.swap(new_pat);
  st_shift.swap(new_shift);
  rp.st_off.swap(new_off);
  finish();
  return rp;
}

}  // namespace mha
