#include "mesh.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

#include "common.hpp"
#include "ref_tables.hpp"

namespace mha {

namespace {
// Host-side setup parallelism (std::thread; the library carries no OpenMP runtime).
template <class F>
void parallel_for_rows(int nrows, F &&body) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 32) nt = 32;
  if (nrows < 65536) nt = 1;
  std::atomic<int> next(0);
  const int chunk = 4096;
  auto work = [&]() {
    std::vector<int32_t> buf;
    for (;;) {
      const int r0 = next.fetch_add(chunk);
      if (r0 >= nrows) break;
      const int r1 = std::min(nrows, r0 + chunk);
      for (int r = r0; r < r1; ++r) body(r, buf);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
}
}  // namespace

void mesh_sizes(int dim, int order, const int *nc, int *nverts, int *nelem, int64_t *ndof) {
  MHA_REQUIRE(dim == 2 || dim == 3, MHA_ERR_INVALID, "dimension must be 2 or 3");
  int64_t nv = 1, ne = 1, nd = 1;
  for (int d = 0; d < dim; ++d) {
    MHA_REQUIRE(nc[d] >= 1, MHA_ERR_INVALID, "cell counts must be positive");
    nv *= nc[d] + 1;
    ne *= nc[d];
    nd *= static_cast<int64_t>(order) * nc[d] + 1;
  }
  MHA_REQUIRE(nd < (int64_t(1) << 31) && ne < (int64_t(1) << 31), MHA_ERR_INVALID,
              "mesh too large for int32 local ordinals (LO=int)");
  *nverts = static_cast<int>(nv);
  *nelem = static_cast<int>(ne);
  *ndof = nd;
}

void mesh_structured(int dim, int order, const int *nc, const double *lo, const double *hi,
                     double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                     uint8_t *boundary_dof) {
  int nv, ne;
  int64_t nd;
  mesh_sizes(dim, order, nc, &nv, &ne, &nd);
  const int m = order + 1, n = ipow(m, dim), nn = 1 << dim;
  const int N[3] = {nc[0], nc[1], dim == 3 ? nc[2] : 1};
  double h[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) h[d] = (hi[d] - lo[d]) / nc[d];

  // vertex coordinates, x fastest
  {
    int v = 0;
    for (int k = 0; k <= (dim == 3 ? N[2] : 0); ++k)
      for (int j = 0; j <= N[1]; ++j)
        for (int i = 0; i <= N[0]; ++i, ++v) {
          const size_t vo = static_cast<size_t>(v) * dim;
          verts[vo + 0] = lo[0] + i * h[0];
          verts[vo + 1] = lo[1] + j * h[1];
          if (dim == 3) verts[vo + 2] = lo[2] + k * h[2];
        }
  }
  // offsets: tensor dof -> slot in the element's LID list (vertices in shards order first)
  std::vector<int> slot(n, -1);
  for (int v = 0; v < nn; ++v) {
    int t = 0, stride = 1;
    for (int d = 0; d < dim; ++d) {
      if (ref_vertex_sign(dim, v, d) > 0) t += stride * order;
      stride *= m;
    }
    slot[t] = v;
  }
  for (int t = 0, next = nn; t < n; ++t)
    if (slot[t] < 0) slot[t] = next++;
  for (int t = 0; t < n; ++t) offsets[t] = slot[t];

  const int64_t Dx = int64_t(order) * N[0] + 1, Dy = int64_t(order) * N[1] + 1,
                Dz = dim == 3 ? int64_t(order) * N[2] + 1 : 1;
  int e = 0;
  for (int k = 0; k < N[2]; ++k)
    for (int j = 0; j < N[1]; ++j)
      for (int i = 0; i < N[0]; ++i, ++e) {
        for (int v = 0; v < nn; ++v) {
          const int vi = i + (ref_vertex_sign(dim, v, 0) > 0), vj = j + (ref_vertex_sign(dim, v, 1) > 0);
          const int vk = dim == 3 ? k + (ref_vertex_sign(dim, v, 2) > 0) : 0;
          cell2vert[static_cast<size_t>(e) * nn + v] = (vk * (N[1] + 1) + vj) * (N[0] + 1) + vi;
        }
        for (int t = 0; t < n; ++t) {
          const int64_t gi = int64_t(order) * i + t % m, gj = int64_t(order) * j + (t / m) % m;
          const int64_t gk = dim == 3 ? int64_t(order) * k + t / (m * m) : 0;
          lids[static_cast<size_t>(e) * n + slot[t]] = static_cast<int32_t>((gk * Dy + gj) * Dx + gi);
        }
      }
  if (boundary_dof) {
    for (int64_t gk = 0; gk < Dz; ++gk)
      for (int64_t gj = 0; gj < Dy; ++gj)
        for (int64_t gi = 0; gi < Dx; ++gi) {
          bool b = gi == 0 || gi == Dx - 1 || gj == 0 || gj == Dy - 1;
          if (dim == 3) b = b || gk == 0 || gk == Dz - 1;
          boundary_dof[(gk * Dy + gj) * Dx + gi] = b ? 1 : 0;
        }
  }
}

void build_row_incidence(int nrows, int nelem, int n, const int32_t *lids, std::vector<int32_t> &ptr,
                         std::vector<int32_t> &elem, std::vector<int32_t> &lpos) {
  ptr.assign(static_cast<size_t>(nrows) + 1, 0);
  const size_t tot = static_cast<size_t>(nelem) * n;
  for (size_t k = 0; k < tot; ++k) {
    MHA_REQUIRE(lids[k] >= 0 && lids[k] < nrows, MHA_ERR_INVALID, "LID out of range: " << lids[k]);
    ptr[lids[k] + 1]++;
  }
  for (int r = 0; r < nrows; ++r) ptr[r + 1] += ptr[r];
  elem.assign(tot, 0);
  lpos.assign(tot, 0);
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int e = 0; e < nelem; ++e)
    for (int i = 0; i < n; ++i) {
      const int r = lids[static_cast<size_t>(e) * n + i];
      elem[fill[r]] = e;
      lpos[fill[r]] = i;
      fill[r]++;
    }
}

// A caller-supplied graph must be a CRS pattern the scatter can use: rowptr from 0, non-decreasing, nnz < 2^31; columns
// strictly ascending inside [0, nrows); and every coupling (LID_i, LID_j) of every element present -- the slot maps of
// the device paths are built by a column search, and a missing column has no slot (the reference's sumIntoValues would
// drop such a contribution silently; here it is an input error).
void validate_crs_graph(int nrows, int nelem, int n, const int32_t *lids, const int32_t *rowptr, const int32_t *colind) {
  MHA_REQUIRE(rowptr[0] == 0, MHA_ERR_INVALID, "rowptr[0] must be 0");
  for (int r = 0; r < nrows; ++r)
    MHA_REQUIRE(rowptr[r + 1] >= rowptr[r], MHA_ERR_INVALID, "rowptr must be non-decreasing (row " << r << ")");
  for (int r = 0; r < nrows; ++r)
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      MHA_REQUIRE(colind[p] >= 0 && colind[p] < nrows, MHA_ERR_INVALID, "column " << colind[p] << " of row " << r << " outside [0," << nrows << ")");
      MHA_REQUIRE(p == rowptr[r] || colind[p] > colind[p - 1], MHA_ERR_INVALID, "colind must be strictly ascending in row " << r);
    }
  std::atomic<int> bad_elem(-1);
  parallel_for_rows(nelem, [&](int e, std::vector<int32_t> &) {
    if (bad_elem.load(std::memory_order_relaxed) >= 0) return;
    const int32_t *l = lids + static_cast<size_t>(e) * n;
    for (int i = 0; i < n; ++i) {
      const int32_t *lo = colind + rowptr[l[i]], *hi = colind + rowptr[l[i] + 1];
      for (int j = 0; j < n; ++j)
        if (!std::binary_search(lo, hi, l[j])) { bad_elem.store(e); return; }
    }
  });
  MHA_REQUIRE(bad_elem.load() < 0, MHA_ERR_INVALID,
              "the CRS graph misses a coupling of element " << bad_elem.load() << " (every pair of dofs of an element must be a graph entry)");
}

void build_crs_graph(int nrows, int nelem, int n, const int32_t *lids, std::vector<int32_t> &rowptr,
                     std::vector<int32_t> &colind) {
  std::vector<int32_t> ptr, elem, lpos;
  build_row_incidence(nrows, nelem, n, lids, ptr, elem, lpos);
  rowptr.assign(static_cast<size_t>(nrows) + 1, 0);
  // pass 1: count unique columns per row; pass 2: fill
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      int64_t nnz = 0;  // the counts are summed in 64 bits: local ordinals are int32 (LO = int), so nnz must stay below 2^31
      for (int r = 0; r < nrows; ++r) nnz += rowptr[r + 1];
      MHA_REQUIRE(nnz < (int64_t(1) << 31), MHA_ERR_INVALID,
                  "the block's CRS graph has " << nnz << " entries: more than int32 row offsets hold (2^31 - 1)");
      for (int r = 0; r < nrows; ++r) rowptr[r + 1] += rowptr[r];
      colind.assign(static_cast<size_t>(rowptr[nrows]), 0);
    }
    parallel_for_rows(nrows, [&](int r, std::vector<int32_t> &buf) {
      buf.clear();
      int last = -1;
      for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
        if (elem[k] == last) continue;
        last = elem[k];
        const int32_t *l = lids + static_cast<size_t>(last) * n;
        buf.insert(buf.end(), l, l + n);
      }
      std::sort(buf.begin(), buf.end());
      buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
      if (pass == 0) rowptr[r + 1] = static_cast<int32_t>(buf.size());
      else std::copy(buf.begin(), buf.end(), colind.begin() + rowptr[r]);
    });
  }
}

}  // namespace mha
