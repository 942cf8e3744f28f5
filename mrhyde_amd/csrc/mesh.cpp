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

// ---- multi-variable blocks -------------------------------------------------------------------------------------------
namespace {
int basis_card(int dim, int type, int order) {
  if (type == MHA_BASIS_HGRAD) return order >= 1 ? ipow(order + 1, dim) : -1;
  if (type == MHA_BASIS_HVOL) return order == 0 ? 1 : -1;
  if (type == MHA_BASIS_HDIV) return (order == 1 && dim >= 2) ? 2 * dim : -1;
  return -1;
}
int max_hgrad_order(int nvars, const int *types, const int *orders) {
  int K = 0;
  for (int v = 0; v < nvars; ++v)
    if (types[v] == MHA_BASIS_HGRAD) K = std::max(K, orders[v]);
  return K;
}
}  // namespace

void mesh_multi_sizes(int dim, const int *nc, int nvars, const int *types, const int *orders, int *nverts, int *nelem,
                      int *n_tot, int64_t *ndof) {
  MHA_REQUIRE((dim == 2 || dim == 3) && nvars >= 1 && nvars <= 8, MHA_ERR_INVALID, "mesh_multi: bad dimension or variable count");
  const int K = max_hgrad_order(nvars, types, orders);
  int64_t nd = 0, ne = 1, nv = 1;
  int nt = 0;
  for (int d = 0; d < dim; ++d) {
    MHA_REQUIRE(nc[d] >= 1, MHA_ERR_INVALID, "cell counts must be positive");
    ne *= nc[d];
    nv *= nc[d] + 1;
  }
  for (int v = 0; v < nvars; ++v) {
    const int card = basis_card(dim, types[v], orders[v]);
    MHA_REQUIRE(card > 0, MHA_ERR_INVALID, "mesh_multi: unsupported (basis type, order) of variable " << v);
    nt += card;
    if (types[v] == MHA_BASIS_HGRAD) {
      MHA_REQUIRE(K % orders[v] == 0, MHA_ERR_INVALID, "mesh_multi: HGRAD orders must divide the largest one");
      int64_t m = 1;
      for (int d = 0; d < dim; ++d) m *= static_cast<int64_t>(orders[v]) * nc[d] + 1;
      nd += m;
    } else if (types[v] == MHA_BASIS_HVOL) {
      nd += ne;
    } else {
      for (int c = 0; c < dim; ++c) {
        int64_t m = 1;
        for (int d = 0; d < dim; ++d) m *= nc[d] + (d == c);
        nd += m;
      }
    }
  }
  MHA_REQUIRE(nd < (int64_t(1) << 31) && ne < (int64_t(1) << 31), MHA_ERR_INVALID, "mesh too large for int32 local ordinals");
  *nverts = static_cast<int>(nv);
  *nelem = static_cast<int>(ne);
  *n_tot = nt;
  *ndof = nd;
}

// Subcell-major dof map of a structured mesh for a block of HGRAD / HVOL / HDIV variables (what panzer's DOFManager
// plays in the reference, discretizationInterface.cpp:2336-2413; the numbering itself is this build's: nodes of the
// finest HGRAD lattice in lexicographic order, interleaved over the variables living on a node; then the cells (HVOL);
// then the faces direction by direction (HDIV)).  Element LID lists: vertices in shards order first, then the other
// lattice sites, then cell and face dofs; offsets[varptr[v] + dof] = position of the variable's dof in that list.
// orient: lowest-order HDIV signs (phi . n_out of the reference side, flipped when the side's global vertex ids are a
// reflected permutation); side_mask[dof]: bit 2d / 2d+1 = the dof lies on the low / high boundary in direction d.
void mesh_structured_multi(int dim, const int *nc, const double *lo, const double *hi, int nvars, const int *types,
                           const int *orders, double *verts, int32_t *cell2vert, int32_t *lids, int32_t *offsets,
                           int8_t *orient, uint8_t *side_mask, int32_t *dof_var) {
  int nverts, nelem, n_tot;
  int64_t ndof;
  mesh_multi_sizes(dim, nc, nvars, types, orders, &nverts, &nelem, &n_tot, &ndof);
  const int K = max_hgrad_order(nvars, types, orders), nn = 1 << dim;
  std::vector<int> varptr(nvars + 1, 0);
  for (int v = 0; v < nvars; ++v) varptr[v + 1] = varptr[v] + basis_card(dim, types[v], orders[v]);
  {  // vertices + cell -> vertex map from the single-variable generator
    std::vector<int32_t> tl(static_cast<size_t>(nelem) * nn), to(nn);
    mesh_structured(dim, 1, nc, lo, hi, verts, cell2vert, tl.data(), to.data(), nullptr);
  }
  int fd[3] = {1, 1, 1};
  size_t nfine = 1;
  for (int d = 0; d < dim; ++d) { fd[d] = K * nc[d] + 1; nfine *= static_cast<size_t>(fd[d]); }
  std::vector<int32_t> gnode;
  int32_t next = 0;
  if (K > 0) {
    gnode.assign(static_cast<size_t>(nvars) * nfine, -1);
    for (size_t s = 0; s < nfine; ++s) {
      const int a[3] = {static_cast<int>(s % fd[0]), static_cast<int>((s / fd[0]) % fd[1]), static_cast<int>(s / (static_cast<size_t>(fd[0]) * fd[1]))};
      for (int v = 0; v < nvars; ++v) {
        if (types[v] != MHA_BASIS_HGRAD) continue;
        const int st = K / orders[v];
        bool in = true;
        for (int d = 0; d < dim; ++d) in = in && (a[d] % st == 0);
        if (!in) continue;
        uint8_t m = 0;
        for (int d = 0; d < dim; ++d) {
          if (a[d] == 0) m |= 1u << (2 * d);
          if (a[d] == fd[d] - 1) m |= 1u << (2 * d + 1);
        }
        if (side_mask) side_mask[next] = m;
        if (dof_var) dof_var[next] = v;
        gnode[static_cast<size_t>(v) * nfine + s] = next++;
      }
    }
  }
  const int ncx = nc[0], ncy = nc[1], ncz = dim == 3 ? nc[2] : 1;
  int nhvol = 0, nhdiv = 0;
  std::vector<int> hvol_rank(nvars, 0), hdiv_rank(nvars, 0);
  for (int v = 0; v < nvars; ++v) {
    hvol_rank[v] = nhvol;
    hdiv_rank[v] = nhdiv;
    if (types[v] == MHA_BASIS_HVOL) ++nhvol;
    if (types[v] == MHA_BASIS_HDIV) ++nhdiv;
  }
  const int32_t cell_base = next;
  for (int e = 0; e < nelem && nhvol; ++e)
    for (int v = 0; v < nvars; ++v)
      if (types[v] == MHA_BASIS_HVOL) {
        if (side_mask) side_mask[next] = 0;
        if (dof_var) dof_var[next] = v;
        ++next;
      }
  int32_t face_base[3] = {0, 0, 0};
  for (int c = 0; c < dim && nhdiv; ++c) {
    face_base[c] = next;
    const int ex[3] = {ncx + (c == 0), ncy + (c == 1), dim == 3 ? ncz + (c == 2) : 1};
    for (int k = 0; k < ex[2]; ++k)
      for (int j = 0; j < ex[1]; ++j)
        for (int i = 0; i < ex[0]; ++i)
          for (int v = 0; v < nvars; ++v)
            if (types[v] == MHA_BASIS_HDIV) {
              const int idx[3] = {i, j, k};
              uint8_t m = 0;
              if (idx[c] == 0) m |= 1u << (2 * c);
              if (idx[c] == ex[c] - 1) m |= 1u << (2 * c + 1);
              if (side_mask) side_mask[next] = m;
              if (dof_var) dof_var[next] = v;
              ++next;
            }
  }
  MHA_REQUIRE(next == static_cast<int32_t>(ndof), MHA_ERR_STATE, "mesh_multi: dof count mismatch");
  const int kp = K + 1, nsite = K > 0 ? ipow(kp, dim) : 0;
  std::vector<int> site_order;
  {  // vertices in shards order, then the remaining tensor sites
    std::vector<char> isv(nsite + 1, 0);
    for (int v = 0; v < nn && K > 0; ++v) {
      int t = 0, mul = 1;
      for (int d = 0; d < dim; ++d) { t += (ref_vertex_sign(dim, v, d) > 0 ? K : 0) * mul; mul *= kp; }
      site_order.push_back(t);
      isv[t] = 1;
    }
    for (int t = 0; t < nsite; ++t)
      if (!isv[t]) site_order.push_back(t);
  }
  // shards side -> vertices (quad: 4 edges, hex: 6 faces), as in ref_tables.cpp
  static const int quad_side[4][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}};
  static const int hex_side[6][4] = {{0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {0, 4, 7, 3}, {0, 3, 2, 1}, {4, 5, 6, 7}};
  static const int face_of_dof3[6] = {3, 1, 0, 2, 4, 5}, edge_of_dof2[4] = {3, 1, 0, 2};
  for (int e = 0; e < nelem; ++e) {
    const int ci[3] = {e % ncx, (e / ncx) % ncy, e / (ncx * ncy)};
    int32_t *L = lids + static_cast<size_t>(e) * n_tot;
    int pos = 0;
    for (int so = 0; so < nsite; ++so) {
      const int t = site_order[so];
      const int a[3] = {t % kp, (t / kp) % kp, t / (kp * kp)};
      size_t s = 0, mul = 1;
      for (int d = 0; d < dim; ++d) { s += static_cast<size_t>(ci[d] * K + a[d]) * mul; mul *= static_cast<size_t>(fd[d]); }
      for (int v = 0; v < nvars; ++v) {
        if (types[v] != MHA_BASIS_HGRAD) continue;
        const int st = K / orders[v], p1 = orders[v] + 1;
        bool in = true;
        for (int d = 0; d < dim; ++d) in = in && (a[d] % st == 0);
        if (!in) continue;
        int dof = 0, m2 = 1;
        for (int d = 0; d < dim; ++d) { dof += (a[d] / st) * m2; m2 *= p1; }
        if (e == 0) offsets[varptr[v] + dof] = pos;
        L[pos++] = gnode[static_cast<size_t>(v) * nfine + s];
      }
    }
    for (int v = 0; v < nvars; ++v)
      if (types[v] == MHA_BASIS_HVOL) {
        if (e == 0) offsets[varptr[v]] = pos;
        L[pos++] = cell_base + e * nhvol + hvol_rank[v];
      }
    for (int f = 0; f < 2 * dim && nhdiv; ++f) {
      const int c = f / 2, sgn = f % 2;
      const int ex[3] = {ncx + (c == 0), ncy + (c == 1), dim == 3 ? ncz + (c == 2) : 1};
      int idx[3] = {ci[0], ci[1], ci[2]};
      idx[c] += sgn;
      const int lin = idx[0] + ex[0] * (idx[1] + ex[1] * idx[2]);
      for (int v = 0; v < nvars; ++v)
        if (types[v] == MHA_BASIS_HDIV) {
          if (e == 0) offsets[varptr[v] + f] = pos;
          L[pos++] = face_base[c] + lin * nhdiv + hdiv_rank[v];
        }
    }
    MHA_REQUIRE(pos == n_tot, MHA_ERR_STATE, "mesh_multi: LID list length mismatch");
    if (orient) {
      int8_t *o = orient + static_cast<size_t>(e) * n_tot;
      for (int k = 0; k < n_tot; ++k) o[k] = 1;
      const int32_t *cv = cell2vert + static_cast<size_t>(e) * nn;
      for (int v = 0; v < nvars; ++v) {
        if (types[v] != MHA_BASIS_HDIV) continue;
        for (int f = 0; f < 2 * dim; ++f) {
          bool flip;
          if (dim == 2) {
            const int *sn = quad_side[edge_of_dof2[f]];
            flip = cv[sn[0]] > cv[sn[1]];
          } else {
            const int *sn = hex_side[face_of_dof3[f]];
            int rot = 0;
            for (int k = 1; k < 4; ++k)
              if (cv[sn[k]] < cv[sn[rot]]) rot = k;
            flip = cv[sn[(rot + 1) % 4]] > cv[sn[(rot + 3) % 4]];
          }
          const int sigma = (f % 2) ? 1 : -1;  // phi_raw . n_out on its own face
          o[varptr[v] + f] = static_cast<int8_t>(flip ? -sigma : sigma);
        }
      }
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
