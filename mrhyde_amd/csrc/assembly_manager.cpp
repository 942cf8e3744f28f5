#include "assembly_manager.hpp"

#include <cstring>
#include <unordered_map>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mesh.hpp"

namespace mha {

AssemblyManager::AssemblyManager(const mha_block_desc &desc) {
  MHA_REQUIRE(desc.dimension == 2 || desc.dimension == 3, MHA_ERR_INVALID, "dimension must be 2 or 3");
  MHA_REQUIRE((desc.dimension == 2 && desc.topology == MHA_TOPO_QUAD4) ||
                  (desc.dimension == 3 && desc.topology == MHA_TOPO_HEX8),
              MHA_ERR_INVALID, "supported cell topologies: Quadrilateral_4 (2-D), Hexahedron_8 (3-D)");
  MHA_REQUIRE(desc.num_vars >= 1 && desc.num_vars <= MHA_MAX_VARS, MHA_ERR_INVALID,
              "num_vars must be in [1," << MHA_MAX_VARS << "]; got " << desc.num_vars);
  dim_ = desc.dimension;
  n_ = 0;
  for (int v = 0; v < desc.num_vars; ++v) {
    VarInfo vi{desc.basis_type[v], desc.basis_order[v], 0};
    switch (vi.type) {
      case MHA_BASIS_HGRAD:
        MHA_REQUIRE(vi.order >= 1 && vi.order <= 8, MHA_ERR_INVALID, "HGRAD order must be in [1,8]");
        vi.card = ipow(vi.order + 1, dim_);
        break;
      case MHA_BASIS_HVOL:
        MHA_REQUIRE(vi.order == 0, MHA_ERR_INVALID, "HVOL is available at order 0 (Basis_HVOL_C0_FEM)");
        vi.card = 1;
        break;
      case MHA_BASIS_HDIV:
        MHA_REQUIRE(vi.order == 1, MHA_ERR_INVALID, "HDIV is available at order 1 (lowest-order In_FEM)");
        vi.card = 2 * dim_;
        break;
      default:
        MHA_REQUIRE(false, MHA_ERR_INVALID, "unknown basis type " << vi.type << " for variable " << v);
    }
    vars_.push_back(vi);
    n_ += vi.card;
  }
  single_hgrad_ = vars_.size() == 1 && vars_[0].type == MHA_BASIS_HGRAD;
  int max_order = 0;
  for (const auto &vi : vars_) max_order = std::max(max_order, vi.order);
  order_ = single_hgrad_ ? vars_[0].order : std::max(1, max_order);
  // default quadrature = 2*max order (reference: discretizationInterface.cpp:166)
  qdeg_ = desc.quadrature_degree > 0 ? desc.quadrature_degree : 2 * std::max(1, max_order);
  // reference tables of the (first) HGRAD variable + the geometry basis; the other variables' tables are built in
  // buildVarLayout()
  ref_ = make_ref_tables(dim_, order_, qdeg_);
  nq_ = ref_.nq;
  nnodes_ = ref_.nnodes;
  workset_size_ = desc.workset_size;

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  MHA_REQUIRE(e == hipSuccess && ndev > 0, MHA_ERR_DEVICE,
              "no HIP device available (" << hipGetErrorString(e) << "): the MI355X path has no CPU fallback");
  MHA_REQUIRE(desc.device >= 0 && desc.device < ndev, MHA_ERR_INVALID, "device ordinal out of range");
  device_ = desc.device;
  DeviceGuard guard(device_);  // the caller's current device is restored when the constructor returns

  d_ref_basis_.upload(ref_.basis);
  d_ref_grad_.upload(ref_.grad);
  d_ref_wts_.upload(ref_.wts);
  d_nodeval_.upload(ref_.nodeval);
  d_nodegrad_.upload(ref_.nodegrad);
  d_phi1d_.upload(ref_.phi1d);
  d_dphi1d_.upload(ref_.dphi1d);
  d_gw1d_.upload(ref_.gauss_wts);
  d_gp1d_.upload(ref_.gauss_pts);
  MHA_HIP(hipEventCreate(&ev0_));
  MHA_HIP(hipEventCreate(&ev1_));

  wkset_.block = 0;
  wkset_.dimension = dim_;
  wkset_.numip = nq_;
  wkset_.numVars = static_cast<int>(vars_.size());
  wkset_.single_hgrad = single_hgrad_;
  buildVarLayout();
  wkset_.order = order_;
  wkset_.nq1 = ref_.nq1;
}

AssemblyManager::~AssemblyManager() {
  if (side_stream_) (void)hipStreamDestroy(side_stream_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
}

// createGroups' copies of LIDs / nodes (reference: assemblyManager.cpp:656-688) + createFixedDOFs (:185-265)
void AssemblyManager::setMesh(int nelem, const double *nodes, const int32_t *lids, const int32_t *offsets,
                              int nrows, const uint8_t *fixed) {
  boundary_groups_.clear();  // entries refer to the previous mesh's element ids
  has_orient_ = false;
  d_orient_.resize(0);
  MHA_REQUIRE(nelem > 0 && nrows > 0 && nodes && lids && offsets, MHA_ERR_INVALID, "mha_set_mesh: null or empty input");
  const size_t nl = static_cast<size_t>(nelem) * n_;
  for (size_t k = 0; k < nl; ++k)
    MHA_REQUIRE(lids[k] >= 0 && lids[k] < nrows, MHA_ERR_INVALID, "LID " << lids[k] << " outside [0," << nrows << ")");
  std::vector<char> seen(n_, 0);
  for (int t = 0; t < n_; ++t) {
    MHA_REQUIRE(offsets[t] >= 0 && offsets[t] < n_ && !seen[offsets[t]], MHA_ERR_INVALID,
                "offsets must be a permutation of 0.." << n_ - 1);
    seen[offsets[t]] = 1;
  }
  nelem_ = nelem;
  nrows_ = nrows;
  h_lids_.assign(lids, lids + nl);
  d_nodes_.upload(nodes, static_cast<size_t>(nelem) * nnodes_ * dim_);
  d_lids_.upload(lids, nl);
  d_offsets_.upload(offsets, n_);
  has_fixed_ = fixed != nullptr;
  if (fixed) h_fixed_.assign(fixed, fixed + nrows);
  if (fixed) d_fixed_.upload(fixed, nrows);
  else d_fixed_.resize(0);
  has_mesh_ = true;
  subgrid_checked_ = false;
  has_graph_ = false;
  porous_direct_ = -1;
  porous_db_ = PorousDatabase();
  // workset size <= 0 or larger than the block => one workset (reference: assemblyManager.cpp:326-332)
  const int ws = (workset_size_ <= 0 || workset_size_ > nelem_) ? nelem_ : workset_size_;
  wkset_.maxElem = ws;
}

// slot tables of the point engine: per distinct (type, order) the reference values [nq][nslot][card padded to 4] of
// value / gradient components (HGRAD), value (HVOL), vector components + divergence (HDIV, raw In_FEM functions:
// dof 2c = (1-x_c)/2 e_c, dof 2c+1 = (1+x_c)/2 e_c)
void AssemblyManager::buildVarLayout() {
  VarLayoutDev &L = layout_;
  L = VarLayoutDev();
  L.nvars = static_cast<int>(vars_.size());
  L.nq = nq_;
  std::vector<double> tables;
  std::vector<std::pair<std::pair<int, int>, int>> built;  // (type, order) -> offset
  for (int v = 0; v < L.nvars; ++v) {
    const VarInfo &vi = vars_[v];
    L.type[v] = vi.type;
    L.card[v] = vi.card;
    L.nslot[v] = vi.type == MHA_BASIS_HVOL ? 1 : 1 + dim_;
    L.cardpad[v] = (vi.card + 3) & ~3;
    L.varptr[v + 1] = L.varptr[v] + vi.card;
    L.slotptr[v + 1] = L.slotptr[v] + L.nslot[v];
    int off = -1;
    for (const auto &b : built)
      if (b.first == std::make_pair(vi.type, vi.order)) off = b.second;
    if (off < 0) {
      off = static_cast<int>(tables.size());
      built.push_back({{vi.type, vi.order}, off});
      const int ns = L.nslot[v], cp = L.cardpad[v];
      tables.resize(tables.size() + static_cast<size_t>(nq_) * ns * cp, 0.0);
      double *T = tables.data() + off;  // [q][slot][dof], rows padded with zeros
      if (vi.type == MHA_BASIS_HGRAD) {
        const RefTables rt = (vi.order == order_) ? ref_ : make_ref_tables(dim_, vi.order, qdeg_);
        for (int f = 0; f < vi.card; ++f)
          for (int q = 0; q < nq_; ++q) {
            T[(q * ns) * cp + f] = rt.basis[f * nq_ + q];
            for (int d = 0; d < dim_; ++d) T[(q * ns + 1 + d) * cp + f] = rt.grad[(f * nq_ + q) * dim_ + d];
          }
      } else if (vi.type == MHA_BASIS_HVOL) {
        for (int q = 0; q < nq_; ++q) T[q * cp] = 1.0;
      } else {
        for (int c = 0; c < dim_; ++c)
          for (int sd = 0; sd < 2; ++sd)
            for (int q = 0; q < nq_; ++q) {
              const double x = ref_.ip[q * dim_ + c];
              T[(q * ns + c) * cp + 2 * c + sd] = sd ? 0.5 * (1.0 + x) : 0.5 * (1.0 - x);
              T[(q * ns + dim_) * cp + 2 * c + sd] = sd ? 0.5 : -0.5;
            }
      }
    }
    L.table_off[v] = off;
  }
  L.n_tot = L.varptr[L.nvars];
  L.ns_tot = L.slotptr[L.nvars];
  MHA_REQUIRE(L.ns_tot <= kMaxSlots, MHA_ERR_INVALID, "too many field slots (" << L.ns_tot << ")");
  L.tables_size = static_cast<int>(tables.size());
  d_slot_tables_.upload(tables);
  L.tables = d_slot_tables_.data();
}

void AssemblyManager::setOrientation(const int8_t *signs) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "mha_set_orientation before mha_set_mesh");
  has_orient_ = signs != nullptr;
  porous_db_ = PorousDatabase();  // (its uniformity check and the cached element matrix depend on the signs)
  if (!signs) { d_orient_.resize(0); h_orient_.clear(); db_index_.clear(); return; }
  const size_t cnt = static_cast<size_t>(nelem_) * n_;
  for (size_t k = 0; k < cnt; ++k)
    MHA_REQUIRE(signs[k] == 1 || signs[k] == -1, MHA_ERR_INVALID, "orientation signs must be +1 or -1");
  d_orient_.upload(signs, cnt);
  h_orient_.assign(signs, signs + cnt);
  db_index_.clear();
}

void AssemblyManager::prepareRowGather(bool need_jacobian, bool dense) {
  if (!has_incidence_) {
    std::vector<int32_t> ptr, elem, lpos;
    build_row_incidence(nrows_, nelem_, n_, h_lids_.data(), ptr, elem, lpos);
    d_inc_ptr_.upload(ptr);
    d_inc_elem_.upload(elem);
    d_inc_pos_.upload(lpos);
    std::vector<int32_t> offs(n_), p2d(n_, 0);
    d_offsets_.download(offs.data());
    for (int f = 0; f < n_; ++f) p2d[offs[f]] = f;
    std::vector<int32_t> ldof(lpos.size());
    for (size_t k = 0; k < lpos.size(); ++k) ldof[k] = p2d[lpos[k]];
    d_inc_dof_.upload(ldof);
    max_row_ = 0;
    for (int r = 0; r < nrows_; ++r) max_row_ = std::max(max_row_, h_rowptr_[r + 1] - h_rowptr_[r]);
    has_incidence_ = true;
  }
  prepareElemSlots();
  if (!dense) return;  // (the direct form of porousMixed needs the incidences and the slot map only)
  if (need_jacobian) d_gather_J_.resize(static_cast<size_t>(nelem_) * n_ * n_);
  d_gather_res_.resize(static_cast<size_t>(nelem_) * n_);
}

// Database mode of the direct form.  Preconditions, checked once per mesh / graph (the coefficient kinds per assembly):
// every element has the same vertex offsets from its first vertex and the same orientation signs, bit for bit (then the
// direct kernel, which works on relative coordinates, produces the same matrix for every element).  Rows are classified
// by what determines their values: fixed flag, and per incident element its local dof and the slots of the element's
// columns in the row.  Per class the first run of >= 2 K consecutive rows gives K = ceil(128 / len) + 2 representative
// rows (enough for any 1 KB chunk to be sourced contiguously from their periodic image); every other row of a class that
// has representatives is REPLICATED; the rest (fixed rows, short or rare classes, the representatives) are COMPUTED as
// before, by the elements incident to them.
bool AssemblyManager::porousDatabaseUsable() {
  PorousDatabase &db = porous_db_;
  if (db.state >= 0) return db.state == 1;
  db.state = 0;
  const char *m = std::getenv("MHA_POROUS_DATABASE");
  if (m && m[0] == '0') { db.why = "MHA_POROUS_DATABASE=0"; return false; }
  if (!porousDirectUsable() || elem_slot_bytes_ != 1) { db.why = "direct form not usable"; return false; }
  const int nn = nnodes_, d = dim_;
  std::vector<double> nodes(static_cast<size_t>(nelem_) * nn * d);
  d_nodes_.download(nodes.data());
  for (int e = 1; e < nelem_; ++e)
    for (int k = 1; k < nn; ++k)
      for (int c = 0; c < d; ++c) {
        const double a = nodes[(static_cast<size_t>(e) * nn + k) * d + c] - nodes[static_cast<size_t>(e) * nn * d + c];
        const double b0 = nodes[static_cast<size_t>(k) * d + c] - nodes[c];
        if (std::memcmp(&a, &b0, sizeof(double)) != 0) { db.why = "elements of different shapes"; return false; }
      }
  if (has_orient_)
    for (int e = 1; e < nelem_; ++e)
      if (std::memcmp(&h_orient_[static_cast<size_t>(e) * n_], &h_orient_[0], n_) != 0) { db.why = "orientation signs differ between elements"; return false; }
  {  // is the common shape an axis-aligned box?  (shards vertex order: bit pattern of vertex k = (k in {1,2,5,6}, k in {2,3,6,7}, k >= 4))
    db.axis_aligned = true;
    for (int k = 0; k < nn; ++k) {
      const bool bit[3] = {k == 1 || k == 2 || k == 5 || k == 6, k == 2 || k == 3 || k == 6 || k == 7, k >= 4};
      const int ref[3] = {1, 3, 4};  // the vertices one step from vertex 0 in x, y, z
      for (int c = 0; c < d; ++c) {
        const double rel = nodes[static_cast<size_t>(k) * d + c] - nodes[c];
        const double want = bit[c] ? nodes[static_cast<size_t>(ref[c]) * d + c] - nodes[c] : 0.0;
        if (rel != want) db.axis_aligned = false;
      }
    }
  }
  // ---- row classes ----
  prepareElemSlots();
  std::vector<uint8_t> slot(static_cast<size_t>(nelem_) * n_ * n_);
  MHA_HIP(hipStreamSynchronize(stream_));
  d_elem_slot_.download(slot.data());
  std::vector<int32_t> ptr, elem, lpos, offs(n_), p2d(n_, 0);
  build_row_incidence(nrows_, nelem_, n_, h_lids_.data(), ptr, elem, lpos);
  d_offsets_.download(offs.data());
  for (int f = 0; f < n_; ++f) p2d[offs[f]] = f;
  std::unordered_map<std::string, int32_t> classes;
  std::vector<int32_t> cls(nrows_, -1);
  std::string key;
  for (int r = 0; r < nrows_; ++r) {
    if (has_fixed_ && h_fixed_[r]) continue;  // fixed rows: computed (zeroed) by the finishing pass
    key.clear();
    key.push_back(static_cast<char>(h_rowptr_[r + 1] - h_rowptr_[r]));
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      key.push_back(static_cast<char>(p2d[lpos[k]]));
      const uint8_t *srow = &slot[(static_cast<size_t>(elem[k]) * n_ + lpos[k]) * n_];
      for (int f = 0; f < n_; ++f) key.push_back(static_cast<char>(srow[offs[f]]));
    }
    cls[r] = classes.emplace(key, static_cast<int32_t>(classes.size())).first->second;
  }
  const int nc = static_cast<int>(classes.size());
  std::vector<int32_t> rep_entry(nc, -1), len_of(nc, 0), K_of(nc, 0);
  std::vector<uint8_t> replicated(nrows_, 0);
  // pass 1: representatives = the first K rows of the first long run of a class
  for (int r = 0; r < nrows_;) {
    int r1 = r + 1;
    while (r1 < nrows_ && cls[r1] == cls[r]) ++r1;
    const int c = cls[r];
    if (c >= 0 && rep_entry[c] < 0) {
      const int len = h_rowptr_[r + 1] - h_rowptr_[r];
      const int K = len > 0 ? (128 + len - 1) / len + 2 : 0;
      if (len > 0 && r1 - r >= 2 * K) { rep_entry[c] = h_rowptr_[r]; len_of[c] = len; K_of[c] = K; for (int q = r + K; q < r1; ++q) replicated[q] = 1; }
    }
    r = r1;
  }
  // pass 2: every other row of a class that has representatives
  for (int r = 0; r < nrows_; ++r) {
    const int c = cls[r];
    if (c < 0 || rep_entry[c] < 0 || replicated[r]) continue;
    const bool is_rep = h_rowptr_[r] >= rep_entry[c] && h_rowptr_[r] < rep_entry[c] + K_of[c] * len_of[c];
    if (!is_rep) replicated[r] = 1;
  }
  // chunks of the replicated ranges (maximal runs of replicated rows of one class), 1 KB on 128-byte lines
  std::vector<int32_t> chunks;
  int64_t computed = 0;
  for (int r = 0; r < nrows_;) {
    if (!replicated[r]) { ++computed; ++r; continue; }
    int r1 = r + 1;
    while (r1 < nrows_ && replicated[r1] && cls[r1] == cls[r]) ++r1;
    const int c = cls[r], len = len_of[c];
    const int64_t dbeg = h_rowptr_[r], dend = h_rowptr_[r1];
    for (int64_t c0 = dbeg / 16 * 16; c0 < dend; c0 += 128) {
      const int64_t ph = ((c0 - dbeg) % len + len) % len;
      chunks.push_back(static_cast<int32_t>(c0 / 2));
      chunks.push_back(static_cast<int32_t>(rep_entry[c] + ph));
      chunks.push_back(static_cast<int32_t>(dbeg));
      chunks.push_back(static_cast<int32_t>(dend));
    }
    r = r1;
  }
  if (chunks.empty()) { db.why = "no class has a run long enough to replicate"; return false; }
  // elements incident to computed rows store their entries; diagonal positions of the computed face rows only
  std::vector<uint8_t> jacflag(nelem_, 0);
  std::vector<int32_t> diag(nrows_, -1);
  for (int r = 0; r < nrows_; ++r) {
    if (replicated[r]) continue;
    bool face = false;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) { jacflag[elem[k]] = 1; face = face || p2d[lpos[k]] > 0; }
    if (face)
      for (int k = h_rowptr_[r]; k < h_rowptr_[r + 1]; ++k)
        if (h_colind_[k] == r) diag[r] = k;
  }
  db.jacflag.upload(jacflag);
  {
    std::vector<int32_t> elist;
    for (int e = 0; e < nelem_; ++e)
      if (jacflag[e]) elist.push_back(e);
    db.num_listed = static_cast<int>(elist.size());
    if (elist.empty()) elist.push_back(0);
    db.elist.upload(elist);
  }
  db.diag.upload(diag);
  db.chunks.upload(chunks);
  db.num_chunks = static_cast<int>(chunks.size() / 4);
  db.num_classes = nc;
  db.computed_rows = computed;
  db.state = 1;
  if (std::getenv("MHA_VERBOSE")) {
    int64_t flagged = 0;
    for (uint8_t f : jacflag) flagged += f;
    fprintf(stderr, "[mrhyde_amd] porousMixed database mode: %d row classes, %lld of %d rows computed, %lld of %d elements store entries, %d chunks\n",
            nc, (long long)computed, nrows_, (long long)flagged, nelem_, db.num_chunks);
  }
  return true;
}

// The direct form of the porousMixed assembly (kernels/porous_element.hip) rests on one property of the mesh: any two
// elements share at most ONE dof (a face), so that a matrix entry (i, j), i != j, has one contributing element and a
// row at most two.  Checked here on the LID lists, once per mesh / graph; anything else keeps the row gather.
bool AssemblyManager::porousDirectUsable() {
  if (porous_direct_ >= 0) return porous_direct_ == 1;
  porous_direct_ = 0;
  const char *m = std::getenv("MHA_POROUS_DIRECT"), *k = std::getenv("MHA_POROUS_KERNEL"), *g = std::getenv("MHA_GATHER_ORDER");
  if (m && m[0] == '0') { porous_direct_why_ = "MHA_POROUS_DIRECT=0"; return false; }
  if ((k && k[0] == 'e') || (g && g[0] == 'p')) { porous_direct_why_ = "point engine / position order forced"; return false; }
  if (!physics_ || physics_->label != "porousMixed" || n_ != 1 + 2 * dim_) { porous_direct_why_ = "not the lowest-order mixed element"; return false; }
  std::vector<int32_t> ptr, elem, lpos;
  build_row_incidence(nrows_, nelem_, n_, h_lids_.data(), ptr, elem, lpos);
  for (int r = 0; r < nrows_; ++r) {
    const int ni = ptr[r + 1] - ptr[r];
    if (ni > 2) { porous_direct_why_ = "a row with more than two incident elements"; return false; }
    if (ni == 2) {
      const int32_t *a = &h_lids_[static_cast<size_t>(elem[ptr[r]]) * n_], *b = &h_lids_[static_cast<size_t>(elem[ptr[r] + 1]) * n_];
      int shared = 0;
      for (int i = 0; i < n_; ++i)
        for (int j = 0; j < n_; ++j) shared += a[i] == b[j];
      if (shared != 1) { porous_direct_why_ = "two elements share more than one dof"; return false; }
    }
  }
  for (int e = 0; e < nelem_; ++e)  // (an element listing a dof twice would add twice into one entry)
    for (int i = 0; i < n_; ++i)
      for (int j = i + 1; j < n_; ++j)
        if (h_lids_[static_cast<size_t>(e) * n_ + i] == h_lids_[static_cast<size_t>(e) * n_ + j]) { porous_direct_why_ = "repeated dof in an element"; return false; }
  // which incidence of its row an element is (dof order), and where the diagonal of a face row sits in the CRS
  std::vector<int32_t> offs(n_), p2d(n_, 0);
  d_offsets_.download(offs.data());
  for (int f = 0; f < n_; ++f) p2d[offs[f]] = f;
  std::vector<uint8_t> side(static_cast<size_t>(nelem_) * n_, 0);
  std::vector<int32_t> diag(nrows_, -1);
  for (int r = 0; r < nrows_; ++r) {
    bool face = false;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      const int d = p2d[lpos[k]];
      side[static_cast<size_t>(elem[k]) * n_ + d] = static_cast<uint8_t>(k - ptr[r]);
      face = face || d > 0;
    }
    if (face)
      for (int k = h_rowptr_[r]; k < h_rowptr_[r + 1]; ++k)
        if (h_colind_[k] == r) diag[r] = k;
  }
  d_direct_side_.upload(side);
  d_direct_diag_.upload(diag);
  porous_direct_ = 1;
  return true;
}

void AssemblyManager::launchPointEngine(int compute_jacobian, const ElemOut &out, int e_begin, int e_count) {
  // CRS scatter through the element-major slot map (built once per graph) instead of a column search per entry
  if (compute_jacobian && out.crs_vals) {
    prepareElemSlots();
    wkset_.elem_slot = d_elem_slot_.data();
    wkset_.elem_slot_bytes = elem_slot_bytes_;
  } else {
    wkset_.elem_slot = nullptr;
  }
  wkset_.layout = layout_;
  wkset_.layout.orient = has_orient_ ? d_orient_.data() : nullptr;
  wkset_.use_point_engine = true;
  wkset_.first_elem = e_begin;
  wkset_.numElem = e_count;
  wkset_.res = out;
  physics_->volumeResidual();
  wkset_.use_point_engine = false;
}

void AssemblyManager::setGraph(const int32_t *rowptr, const int32_t *colind) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "mha_set_graph before mha_set_mesh");
  if (rowptr && colind) {
    MHA_REQUIRE(rowptr[0] == 0, MHA_ERR_INVALID, "rowptr[0] must be 0");
    for (int r = 0; r < nrows_; ++r)
      MHA_REQUIRE(rowptr[r + 1] >= rowptr[r], MHA_ERR_INVALID, "rowptr must be non-decreasing");
    validate_crs_graph(nrows_, nelem_, n_, h_lids_.data(), rowptr, colind);
    h_rowptr_.assign(rowptr, rowptr + nrows_ + 1);
    h_colind_.assign(colind, colind + h_rowptr_[nrows_]);
  } else {
    MHA_REQUIRE(!rowptr && !colind, MHA_ERR_INVALID, "pass both rowptr and colind, or neither");
    build_crs_graph(nrows_, nelem_, n_, h_lids_.data(), h_rowptr_, h_colind_);
  }
  d_rowptr_.upload(h_rowptr_);
  d_colind_.upload(h_colind_);
  has_graph_ = true;
  ro_ = RowOwnerData();
  bpat_ = BlockPatternData();
  gro_ = GeneralRowOwnerData();
  has_elem_slot_ = false;
  has_incidence_ = false;
  porous_direct_ = -1;
  porous_db_ = PorousDatabase();
}

void AssemblyManager::selectPhysics(int physics_id) {
  // the module's myvars / mybasistypes must be what the block was created with (physicsInterface.cpp:537-610)
  auto expect = [&](std::initializer_list<int> types) {
    bool ok = types.size() == vars_.size();
    size_t k = 0;
    for (int t : types) { if (ok && vars_[k].type != t) ok = false; ++k; }
    return ok;
  };
  if (physics_id == MHA_PHYSICS_THERMAL)
    MHA_REQUIRE(expect({MHA_BASIS_HGRAD}), MHA_ERR_INVALID, "thermal needs one HGRAD variable (e)");
  else if (physics_id == MHA_PHYSICS_POROUS_MIXED)
    MHA_REQUIRE(expect({MHA_BASIS_HVOL, MHA_BASIS_HDIV}), MHA_ERR_INVALID,
                "porousMixed needs the variables p (HVOL) and u (HDIV), in that order");
  else if (physics_id == MHA_PHYSICS_NAVIERSTOKES)
    MHA_REQUIRE(dim_ == 2 ? expect({MHA_BASIS_HGRAD, MHA_BASIS_HGRAD, MHA_BASIS_HGRAD})
                          : expect({MHA_BASIS_HGRAD, MHA_BASIS_HGRAD, MHA_BASIS_HGRAD, MHA_BASIS_HGRAD}),
                MHA_ERR_INVALID, "navierstokes needs the HGRAD variables ux, pr, uy[, uz], in that order");
  else if (physics_id == MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED)
    MHA_REQUIRE(dim_ == 2 && expect({MHA_BASIS_HGRAD, MHA_BASIS_HGRAD, MHA_BASIS_HGRAD}), MHA_ERR_INVALID,
                "shallowwaterHybridized needs the HGRAD variables H, Hux, Huy in 2-D");
  physics_id_ = physics_id;
  physics_ = import_physics(physics_id);
  physics_->defineFunctions(functions_);
  physics_->setWorkset(&wkset_);
  // names of the solution fields a deck string may read (Workset::getSolutionField, workset.cpp:314-379) -> slots of
  // the point engine's field arrays (physics_points.hpp): HGRAD [value, d/dx, d/dy(, d/dz)], HVOL [value],
  // HDIV [v_x, v_y(, v_z), div]; `_t` names index the time-derivative array at the value slots
  std::map<std::string, int> fields, fields_t;
  const char *comp[3] = {"[x]", "[y]", "[z]"};
  int slot = 0;
  for (size_t v = 0; v < vars_.size() && v < physics_->myvars.size(); ++v) {
    const std::string &nm = physics_->myvars[v];
    if (vars_[v].type == MHA_BASIS_HGRAD) {
      fields[nm] = slot;
      fields_t[nm + "_t"] = slot;
      for (int d = 0; d < dim_; ++d) fields["grad(" + nm + ")" + comp[d]] = slot + 1 + d;
      slot += 1 + dim_;
    } else if (vars_[v].type == MHA_BASIS_HVOL) {
      fields[nm] = slot;
      fields_t[nm + "_t"] = slot;
      slot += 1;
    } else {
      for (int d = 0; d < dim_; ++d) { fields[nm + comp[d]] = slot + d; fields_t[nm + "_t" + comp[d]] = slot + d; }
      fields["div(" + nm + ")"] = slot + dim_;
      slot += dim_ + 1;
    }
  }
  functions_.setFieldSlots(fields, fields_t);
}

void AssemblyManager::setFunction(const std::string &name, int kind, double amp, const double *freq3,
                                  const double *ip_dev) {
  MHA_REQUIRE(kind == MHA_FUNC_CONSTANT || kind == MHA_FUNC_IP_ARRAY || kind == MHA_FUNC_SINPROD, MHA_ERR_INVALID,
              "unknown function kind " << kind);
  MHA_REQUIRE(kind != MHA_FUNC_IP_ARRAY || ip_dev, MHA_ERR_INVALID, "MHA_FUNC_IP_ARRAY needs a device array");
  MHA_REQUIRE(kind != MHA_FUNC_SINPROD || freq3, MHA_ERR_INVALID, "MHA_FUNC_SINPROD needs freq[3]");
  FuncDesc f;
  f.kind = kind;
  f.amp = amp;
  if (freq3) for (int d = 0; d < 3; ++d) f.freq[d] = freq3[d];
  f.ip = ip_dev;
  functions_.addFunction(name, f);
}

// Butcher / BDF data consumed by the seeding (reference: src/tools/workset.cpp:571-598)
void AssemblyManager::setTimeIntegration(int transient, int nsteps, int nstages, int stage, double dt,
                                         const double *A, const double *b, const double *bdf) {
  TimeDev t;
  if (transient) {
    MHA_REQUIRE(nsteps >= 1 && nsteps <= kMaxSteps && nstages >= 1 && nstages <= kMaxStages, MHA_ERR_INVALID,
                "num_steps/num_stages out of range");
    MHA_REQUIRE(stage >= 0 && stage < nstages && A && b && bdf && dt > 0.0, MHA_ERR_INVALID,
                "bad time-integration tables");
    t.transient = 1;
    t.nsteps = nsteps;
    t.nstages = nstages;
    t.stage = stage;
    t.alpha_u = A[stage * nstages + stage] / b[stage];
    t.timewt = 1.0 / dt / b[stage];
    t.dt = dt;
    t.alpha_t = bdf[0] * t.timewt;
    for (int s = 0; s < stage; ++s) t.stage_ratio[s] = A[stage * nstages + s] / b[s];
    for (int s = 0; s <= nsteps; ++s) t.bdf[s] = bdf[s];
  }
  time_ = t;
  wkset_.isTransient = transient != 0;
  wkset_.deltat = dt;
  wkset_.current_stage = stage;
}

void AssemblyManager::requireReady(bool need_graph) const {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  MHA_REQUIRE(physics_ != nullptr, MHA_ERR_STATE, "no physics module: call mha_physics_select first");
  MHA_REQUIRE(!need_graph || has_graph_, MHA_ERR_STATE, "no CRS graph: call mha_set_graph first");
}

BlockDev AssemblyManager::blockDev() const {
  BlockDev b;
  b.dim = dim_;
  b.nelem = nelem_;
  b.nrows = nrows_;
  b.n = n_;
  b.nq = nq_;
  b.nnodes = nnodes_;
  b.e_begin = 0;
  b.e_count = nelem_;
  b.nodes = d_nodes_.data();
  b.lids = d_lids_.data();
  b.offsets = d_offsets_.data();
  b.fixed = has_fixed_ ? d_fixed_.data() : nullptr;
  b.rowptr = d_rowptr_.data();
  b.colind = d_colind_.data();
  b.ref_basis = d_ref_basis_.data();
  b.ref_grad = d_ref_grad_.data();
  b.ref_wts = d_ref_wts_.data();
  b.nodeval = d_nodeval_.data();
  b.nodegrad = d_nodegrad_.data();
  return b;
}

void AssemblyManager::bindState(const double *u, const double *u_prev, const double *u_stage) {
  MHA_REQUIRE(u != nullptr, MHA_ERR_INVALID, "solution vector is null");
  MHA_REQUIRE(!time_.transient || (u_prev && u_stage), MHA_ERR_INVALID,
              "transient assembly needs u_prev and u_stage");
  time_.u = u;
  time_.u_prev = u_prev;
  time_.u_stage = u_stage;
  wkset_.dev = blockDev();
  wkset_.time_dev = time_;
  wkset_.stream = stream_;
}

// element-major slot map: position of (row LIDs[e][si], col LIDs[e][sj]) inside the CRS row -- replaces the
// column search of sumIntoValues (assemblyManager.cpp:4138) in the general-element kernel
void AssemblyManager::prepareElemSlots() {
  if (has_elem_slot_) return;
  int max_row = 0;
  for (int r = 0; r < nrows_; ++r) max_row = std::max(max_row, h_rowptr_[r + 1] - h_rowptr_[r]);
  MHA_REQUIRE(max_row <= 65536, MHA_ERR_INVALID, "CRS rows longer than 65536 entries are not supported");
  elem_slot_bytes_ = max_row <= 256 ? 1 : 2;
  d_elem_slot_.resize(static_cast<size_t>(nelem_) * n_ * n_ * elem_slot_bytes_);
  launch_build_elem_slot_map(blockDev(), d_elem_slot_.data(), elem_slot_bytes_, stream_);
  has_elem_slot_ = true;
}

void AssemblyManager::useGeneralKernel(bool need_slots) {
  const char *off = std::getenv("MHA_BASELINE_ELEMENT_KERNEL");  // cross-check knob: force the baseline kernel
  wkset_.use_general = !(off && off[0] == '1') && thermal_row_owner_supported(dim_, order_, ref_.nq1);
  if (!wkset_.use_general) return;
  if (need_slots) prepareElemSlots();
  wkset_.tables = AffineDev();
  wkset_.tables.phi1d = d_phi1d_.data();
  wkset_.tables.dphi1d = d_dphi1d_.data();
  wkset_.tables.gw1d = d_gw1d_.data();
  wkset_.tables.gp1d = d_gp1d_.data();
  wkset_.elem_slot = need_slots ? d_elem_slot_.data() : nullptr;
  wkset_.elem_slot_bytes = elem_slot_bytes_;
}

void AssemblyManager::timedBegin() {
  if (timing_) MHA_HIP(hipEventRecord(ev0_, stream_));
}

void AssemblyManager::timedEnd() {
  if (!timing_) return;
  MHA_HIP(hipEventRecord(ev1_, stream_));
  MHA_HIP(hipEventSynchronize(ev1_));
  float ms = 0.f;
  MHA_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
  last_ms_ = ms;
}

// reference: AssemblyManager::assembleJacRes<EvalT> volume loop (assemblyManager.cpp:2357-2509) and
// assembleRes (:2946-3151) when compute_jacobian == 0.
void AssemblyManager::assembleJacRes(int flags, int path, const double *u, const double *u_prev,
                                     const double *u_stage, double *res, double *crs_vals) {
  const int compute_jacobian = (flags & MHA_ASSEMBLE_JACOBIAN) ? 1 : 0;
  const bool overwrite = (flags & MHA_ASSEMBLE_OVERWRITE) != 0;
  requireReady(true);
  MHA_REQUIRE(res != nullptr, MHA_ERR_INVALID, "residual vector is null");
  MHA_REQUIRE(!compute_jacobian || crs_vals, MHA_ERR_INVALID, "compute_jacobian set but crs_vals is null");
  bindState(u, u_prev, u_stage);
  if (engineOnly()) {
    // multi-variable modules (and thermal with its advection term) run on the point engine
    if (path == MHA_PATH_AUTO) path = MHA_PATH_ROW_GATHER;
    if (path == MHA_PATH_ELEMENT_ATOMIC) path = MHA_PATH_POINT_ENGINE;
    MHA_REQUIRE(path == MHA_PATH_POINT_ENGINE || path == MHA_PATH_LOCAL_THEN_SCATTER || path == MHA_PATH_ROW_GATHER,
                MHA_ERR_INVALID,
                "assembly path " << path << " is not available for this physics module");
  }
  const bool adjoint = (flags & MHA_ASSEMBLE_ADJOINT) != 0, lump_mass = (flags & MHA_ASSEMBLE_LUMP_MASS) != 0;
  if (adjoint || lump_mass) {  // scatter options: element matrices + row gather
    MHA_REQUIRE(path == MHA_PATH_AUTO || path == MHA_PATH_ROW_GATHER, MHA_ERR_INVALID,
                "the adjoint / lumped-mass scatter options need MHA_PATH_AUTO or MHA_PATH_ROW_GATHER");
    path = MHA_PATH_ROW_GATHER;
  }
  if (path == MHA_PATH_AUTO) {
    if (!ro_.ready && !ro_.failed && thermal_row_owner_supported(dim_, order_, ref_.nq1)) {
      try {
        prepareRowOwner();
      } catch (const Error &) {  // e.g. a row exceeds the row-block caps: AUTO keeps the general path, once and for all
        ro_ = RowOwnerData();
        ro_.failed = true;
      }
    }
    // affine elements with constant coefficients: the fused affine row-owner kernels; general elements / variable
    // coefficients: the general row-owner kernel; what neither covers: dense element matrices + the row gather
    static const bool no_general = [] { const char *m = std::getenv("MHA_GENERAL"); return m && m[0] == 'g'; }();  // "gather"
    if (rowOwnerUsable(nullptr)) {
      path = MHA_PATH_ROW_OWNER;
    } else {
      if (!no_general) prepareGeneralRowOwner();
      path = (!no_general && gro_.usable) ? MHA_PATH_ROW_OWNER : MHA_PATH_ROW_GATHER;
    }
  }
  int row_owner_kind = 0;  // 1: affine pair of kernels, 2: general-element kernel
  if (path == MHA_PATH_ROW_OWNER) {
    std::string why;
    if (!ro_.ready && !ro_.failed && thermal_row_owner_supported(dim_, order_, ref_.nq1)) {
      try {
        prepareRowOwner();
      } catch (const Error &) {
        ro_ = RowOwnerData();
        ro_.failed = true;
      }
    }
    if (rowOwnerUsable(&why)) {
      row_owner_kind = 1;
    } else {
      prepareGeneralRowOwner();
      MHA_REQUIRE(gro_.usable, MHA_ERR_INVALID,
                  "row-owner path not available: affine kernels: " << why << "; general kernel: " << gro_.why);
      row_owner_kind = 2;
    }
  }
  const bool ro_all_rows = row_owner_kind == 2 ? gro_.all_rows_covered : ro_.all_rows_covered;
  const bool deterministic = (flags & MHA_ASSEMBLE_DETERMINISTIC) != 0;
  MHA_REQUIRE(!deterministic || row_owner_kind == 1, MHA_ERR_INVALID,
              "MHA_ASSEMBLE_DETERMINISTIC is available on the affine row-owner path (thermal, affine elements, constant "
              "coefficients; MHA_PATH_AUTO or MHA_PATH_ROW_OWNER)");
  // porousMixed behind MHA_PATH_ROW_GATHER: the direct form (element threads store into the CRS, a finishing pass sums
  // residual and diagonal parts) whenever the mesh allows it and no scatter option asks for the dense matrices
  const bool porous_direct = path == MHA_PATH_ROW_GATHER && engineOnly() && !adjoint && !lump_mass && porousDirectUsable();
  if (path == MHA_PATH_ROW_GATHER) prepareRowGather(compute_jacobian != 0, !porous_direct);
  timedBegin();
  if (overwrite && path != MHA_PATH_ROW_GATHER && !(path == MHA_PATH_ROW_OWNER && ro_all_rows)) {
    // the accumulate-only kernels get the fused zeroing as an explicit memset on the same stream
    MHA_HIP(hipMemsetAsync(res, 0, sizeof(double) * nrows_, stream_));
    if (compute_jacobian) MHA_HIP(hipMemsetAsync(crs_vals, 0, sizeof(double) * h_rowptr_[nrows_], stream_));
  }
  switch (path) {
    case MHA_PATH_ROW_OWNER:
      if (row_owner_kind == 2) launchGeneralRowOwner(compute_jacobian != 0, overwrite && ro_all_rows, res, crs_vals);
      else launchRowOwner(compute_jacobian != 0, overwrite && ro_all_rows, res, crs_vals, deterministic);
      break;
    case MHA_PATH_ELEMENT_ATOMIC: {
      // one launch over the whole block: the worksets of the reference are an execution detail
      // (sequential reuse of one Workset, assemblyManager.cpp:2355-2357) that does not change results
      wkset_.first_elem = 0;
      wkset_.numElem = nelem_;
      wkset_.res = ElemOut();
      wkset_.res.compute_jacobian = compute_jacobian ? 1 : 0;
      wkset_.res.res = res;
      wkset_.res.crs_vals = compute_jacobian ? crs_vals : nullptr;
      useGeneralKernel(compute_jacobian != 0);
      physics_->volumeResidual();
      break;
    }
    case MHA_PATH_ROW_GATHER: {
      // dense element matrices from the element kernel (stored, not accumulated), then one wavefront per CRS row
      ElemOut o;
      bool dof_order = false;
      o.compute_jacobian = compute_jacobian ? 1 : 0;
      o.local_store = 1;
      o.local_J = compute_jacobian ? d_gather_J_.data() : nullptr;
      o.local_res = d_gather_res_.data();
      // thermal: the element matrices come from the specialised general-element kernel (kernels/thermal_general.hip;
      // perturbed config 2: 4.1 ms against 7.7 ms with the point engine); MHA_ROW_GATHER_KERNEL=engine forces the engine
      static const bool use_general = [] { const char *m = std::getenv("MHA_ROW_GATHER_KERNEL"); return !(m && m[0] == 'e'); }();
      if (!engineOnly() && use_general && thermal_row_owner_supported(dim_, order_, ref_.nq1)) {
        wkset_.first_elem = 0;
        wkset_.numElem = nelem_;
        wkset_.res = o;
        useGeneralKernel(false);
        if (!wkset_.use_general) {
          // MHA_BASELINE_ELEMENT_KERNEL=1 (cross-check knob): the baseline kernel ACCUMULATES into local_J / local_res
          // (updateJac / updateRes convention) and ignores local_store, so the scratch must start from zero
          if (compute_jacobian) MHA_HIP(hipMemsetAsync(d_gather_J_.data(), 0, sizeof(double) * d_gather_J_.size(), stream_));
          MHA_HIP(hipMemsetAsync(d_gather_res_.data(), 0, sizeof(double) * d_gather_res_.size(), stream_));
        }
        physics_->volumeResidual();
      } else {
        // porousMixed: its thread-per-element kernel writes the element arrays in dof order, straight from registers
        // (kernels/porous_element.hip); MHA_GATHER_ORDER=pos keeps the LID-position order and the LDS staging
        static const bool by_pos = [] {
          const char *m = std::getenv("MHA_GATHER_ORDER"), *k = std::getenv("MHA_POROUS_KERNEL");
          return (m && m[0] == 'p') || (k && k[0] == 'e');  // the point engine writes LID-position order only
        }();
        // (the same predicate launch_row_gather has for dof-ordered arrays -- short rows, one-byte slots: a porousMixed
        // block with a wider caller graph or 16-bit slots keeps the position order instead of failing)
        dof_order = !by_pos && !adjoint && !lump_mass && physics_->label == "porousMixed" && max_row_ <= 32 && n_ <= 16 &&
                    elem_slot_bytes_ == 1;
        if (porous_direct && elem_slot_bytes_ == 1) {
          d_direct_part_.resize(static_cast<size_t>(nrows_) * 4);
          o.local_J = nullptr;
          o.local_res = nullptr;
          o.direct_part = d_direct_part_.data();
          o.direct_slot = static_cast<const uint8_t *>(d_elem_slot_.data());
          o.direct_side = d_direct_side_.data();
          o.direct_vals = compute_jacobian ? crs_vals : nullptr;
          static const bool porous_nt = [] { const char *m = std::getenv("MHA_POROUS_NT"); return m && m[0] == '1'; }();  // experiment: nontemporal entry stores
          o.direct_overwrite = overwrite ? (porous_nt ? 2 : 1) : 0;
          // database mode: overwriting assemblies of a uniform block with constant permeability / mobility
          bool pdb = overwrite && compute_jacobian && (reinterpret_cast<uintptr_t>(crs_vals) & 127u) == 0;
          for (const char *name : {"Kinv_xx", "Kinv_yy", "Kinv_zz", "total_mobility"})
            pdb = pdb && functions_.has(name) && functions_.evaluate(name).kind == MHA_FUNC_CONSTANT;
          pdb = pdb && porousDatabaseUsable();
          static const bool two_kernels = [] { const char *m = std::getenv("MHA_POROUS_DB_LEAN"); return !(m && m[0] == '0'); }();
          if (pdb && two_kernels && functions_.evaluate("source").kind != MHA_FUNC_EXPRESSION) {
            // database mode: the lean build (residual parts only) over all elements, then the full build over the few
            // elements incident to computed rows (same records, rewritten with the diagonal parts; their entries)
            o.direct_axis_aligned = porous_db_.axis_aligned ? 1 : 0;
            ElemOut lean = o;
            lean.direct_res_only = 1;
            static const bool matvec = [] { const char *m = std::getenv("MHA_POROUS_DB_MATVEC"); return !(m && m[0] == '0'); }();
            if (matvec) {
              // the residual of a linear module from the element matrix: the dense kernel on element 0 leaves the matrix
              // of the uniform block (dof order) in front of the point tables (kernels/porous_element.hip)
              // (made again only when a coefficient or the time-integration factor has changed)
              const double key[5] = {functions_.evaluate("Kinv_xx").amp, functions_.evaluate("Kinv_yy").amp,
                                     functions_.evaluate("Kinv_zz").amp, functions_.evaluate("total_mobility").amp, wkset_.time_dev.alpha_u};
              if (!porous_db_.uniform_valid || std::memcmp(key, porous_db_.uniform_key, sizeof(key)) != 0) {
                porous_db_.uniform.resize(static_cast<size_t>(n_) * n_ + static_cast<size_t>(nq_) * (dim_ + 1) + n_);
                ElemOut dense;
                dense.compute_jacobian = 1;
                dense.local_store = 1;
                dense.local_dof_order = 1;
                dense.local_J = porous_db_.uniform.data();
                dense.local_res = porous_db_.uniform.data() + static_cast<size_t>(n_) * n_ + static_cast<size_t>(nq_) * (dim_ + 1);
                launchPointEngine(1, dense, 0, 1);
                launch_porous_uniform_points(blockDev(), porous_db_.uniform.data(), stream_);
                std::memcpy(porous_db_.uniform_key, key, sizeof(key));
                porous_db_.uniform_valid = true;
              }
              lean.direct_uniform = porous_db_.uniform.data();
            }
            o.direct_elist = porous_db_.elist.data();
            if (lean.direct_uniform) {
              // the two kernels write the records of disjoint sets of elements: side by side on two streams
              lean.direct_jacflag = porous_db_.jacflag.data();
              if (!side_stream_) {
                MHA_HIP(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
                MHA_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
                MHA_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
              }
              MHA_HIP(hipEventRecord(ev_fork_, stream_));
              MHA_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
              wkset_.stream = side_stream_;
              launchPointEngine(compute_jacobian, o, 0, porous_db_.num_listed);
              wkset_.stream = stream_;
              MHA_HIP(hipEventRecord(ev_join_, side_stream_));
              launchPointEngine(compute_jacobian, lean, 0, nelem_);
              MHA_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
            } else {
              launchPointEngine(compute_jacobian, lean, 0, nelem_);
              launchPointEngine(compute_jacobian, o, 0, porous_db_.num_listed);
            }
            launch_porous_direct_finish(blockDev(), d_inc_ptr_.data(), porous_db_.diag.data(), o.direct_part, res, o.direct_vals, 1, stream_);
            launch_replicate_runs(porous_db_.chunks.data(), porous_db_.num_chunks, crs_vals, stream_);
            last_porous_direct_ = 2;
            break;
          }
          if (pdb) o.direct_jacflag = porous_db_.jacflag.data();
          if (!compute_jacobian) o.direct_res_only = 1;  // residual-only assemblies: the lean build
          launchPointEngine(compute_jacobian, o, 0, nelem_);
          launch_porous_direct_finish(blockDev(), d_inc_ptr_.data(), pdb ? porous_db_.diag.data() : d_direct_diag_.data(), o.direct_part,
                                      res, o.direct_vals, overwrite ? 1 : 0, stream_);
          if (pdb) launch_replicate_runs(porous_db_.chunks.data(), porous_db_.num_chunks, crs_vals, stream_);
          last_porous_direct_ = pdb ? 2 : 1;
          break;
        }
        last_porous_direct_ = 0;
        o.local_dof_order = dof_order ? 1 : 0;
        launchPointEngine(compute_jacobian, o, 0, nelem_);
      }
      RowGatherDev g;
      g.inc_dof = dof_order ? d_inc_dof_.data() : nullptr;
      g.inc_ptr = d_inc_ptr_.data();
      g.inc_elem = d_inc_elem_.data();
      g.inc_pos = d_inc_pos_.data();
      g.slot = d_elem_slot_.data();
      g.slot_bytes = elem_slot_bytes_;
      g.max_row = max_row_;
      g.adjoint = adjoint ? 1 : 0;
      g.lump_mass = lump_mass ? 1 : 0;
      launch_row_gather(blockDev(), g, o.local_J, o.local_res, res, compute_jacobian ? crs_vals : nullptr,
                        overwrite ? 1 : 0, stream_);
      break;
    }
    case MHA_PATH_POINT_ENGINE: {
      ElemOut o;
      o.compute_jacobian = compute_jacobian ? 1 : 0;
      o.res = res;
      o.crs_vals = compute_jacobian ? crs_vals : nullptr;
      launchPointEngine(compute_jacobian, o, 0, nelem_);
      break;
    }
    case MHA_PATH_LOCAL_THEN_SCATTER: {
      // the reference's two-step path, workset by workset (assemblyManager.cpp:2442-2509)
      const int ws = wkset_.maxElem;
      d_local_J_.resize(static_cast<size_t>(ws) * n_ * n_);
      d_local_res_.resize(static_cast<size_t>(ws) * n_);
      for (int e0 = 0; e0 < nelem_; e0 += ws) {
        const int ne = std::min(ws, nelem_ - e0);
        MHA_HIP(hipMemsetAsync(d_local_J_.data(), 0, sizeof(double) * ne * n_ * n_, stream_));
        MHA_HIP(hipMemsetAsync(d_local_res_.data(), 0, sizeof(double) * ne * n_, stream_));
        wkset_.first_elem = e0;
        wkset_.numElem = ne;
        wkset_.res = ElemOut();
        wkset_.res.compute_jacobian = compute_jacobian ? 1 : 0;
        // dense arrays are indexed by the element's position inside the workset
        wkset_.res.local_base = e0;
        wkset_.res.local_J = d_local_J_.data();
        wkset_.res.local_res = d_local_res_.data();
        wkset_.use_general = false;  // this path keeps the baseline element kernel: an independent implementation
        if (engineOnly()) launchPointEngine(compute_jacobian, wkset_.res, e0, ne);
        else physics_->volumeResidual();
        BlockDev b = blockDev();
        b.e_begin = e0;
        b.e_count = ne;
        launch_scatter_local(b, compute_jacobian ? wkset_.res.local_J : nullptr, wkset_.res.local_res, res,
                             compute_jacobian ? crs_vals : nullptr, e0, stream_);
      }
      break;
    }
    default:
      MHA_REQUIRE(false, MHA_ERR_INVALID, "assembly path " << path << " is not available");
  }
  timedEnd();
  last_path_ = path;
  last_row_owner_kind_ = row_owner_kind;
}

// reference: updateJac / updateRes on the whole block (assemblyManager.cpp:7412-7455, 7115-7152)
void AssemblyManager::computeLocalJacRes(int compute_jacobian, const double *u, const double *u_prev,
                                         const double *u_stage, double *local_J, double *local_res) {
  requireReady(false);
  MHA_REQUIRE(local_res != nullptr, MHA_ERR_INVALID, "local_res is null");
  MHA_REQUIRE(!compute_jacobian || local_J, MHA_ERR_INVALID, "compute_jacobian set but local_J is null");
  bindState(u, u_prev, u_stage);
  wkset_.first_elem = 0;
  wkset_.numElem = nelem_;
  wkset_.res = ElemOut();
  wkset_.res.compute_jacobian = compute_jacobian ? 1 : 0;
  wkset_.res.local_J = compute_jacobian ? local_J : nullptr;
  wkset_.res.local_res = local_res;
  timedBegin();
  if (engineOnly()) {
    launchPointEngine(compute_jacobian, wkset_.res, 0, nelem_);
  } else {
    useGeneralKernel(false);
    physics_->volumeResidual();
  }
  timedEnd();
}

// reference: identifyVolumetricDatabase (assemblyManager.cpp:4314-4467) with exact matching: key = the vertex offsets
// from the element's first vertex (bit patterns) + the orientation signs; representatives in order of first appearance
int AssemblyManager::databaseBuild() {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  std::vector<double> nodes(static_cast<size_t>(nelem_) * nnodes_ * dim_);
  d_nodes_.download(nodes.data());
  const size_t kd = static_cast<size_t>(nnodes_ - 1) * dim_, ko = has_orient_ ? static_cast<size_t>(n_) : 0;
  const size_t kb = kd * sizeof(double) + ko;
  std::unordered_map<std::string, int32_t> seen;
  seen.reserve(1024);
  db_index_.assign(nelem_, 0);
  db_first_users_.clear();
  std::string key(kb, '\0');
  std::vector<double> rel(kd);
  for (int e = 0; e < nelem_; ++e) {
    const double *xn = nodes.data() + static_cast<size_t>(e) * nnodes_ * dim_;
    for (int v = 1; v < nnodes_; ++v)
      for (int d = 0; d < dim_; ++d) {
        double r = xn[v * dim_ + d] - xn[d];
        if (r == 0.0) r = 0.0;  // -0.0 and +0.0 are the same offset
        rel[static_cast<size_t>(v - 1) * dim_ + d] = r;
      }
    std::memcpy(&key[0], rel.data(), kd * sizeof(double));
    if (ko) std::memcpy(&key[kd * sizeof(double)], h_orient_.data() + static_cast<size_t>(e) * n_, ko);
    auto it = seen.find(key);
    if (it == seen.end()) {
      it = seen.emplace(key, static_cast<int32_t>(db_first_users_.size())).first;
      db_first_users_.push_back(e);
    }
    db_index_[e] = it->second;
  }
  d_db_index_.upload(db_index_);
  return static_cast<int>(db_first_users_.size());
}

void AssemblyManager::databaseGet(int32_t *index, int32_t *first_users) const {
  MHA_REQUIRE(!db_index_.empty(), MHA_ERR_STATE, "no database: call mha_database_build first");
  if (index) std::copy(db_index_.begin(), db_index_.end(), index);
  if (first_users) std::copy(db_first_users_.begin(), db_first_users_.end(), first_users);
}

// reference: AssemblyManager::applyMassMatrixFree (assemblyManager.cpp:1582-1778)
void AssemblyManager::applyMassMatrixFree(int mode, const double *masswts, const double *mass, int maxent,
                                          const int32_t *nnz_row, const double *values, const int32_t *columns,
                                          const double *x, double *y) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  MHA_REQUIRE(x && y, MHA_ERR_INVALID, "null vector");
  BlockDev b = blockDev();
  b.e_begin = 0;
  b.e_count = nelem_;
  VarLayoutDev vl = layout_;
  vl.orient = has_orient_ ? d_orient_.data() : nullptr;
  if (mode == MHA_MASS_ON_THE_FLY) {
    launch_mass_apply_free(b, vl, masswts, x, y, stream_);
    return;
  }
  const bool db = mode == MHA_MASS_DATABASE || mode == MHA_MASS_DATABASE_SPARSE;
  MHA_REQUIRE(mode == MHA_MASS_LOCAL || db, MHA_ERR_INVALID, "unknown mass mode " << mode);
  MHA_REQUIRE(!db || !db_index_.empty(), MHA_ERR_STATE, "database mass needs mha_database_build first");
  if (mode == MHA_MASS_DATABASE_SPARSE) {
    MHA_REQUIRE(values && columns && nnz_row, MHA_ERR_INVALID, "sparse mass format needs a Sparse3DView");
    if (d_pos_var_.size() != static_cast<size_t>(n_)) {  // variable of every LID position (Sparse3DView::setLocalColumns)
      std::vector<int32_t> offs(n_), pv(n_, 0);
      d_offsets_.download(offs.data());
      for (int v = 0; v < layout_.nvars; ++v)
        for (int j = layout_.varptr[v]; j < layout_.varptr[v + 1]; ++j) pv[offs[j]] = v;
      d_pos_var_.upload(pv);
    }
  } else {
    MHA_REQUIRE(mass != nullptr, MHA_ERR_INVALID, "stored mass is null");
  }
  launch_mass_apply_stored(b, vl, db ? d_db_index_.data() : nullptr, mode == MHA_MASS_DATABASE_SPARSE ? nullptr : mass,
                           maxent, nnz_row, values, columns, d_pos_var_.data(), x, y, stream_);
}

// reference: AssemblyManager::getMass / getWeightedMass (assemblyManager.cpp:7776-7925): dense element mass matrices,
// accumulated (+=) into local_mass[E][n][n] in LID-position order.  Runs the point engine with the value slots as
// the "flux" (kernels/point_engine.hip, mass mode): the mass matrix is the B^T C B product with C = weights.
void AssemblyManager::getMass(const double *masswts, double *local_mass) {
  requireReady(false);
  MHA_REQUIRE(local_mass != nullptr, MHA_ERR_INVALID, "local_mass is null");
  DeviceBuffer<double> zero_u(static_cast<size_t>(nrows_));
  MHA_HIP(hipMemsetAsync(zero_u.data(), 0, sizeof(double) * nrows_, stream_));
  TimeDev steady;
  steady.u = zero_u.data();
  BlockDev b = blockDev();
  VarLayoutDev vl = layout_;
  vl.orient = has_orient_ ? d_orient_.data() : nullptr;
  PhysParamsDev pp;
  pp.physics = -physics_id_;  // negative id = mass mode on the module's variable layout
  for (size_t v = 0; v < vars_.size(); ++v) pp.p[v] = masswts ? masswts[v] : 1.0;
  ElemOut o;
  o.compute_jacobian = 1;
  o.local_J = local_mass;
  timedBegin();
  launch_point_engine(b, vl, pp, steady, o, nullptr, 1, stream_);
  timedEnd();
  MHA_HIP(hipStreamSynchronize(stream_));  // zero_u is released on return
}

// HDG element of shallowwaterHybridized, side part (kernels/swhdg_element.hip): residual and derivative blocks of the
// interior + trace unknowns; settings g / stabilisation from the module
void AssemblyManager::swhdgElementBlocks(const double *u, const double *u_prev, const double *u_stage, const double *lambda,
                                         const uint8_t *side_types, const double *farfield, double *res, double *blocks) {
  requireReady(false);
  shallowwaterHybridized *sw = dynamic_cast<shallowwaterHybridized *>(physics_.get());
  MHA_REQUIRE(sw != nullptr, MHA_ERR_INVALID, "the block's physics module is not shallowwaterHybridized");
  MHA_REQUIRE(lambda && (res || blocks), MHA_ERR_INVALID, "null trace values or outputs");
  for (const auto &vi : vars_) MHA_REQUIRE(vi.order == 1, MHA_ERR_INVALID, "the HDG element is built for order-1 variables");
  bindState(u, u_prev, u_stage);
  prepareSideTables();
  SwhElementDev a;
  a.lambda = lambda;
  a.side_types = side_types;
  if (farfield) for (int i = 0; i < 3; ++i) a.farfield[i] = farfield[i];
  a.g = sw->gravity;
  a.roe = sw->roestab ? 1 : 0;
  a.res = res;
  a.blocks = blocks;
  timedBegin();
  launch_swhdg_element(blockDev(), sideTablesDev(), a, time_, stream_);
  timedEnd();
}

// The fused element step (kernels/swhdg_fused.hip): side + volume assembly + static condensation, nothing but S, g, du
// (and the loop state) leaves the chip.  Deck-string sources are the one thing it does not take.
bool AssemblyManager::swhdgFusedUsable() const {
  const shallowwaterHybridized *sw = dynamic_cast<const shallowwaterHybridized *>(physics_.get());
  if (!sw || std::getenv("MHA_SUBGRID_UNFUSED")) return false;
  for (const char *k : {"source H", "source Hux", "source Huy"})
    if (functions_.has(k) && functions_.evaluate(k).kind == MHA_FUNC_EXPRESSION) return false;
  return dim_ == 2 && n_ == 12;
}

void AssemblyManager::swhdgCondensedElement(const double *u, const double *u_prev, const double *u_stage, const double *lambda,
                                            const uint8_t *side_types, const double *farfield, SwhFusedOut o) {
  requireReady(false);
  shallowwaterHybridized *sw = dynamic_cast<shallowwaterHybridized *>(physics_.get());
  MHA_REQUIRE(sw != nullptr, MHA_ERR_INVALID, "the block's physics module is not shallowwaterHybridized");
  MHA_REQUIRE(lambda != nullptr, MHA_ERR_INVALID, "null trace values");
  for (const auto &vi : vars_) MHA_REQUIRE(vi.order == 1, MHA_ERR_INVALID, "the HDG element is built for order-1 variables");
  bindState(u, u_prev, u_stage);
  prepareSideTables();
  SwhElementDev a;
  a.lambda = lambda;
  a.side_types = side_types;
  if (farfield) for (int i = 0; i < 3; ++i) a.farfield[i] = farfield[i];
  a.g = sw->gravity;
  a.roe = sw->roestab ? 1 : 0;
  PhysParamsDev pp;
  pp.physics = MHA_PHYSICS_SHALLOWWATER_HYBRIDIZED;
  const char *names[3] = {"source H", "source Hux", "source Huy"};
  for (int k = 0; k < 3; ++k) pp.f[k] = functions_.evaluate(names[k]);
  pp.p[0] = sw->gravity;
  timedBegin();
  launch_swhdg_fused(blockDev(), sideTablesDev(), a, time_, pp, o, stream_);
  timedEnd();
}

// Workspace of subgridSolve (doubles unless noted): blocks [E][36][36], res [E][36], local_J [E][12][12], local_res
// [E][12], du [E][12], rn0 [E], then int32 active [E].
size_t AssemblyManager::subgridWorkspaceBytes() const {
  const size_t E = static_cast<size_t>(nelem_), ni = static_cast<size_t>(n_), n = ni + 24;
  return sizeof(double) * E * (n * n + n + ni * ni + ni + ni + 1) + sizeof(int32_t) * E + 64;
}

// The sub-iteration loop of the subgrid solver on the device: max_iter passes of
//   side blocks (mha_swhdg_element_blocks) + volume block (mha_compute_local_jacres) -> combine + loop bookkeeping ->
//   element-local solve (the condensation kernel's du = A_uu^-1 r_u) -> sol += du for the elements still iterating,
// then one closing assembly + condensation at the final state for the Schur complement / condensed right-hand side the
// macro trace system takes (updateFlux's sensitivities, :1542-1616).  Everything is enqueued on the context's stream:
// no host synchronisation, no allocation (the caller provides the workspace).  An element leaves its loop when its
// scaled residual norm drops to tol, exactly as the reference's while condition; the passes it no longer needs are
// still computed for it and ignored (a uniform schedule is what keeps the host out of the loop).
void AssemblyManager::subgridSolve(double *u, const double *u_prev, const double *u_stage, const double *lambda,
                                   const uint8_t *side_types, const double *farfield, int max_iter, double tol,
                                   void *workspace, size_t workspace_bytes, double *schur, double *gvec, int32_t *iters,
                                   double *resnorm_scaled, int32_t *num_singular) {
  requireReady(false);
  MHA_REQUIRE(dynamic_cast<shallowwaterHybridized *>(physics_.get()) != nullptr, MHA_ERR_INVALID,
              "the subgrid driver is built for shallowwaterHybridized blocks");
  MHA_REQUIRE(u && lambda && workspace && iters && resnorm_scaled && num_singular, MHA_ERR_INVALID, "null argument");
  MHA_REQUIRE(max_iter >= 1 && tol >= 0.0, MHA_ERR_INVALID, "bad iteration limits");
  MHA_REQUIRE(workspace_bytes >= subgridWorkspaceBytes(), MHA_ERR_INVALID,
              "workspace too small: " << workspace_bytes << " < " << subgridWorkspaceBytes());
  if (!subgrid_checked_) {  // interior unknowns must be element-local (discontinuous): checked once per mesh
    std::vector<char> seen(nrows_, 0);
    for (size_t k = 0; k < h_lids_.size(); ++k) {
      MHA_REQUIRE(!seen[h_lids_[k]], MHA_ERR_INVALID,
                  "the subgrid driver needs element-local interior unknowns: row " << h_lids_[k] << " belongs to two elements");
      seen[h_lids_[k]] = 1;
    }
    subgrid_checked_ = true;
  }
  const size_t E = static_cast<size_t>(nelem_), ni = static_cast<size_t>(n_), n = ni + 24;
  double *blocks = static_cast<double *>(workspace);
  double *res = blocks + E * n * n, *lJ = res + E * n, *lr = lJ + E * ni * ni, *du = lr + E * ni, *rn0 = du + E * ni;
  int32_t *active = reinterpret_cast<int32_t *>(rn0 + E);
  MHA_HIP(hipMemsetAsync(num_singular, 0, sizeof(int32_t), stream_));
  if (swhdgFusedUsable()) {
    // one kernel per pass: assembly, bookkeeping, element-local solve and sol += du fused (kernels/swhdg_fused.hip); the
    // [36 x 36] blocks never reach memory
    SwhFusedOut o;
    o.singular = num_singular;
    o.tol = tol;
    o.rn0 = rn0;
    o.scaled = resnorm_scaled;
    o.iters = iters;
    o.active = active;
    for (int pass = 0; pass < max_iter; ++pass) {
      o.pass = pass;
      o.update_u = u;
      swhdgCondensedElement(u, u_prev, u_stage, lambda, side_types, farfield, o);
    }
    if (schur || gvec) {
      o.pass = -1;
      o.update_u = nullptr;
      o.schur = schur;
      o.gvec = gvec;
      swhdgCondensedElement(u, u_prev, u_stage, lambda, side_types, farfield, o);
    }
    return;
  }
  auto assemble = [&](int pass) {
    swhdgElementBlocks(u, u_prev, u_stage, lambda, side_types, farfield, res, blocks);
    MHA_HIP(hipMemsetAsync(lJ, 0, sizeof(double) * E * (ni * ni + ni), stream_));  // local_J and local_res are contiguous
    computeLocalJacRes(1, u, u_prev, u_stage, lJ, lr);
    launch_subgrid_combine(nelem_, n_, static_cast<int>(n), d_offsets_.data(), lJ, lr, blocks, res, pass, tol, rn0,
                           resnorm_scaled, iters, active, stream_);
  };
  for (int pass = 0; pass < max_iter; ++pass) {
    assemble(pass);
    launch_condense(n_, 24, nelem_, blocks, res, nullptr, nullptr, du, num_singular, stream_);
    launch_subgrid_update(nelem_, n_, d_lids_.data(), d_offsets_.data(), du, active, u, stream_);
  }
  if (schur || gvec) {
    assemble(-1);
    launch_condense(n_, 24, nelem_, blocks, res, schur, gvec, nullptr, num_singular, stream_);
  }
}

void AssemblyManager::scatterLocal(const double *local_J, const double *local_res, double *res, double *crs_vals) {
  requireReady(true);
  timedBegin();
  launch_scatter_local(blockDev(), local_J, local_res, res, crs_vals, 0, stream_);
  timedEnd();
}

void AssemblyManager::dirichletLift(double *u, const double *vals, double scalar) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  MHA_REQUIRE(u != nullptr, MHA_ERR_INVALID, "solution vector is null");
  if (!has_fixed_) return;
  launch_dirichlet_lift(nrows_, d_fixed_.data(), vals, scalar, u, stream_);
}

void AssemblyManager::applyDbcDiag(double *crs_vals) {
  requireReady(true);
  MHA_REQUIRE(crs_vals != nullptr, MHA_ERR_INVALID, "crs_vals is null");
  launch_dbc_diag(blockDev(), crs_vals, stream_);
}

void AssemblyManager::gather(const double *vec, double *elem_vals) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  MHA_REQUIRE(vec && elem_vals, MHA_ERR_INVALID, "null pointer");
  launch_gather(blockDev(), vec, elem_vals, stream_);
}

int AssemblyManager::numWorksets() const {
  if (!has_mesh_) return 0;
  return (nelem_ + wkset_.maxElem - 1) / wkset_.maxElem;
}

// reference: updateWorkset<EvalT> pointing the workset at group `index` (assemblyManager.cpp:6512-6596)
void AssemblyManager::worksetUpdate(int index) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  MHA_REQUIRE(index >= 0 && index < numWorksets(), MHA_ERR_INVALID, "workset index out of range");
  wkset_.dev = blockDev();
  wkset_.stream = stream_;
  wkset_.first_elem = index * wkset_.maxElem;
  wkset_.numElem = std::min(wkset_.maxElem, nelem_ - wkset_.first_elem);
  wkset_.update_views();
  ws_fields_first_ = -1;
  ws_res_first_ = -1;
  if (!single_hgrad_) {
    // every variable's basis / basis_grad / basis_div on this workset (Group::computeBasis keeps one set per basis)
    ws_var_views_.resize(vars_.size());
    const size_t cap = static_cast<size_t>(std::max(wkset_.maxElem, wkset_.numElem));
    for (size_t v = 0; v < vars_.size(); ++v) {
      const VarInfo &vi = vars_[v];
      VarViews &vv = ws_var_views_[v];
      vv.basis.resize(cap * vi.card * nq_ * varComps(static_cast<int>(v)));
      if (vi.type == MHA_BASIS_HGRAD) vv.grad.resize(cap * vi.card * nq_ * dim_);
      if (vi.type == MHA_BASIS_HDIV) vv.div.resize(cap * vi.card * nq_);
      VarViewsDev out;
      out.basis = vv.basis.data();
      out.grad = vv.grad.data();
      out.div = vv.div.data();
      launch_var_views(blockDev(), varPoints(static_cast<int>(v), false), nullptr, nullptr, wkset_.first_elem,
                       wkset_.numElem, out, stream_);
    }
  }
}

// variable names are the module's myvars (reference: var_list[set][block]); without a module: "var0", "var1", ...
std::string AssemblyManager::varName(int v) const {
  if (physics_ && v < static_cast<int>(physics_->myvars.size())) return physics_->myvars[v];
  return "var" + std::to_string(v);
}

int AssemblyManager::varIndex(const std::string &name) const {
  for (int v = 0; v < static_cast<int>(vars_.size()); ++v)
    if (varName(v) == name) return v;
  return -1;
}

// reference values of variable v at the volume points (one set) or at the side points (one set per local side);
// built on first use, kept for the life of the block
VarPointsDev AssemblyManager::varPoints(int v, bool side) {
  std::vector<VarTables> &tabs = side ? var_side_tables_ : var_vol_tables_;
  if (tabs.size() != vars_.size()) tabs.resize(vars_.size());
  if (side) prepareSideTables();
  const VarInfo &vi = vars_[v];
  const int nsets = side ? side_ref_.nsides : 1, np = side ? side_ref_.nqs : nq_;
  VarTables &t = tabs[v];
  if (!t.ready) {
    std::vector<double> val, grad, div, a, b, c;
    for (int s = 0; s < nsets; ++s) {
      const double *pts = side ? side_ref_.ip.data() + static_cast<size_t>(s) * np * dim_ : ref_.ip.data();
      const int card = ref_basis_var(dim_, vi.type, vi.order, np, pts, a, b, c);
      MHA_REQUIRE(card == vi.card, MHA_ERR_INVALID, "basis cardinality mismatch");
      val.insert(val.end(), a.begin(), a.end());
      grad.insert(grad.end(), b.begin(), b.end());
      div.insert(div.end(), c.begin(), c.end());
    }
    t.val.upload(val);
    t.grad.upload(grad);
    t.div.upload(div);
    t.ready = true;
  }
  VarPointsDev d;
  d.type = vi.type;
  d.card = vi.card;
  d.npts = np;
  d.val = t.val.data();
  d.grad = t.grad.data();
  d.div = t.div.data();
  d.nodegrad = side ? d_side_nodegrad_.data() : d_nodegrad_.data();
  d.orient = has_orient_ ? d_orient_.data() : nullptr;
  d.n_tot = n_;
  d.var_off = layout_.varptr[v];
  return d;
}

void AssemblyManager::worksetVarArrays(int v, const double **basis, const double **grad, const double **div) const {
  if (single_hgrad_) {
    *basis = static_cast<const double *>(wkset_.get("basis").ptr);
    *grad = static_cast<const double *>(wkset_.get("basis_grad").ptr);
    *div = nullptr;
  } else {
    MHA_REQUIRE(ws_var_views_.size() == vars_.size(), MHA_ERR_STATE, "workset views requested before mha_workset_update");
    *basis = ws_var_views_[v].basis.data();
    *grad = ws_var_views_[v].grad.data();
    *div = ws_var_views_[v].div.data();
  }
}

// reference: Workset::getBasis(var) / getBasisGrad(var) / getBasisDiv(var) (workset.hpp:241-275), getResidual (:193),
// getSolutionField (:229).  Names: "basis <var>" (numElem,card,numip,ncomp) "basis_grad <var>" (numElem,card,numip,dim)
// "basis_div <var>" (numElem,card,numip); "res" (numElem,n) "res.dx" (numElem,n,n); field names as the reference
// spells them: "<var>", "grad(<var>)[x]", "<var>_t", "<var>[x]", "<var>_t[x]", "div(<var>)" (numElem,numip).
View AssemblyManager::worksetView(const std::string &name) const {
  View v;
  const int64_t ne = wkset_.numElem;
  auto var_of = [&](const std::string &prefix) -> int {
    if (name.compare(0, prefix.size(), prefix) != 0) return -1;
    return varIndex(name.substr(prefix.size()));
  };
  int k;
  if ((k = var_of("basis_grad ")) >= 0 || (k = var_of("basis_div ")) >= 0 || (k = var_of("basis ")) >= 0) {
    const double *b, *g, *d;
    worksetVarArrays(k, &b, &g, &d);
    const VarInfo &vi = vars_[k];
    v.extent[0] = ne; v.extent[1] = vi.card; v.extent[2] = nq_;
    if (name[5] == ' ') { v.ptr = const_cast<double *>(b); v.rank = 4; v.extent[3] = varComps(k); }
    else if (name[6] == 'g') { v.ptr = const_cast<double *>(g); v.rank = 4; v.extent[3] = dim_; }
    else { v.ptr = const_cast<double *>(d); v.rank = 3; }
    if (!v.ptr) throw Error(MHA_ERR_UNKNOWN_FIELD, "'" + name + "' is not defined for this basis type");
    return v;
  }
  if (name == "res" || name == "res.dx") {
    MHA_REQUIRE(ws_res_first_ == wkset_.first_elem && ws_res_first_ >= 0, MHA_ERR_STATE,
                "'" << name << "' requested before mha_workset_compute_residual on this workset");
    if (name == "res") { v.ptr = ws_res_.data(); v.rank = 2; v.extent[0] = ws_res_num_; v.extent[1] = n_; }
    else {
      MHA_REQUIRE(ws_res_has_dx_, MHA_ERR_STATE, "the residual was computed without its derivative array");
      v.ptr = ws_res_dx_.data(); v.rank = 3; v.extent[0] = ws_res_num_; v.extent[1] = n_; v.extent[2] = n_;
    }
    return v;
  }
  auto it = ws_fields_.find(name);
  if (it != ws_fields_.end()) {
    MHA_REQUIRE(ws_fields_first_ == wkset_.first_elem, MHA_ERR_STATE,
                "solution field '" << name << "' requested before mha_workset_compute_solution on this workset");
    v.ptr = it->second.data(); v.rank = 2; v.extent[0] = ne; v.extent[1] = nq_;
    return v;
  }
  return wkset_.get(name);
}

void AssemblyManager::worksetComputeSolution(const double *u, const double *u_prev, const double *u_stage) {
  requireReady(false);
  MHA_REQUIRE(wkset_.get("LIDs").ptr != nullptr, MHA_ERR_STATE, "mha_workset_compute_solution before mha_workset_update");
  bindState(u, u_prev, u_stage);
  static const char *xyz[3] = {"[x]", "[y]", "[z]"};
  const size_t cnt = static_cast<size_t>(std::max(wkset_.maxElem, wkset_.numElem)) * nq_;
  auto buf = [&](const std::string &nm) {
    DeviceBuffer<double> &b = ws_fields_[nm];
    if (b.size() < cnt) b.resize(cnt);
    return b.data();
  };
  for (int v = 0; v < static_cast<int>(vars_.size()); ++v) {
    const VarInfo &vi = vars_[v];
    const std::string var = varName(v);
    const double *b, *g, *d;
    worksetVarArrays(v, &b, &g, &d);
    VarFieldsDev out;
    if (vi.type == MHA_BASIS_HDIV) {
      for (int c = 0; c < dim_; ++c) { out.val[c] = buf(var + xyz[c]); out.dot[c] = buf(var + "_t" + xyz[c]); }
      out.div = buf("div(" + var + ")");
    } else {
      out.val[0] = buf(var);
      out.dot[0] = buf(var + "_t");
      if (vi.type == MHA_BASIS_HGRAD)
        for (int c = 0; c < dim_; ++c) out.grad[c] = buf("grad(" + var + ")" + xyz[c]);
    }
    launch_var_fields(blockDev(), time_, wkset_.first_elem, wkset_.numElem, vi.card, layout_.varptr[v], nq_, varComps(v),
                      b, vi.type == MHA_BASIS_HGRAD ? g : nullptr, vi.type == MHA_BASIS_HDIV ? d : nullptr, out, stream_);
  }
  ws_fields_first_ = wkset_.first_elem;
}

void AssemblyManager::worksetComputeResidual(int compute_jacobian, const double *u, const double *u_prev,
                                             const double *u_stage) {
  requireReady(false);
  MHA_REQUIRE(physics_ != nullptr, MHA_ERR_STATE, "no physics module: call mha_physics_select first");
  bindState(u, u_prev, u_stage);
  const int e0 = wkset_.first_elem, ne = wkset_.numElem;
  MHA_REQUIRE(ne > 0 && e0 + ne <= nelem_, MHA_ERR_STATE, "mha_workset_compute_residual before mha_workset_update");
  const size_t cap = static_cast<size_t>(std::max(wkset_.maxElem, ne));
  if (ws_res_.size() < cap * n_) ws_res_.resize(cap * n_);
  if (compute_jacobian && ws_res_dx_.size() < cap * n_ * n_) ws_res_dx_.resize(cap * n_ * n_);
  // Workset::resetResidual (workset.cpp:449-459)
  MHA_HIP(hipMemsetAsync(ws_res_.data(), 0, sizeof(double) * ne * n_, stream_));
  if (compute_jacobian) MHA_HIP(hipMemsetAsync(ws_res_dx_.data(), 0, sizeof(double) * ne * n_ * n_, stream_));
  wkset_.res = ElemOut();
  wkset_.res.compute_jacobian = compute_jacobian ? 1 : 0;
  wkset_.res.local_base = e0;
  wkset_.res.local_J = compute_jacobian ? ws_res_dx_.data() : nullptr;
  wkset_.res.local_res = ws_res_.data();
  if (engineOnly()) {
    launchPointEngine(compute_jacobian, wkset_.res, e0, ne);
  } else {
    wkset_.use_general = false;
    physics_->volumeResidual();
  }
  // the element kernels follow updateRes (local_res -= res.val()); the view holds res.val()
  launch_negate(ws_res_.data(), static_cast<size_t>(ne) * n_, stream_);
  ws_res_first_ = e0;
  ws_res_num_ = ne;
  ws_res_has_dx_ = compute_jacobian != 0;
}

// ---------------------------------------------------------------------------------------------
// boundary groups
// ---------------------------------------------------------------------------------------------

void AssemblyManager::prepareSideTables() {
  if (has_side_tables_) return;
  side_ref_ = make_side_tables(ref_);
  d_side_ip_.upload(side_ref_.ip);
  d_side_wts_.upload(side_ref_.wts);
  d_side_tanU_.upload(side_ref_.tanU);
  d_side_tanV_.upload(side_ref_.tanV);
  d_side_basis_.upload(side_ref_.basis);
  d_side_grad_.upload(side_ref_.grad);
  d_side_nodeval_.upload(side_ref_.nodeval);
  d_side_nodegrad_.upload(side_ref_.nodegrad);
  has_side_tables_ = true;
}

SideTablesDev AssemblyManager::sideTablesDev() const {
  SideTablesDev t;
  t.nsides = side_ref_.nsides;
  t.nqs = side_ref_.nqs;
  t.ip = d_side_ip_.data();
  t.wts = d_side_wts_.data();
  t.tanU = d_side_tanU_.data();
  t.tanV = d_side_tanV_.data();
  t.basis = d_side_basis_.data();
  t.grad = d_side_grad_.data();
  t.nodeval = d_side_nodeval_.data();
  t.nodegrad = d_side_nodegrad_.data();
  return t;
}

BoundaryDev AssemblyManager::boundaryDev(const BoundaryGroupData &g) const {
  BoundaryDev b;
  b.num = g.num;
  b.elem = g.elem.data();
  b.side = g.side.data();
  b.bc_type = g.bc_type;
  return b;
}

// reference: the BoundaryGroup constructor keeps (localElemID, localSideID, sidename) of one side set
// (src/tools/boundaryGroup.cpp:25-108); the BC type is what bcs(var, side) holds for it.
int AssemblyManager::addBoundaryGroup(const std::string &sidename, int bc_type, int num, const int32_t *elem_ids,
                                      const int32_t *side_ids) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  MHA_REQUIRE(bc_type == MHA_BC_NEUMANN || bc_type == MHA_BC_WEAK_DIRICHLET || bc_type == MHA_BC_INTERFACE ||
                  (bc_type >= MHA_BC_SWH_INTERFACE && bc_type <= MHA_BC_SWH_SLIP),
              MHA_ERR_INVALID, "boundary-condition type must be one of MHA_BC_*");
  MHA_REQUIRE(num >= 0 && (num == 0 || (elem_ids && side_ids)), MHA_ERR_INVALID, "bad boundary entry arrays");
  MHA_REQUIRE(!sidename.empty(), MHA_ERR_INVALID, "side name is empty");
  prepareSideTables();
  for (int k = 0; k < num; ++k) {
    MHA_REQUIRE(elem_ids[k] >= 0 && elem_ids[k] < nelem_, MHA_ERR_INVALID,
                "boundary entry " << k << ": element id " << elem_ids[k] << " out of range");
    MHA_REQUIRE(side_ids[k] >= 0 && side_ids[k] < side_ref_.nsides, MHA_ERR_INVALID,
                "boundary entry " << k << ": local side id " << side_ids[k] << " out of range");
  }
  MHA_REQUIRE(thermal_boundary_supported(n_, side_ref_.nqs), MHA_ERR_INVALID,
              "boundary terms are not available for " << n_ << " dofs per element with " << side_ref_.nqs
                                                      << " side integration points");
  std::unique_ptr<BoundaryGroupData> g(new BoundaryGroupData());
  g->sidename = sidename;
  g->bc_type = bc_type;
  g->num = num;
  g->elem.upload(elem_ids, num);
  g->side.upload(side_ids, num);
  boundary_groups_.push_back(std::move(g));
  return static_cast<int>(boundary_groups_.size()) - 1;
}

// reference: wkset->var_bcs(var, side) == "Flux" / "Dirichlet" for `varname` on this side set (physicsInterface.cpp:1705-1712,
// assemblyManager.cpp:6300): the data is the function "Flux <var> <sidename>" / "Dirichlet <var> <sidename>"
int AssemblyManager::addVarGroup(int bc_type, const std::string &sidename, const std::string &varname, int num,
                                 const int32_t *elem_ids, const int32_t *side_ids) {
  MHA_REQUIRE(has_mesh_, MHA_ERR_STATE, "no mesh: call mha_set_mesh first");
  const int var = varIndex(varname);
  MHA_REQUIRE(var >= 0, MHA_ERR_INVALID, "the block has no variable named '" << varname << "'");
  MHA_REQUIRE(num >= 0 && (num == 0 || (elem_ids && side_ids)), MHA_ERR_INVALID, "bad boundary entry arrays");
  MHA_REQUIRE(!sidename.empty(), MHA_ERR_INVALID, "side name is empty");
  prepareSideTables();
  for (int k = 0; k < num; ++k) {
    MHA_REQUIRE(elem_ids[k] >= 0 && elem_ids[k] < nelem_, MHA_ERR_INVALID,
                "boundary entry " << k << ": element id " << elem_ids[k] << " out of range");
    MHA_REQUIRE(side_ids[k] >= 0 && side_ids[k] < side_ref_.nsides, MHA_ERR_INVALID,
                "boundary entry " << k << ": local side id " << side_ids[k] << " out of range");
  }
  std::unique_ptr<BoundaryGroupData> g(new BoundaryGroupData());
  g->sidename = sidename;
  g->bc_type = bc_type;
  g->var = var;
  g->num = num;
  g->elem.upload(elem_ids, num);
  g->side.upload(side_ids, num);
  boundary_groups_.push_back(std::move(g));
  return static_cast<int>(boundary_groups_.size()) - 1;
}

int AssemblyManager::addFluxGroup(const std::string &sidename, const std::string &varname, int num, const int32_t *elem_ids,
                                  const int32_t *side_ids) {
  return addVarGroup(MHA_BC_FLUX, sidename, varname, num, elem_ids, side_ids);
}

int AssemblyManager::addDirichletGroup(const std::string &sidename, const std::string &varname, int num,
                                       const int32_t *elem_ids, const int32_t *side_ids) {
  return addVarGroup(MHA_BC_DIRICHLET, sidename, varname, num, elem_ids, side_ids);
}

// PhysicsInterface::getInitial (physicsInterface.cpp:911-958): "initial <var>" for HGRAD / HVOL, "initial <var>[x]", "[y]",
// "[z]" for HDIV
void AssemblyManager::initialFunctions(int v, FuncDesc f[3]) const {
  const std::string base = "initial " + varName(v);
  if (vars_[v].type == MHA_BASIS_HDIV) {
    static const char *comp[3] = {"[x]", "[y]", "[z]"};
    for (int d = 0; d < dim_; ++d) f[d] = functions_.evaluate(base + comp[d]);
  } else {
    f[0] = functions_.evaluate(base);
  }
}

// reference: AssemblyManager::setInitial(set, rhs, mass, useadjoint, lumpmass, scale) (assemblyManager.cpp:1185-1305): per
// group getInitial(project = true) and getMass, summed into the vector and the matrix (no sign change, fixed rows
// included), then rows without entries get a one on the diagonal.  `scale` is not used by the reference either.
void AssemblyManager::setInitial(int lump_mass, double *rhs, double *mass_vals) {
  requireReady(false);
  MHA_REQUIRE(rhs && mass_vals, MHA_ERR_INVALID, "null pointer");
  MHA_REQUIRE(has_graph_, MHA_ERR_STATE, "no CRS graph: call mha_set_graph first");
  for (int w = 0; w < numWorksets(); ++w) {
    worksetUpdate(w);
    for (int v = 0; v < static_cast<int>(vars_.size()); ++v) {
      const double *basis, *grad, *div;
      worksetVarArrays(v, &basis, &grad, &div);
      ProjectDev p;
      p.num = wkset_.numElem;
      p.card = vars_[v].card;
      p.var_off = layout_.varptr[v];
      p.np = nq_;
      p.ncomp = varComps(v);
      p.e0 = wkset_.first_elem;
      p.wts = static_cast<const double *>(wkset_.get("wts").ptr);
      static const char *xyz[3] = {"x", "y", "z"};
      for (int d = 0; d < dim_; ++d) p.xyz[d] = static_cast<const double *>(wkset_.get(xyz[d]).ptr);
      p.basis = basis;
      FuncDesc f[3];
      initialFunctions(v, f);
      launch_project_rhs(blockDev(), p, f, rhs, stream_);
      launch_project_mass(blockDev(), p, lump_mass ? 1 : 0, mass_vals, stream_);
    }
  }
  launch_fix_zero_rows(blockDev(), mass_vals, stream_);
}

// reference: AssemblyManager::setInitial(set, initial, useadjoint) (assemblyManager.cpp:1830-1850) with getInitial(project =
// false) (:7683-7727) -- "only works if using HGRAD linear basis": the function's value at the element's vertices replaces
// the vector entries of the vertex dofs
void AssemblyManager::setInitialNodal(double *initial) {
  requireReady(false);
  MHA_REQUIRE(initial, MHA_ERR_INVALID, "null pointer");
  for (size_t v = 0; v < vars_.size(); ++v)
    MHA_REQUIRE(vars_[v].type == MHA_BASIS_HGRAD && vars_[v].order == 1, MHA_ERR_INVALID,
                "nodal initial values need HGRAD variables of order 1 (variable '" << varName(static_cast<int>(v)) << "')");
  // the vertex each order-1 basis function sits on, from the reference values at the reference vertices (shards order)
  static const double quad[8] = {-1, -1, 1, -1, 1, 1, -1, 1};
  static const double hex[24] = {-1, -1, -1, 1, -1, -1, 1, 1, -1, -1, 1, -1, -1, -1, 1, 1, -1, 1, 1, 1, 1, -1, 1, 1};
  const int nn = 1 << dim_;
  std::vector<double> val, g, dv;
  const int card = ref_basis_var(dim_, MHA_BASIS_HGRAD, 1, nn, dim_ == 2 ? quad : hex, val, g, dv);
  MHA_REQUIRE(card == nn, MHA_ERR_INVALID, "order-1 HGRAD basis has " << card << " functions on " << nn << " vertices");
  int vert_of_dof[8] = {0};
  for (int k = 0; k < nn; ++k) {
    int at = -1;
    for (int v = 0; v < nn; ++v)
      if (std::fabs(val[static_cast<size_t>(k) * nn + v] - 1.0) < 1e-12) at = v;
    MHA_REQUIRE(at >= 0, MHA_ERR_INVALID, "order-1 HGRAD basis function " << k << " is not nodal");
    vert_of_dof[k] = at;
  }
  for (int v = 0; v < static_cast<int>(vars_.size()); ++v) {
    const FuncDesc f = functions_.evaluate("initial " + varName(v));
    launch_interpolate_nodes(blockDev(), f, layout_.varptr[v], vert_of_dof, initial, stream_);
  }
}

// reference: AssemblyManager::setDirichlet (assemblyManager.cpp:1855-1943): on the Dirichlet groups, the fixed rows get
// getDirichletBoundary (:6288-6350) in the vector and getMassBoundary (:6360-6425) in the matrix; every other row
// touched by an element gets a one on its diagonal
void AssemblyManager::setDirichlet(int lump_mass, double *rhs, double *mass_vals) {
  requireReady(false);
  MHA_REQUIRE(rhs && mass_vals, MHA_ERR_INVALID, "null pointer");
  MHA_REQUIRE(has_graph_, MHA_ERR_STATE, "no CRS graph: call mha_set_graph first");
  for (size_t gi = 0; gi < boundary_groups_.size(); ++gi) {
    BoundaryGroupData &g = *boundary_groups_[gi];
    if (g.bc_type != MHA_BC_DIRICHLET) continue;
    if (!g.has_views) boundaryUpdate(static_cast<int>(gi));
    const VarInfo &vi = vars_[g.var];
    ProjectDev p;
    p.num = g.num;
    p.card = vi.card;
    p.var_off = layout_.varptr[g.var];
    p.np = side_ref_.nqs;
    p.ncomp = varComps(g.var);
    p.elem = g.elem.data();
    p.fixed_only = 1;
    p.normal_trace = vi.type == MHA_BASIS_HDIV ? 1 : 0;
    p.wts = g.wts.data();
    for (int d = 0; d < dim_; ++d) { p.xyz[d] = g.xyz[d].data(); p.nrm[d] = g.nrm[d].data(); }
    p.basis = single_hgrad_ ? g.basis.data() : g.var_views[g.var].basis.data();
    FuncDesc f[3];
    f[0] = functions_.evaluate("Dirichlet " + varName(g.var) + " " + g.sidename);
    launch_project_rhs(blockDev(), p, f, rhs, stream_);
    launch_project_mass(blockDev(), p, lump_mass ? 1 : 0, mass_vals, stream_);
  }
  launch_free_row_identity(blockDev(), mass_vals, stream_);
}

// reference: the boundary-group loop of assembleJacRes (assemblyManager.cpp:2518-2638): per group
// updateWorksetBoundary, performBoundaryGather, physics boundaryResidual, scatter.  Accumulates.
void AssemblyManager::assembleBoundary(int flags, const double *u, const double *u_prev, const double *u_stage,
                                       double *res, double *crs_vals) {
  const int compute_jacobian = (flags & MHA_ASSEMBLE_JACOBIAN) ? 1 : 0;
  requireReady(true);
  MHA_REQUIRE(!(flags & MHA_ASSEMBLE_OVERWRITE), MHA_ERR_INVALID,
              "boundary terms are accumulated after the volume terms: MHA_ASSEMBLE_OVERWRITE is not valid here");
  MHA_REQUIRE(res != nullptr, MHA_ERR_INVALID, "residual vector is null");
  MHA_REQUIRE(!compute_jacobian || crs_vals, MHA_ERR_INVALID, "compute_jacobian set but crs_vals is null");
  bindState(u, u_prev, u_stage);
  timedBegin();
  for (size_t gi = 0; gi < boundary_groups_.size(); ++gi) {
    const auto &g = boundary_groups_[gi];
    if (g->bc_type == MHA_BC_DIRICHLET) continue;  // strong condition: its rows are fixed (setDirichlet builds their system)
    if (g->bc_type == MHA_BC_FLUX) {
      // PhysicsInterface::fluxConditions (physicsInterface.cpp:1702-1762): reads the group's stored side views
      if (!g->has_views) boundaryUpdate(static_cast<int>(gi));
      const VarInfo &vi = vars_[g->var];
      const FuncDesc flux = functions_.evaluate("Flux " + varName(g->var) + " " + g->sidename);
      const double *basis = single_hgrad_ ? g->basis.data() : g->var_views[g->var].basis.data();
      const double *xyz[3] = {g->xyz[0].data(), g->xyz[1].data(), g->xyz[2].data()};
      const double *nrm[3] = {g->nrm[0].data(), g->nrm[1].data(), g->nrm[2].data()};
      launch_flux_condition(blockDev(), flux, g->elem.data(), g->num, vi.card, layout_.varptr[g->var], side_ref_.nqs,
                            varComps(g->var), g->wts.data(), xyz, nrm, basis, res, stream_);
      continue;
    }
    wkset_.sidename = g->sidename;
    wkset_.current_bc = g->bc_type;
    wkset_.bnd = boundaryDev(*g);
    wkset_.side_tables = sideTablesDev();
    wkset_.layout = layout_;
    wkset_.layout.orient = has_orient_ ? d_orient_.data() : nullptr;
    wkset_.res = ElemOut();
    wkset_.res.compute_jacobian = compute_jacobian;
    wkset_.res.res = res;
    wkset_.res.crs_vals = compute_jacobian ? crs_vals : nullptr;
    physics_->boundaryResidual();
  }
  timedEnd();
}

// reference: AssemblyManager::computeFlux -> PhysicsInterface::computeFlux -> the module's computeFlux on a boundary
// group (src/managers/assemblyManager.hpp:573-727, subgridDtN_solver.cpp:1579): fills the workset's flux view
// (elem, auxvar, pt) -- here [num][nqs] of the module's one aux variable -- and, on request, its derivative arrays
void AssemblyManager::computeFlux(int group, const double *u, const double *u_prev, const double *u_stage, double *flux,
                                  double *dflux_du, double *dflux_daux) {
  requireReady(false);
  MHA_REQUIRE(group >= 0 && group < numBoundaryGroups(), MHA_ERR_INVALID, "boundary group id out of range");
  MHA_REQUIRE(flux != nullptr, MHA_ERR_INVALID, "flux array is null");
  const auto &g = boundary_groups_[group];
  bindState(u, u_prev, u_stage);
  if (physics_id_ != MHA_PHYSICS_NAVIERSTOKES) prepareSideTables();
  wkset_.sidename = g->sidename;
  wkset_.current_bc = g->bc_type;
  wkset_.bnd = boundaryDev(*g);
  wkset_.bnd.flux = flux;
  wkset_.bnd.dflux_du = dflux_du;
  wkset_.bnd.dflux_daux = dflux_daux;
  wkset_.side_tables = sideTablesDev();
  wkset_.layout = layout_;
  wkset_.layout.orient = has_orient_ ? d_orient_.data() : nullptr;
  timedBegin();
  physics_->computeFlux();
  timedEnd();
  wkset_.bnd.flux = nullptr;
}

// reference: BoundaryGroup::computeBasis -> getPhysicalBoundaryIntegrationData / getPhysicalBoundaryBasis
// (src/tools/boundaryGroup.cpp:110-178)
void AssemblyManager::boundaryUpdate(int group) {
  MHA_REQUIRE(group >= 0 && group < numBoundaryGroups(), MHA_ERR_INVALID, "boundary group id out of range");
  BoundaryGroupData &g = *boundary_groups_[group];
  const size_t nqs = side_ref_.nqs, num = g.num;
  g.wts.resize(num * nqs);
  for (int d = 0; d < dim_; ++d) { g.xyz[d].resize(num * nqs); g.nrm[d].resize(num * nqs); }
  // basis views are those of the block's single HGRAD variable; multi-variable blocks expose the geometry views only
  if (single_hgrad_) {
    g.basis.resize(num * n_ * nqs);
    g.basis_grad.resize(num * n_ * nqs * dim_);
  }
  BoundaryViewsDev v;
  v.wts = g.wts.data();
  for (int d = 0; d < dim_; ++d) { v.xyz[d] = g.xyz[d].data(); v.nrm[d] = g.nrm[d].data(); }
  v.basis = g.basis.data();
  v.basis_grad = g.basis_grad.data();
  launch_boundary_views(blockDev(), sideTablesDev(), boundaryDev(g), v, stream_);
  if (!single_hgrad_) {
    // getPhysicalBoundaryBasis for every basis of the block (discretizationInterface.cpp:1840-1950)
    g.var_views.resize(vars_.size());
    for (size_t k = 0; k < vars_.size(); ++k) {
      const VarInfo &vi = vars_[k];
      VarViews &vv = g.var_views[k];
      vv.basis.resize(num * vi.card * nqs * varComps(static_cast<int>(k)));
      if (vi.type == MHA_BASIS_HGRAD) vv.grad.resize(num * vi.card * nqs * dim_);
      VarViewsDev out;
      out.basis = vv.basis.data();
      out.grad = vv.grad.data();
      launch_var_views(blockDev(), varPoints(static_cast<int>(k), true), g.elem.data(), g.side.data(), 0, g.num, out,
                       stream_);
    }
  }
  g.has_views = true;
}

View AssemblyManager::boundaryView(int group, const std::string &name) const {
  MHA_REQUIRE(group >= 0 && group < numBoundaryGroups(), MHA_ERR_INVALID, "boundary group id out of range");
  const BoundaryGroupData &g = *boundary_groups_[group];
  MHA_REQUIRE(g.has_views, MHA_ERR_STATE, "boundary views requested before mha_boundary_update");
  View v;
  const int64_t num = g.num, nqs = side_ref_.nqs;
  auto comp = [&](char c) {
    const int k = c - 'x';
    MHA_REQUIRE(k >= 0 && k < dim_, MHA_ERR_UNKNOWN_FIELD, "side field '" << name << "' does not exist in " << dim_ << "-D");
    return k;
  };
  if (name == "wts side") {
    v.ptr = g.wts.data(); v.rank = 2; v.extent[0] = num; v.extent[1] = nqs;
  } else if (name == "x" || name == "y" || name == "z") {
    v.ptr = g.xyz[comp(name[0])].data(); v.rank = 2; v.extent[0] = num; v.extent[1] = nqs;
  } else if (name == "n[x]" || name == "n[y]" || name == "n[z]") {
    v.ptr = g.nrm[comp(name[2])].data(); v.rank = 2; v.extent[0] = num; v.extent[1] = nqs;
  } else if (name.compare(0, 11, "basis side ") == 0 || name.compare(0, 16, "basis_grad side ") == 0) {
    // getBasisSide(var) / getBasisGradSide(var) (workset.hpp:281-293)
    const bool grad = name[5] == '_';
    const int k = varIndex(name.substr(grad ? 16 : 11));
    if (k < 0) throw Error(MHA_ERR_UNKNOWN_FIELD, "unknown variable in boundary view '" + name + "'");
    const VarInfo &vi = vars_[k];
    const double *p = single_hgrad_ ? (grad ? g.basis_grad.data() : g.basis.data())
                                    : (grad ? g.var_views[k].grad.data() : g.var_views[k].basis.data());
    if (!p) throw Error(MHA_ERR_UNKNOWN_FIELD, "'" + name + "' is not defined for this basis type");
    v.ptr = const_cast<double *>(p); v.rank = 4; v.extent[0] = num; v.extent[1] = vi.card; v.extent[2] = nqs;
    v.extent[3] = grad ? dim_ : varComps(k);
  } else if ((name == "basis side" || name == "basis_grad side") && !single_hgrad_) {
    throw Error(MHA_ERR_UNKNOWN_FIELD, "'" + name + "' needs the variable's name on a multi-variable block: '" + name + " <var>'");
  } else if (name == "basis side") {
    v.ptr = g.basis.data(); v.rank = 4; v.extent[0] = num; v.extent[1] = n_; v.extent[2] = nqs; v.extent[3] = 1;
  } else if (name == "basis_grad side") {
    v.ptr = g.basis_grad.data(); v.rank = 4; v.extent[0] = num; v.extent[1] = n_; v.extent[2] = nqs; v.extent[3] = dim_;
  } else {
    throw Error(MHA_ERR_UNKNOWN_FIELD, "unknown boundary view '" + name + "'");
  }
  return v;
}

void AssemblyManager::setPhysicsParameter(const std::string &name, double value) {
  MHA_REQUIRE(physics_ != nullptr, MHA_ERR_STATE, "no physics module: call mha_physics_select first");
  physics_->setParameter(name, value);
}

// ---------------------------------------------------------------------------------------------
// row-owner path
// ---------------------------------------------------------------------------------------------

void AssemblyManager::prepareRowOwner() {
  requireReady(true);
  MHA_REQUIRE(thermal_row_owner_supported(dim_, order_, ref_.nq1), MHA_ERR_INVALID,
              "row-owner kernel: unsupported (dim,order,points/dir)");
  RowOwnerData &ro = ro_;
  const BlockDev b = blockDev();
  // 1. element classification on the device
  ro.flags.resize(nelem_);
  launch_classify_affine(b, ro.flags.data(), 1e-14, stream_);
  std::vector<uint8_t> flags(nelem_);
  MHA_HIP(hipStreamSynchronize(stream_));
  ro.flags.download(flags.data());
  ro.num_affine_elems = 0;
  for (int e = 0; e < nelem_; ++e) ro.num_affine_elems += flags[e];
  // 2. row blocks
  std::vector<double> nodes(static_cast<size_t>(nelem_) * nnodes_ * dim_);
  d_nodes_.download(nodes.data());
  for (int d = 0; d < 3; ++d) ro.max_abs_coord[d] = 0.0;
  for (size_t i = 0; i < nodes.size(); ++i) ro.max_abs_coord[i % dim_] = std::max(ro.max_abs_coord[i % dim_], std::fabs(nodes[i]));
  int max_row = 0;
  for (int r = 0; r < nrows_; ++r) max_row = std::max(max_row, h_rowptr_[r + 1] - h_rowptr_[r]);
  MHA_REQUIRE(max_row <= 65536, MHA_ERR_INVALID, "CRS rows longer than 65536 entries are not supported");
  ro.slot_bytes = max_row <= 256 ? 1 : 2;
  RowBlockCaps caps = default_caps(dim_, n_);
  // LDS budget: two workgroups per CU (80 KiB each).  An aligned interior chunk touches 3^dim elements
  // and owns (2*order)^dim rows; the accumulator gets whatever the fixed parts leave.
  {
    // LDS budget of K2: four workgroups per CU (40 KiB each).  An aligned interior chunk touches
    // 3^dim elements; the accumulator gets whatever the pair tables leave.
    const int neigh = (dim_ == 3) ? 27 : 25;
    caps.max_elems = neigh;
    caps.max_rows = 96;
    caps.max_pairs = (dim_ == 3) ? 224 : 256;
    caps.max_acc = 65534;
    RowBlocksDev probe;
    probe.lds_rows = caps.max_rows;
    probe.lds_elems = caps.max_elems;
    probe.lds_pairs = caps.max_pairs;
    probe.lds_acc = 0;
    const long fixed_bytes = static_cast<long>(row_owner_jacobian_lds(probe, n_, ro.slot_bytes));
    const long budget = 40 * 1024 - fixed_bytes;
    MHA_REQUIRE(budget >= 8 * 2 * max_row, MHA_ERR_INVALID, "row-owner kernel does not fit the LDS budget for this element");
    caps.max_acc = std::min(65534, static_cast<int>(budget / 8) / 2 * 2);
    // tuning knobs (experiments): elements per Morton chunk and the accumulator cap
    if (const char *e = std::getenv("MHA_RB_CHUNK")) caps.chunk_elems = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("MHA_RB_MAXACC")) caps.max_acc = std::max(2 * max_row, std::atoi(e));
  }
  ro.rb = build_row_blocks(dim_, nnodes_, nelem_, n_, nrows_, nodes.data(), h_lids_.data(), h_rowptr_.data(), caps,
                           has_fixed_ ? h_fixed_.data() : nullptr, ro.slot_bytes);
  const RowBlocks &rb = ro.rb;
  ro.row_ptr.upload(rb.row_ptr);
  ro.rows.upload(rb.rows);
  ro.row_off.upload(rb.row_off);
  ro.acc_size.upload(rb.acc_size);
  ro.elem_ptr.upload(rb.elem_ptr);
  ro.elems.upload(rb.elems);
  ro.pair_ptr.upload(rb.pair_ptr);
  ro.pairs.upload(rb.pairs);
  ro.pair_off.upload(rb.pair_off);
  ro.row_base.upload(rb.row_base);
  ro.row_len.upload(rb.row_len);
  ro.emask.upload(rb.emask);
  ro.epbase.upload(rb.epbase);
  ro.slot_ptr.upload(rb.slot_ptr);
  ro.seg_ptr.upload(rb.seg_ptr);
  ro.seg_acc.upload(rb.seg_acc);
  ro.seg_base.upload(rb.seg_base);
  ro.seg_len.upload(rb.seg_len);
  ro.geo.resize(static_cast<size_t>(nelem_) * kGeoRec);
  launch_affine_geometry(b, ro.geo.data(), stream_);
  {
    std::vector<uint16_t> po(rb.pair_off.size());
    for (size_t i = 0; i < po.size(); ++i) po[i] = static_cast<uint16_t>(rb.pair_off[i]);
    ro.pair_off16.upload(po);
  }
  ro.all_rows_covered = static_cast<int>(rb.rows.size()) == nrows_;
  // 3. block classification: affine blocks touch affine elements only
  std::vector<int32_t> aff, gen;
  for (int k = 0; k < rb.num_blocks; ++k) {
    bool a = true;
    for (int p = rb.elem_ptr[k]; p < rb.elem_ptr[k + 1] && a; ++p) a = flags[rb.elems[p]] != 0;
    (a ? aff : gen).push_back(k);
  }
  ro.num_affine_blocks = static_cast<int>(aff.size());
  ro.num_general_blocks = static_cast<int>(gen.size());
  ro.affine_list.upload(aff);
  ro.general_list.upload(gen);
  // 4. block-major slot table (position of every contribution inside its CRS row)
  ro.slot.resize(std::max<size_t>(16, static_cast<size_t>(rb.slot_ptr.back())));
  launch_build_block_slots(b, rowBlocksDev(), ro.slot.data(), ro.slot_bytes, stream_);
  ro.erec.resize(std::max<size_t>(8, rb.elems.size() * 8));
  launch_build_erec(dim_, rowBlocksDev(), ro.geo.data(), ro.erec.data(), static_cast<int>(rb.elems.size()), stream_);
  // 4b. lane layout of K2: pair up LID slots that are usually owned together by the same (block, element),
  //     so that a wave-instruction working on two slots side by side is either skipped or mostly busy.
  {
    std::vector<double> both(static_cast<size_t>(n_) * n_, 0.0), cnt(n_, 0.0);
    const size_t stride = std::max<size_t>(1, rb.emask.size() / 200000);  // a sample is plenty
    for (size_t i = 0; i < rb.emask.size(); i += stride) {
      const uint32_t m = static_cast<uint32_t>(rb.emask[i]);
      for (int a = 0; a < n_ && a < 32; ++a) {
        if (!((m >> a) & 1u)) continue;
        cnt[a] += 1.0;
        for (int c = a + 1; c < n_ && c < 32; ++c)
          if ((m >> c) & 1u) both[static_cast<size_t>(a) * n_ + c] += 1.0;
      }
    }
    std::vector<int> pairs(2 * ((n_ + 1) / 2), -1);
    std::vector<char> used(n_, 0);
    for (int r = 0; r < n_ / 2; ++r) {  // greedy matching by Jaccard similarity of ownership
      int ba = -1, bc = -1;
      double best = -1.0;
      for (int a = 0; a < n_; ++a)
        for (int c = a + 1; c < n_; ++c) {
          if (used[a] || used[c]) continue;
          const double uni = cnt[a] + cnt[c] - both[static_cast<size_t>(a) * n_ + c];
          const double jac = uni > 0.0 ? both[static_cast<size_t>(a) * n_ + c] / uni : 0.0;
          if (jac > best) { best = jac; ba = a; bc = c; }
        }
      pairs[2 * r] = ba;
      pairs[2 * r + 1] = bc;
      used[ba] = used[bc] = 1;
    }
    if (n_ % 2)
      for (int a = 0; a < n_; ++a)
        if (!used[a]) pairs[2 * (n_ / 2)] = a;
    ro.slot_pair.upload(pairs);
  }
  // 5. reference tables of the affine path, in LID-slot space
  const int nsym = dim_ * (dim_ + 1) / 2;
  std::vector<double> khat(static_cast<size_t>(nsym + 1) * n_ * n_, 0.0);
  std::vector<int32_t> offs(n_);
  d_offsets_.download(offs.data());
  for (int ib = 0; ib < n_; ++ib)
    for (int jb = 0; jb < n_; ++jb) {
      const size_t idx = static_cast<size_t>(offs[ib]) * n_ + offs[jb];
      int k = 0;
      for (int a = 0; a < dim_; ++a)
        for (int c = a; c < dim_; ++c, ++k) {
          double s = 0.0;
          for (int q = 0; q < nq_; ++q) {
            const double *gi = &ref_.grad[(static_cast<size_t>(ib) * nq_ + q) * dim_];
            const double *gj = &ref_.grad[(static_cast<size_t>(jb) * nq_ + q) * dim_];
            s += ref_.wts[q] * (a == c ? gi[a] * gj[a] : gi[a] * gj[c] + gi[c] * gj[a]);
          }
          khat[static_cast<size_t>(k) * n_ * n_ + idx] = s;
        }
      double m = 0.0;
      for (int q = 0; q < nq_; ++q) m += ref_.wts[q] * ref_.basis[ib * nq_ + q] * ref_.basis[jb * nq_ + q];
      khat[static_cast<size_t>(nsym) * n_ * n_ + idx] = m;
    }
  ro.khat.upload(khat);
  // thread-per-element K1: 1-D tables by value; the collocation derivative D = Phi'^T Phi^-T (Gauss-Jordan on the
  // small, well-conditioned point-value matrix)
  ro.k1_thread = false;
  if (thermal_affine_residual_supported(dim_, order_, ref_.nq1)) {
    const int m = order_ + 1;
    AffineTables1D &t = ro.tab1d;
    t = AffineTables1D();
    for (int i = 0; i < m * m; ++i) t.phi[i] = ref_.phi1d[i];
    for (int q = 0; q < m; ++q) { t.gw[q] = ref_.gauss_wts[q]; t.gp[q] = ref_.gauss_pts[q]; }
    // inv = Phi^-1 with Phi[i][q] = phi_i(xi_q)
    std::vector<double> a(ref_.phi1d.begin(), ref_.phi1d.begin() + m * m), inv(m * m, 0.0);
    for (int i = 0; i < m; ++i) inv[i * m + i] = 1.0;
    for (int c = 0; c < m; ++c) {
      int piv = c;
      for (int r = c + 1; r < m; ++r)
        if (std::fabs(a[r * m + c]) > std::fabs(a[piv * m + c])) piv = r;
      for (int k = 0; k < m; ++k) { std::swap(a[c * m + k], a[piv * m + k]); std::swap(inv[c * m + k], inv[piv * m + k]); }
      const double d = 1.0 / a[c * m + c];
      for (int k = 0; k < m; ++k) { a[c * m + k] *= d; inv[c * m + k] *= d; }
      for (int r = 0; r < m; ++r) {
        if (r == c) continue;
        const double f = a[r * m + c];
        for (int k = 0; k < m; ++k) { a[r * m + k] -= f * a[c * m + k]; inv[r * m + k] -= f * inv[c * m + k]; }
      }
    }
    for (int q = 0; q < m; ++q)
      for (int qp = 0; qp < m; ++qp) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += ref_.dphi1d[i * m + q] * inv[qp * m + i];
        t.dcol[q * m + qp] = s;
      }
    const char *k1 = std::getenv("MHA_K1");
    ro.k1_thread = !(k1 && std::string(k1) == "lanes");
    ro.k1_plan = K1PlanDev();
    ro.k1_wg = ro.k1_thread && !(k1 && std::string(k1) == "thread");
  }
  ro.phi.upload(ref_.phi1d);
  ro.dphi.upload(ref_.dphi1d);
  ro.gw.upload(ref_.gauss_wts);
  ro.gp.upload(ref_.gauss_pts);
  MHA_HIP(hipStreamSynchronize(stream_));
  if (ro.k1_wg) {
    // Plan of the workgroup-merged K1 (K1PlanDev): the elements in groups of 256, per group the distinct rows its dofs
    // touch (ascending) and per (element, dof in basis order) the position of its row in that list.
    constexpr int T = kK1PlanThreads;
    std::vector<double> geo(static_cast<size_t>(nelem_) * kGeoRec);
    ro.geo.download(geo.data());
    bool aligned = true;  // every element axis-aligned (J diagonal: the test the kernels make per element)?
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int e = 0; e < nelem_; ++e) {
      const double *g = &geo[static_cast<size_t>(e) * kGeoRec];
      for (int r = 0; r < dim_; ++r) {
        lo[r] = std::min(lo[r], g[kGeoXc + r]);
        hi[r] = std::max(hi[r], g[kGeoXc + r]);
        for (int c = 0; c < dim_; ++c)
          if (r != c && g[kGeoJ + r * dim_ + c] != 0.0) aligned = false;
      }
    }
    // order of the elements: as numbered (coalesced record loads; good whenever consecutive elements are neighbours),
    // or along a Morton curve through the centroids when the numbering scatters a group over the mesh (its distinct
    // rows would not fit the LDS three workgroups deep); MHA_K1_ORDER=natural|morton forces one
    const char *ord = std::getenv("MHA_K1_ORDER");
    const int G = (nelem_ + T - 1) / T;
    std::vector<int32_t> wg_elems(static_cast<size_t>(G) * T), rp, rows, tmp;
    std::vector<uint16_t> loc;
    int max_rows = 0;
    auto build = [&](bool natural) {
      std::vector<std::pair<uint64_t, int32_t>> keyed(nelem_);
      for (int e = 0; e < nelem_; ++e) {
        uint64_t key = 0;
        if (!natural) {
          uint32_t q[3] = {0, 0, 0};
          for (int r = 0; r < dim_; ++r) {
            const double w = hi[r] > lo[r] ? (geo[static_cast<size_t>(e) * kGeoRec + kGeoXc + r] - lo[r]) / (hi[r] - lo[r]) : 0.0;
            q[r] = static_cast<uint32_t>(std::min(1048575.0, std::max(0.0, w * 1048575.0)));
          }
          for (int bit = 19; bit >= 0; --bit)
            for (int r = dim_ - 1; r >= 0; --r) key = (key << 1) | ((q[r] >> bit) & 1u);
        }
        keyed[e] = {key, e};
      }
      std::sort(keyed.begin(), keyed.end());
      for (size_t i = 0; i < wg_elems.size(); ++i) wg_elems[i] = keyed[std::min<size_t>(i, nelem_ - 1)].second;
      rp.assign(static_cast<size_t>(G) + 1, 0);
      rows.clear();
      rows.reserve(static_cast<size_t>(nelem_) * n_ / 2);
      loc.assign(static_cast<size_t>(G) * n_ * T, 0);
      max_rows = 0;
      for (int g = 0; g < G; ++g) {
        const int cnt = std::min(T, nelem_ - g * T);
        tmp.clear();
        for (int t = 0; t < cnt; ++t) {
          const int32_t *L = &h_lids_[static_cast<size_t>(wg_elems[static_cast<size_t>(g) * T + t]) * n_];
          tmp.insert(tmp.end(), L, L + n_);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (int t = 0; t < cnt; ++t) {
          const int32_t *L = &h_lids_[static_cast<size_t>(wg_elems[static_cast<size_t>(g) * T + t]) * n_];
          for (int ib = 0; ib < n_; ++ib)
            loc[(static_cast<size_t>(g) * n_ + ib) * T + t] =
                static_cast<uint16_t>(std::lower_bound(tmp.begin(), tmp.end(), L[offs[ib]]) - tmp.begin());
        }
        rows.insert(rows.end(), tmp.begin(), tmp.end());
        rp[g + 1] = static_cast<int32_t>(rows.size());
        max_rows = std::max(max_rows, static_cast<int>(tmp.size()));
      }
    };
    const bool force_morton = ord && std::string(ord) == "morton", force_natural = ord && std::string(ord) == "natural";
    build(!force_morton);
    if (!force_morton && !force_natural && static_cast<size_t>(max_rows) * 12 > 52 * 1024) {
      const int natural_rows = max_rows;
      build(false);
      if (max_rows >= natural_rows) build(true);
    }
    ro.k1_row_ptr.upload(rp);
    ro.k1_rows.upload(rows);
    ro.k1_loc.upload(loc);
    ro.k1_elems.upload(wg_elems);
    ro.k1_plan.wg_row_ptr = ro.k1_row_ptr.data();
    ro.k1_plan.wg_rows = ro.k1_rows.data();
    ro.k1_plan.loc = ro.k1_loc.data();
    ro.k1_plan.wg_elems = ro.k1_elems.data();
    ro.k1_plan.max_rows = (max_rows + 1) / 2 * 2;
    ro.k1_plan.num_elems = nelem_;
    ro.k1_plan.axis_aligned = aligned ? 1 : 0;
    // geometry database (reference: identifyVolumetricDatabase, assemblyManager.cpp:4314-4467; here exact matching of
    // the 16 shape doubles, bit for bit, so nothing is substituted): distinct shapes in order of first appearance
    {
      static_assert(kGeoXc == 16, "the shape part of a geometry record is its first 16 doubles");
      std::unordered_map<std::string, int32_t> seen;
      std::vector<double> shapes;
      std::vector<int32_t> sidx(nelem_);
      const int nsym_g = dim_ * (dim_ + 1) / 2;
      for (int e = 0; e < nelem_; ++e) {
        double rec[16];
        for (int k = 0; k < 16; ++k) {  // (entries a 2-D record does not use are not part of the shape)
          const bool used = k < nsym_g || k == kGeoDet || (k >= kGeoJ && k < kGeoJ + dim_ * dim_);
          rec[k] = used ? geo[static_cast<size_t>(e) * kGeoRec + k] : 0.0;
        }
        auto it = seen.emplace(std::string(reinterpret_cast<const char *>(rec), sizeof(rec)), static_cast<int32_t>(seen.size()));
        if (it.second) shapes.insert(shapes.end(), rec, rec + 16);
        sidx[e] = it.first->second;
      }
      const char *db = std::getenv("MHA_K1_DATABASE");
      if (!(db && db[0] == '0')) {
        ro.k1_shape.upload(shapes);
        ro.k1_shape_idx.upload(sidx);
        ro.k1_plan.shape = ro.k1_shape.data();
        ro.k1_plan.shape_idx = ro.k1_shape_idx.data();
      }
      ro.k1_plan.num_shapes = static_cast<int>(seen.size());
      ro.num_shapes = ro.k1_plan.num_shapes;
    }
  }
  ro.ready = true;
  prepareBlockPattern();
}

// Matrix-core form of K2: row blocks keyed by assembly pattern (block_pattern.hpp).  Default; MHA_K2=blocks keeps the
// LDS-accumulator row-block kernel, as does a mesh whose blocks share too few patterns (info key "block_patterns" = 0).
void AssemblyManager::prepareBlockPattern() {
  BlockPatternData &bp = bpat_;
  bp.tried = true;
  bp.usable = false;
  const char *mode = std::getenv("MHA_K2");
  if (mode && std::string(mode) == "blocks") { bp.why = "row-block kernel requested (MHA_K2=blocks)"; return; }
  if (ro_.num_general_blocks > 0) { bp.why = "block has non-affine elements"; return; }
  if (static_cast<long long>(h_rowptr_[nrows_]) >= (1ll << 28)) { bp.why = "more than 2^28 CRS entries (32-bit byte offsets)"; return; }
  // its own partition: larger Morton chunks (16 elements = 16 rows of every class of a Q2 hex block: whole MFMA panels),
  // no LDS accumulator to fit
  RowBlockCaps caps = default_caps(dim_, n_);
  caps.chunk_elems = 16;
  if (const char *e = std::getenv("MHA_BP_CHUNK")) caps.chunk_elems = std::max(1, std::atoi(e));
  caps.max_rows = 4096;
  caps.max_elems = 255;
  caps.max_pairs = 1 << 20;
  caps.max_acc = 1 << 30;
  std::vector<double> nodes(static_cast<size_t>(nelem_) * nnodes_ * dim_);
  d_nodes_.download(nodes.data());
  RowBlocks rb;
  try {
    rb = build_row_blocks(dim_, nnodes_, nelem_, n_, nrows_, nodes.data(), h_lids_.data(), h_rowptr_.data(), caps,
                          has_fixed_ ? h_fixed_.data() : nullptr, 1);
  } catch (const Error &e) {
    bp.why = e.what();
    return;
  }
  if (static_cast<int>(rb.rows.size()) != nrows_ && !ro_.all_rows_covered) { /* rows without elements stay untouched on both paths */ }
  prepareElemSlots();
  std::vector<uint8_t> slot(static_cast<size_t>(nelem_) * n_ * n_ * elem_slot_bytes_);
  MHA_HIP(hipStreamSynchronize(stream_));
  d_elem_slot_.download(slot.data());
  const int nsym = dim_ * (dim_ + 1) / 2;
  std::vector<double> khat(static_cast<size_t>(nsym + 1) * n_ * n_);
  ro_.khat.download(khat.data());
  int num_cu = current_device_num_cus();
  if (const char *m = std::getenv("MHA_BP_WGS")) num_cu = std::max(1, std::atoi(m));
  const BlockPatternPlan h = build_block_patterns(rb, n_, nsym, h_rowptr_.data(), has_fixed_ ? h_fixed_.data() : nullptr,
                                                  slot.data(), elem_slot_bytes_, khat.data(), num_cu, size_t(142) * 1024, 256,
                                                  std::getenv("MHA_BP_SEGBLOCKS") ? std::atoi(std::getenv("MHA_BP_SEGBLOCKS")) : 0);
  bp.why = h.why;
  if (std::getenv("MHA_VERBOSE"))
    fprintf(stderr, "[mrhyde_amd] block patterns: usable %d (%s), %d patterns, %d roles (%d with an LDS image), %d parts, %d workgroups, %lld MFMAs per assembly\n",
            int(h.usable), h.why.c_str(), h.num_patterns, h.num_roles, h.num_image_roles, h.num_parts, h.num_wgs, (long long)h.mfma_per_assembly);
  if (std::getenv("MHA_VERBOSE") && h.usable) {  // shapes of the units: (k-steps, tiles, tail, trim class) -> count
    std::map<std::vector<int>, int> shapes;
    for (int p = 0; p < h.num_parts; ++p) {
      const int32_t *hd = h.part_hdr.data() + static_cast<size_t>(p) * kBpHdrInts;
      shapes[{hd[1], hd[5], (hd[6] >> 1) & 1, (hd[6] >> 2) & 7, hd[6] & 1}]++;
    }
    for (const auto &kv : shapes)
      fprintf(stderr, "[mrhyde_amd]   units ks %d tiles %d tail %d trim %d fixed %d: %d\n", kv.first[0], kv.first[1], kv.first[2],
              kv.first[3], kv.first[4], kv.second);
  }
  if (!h.usable) return;
  bp.num_patterns = h.num_patterns;
  bp.num_roles = h.num_roles;
  bp.num_blocks = rb.num_blocks;
  bp.mfma_per_assembly = h.mfma_per_assembly;
  bp.role.upload(h.role);
  bp.seg.upload(h.seg);
  bp.wg_seg_ptr.upload(h.wg_seg_ptr);
  bp.wg_seg_ptr_img.upload(h.wg_seg_ptr_img);
  bp.part_ptr.upload(h.part_ptr);
  bp.part_hdr.upload(h.part_hdr);
  bp.part_lane.upload(h.part_lane);
  bp.rowbase.upload(h.rowbase);
  {
    std::vector<int32_t> ct = h.chunk_tab;
    if (ct.empty()) ct.assign(kBpChunkInts, 0);
    bp.chunk_tab.upload(ct);
  }
  bp.erec_elem.upload(h.erec_elem);
  bp.w.upload(h.w);
  bp.erec2.resize(h.erec_elem.size() * kBpRecDoubles);
  launch_build_erec2(static_cast<int64_t>(h.erec_elem.size()), nsym, bp.erec_elem.data(), ro_.geo.data(), bp.erec2.data(), stream_);
  if (std::getenv("MHA_BP_TIMING")) bp.timing.resize(static_cast<size_t>(h.num_wgs) * kBpWaves * 8);
  BlockPatternDev &d = bp.dev;
  d.num_wgs = h.num_wgs;
  d.max_w_doubles = h.max_w_doubles;
  d.max_rec_doubles = h.max_rec_doubles;
  d.dbg = 0;
  if (const char *m = std::getenv("MHA_BP_DBG")) d.dbg = std::atoi(m);
  d.erec2 = bp.erec2.data();
  d.rowbase = bp.rowbase.data();
  d.w = bp.w.data();
  d.role = bp.role.data();
  d.seg = bp.seg.data();
  d.wg_seg_ptr = bp.wg_seg_ptr.data();
  d.wg_seg_ptr_img = bp.wg_seg_ptr_img.data();
  d.has_direct = h.wg_seg_ptr.back() > h.wg_seg_ptr.front();
  d.has_image = h.wg_seg_ptr_img.back() > h.wg_seg_ptr_img.front();
  d.part_ptr = bp.part_ptr.data();
  d.part_hdr = bp.part_hdr.data();
  d.part_lane = bp.part_lane.data();
  d.chunk_tab = bp.chunk_tab.data();
  d.nnz = h_rowptr_[nrows_];
  d.timing = bp.timing.empty() ? nullptr : bp.timing.data();
  MHA_HIP(hipStreamSynchronize(stream_));
  bp.usable = true;
  // Geometry-database mode (reference: identifyVolumetricDatabase, assemblyManager.cpp:4314-4467,
  // here with exact matching): with ONE geometry shape in the block the rows of a row block depend on its pattern
  // only -- the kernel runs on one representative block per role (the role's first) and the representative's runs are
  // replicated.  MHA_BP_DATABASE=0 keeps the full kernel.
  bp.db_mode = false;
  const char *dbm = std::getenv("MHA_BP_DATABASE");
  if (ro_.num_shapes == 1 && !d.has_image && !(dbm && dbm[0] == '0')) {
    std::vector<int32_t> rseg(static_cast<size_t>(h.num_roles) * 4, 0), rptr(static_cast<size_t>(h.num_roles) + 1, 0);
    struct Run { int32_t src, dst, len; };
    std::vector<Run> runs;
    for (int k = 0; k < h.num_roles; ++k) {
      rseg[4 * k] = k;
      rseg[4 * k + 1] = 0;
      rseg[4 * k + 2] = 1;
      rptr[k + 1] = k + 1;
      const int32_t *ro = &h.role[static_cast<size_t>(k) * kBpRoleInts];
      const int64_t row_base = (static_cast<int64_t>(ro[4]) << 32) | static_cast<uint32_t>(ro[3]);  // R_ROWB_HI / _LO
      const int nruns = ro[5], nblocks = ro[6];                                                        // R_NRUNS, R_NBLOCKS
      const int32_t *rl = &h.runlen[static_cast<size_t>(h.role_runlen_off[k])];
      for (int j = 1; j < nblocks; ++j)
        for (int r = 0; r < nruns; ++r)
          runs.push_back({h.rowbase[static_cast<size_t>(row_base + r)], h.rowbase[static_cast<size_t>(row_base + static_cast<int64_t>(j) * nruns + r)], rl[r]});
    }
    std::sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.dst < b.dst; });
    // 1 KB chunks on 128-byte lines (the CRS values are 128-byte aligned: checked at launch): 16 entries per line
    std::vector<int32_t> chunks;
    for (const Run &r : runs) {
      const int64_t dbeg = r.dst, dend = static_cast<int64_t>(r.dst) + r.len;
      for (int64_t c = dbeg / 16 * 16; c < dend; c += 128) {
        chunks.push_back(static_cast<int32_t>(c / 2));                    // destination / 16 bytes
        chunks.push_back(static_cast<int32_t>(r.src + (c - dbeg)));       // source entry of lane 0's first double
        chunks.push_back(static_cast<int32_t>(dbeg));
        chunks.push_back(static_cast<int32_t>(dend));
      }
    }
    bp.rep_seg.upload(rseg);
    bp.rep_wg_seg_ptr.upload(rptr);
    bp.copy_chunks.upload(chunks);
    bp.copy_runs = static_cast<int>(chunks.size() / 4);
    bp.dev_rep = d;
    bp.dev_rep.seg = bp.rep_seg.data();
    bp.dev_rep.wg_seg_ptr = bp.rep_wg_seg_ptr.data();
    bp.dev_rep.num_wgs = h.num_roles;
    bp.dev_rep.timing = nullptr;
    bp.db_mode = true;
  }
}

bool AssemblyManager::rowOwnerUsable(std::string *why) const {
  auto fail = [&](const char *m) { if (why) *why = m; return false; };
  if (!ro_.ready) return fail("partition not built (unsupported element?)");
  if (ro_.num_general_blocks > 0) return fail("block has non-affine elements (general row-owner kernel not built yet)");
  for (const char *name : {"thermal diffusion", "specific heat", "density"})
    if (functions_.evaluate(name).kind != MHA_FUNC_CONSTANT) return fail("coefficient is not element-wise constant");
  return true;
}

// ---------------------------------------------------------------------------------------------
// general-element row-owner kernel
// ---------------------------------------------------------------------------------------------

RowBlocksDev AssemblyManager::generalRowBlocksDev() const {
  RowBlocksDev rb;
  rb.num_blocks = gro_.rb.num_blocks;
  rb.row_ptr = gro_.row_ptr.data();
  rb.rows = gro_.rows.data();
  rb.elem_ptr = gro_.elem_ptr.data();
  rb.elems = gro_.elems.data();
  rb.pair_ptr = gro_.pair_ptr.data();
  rb.pairs = gro_.pairs.data();
  rb.pair_off = gro_.pair_off.data();
  rb.row_len = gro_.row_len.data();
  rb.slot_ptr = gro_.slot_ptr.data();
  rb.seg_ptr = gro_.seg_ptr.data();
  rb.seg_acc = gro_.seg_acc.data();
  rb.seg_base = gro_.seg_base.data();
  rb.seg_len = gro_.seg_len.data();
  rb.lds_rows = gro_.rb.max_rows;
  rb.lds_elems = gro_.rb.max_elems;
  rb.lds_acc = gro_.rb.max_acc;
  rb.lds_pairs = gro_.rb.max_pairs;
  rb.lds_segs = gro_.rb.max_segs;
  return rb;
}

// Row blocks for the general row-owner kernel: Morton chunks of 2x2x2 (4x4) elements, at most 27 (25) touched elements,
// 256 pairs (16 matrix-core tiles) and an accumulator that leaves room for the per-element point data in LDS.  Any
// failure (a row beyond the caps, rows longer than 256 entries, LDS) leaves usable == false: AUTO keeps the element
// matrices + row gather, an explicit MHA_PATH_ROW_OWNER reports `why`.
void AssemblyManager::prepareGeneralRowOwner() {
  if (gro_.tried) return;
  gro_.tried = true;
  auto fail = [&](const std::string &m) { gro_.why = m; gro_.usable = false; };
  if (!single_hgrad_ || !thermal_general_row_owner_supported(dim_, order_, ref_.nq1))
    return fail("unsupported (dim, order, points/dir) for the general row-owner kernel");
  int max_row = 0;
  for (int r = 0; r < nrows_; ++r) max_row = std::max(max_row, h_rowptr_[r + 1] - h_rowptr_[r]);
  if (max_row > 256) return fail("CRS rows longer than 256 entries");
  RowBlockCaps caps = default_caps(dim_, n_);
  caps.max_elems = (dim_ == 3) ? 27 : 25;
  caps.max_rows = 128;
  caps.max_pairs = 256;
  caps.max_acc = 4352;
  if (const char *e = std::getenv("MHA_GRO_CHUNK")) caps.chunk_elems = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("MHA_GRO_MAXACC")) caps.max_acc = std::max(2 * max_row, std::atoi(e));
  std::vector<double> nodes(static_cast<size_t>(nelem_) * nnodes_ * dim_);
  d_nodes_.download(nodes.data());
  try {
    gro_.rb = build_row_blocks(dim_, nnodes_, nelem_, n_, nrows_, nodes.data(), h_lids_.data(), h_rowptr_.data(), caps,
                               has_fixed_ ? h_fixed_.data() : nullptr, 1);
  } catch (const Error &e) {
    return fail(e.what());
  }
  const RowBlocks &rb = gro_.rb;
  gro_.row_ptr.upload(rb.row_ptr);
  gro_.rows.upload(rb.rows);
  gro_.elem_ptr.upload(rb.elem_ptr);
  gro_.elems.upload(rb.elems);
  gro_.pair_ptr.upload(rb.pair_ptr);
  gro_.pairs.upload(rb.pairs);
  gro_.pair_off.upload(rb.pair_off);
  gro_.row_len.upload(rb.row_len);
  gro_.slot_ptr.upload(rb.slot_ptr);
  gro_.seg_ptr.upload(rb.seg_ptr);
  gro_.seg_acc.upload(rb.seg_acc);
  gro_.seg_base.upload(rb.seg_base);
  gro_.seg_len.upload(rb.seg_len);
  gro_.all_rows_covered = static_cast<int>(rb.rows.size()) == nrows_;
  gro_.lds_bytes = thermal_general_row_owner_lds(dim_, order_, ref_.nq1, generalRowBlocksDev());
  if (gro_.lds_bytes > size_t(160) * 1024) return fail("row blocks do not fit the LDS of the general row-owner kernel");
  gro_.slot.resize(std::max<size_t>(16, static_cast<size_t>(rb.slot_ptr.back())));
  launch_build_block_slots(blockDev(), generalRowBlocksDev(), gro_.slot.data(), 1, stream_);
  {  // block-major row ids of the touched elements' dofs (dof order): the kernel's gather is then two loads deep
    std::vector<int32_t> offs(n_), br(rb.elems.size() * static_cast<size_t>(n_));
    d_offsets_.download(offs.data());
    for (size_t k = 0; k < rb.elems.size(); ++k)
      for (int j = 0; j < n_; ++j) br[k * n_ + j] = h_lids_[static_cast<size_t>(rb.elems[k]) * n_ + offs[j]];
    gro_.blk_rows.upload(br);
    // block headers, 12 ints each: first touched element, T, first pair, NP, first row, NR, first run, NS,
    // slot-table offset / 16, slot-table uint4s
    std::vector<int32_t> hdr(static_cast<size_t>(rb.num_blocks) * 12, 0);
    for (int k = 0; k < rb.num_blocks; ++k) {
      int32_t *h = hdr.data() + static_cast<size_t>(k) * 12;
      h[0] = rb.elem_ptr[k]; h[1] = rb.elem_ptr[k + 1] - rb.elem_ptr[k];
      h[2] = rb.pair_ptr[k]; h[3] = rb.pair_ptr[k + 1] - rb.pair_ptr[k];
      h[4] = rb.row_ptr[k]; h[5] = rb.row_ptr[k + 1] - rb.row_ptr[k];
      h[6] = rb.seg_ptr[k]; h[7] = rb.seg_ptr[k + 1] - rb.seg_ptr[k];
      h[8] = static_cast<int32_t>(rb.slot_ptr[k] / 16); h[9] = static_cast<int32_t>((rb.slot_ptr[k + 1] - rb.slot_ptr[k]) / 16);
    }
    gro_.blk_hdr.upload(hdr);
  }
  gro_.usable = true;
  if (std::getenv("MHA_VERBOSE"))
    fprintf(stderr, "[mha] general row-owner: %d blocks, max rows %d elems %d pairs %d acc %d segs %d, LDS %zu B\n",
            rb.num_blocks, rb.max_rows, rb.max_elems, rb.max_pairs, rb.max_acc, rb.max_segs, gro_.lds_bytes);
}

void AssemblyManager::launchGeneralRowOwner(bool compute_jacobian, bool overwrite, double *res, double *crs_vals, bool ordered) {
  thermal *th = dynamic_cast<thermal *>(physics_.get());
  MHA_REQUIRE(th != nullptr, MHA_ERR_INVALID, "row-owner path: physics module is not thermal");
  const ThermalDev ph = th->device_params();
  RowOut out;
  out.res = res;
  out.vals = compute_jacobian ? crs_vals : nullptr;
  out.overwrite = overwrite ? 1 : 0;
  out.compute_jacobian = compute_jacobian ? 1 : 0;
  out.ordered = ordered ? 1 : 0;
  MHA_REQUIRE(!ordered || !compute_jacobian, MHA_ERR_INVALID, "the ordered residual sums exist in the residual-only pass");
  static const int wgs = [] { const char *m = std::getenv("MHA_GRO_WGS"); return m ? std::atoi(m) : 0; }();
  const int nwg = wgs > 0 ? wgs : current_device_num_cus();
  // profiling aid (env MHA_GRO_TIMING=<file>): cycles every wavefront spent in each phase of the last launch
  static const char *timing_file = std::getenv("MHA_GRO_TIMING");
  if (timing_file && gro_.timing.size() != static_cast<size_t>(nwg) * 8 * 8) gro_.timing.resize(static_cast<size_t>(nwg) * 8 * 8);
  launch_thermal_general_row_owner(dim_, order_, ref_.nq1, blockDev(), ph, generalRowBlocksDev(), gro_.slot.data(),
                                   gro_.blk_rows.data(), d_gp1d_.data(), gro_.blk_hdr.data(),
                                   timing_file ? gro_.timing.data() : nullptr, out, nwg, stream_);
  if (timing_file) {
    MHA_HIP(hipStreamSynchronize(stream_));
    std::vector<long long> t(gro_.timing.size());
    gro_.timing.download(t.data());
    if (FILE *f = fopen(timing_file, "wb")) { fwrite(t.data(), sizeof(long long), t.size(), f); fclose(f); }
  }
}

RowBlocksDev AssemblyManager::rowBlocksDev() const {
  RowBlocksDev rb;
  rb.num_blocks = ro_.rb.num_blocks;
  rb.row_ptr = ro_.row_ptr.data();
  rb.rows = ro_.rows.data();
  rb.row_off = ro_.row_off.data();
  rb.acc_size = ro_.acc_size.data();
  rb.elem_ptr = ro_.elem_ptr.data();
  rb.elems = ro_.elems.data();
  rb.pair_ptr = ro_.pair_ptr.data();
  rb.pairs = ro_.pairs.data();
  rb.pair_off = ro_.pair_off.data();
  rb.row_base = ro_.row_base.data();
  rb.row_len = ro_.row_len.data();
  rb.emask = ro_.emask.data();
  rb.epbase = ro_.epbase.data();
  rb.slot_ptr = ro_.slot_ptr.data();
  rb.seg_ptr = ro_.seg_ptr.data();
  rb.seg_acc = ro_.seg_acc.data();
  rb.seg_base = ro_.seg_base.data();
  rb.seg_len = ro_.seg_len.data();
  rb.lds_rows = ro_.rb.max_rows;
  rb.lds_elems = ro_.rb.max_elems;
  rb.lds_acc = ro_.rb.max_acc;
  rb.lds_pairs = ro_.rb.max_pairs;
  rb.lds_segs = ro_.rb.max_segs;
  return rb;
}

void AssemblyManager::launchRowOwner(bool compute_jacobian, bool overwrite, double *res, double *crs_vals, bool deterministic) {
  const RowBlocksDev rb = rowBlocksDev();
  AffineDev af;
  af.khat = ro_.khat.data();
  af.phi1d = ro_.phi.data();
  af.dphi1d = ro_.dphi.data();
  af.gw1d = ro_.gw.data();
  af.gp1d = ro_.gp.data();
  af.slot = ro_.slot.data();
  af.slot_bytes = ro_.slot_bytes;
  af.geo = ro_.geo.data();
  af.erec = ro_.erec.data();
  af.pair_off16 = ro_.pair_off16.data();
  af.slot_pair = ro_.slot_pair.data();
  RowOut out;
  out.res = res;
  out.vals = crs_vals;
  out.overwrite = overwrite ? 1 : 0;
  out.compute_jacobian = compute_jacobian ? 1 : 0;
  thermal *th = dynamic_cast<thermal *>(physics_.get());
  MHA_REQUIRE(th != nullptr, MHA_ERR_INVALID, "row-owner path: physics module is not thermal");
  const ThermalDev ph = th->device_params();
  if (deterministic) {
    // MHA_ASSEMBLE_DETERMINISTIC: Jacobian rows as fixed-order register sums, residual entries summed in pair order by
    // their owner (the general row-owner kernel, residual only) -- no atomics anywhere, one stream
    MHA_REQUIRE(!compute_jacobian || bpat_.usable, MHA_ERR_INVALID,
                "deterministic mode needs the block-pattern Jacobian kernel (" << bpat_.why << ")");
    prepareGeneralRowOwner();
    MHA_REQUIRE(gro_.usable, MHA_ERR_INVALID, "deterministic mode needs the general row-owner kernel: " << gro_.why);
    if (overwrite && !gro_.all_rows_covered) MHA_HIP(hipMemsetAsync(res, 0, sizeof(double) * nrows_, stream_));
    launchGeneralRowOwner(false, overwrite, res, nullptr, true);
    if (compute_jacobian) launch_block_pattern_jacobian(bpat_.dev, out, ph.time.alpha_u * ph.diff.amp, ph.time.alpha_t * ph.rho.amp * ph.cp.amp, stream_);
    return;
  }
  // K1 accumulates the residual with atomics: the fused zeroing becomes a (small) memset -- on the stream K1 runs on (with
  // the side stream below it no longer sits in front of the Jacobian kernels)
  static const int overlap_early = [] { const char *m = std::getenv("MHA_K1K2_OVERLAP"); return m ? std::atoi(m) : 1; }();
  const bool memset_on_side = overwrite && overlap_early && compute_jacobian;
  if (overwrite && !memset_on_side) MHA_HIP(hipMemsetAsync(res, 0, sizeof(double) * nrows_, stream_));
  // K1 (residual, VALU-bound) and K2 (Jacobian, latency-bound) write different arrays: K2 goes first on the
  // context's stream and K1 on a side stream, so that K1's workgroups fill the wave slots K2 leaves free
  // (0.743 -> 0.686 ms per assembly on config 2, profiles/r1_ab_k1k2_overlap.log); MHA_K1K2_OVERLAP=0 serialises them
  static const int overlap = [] { const char *m = std::getenv("MHA_K1K2_OVERLAP"); return m ? std::atoi(m) : 1; }();
  const double su = ph.time.alpha_u * ph.diff.amp, st = ph.time.alpha_t * ph.rho.amp * ph.cp.amp;
  auto residual = [&](hipStream_t s) {  // K1: one thread per element (default) or the 32-lanes-per-element form (MHA_K1=lanes)
    if (ro_.k1_thread) launch_thermal_affine_residual(dim_, order_, blockDev(), ph, ro_.geo.data(), ro_.tab1d, &ro_.k1_plan, res, ro_.max_abs_coord, s);
    else launch_thermal_affine_element(dim_, order_, ref_.nq1, blockDev(), ph, af, res, s);
  };
  auto jacobian = [&](hipStream_t s) {  // K2: pattern GEMMs on the matrix cores when the rows group, row blocks otherwise
    if (bpat_.usable && bpat_.db_mode && out.overwrite && !bpat_.dev.timing && (reinterpret_cast<uintptr_t>(out.vals) & 127u) == 0) {  // (the copy's chunks sit on 128-byte lines of the caller's array)
      // geometry-database mode: the representatives' rows, then their copies (same stream: ordered)
      launch_block_pattern_jacobian(bpat_.dev_rep, out, su, st, s);
      launch_replicate_runs(bpat_.copy_chunks.data(), bpat_.copy_runs, out.vals, s);
      last_db_mode_ = 1;
    } else if (bpat_.usable) {
      last_db_mode_ = 0;
      launch_block_pattern_jacobian(bpat_.dev, out, su, st, s);
      if (bpat_.dev.timing) {  // profiling aid: wall-clock stamps of every wavefront of the last launch -> $MHA_BP_TIMING
        MHA_HIP(hipStreamSynchronize(s));
        std::vector<long long> t(bpat_.timing.size());
        bpat_.timing.download(t.data());
        if (FILE *f = fopen(std::getenv("MHA_BP_TIMING"), "wb")) { fwrite(t.data(), sizeof(long long), t.size(), f); fclose(f); }
      }
    } else {
      launch_row_owner_jacobian(dim_, n_, rb, af, out, su, st, s);
    }
  };
  if (overlap && compute_jacobian) {
    if (!side_stream_) {
      MHA_HIP(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
      MHA_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
      MHA_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    }
    MHA_HIP(hipEventRecord(ev_fork_, stream_));
    MHA_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
    if (memset_on_side) MHA_HIP(hipMemsetAsync(res, 0, sizeof(double) * nrows_, side_stream_));
    if (overlap == 2) residual(side_stream_);
    jacobian(stream_);
    if (overlap != 2) residual(side_stream_);
    MHA_HIP(hipEventRecord(ev_join_, side_stream_));
    MHA_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
    return;
  }
  residual(stream_);
  if (compute_jacobian) jacobian(stream_);
}

int64_t AssemblyManager::info(const std::string &key) const {
  if (key == "num_elems") return nelem_;
  if (key == "num_rows") return nrows_;
  if (key == "nnz") return has_graph_ ? h_rowptr_[nrows_] : 0;
  if (key == "dofs_per_elem") return n_;
  if (key == "num_ip") return nq_;
  if (key == "last_path") return last_path_;
  if (key == "jacobian_database_mode") return last_db_mode_;  // the last affine row-owner Jacobian replicated one block per pattern
  if (key == "affine_shapes") return ro_.ready ? ro_.k1_plan.num_shapes : 0;  // distinct geometry records behind the residual kernel's database index
  if (key == "porous_direct") return last_porous_direct_;  // the last row-gather assembly of a porousMixed block stored straight into the CRS
  if (key == "workset_size") return wkset_.maxElem;
  if (key == "row_blocks") return ro_.ready ? ro_.rb.num_blocks : 0;
  if (key == "row_owner_kind") return last_row_owner_kind_;  // of the last assembly: 1 affine kernels, 2 general-element kernel
  if (key == "general_row_blocks") return gro_.usable ? gro_.rb.num_blocks : 0;
  if (key == "general_row_owner_lds_bytes") return gro_.usable ? static_cast<int64_t>(gro_.lds_bytes) : 0;
  if (key == "block_patterns") return bpat_.usable ? bpat_.num_patterns : 0;
  if (key == "block_pattern_roles") return bpat_.usable ? bpat_.num_roles : 0;
  if (key == "block_pattern_blocks") return bpat_.usable ? bpat_.num_blocks : 0;
  if (key == "block_pattern_mfma") return bpat_.usable ? bpat_.mfma_per_assembly : 0;
  if (key == "num_affine_elems") return ro_.ready ? ro_.num_affine_elems : -1;
  if (key == "row_block_max_rows") return ro_.rb.max_rows;
  if (key == "row_block_max_elems") return ro_.rb.max_elems;
  if (key == "row_block_max_acc") return ro_.rb.max_acc;
  if (key == "row_block_max_pairs") return ro_.rb.max_pairs;
  if (key == "row_owner_lds_bytes")
    return ro_.ready ? static_cast<int64_t>(row_owner_jacobian_lds(rowBlocksDev(), n_, ro_.slot_bytes)) : 0;
  throw Error(MHA_ERR_INVALID, "unknown info key '" + key + "'");
}

}  // namespace mha
