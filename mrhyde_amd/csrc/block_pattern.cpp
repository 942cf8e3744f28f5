// block_pattern.cpp -- see block_pattern.hpp.
#include "block_pattern.hpp"

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <unordered_map>

#include "common.hpp"

namespace mha {
namespace {

// role record (kBpRoleInts ints)
enum { R_EREC_LO = 0, R_EREC_HI, R_ESTRIDE, R_ROWB_LO, R_ROWB_HI, R_NRUNS, R_NBLOCKS, R_COST, R_WOFF_LO, R_WOFF_HI,
       R_WDOUBLES, R_PATTERN, R_NELEMS,
       R_IMG,        // image roles: doubles of a block's LDS image (0: rows are stored straight from the registers)
       R_WLDS,       // doubles of the role's W image that live in LDS (the rest belongs to REGW units)
       R_NCHUNK, R_CHUNK_LO, R_CHUNK_HI };
// part header (kBpHdrInts ints)
enum { H_WOFF = 0, H_KS, H_NCT, H_LEN, H_CT0, H_NTILE, H_FLAGS, H_CLASS, H_MASK0, H_MASK1, H_MASK2 };  // flags: 1 rows of fixed dofs, 2 a plain tail tile follows the group, 4 / 8 / 16 trim class (see the mask loop), 32 W in registers (REGW)

uint64_t fnv1a(const std::vector<uint8_t> &b) {
  uint64_t h = 1469598103934665603ull;
  for (uint8_t c : b) { h ^= c; h *= 1099511628211ull; }
  return h;
}

inline void put16(std::vector<uint8_t> &o, int v) { o.push_back(v & 0xff); o.push_back((v >> 8) & 0xff); }
inline void put24(std::vector<uint8_t> &o, int v) { put16(o, v); o.push_back((v >> 16) & 0xff); }

// Everything the kernel's static tables depend on, as bytes: rows (length, fixed), touched elements (count) and per
// contribution pair the owned row, the element, the local dof and the slot of every column of that element.
void block_signature(const RowBlocks &rb, int b, int n, const int32_t *rowptr, const uint8_t *fixed, const uint8_t *slot,
                     int slot_bytes, std::vector<uint8_t> &out) {
  out.clear();
  const int r0 = rb.row_ptr[b], R = rb.row_ptr[b + 1] - r0;
  const int t0 = rb.elem_ptr[b], T = rb.elem_ptr[b + 1] - t0;
  put16(out, R);
  put16(out, T);
  for (int o = 0; o < R; ++o) {
    const int r = rb.rows[r0 + o];
    put24(out, rowptr[r + 1] - rowptr[r]);
    out.push_back((fixed && fixed[r]) ? 1 : 0);
    out.push_back((o > 0 && rb.rows[r0 + o - 1] + 1 == r) ? 0 : 1);  // 1: the row starts a new run of consecutive rows
  }
  for (int p = rb.pair_ptr[b]; p < rb.pair_ptr[b + 1]; ++p) {
    const uint32_t pk = rb.pairs[p];
    const int o = pk >> 16, t = (pk >> 8) & 0xff, si = pk & 0xff;
    put16(out, o);
    out.push_back(static_cast<uint8_t>(t));
    out.push_back(static_cast<uint8_t>(si));
    const uint8_t *s = slot + (static_cast<size_t>(rb.elems[t0 + t]) * n + si) * n * slot_bytes;
    out.insert(out.end(), s, s + static_cast<size_t>(n) * slot_bytes);
  }
}

struct Inc { int si, t; const uint8_t *slots; };

struct RowClass {
  int len = 0, ni = 0;
  bool fixed = false;
  std::vector<int> inst_row;             // block-local rows of the instances
  std::vector<std::vector<int>> inst_t;  // per instance: block-local element of incidence k
  std::vector<int> si;                   // local dof of incidence k
  std::vector<std::vector<int>> slots;   // per incidence k: slot of every column of that element
  int ks = 0, nct = 0, stride = 0, w_doubles = 0;
  int bin = 0, w_off = 0;                // role (LDS bin) and offset inside it
  bool regw = false;                     // image roles: units of two tiles with W in registers
};

inline int stride_for(int nct) { return (nct % 2 == 1) ? 16 * nct : 16 * nct + 16; }

}  // namespace

BlockPatternPlan build_block_patterns(const RowBlocks &rb, int n, int nsym, const int32_t *rowptr, const uint8_t *fixed,
                                      const void *elem_slot, int slot_bytes, const double *khat, int num_cus,
                                      size_t lds_budget_bytes, int max_patterns, int seg_blocks) {
  BlockPatternPlan pl;
  pl.nsym = nsym;
  pl.ke = nsym + 1;
  const int ke = pl.ke;
  const uint8_t *slot = static_cast<const uint8_t *>(elem_slot);
  auto fail = [&](const std::string &m) { pl.usable = false; pl.why = m; return pl; };
  if (rb.num_blocks == 0) return fail("no row blocks");
  if (ke > kBpRecDoubles) return fail("more geometry components than an element record holds");
  const size_t nn = static_cast<size_t>(n) * n;

  // ---- 1. pattern of every block ----
  std::unordered_map<uint64_t, std::vector<int32_t>> by_hash;  // hash -> pattern ids (collisions resolved by comparing bytes)
  std::vector<int32_t> rep;                                    // representative block of each pattern
  std::vector<std::vector<int32_t>> members;
  std::vector<uint8_t> cur, other;
  for (int b = 0; b < rb.num_blocks; ++b) {
    block_signature(rb, b, n, rowptr, fixed, slot, slot_bytes, cur);
    std::vector<int32_t> &ids = by_hash[fnv1a(cur)];
    int found = -1;
    for (int id : ids) {
      block_signature(rb, rep[id], n, rowptr, fixed, slot, slot_bytes, other);
      if (other == cur) { found = id; break; }
    }
    if (found < 0) {
      found = static_cast<int>(rep.size());
      if (found >= max_patterns) return fail("row blocks share too few assembly patterns (unstructured numbering)");
      rep.push_back(b);
      members.emplace_back();
      ids.push_back(found);
    }
    members[found].push_back(b);
  }
  pl.num_patterns = static_cast<int>(rep.size());

  // ---- 2. per pattern: classes, tiles, parts, roles ----
  struct RoleBuild {
    int pattern, bin, R, T, nruns;
    int w_doubles = 0;
    int img_doubles = 0, w_lds = 0, nchunk = 0;  // image roles (block_pattern.hpp)
    int64_t chunk_off = 0;
    int64_t w_off = 0;             // start of the role's LDS image inside pl.w
    int64_t cost = 0;              // per block
    std::vector<std::vector<int32_t>> wave_parts;  // [kBpWaves] -> part indices (global)
  };
  std::vector<RoleBuild> roles;
  const int budget_doubles = static_cast<int>(lds_budget_bytes / sizeof(double));

  for (int pat = 0; pat < pl.num_patterns; ++pat) {
    const int b = rep[pat];
    const int r0 = rb.row_ptr[b], R = rb.row_ptr[b + 1] - r0;
    const int t0 = rb.elem_ptr[b], T = rb.elem_ptr[b + 1] - t0;
    if ((T + 1) * kBpRecDoubles > 512) return fail("a block touches more elements than the record loader fetches (63)");
    pl.max_rec_doubles = std::max(pl.max_rec_doubles, (T + 1) * kBpRecDoubles);
    // runs of consecutive rows: contiguous in the CRS value array, so one offset per (block, run) locates every row
    std::vector<int> run_of(R), run_first;
    for (int o = 0; o < R; ++o) {
      if (o == 0 || rb.rows[r0 + o - 1] + 1 != rb.rows[r0 + o]) run_first.push_back(o);
      run_of[o] = static_cast<int>(run_first.size()) - 1;
      if (rowptr[rb.rows[r0 + o]] - rowptr[rb.rows[r0 + run_first.back()]] >= (1 << 20)) return fail("a run of rows is longer than 2^20 entries");
    }
    const int nruns = static_cast<int>(run_first.size());
    if (nruns > kBpSegInts || nruns >= 2048) return fail("the rows of a block form too many runs");
    // incidences of every owned row
    std::vector<std::vector<Inc>> inc(R);
    for (int p = rb.pair_ptr[b]; p < rb.pair_ptr[b + 1]; ++p) {
      const uint32_t pk = rb.pairs[p];
      const int o = pk >> 16, t = (pk >> 8) & 0xff, si = pk & 0xff;
      inc[o].push_back({si, t, slot + (static_cast<size_t>(rb.elems[t0 + t]) * n + si) * n * slot_bytes});
    }
    std::map<std::vector<uint8_t>, int> class_of;
    std::vector<RowClass> classes;
    std::vector<uint8_t> key;
    for (int o = 0; o < R; ++o) {
      const int r = rb.rows[r0 + o];
      const int len = rowptr[r + 1] - rowptr[r];
      const bool fx = fixed && fixed[r];
      std::vector<Inc> &v = inc[o];
      const size_t sb = static_cast<size_t>(n) * slot_bytes;
      std::sort(v.begin(), v.end(), [&](const Inc &a, const Inc &c) {
        if (a.si != c.si) return a.si < c.si;
        return std::memcmp(a.slots, c.slots, sb) < 0;
      });
      key.clear();
      put24(key, len);
      key.push_back(fx ? 1 : 0);
      if (!fx)
        for (const Inc &i : v) { key.push_back(static_cast<uint8_t>(i.si)); key.insert(key.end(), i.slots, i.slots + sb); }
      auto it = class_of.find(key);
      int c;
      if (it == class_of.end()) {
        c = static_cast<int>(classes.size());
        class_of[key] = c;
        RowClass rc;
        rc.len = len;
        rc.fixed = fx;
        rc.ni = fx ? 0 : static_cast<int>(v.size());
        if (!fx)
          for (const Inc &i : v) {
            rc.si.push_back(i.si);
            std::vector<int> s(n);
            for (int sj = 0; sj < n; ++sj)
              s[sj] = slot_bytes == 1 ? i.slots[sj] : reinterpret_cast<const uint16_t *>(i.slots)[sj];
            rc.slots.push_back(std::move(s));
          }
        rc.nct = (len + 15) / 16;
        rc.stride = stride_for(rc.nct);
        rc.ks = (rc.ni * ke + 3) / 4;
        if (rc.ks > kBpMaxKSteps) return fail("a dof is shared by more elements than the pattern kernel holds");
        rc.w_doubles = rc.ks * 4 * rc.stride;
        if (rc.w_doubles > budget_doubles) return fail("a pattern matrix does not fit the LDS");
        classes.push_back(std::move(rc));
      } else {
        c = it->second;
      }
      classes[c].inst_row.push_back(o);
      std::vector<int> ts;
      if (!fx)
        for (const Inc &i : v) ts.push_back(i.t);
      classes[c].inst_t.push_back(std::move(ts));
    }
    // ---- image form?  (block_pattern.hpp)  The block's runs, each in a slot of whole 16-entry lines with room for its
    // global offset mod 16; all W that stays in LDS in ONE bin beside the image; at most kBpStreamWaves units, one per
    // wavefront; no rows of fixed dofs (their zeros would need a unit kind of their own) ----
    std::vector<int> run_len(nruns, 0), run_slot(nruns + 1, 0);
    for (int o = 0; o < R; ++o) run_len[run_of[o]] += rowptr[rb.rows[r0 + o] + 1] - rowptr[rb.rows[r0 + o]];
    for (int r = 0; r < nruns; ++r) run_slot[r + 1] = run_slot[r] + (run_len[r] + 15 + 15) / 16 * 16;
    const int img_doubles = run_slot[nruns];
    // (opt-in, MHA_BP_IMAGE=1: at the end of round 3 the image kernel is correct but slower than the direct stores --
    // 253 us for the interior blocks of config 2 against ~235 -- see DESIGN.md)
    bool image = std::getenv("MHA_BP_IMAGE") && std::string(std::getenv("MHA_BP_IMAGE")) == "1" && !classes.empty();
    {
      int w_lds = 0, units = 0, nchunk = 0;
      for (const RowClass &rc : classes) { w_lds += rc.w_doubles; image = image && !rc.fixed; }
      image = image && w_lds <= budget_doubles;  // (in accumulate mode the role runs the plain form with all of W in LDS)
      for (int r = 0; r < nruns; ++r) nchunk += (run_len[r] + 15 + 127) / 128;
      image = image && nchunk <= 64 * kBpStreamWaves && img_doubles < (1 << 16) && nruns < 256;
      while (image && w_lds + img_doubles + 64 > budget_doubles) {  // largest W first into registers
        int best = -1;
        for (size_t c = 0; c < classes.size(); ++c)
          if (!classes[c].regw && classes[c].nct % 2 == 0 && classes[c].ks * 2 <= 28 && (best < 0 || classes[c].w_doubles > classes[best].w_doubles))
            best = static_cast<int>(c);
        if (best < 0) { image = false; break; }
        classes[best].regw = true;
        w_lds -= classes[best].w_doubles;
      }
      for (const RowClass &rc : classes) {
        const int ntile = (static_cast<int>(rc.inst_row.size()) + 15) / 16;
        const int ng = rc.nct / 4, rem = rc.nct % 4;
        units += ntile * (rc.regw ? rc.nct / 2 : ng + ((rem >= 2 || (ng == 0 && rem == 1)) ? 1 : 0));
      }
      if (std::getenv("MHA_BP_STATS")) {
        int nfixed_rows = 0, nfixed_cls = 0, prod_units = 0;
        for (const RowClass &rc : classes) {
          const int ntile = (static_cast<int>(rc.inst_row.size()) + 15) / 16;
          const int ng = rc.nct / 4, rem = rc.nct % 4;
          if (rc.fixed) { nfixed_rows += static_cast<int>(rc.inst_row.size()); ++nfixed_cls; }
          else prod_units += ntile * (rc.regw ? rc.nct / 2 : ng + ((rem >= 2 || (ng == 0 && rem == 1)) ? 1 : 0));
        }
        fprintf(stderr, "pattern %d: %zu blocks, R %d T %d runs %d image %d doubles, W %d doubles, classes %zu (fixed %d, %d rows), product units %d, image %d\n",
                pat, members[pat].size(), R, T, nruns, img_doubles, w_lds, classes.size(), nfixed_cls, nfixed_rows, prod_units, int(image));
      }
      image = image && units <= kBpStreamWaves;
      if (!image)
        for (RowClass &rc : classes) rc.regw = false;
    }
    // LDS bins: first fit, largest W first (an image role: one bin, the REGW classes behind the part that is copied to LDS)
    std::vector<int> order(classes.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
      if (classes[a].regw != classes[c].regw) return !classes[a].regw;
      return classes[a].w_doubles > classes[c].w_doubles;
    });
    std::vector<int> bin_fill;
    int w_lds_image = 0;
    for (int c : order) {
      RowClass &rc = classes[c];
      int bin = -1;
      if (image) {
        if (bin_fill.empty()) bin_fill.push_back(0);
        bin = 0;
      }
      for (size_t k = 0; k < bin_fill.size() && bin < 0; ++k)
        if (bin_fill[k] + rc.w_doubles <= budget_doubles) bin = static_cast<int>(k);
      if (bin < 0) { bin = static_cast<int>(bin_fill.size()); bin_fill.push_back(0); }
      rc.bin = bin;
      rc.w_off = bin_fill[bin];
      bin_fill[bin] += rc.w_doubles;
      if (image && !rc.regw) w_lds_image = bin_fill[bin];
    }
    if (bin_fill.empty()) bin_fill.push_back(0);
    const int first_role = static_cast<int>(roles.size());
    std::vector<size_t> role_woff(bin_fill.size());
    for (size_t k = 0; k < bin_fill.size(); ++k) {
      role_woff[k] = pl.w.size();
      pl.w.resize(pl.w.size() + 2 * static_cast<size_t>(bin_fill[k]), 0.0);  // stiffness rows | mass rows
      pl.max_w_doubles = std::max(pl.max_w_doubles, image ? w_lds_image + img_doubles + 64 : bin_fill[k]);  // (+ 64: the trash slot behind the image)
      RoleBuild rbld;
      if (image) {
        rbld.img_doubles = img_doubles;
        rbld.w_lds = w_lds_image;
        rbld.chunk_off = static_cast<int64_t>(pl.chunk_tab.size() / kBpChunkInts);
        for (int r = 0; r < nruns; ++r)
          for (int kk = 0; kk < (run_len[r] + 15 + 127) / 128; ++kk) {
            const int32_t ch[kBpChunkInts] = {r, run_slot[r] + 128 * kk, 128 * kk, run_len[r]};
            pl.chunk_tab.insert(pl.chunk_tab.end(), ch, ch + kBpChunkInts);
            ++rbld.nchunk;
          }
        ++pl.num_image_roles;
      }
      rbld.pattern = pat;
      rbld.bin = static_cast<int>(k);
      rbld.R = R;
      rbld.T = T;
      rbld.nruns = nruns;
      rbld.w_doubles = bin_fill[k];
      rbld.w_off = static_cast<int64_t>(role_woff[k]);
      rbld.wave_parts.assign(kBpWaves, {});
      roles.push_back(std::move(rbld));
    }
    // W images
    for (RowClass &rc : classes) {
      if (rc.fixed) continue;
      double *W = pl.w.data() + role_woff[rc.bin] + rc.w_off;
      const size_t mass = static_cast<size_t>(bin_fill[rc.bin]);  // offset of the mass half of the role's image
      for (int k = 0; k < rc.ni; ++k)
        for (int sj = 0; sj < n; ++sj)
          for (int m = 0; m < ke; ++m)
            W[(m == nsym ? mass : 0) + static_cast<size_t>(k * ke + m) * rc.stride + rc.slots[k][sj]] +=
                khat[m * nn + static_cast<size_t>(rc.si[k]) * n + sj];
    }
    // tiles of 16 instances, cut into UNITS of up to four column tiles (what one wavefront computes and stores per
    // block: the products of a unit are transposed so that every store writes 64 consecutive entries of one row); a
    // single left-over tile rides with the group before it, two or three form a unit of their own.  All wavefronts of
    // a workgroup work on the same block at the same time (the rows of a block are neighbours in memory: written
    // within microseconds of each other they leave the L2 as whole DRAM pages), so the units -- not blocks -- are what
    // is dealt to the wavefronts, heaviest first.
    struct Tile { int cls, tile; int64_t cost; };
    std::vector<std::vector<Tile>> tiles(bin_fill.size());
    for (size_t c = 0; c < classes.size(); ++c) {
      const RowClass &rc = classes[c];
      const int ntile = (static_cast<int>(rc.inst_row.size()) + 15) / 16;
      for (int t = 0; t < ntile; ++t)
        tiles[rc.bin].push_back({static_cast<int>(c), t, static_cast<int64_t>(rc.nct) * (rc.ks + 4)});
    }
    for (size_t k = 0; k < bin_fill.size(); ++k) {
      RoleBuild &role = roles[first_role + k];
      int64_t total = 0;
      for (const Tile &t : tiles[k]) total += t.cost;
      role.cost = total;
      struct PartBuild { int cls, tile, ct0, ntile, tail; double cost; };
      std::vector<PartBuild> parts;
      for (const Tile &t : tiles[k]) {
        const RowClass &rc = classes[t.cls];
        if (rc.regw) {  // units of one tile pair, W in registers
          for (int g = 0; g < rc.nct / 2; ++g) parts.push_back({t.cls, t.tile, 2 * g, 2, 0, static_cast<double>(2 * (rc.ks + 1))});
          continue;
        }
        const int ng = rc.nct / 4, rem = rc.nct % 4;
        for (int g = 0; g < ng; ++g) {
          const int tail = (g == ng - 1 && rem == 1) ? 1 : 0;
          parts.push_back({t.cls, t.tile, 4 * g, 4, tail, static_cast<double>((4 + tail) * (rc.ks + 1))});
        }
        if (ng == 0 && rem == 1) parts.push_back({t.cls, t.tile, 0, 0, 1, static_cast<double>(rc.ks + 1)});
        if (rem >= 2) parts.push_back({t.cls, t.tile, 4 * ng, rem, 0, static_cast<double>(rem * (rc.ks + 1) + 2)});
      }
      // cost of a product unit = its non-zero 4 x 16 blocks of W (the products the kernel's trimmed forms issue)
      for (PartBuild &pb : parts) {
        const RowClass &rc = classes[pb.cls];
        if (rc.fixed) continue;
        const double *Wk = pl.w.data() + role_woff[rc.bin] + rc.w_off, *Wm = Wk + bin_fill[rc.bin];
        int nzb = 0;
        for (int sidx = 0; sidx < rc.ks; ++sidx)
          for (int q = 0; q < pb.ntile + pb.tail; ++q) {
            bool any = false;
            for (int r = 4 * sidx; r < 4 * sidx + 4 && !any; ++r)
              for (int lc = 0; lc < 16 && !any; ++lc) {
                const int cc = bp_unit_col(pb.ct0, pb.ntile, q, lc);
                any = Wk[static_cast<size_t>(r) * rc.stride + cc] != 0.0 || Wm[static_cast<size_t>(r) * rc.stride + cc] != 0.0;
              }
            nzb += any ? 1 : 0;
          }
        pb.cost = nzb + 1;
      }
      std::stable_sort(parts.begin(), parts.end(), [](const PartBuild &a, const PartBuild &c) { return a.cost > c.cost; });
      // units of fixed rows (zeros, no products) go to wavefronts that carry no products when there are any: the
      // kernel's fast forms take wavefronts whose units are all of one kind
      std::stable_partition(parts.begin(), parts.end(), [&](const PartBuild &a) { return !classes[a.cls].fixed; });
      std::vector<double> load(kBpWaves, 0.0), simd_load(4, 0.0);
      std::vector<char> has_products(kBpWaves, 0);
      for (const PartBuild &pb : parts) {
        int wv = -1;
        if (classes[pb.cls].fixed) {
          for (int w = 0; w < kBpWaves; ++w)
            if (!has_products[w] && (wv < 0 || load[w] < load[wv])) wv = w;
        }
        if (wv < 0) {
          // a product unit: a wavefront of its own while there are free ones, on the SIMD (wavefront id mod 4: the four
          // SIMDs of a CU take a workgroup's wavefronts round robin) whose matrix pipe has the least work so far
          const int w0 = role.img_doubles > 0 ? 1 : 0;  // image roles: wave 0 is the record loader
          for (int w = w0; w < kBpWaves; ++w)
            if (load[w] == 0.0 && (wv < 0 || simd_load[w % 4] < simd_load[wv % 4])) wv = w;
          if (wv < 0) wv = static_cast<int>(std::min_element(load.begin() + w0, load.end()) - load.begin());
          simd_load[wv % 4] += pb.cost;
        }
        load[wv] += pb.cost;
        if (!classes[pb.cls].fixed) has_products[wv] = 1;
        const RowClass &rc = classes[pb.cls];
        const int pidx = pl.num_parts++;
        role.wave_parts[wv].push_back(pidx);
        int32_t hdr[kBpHdrInts] = {0};
        hdr[H_WOFF] = rc.w_off;
        hdr[H_KS] = rc.ks;
        hdr[H_NCT] = rc.nct;
        hdr[H_LEN] = rc.len;
        hdr[H_CT0] = pb.ct0;
        hdr[H_NTILE] = pb.ntile;
        hdr[H_FLAGS] = (rc.fixed ? 1 : 0) | (pb.tail ? 2 : 0) | (rc.regw ? 32 : 0);
        hdr[H_CLASS] = pl.num_classes + pb.cls;
        if (!rc.fixed) {  // which 4 x 16 blocks of the unit's part of W hold anything: the kernel skips the products of the others
          const double *Wk = pl.w.data() + role_woff[rc.bin] + rc.w_off, *Wm = Wk + bin_fill[rc.bin];
          for (int sidx = 0; sidx < rc.ks; ++sidx)
            for (int q = 0; q < pb.ntile + pb.tail; ++q) {
              bool any = false;
              for (int r = 4 * sidx; r < 4 * sidx + 4 && !any; ++r)
                for (int lc = 0; lc < 16 && !any; ++lc) {
                  const int cc = bp_unit_col(pb.ct0, pb.ntile, q, lc);
                  any = Wk[static_cast<size_t>(r) * rc.stride + cc] != 0.0 || Wm[static_cast<size_t>(r) * rc.stride + cc] != 0.0;
                }
              const int bit = sidx * 5 + q;
              if (any) hdr[H_MASK0 + bit / 32] |= static_cast<int32_t>(1u << (bit % 32));
            }
          // trim class of a unit of two tile pairs (flags bits 2..4, read by the kernel's specialised form): the first pair
          // (classes 1, 3) or the second (2, 4) has only zero blocks in the second (1, 4) or first (2, 3) half of the k-steps
          const int nq = pb.ntile + pb.tail;
          if (nq >= 2 && rc.ks >= 2) {
            bool t[5] = {false, true, true, true, true};
            if (pb.ntile == 2 && rc.regw) t[2] = t[4] = false;  // one pair: classes 1 (second half zero) and 3 (first half zero)
            else if (pb.ntile != 4 || pb.tail) t[1] = t[2] = t[3] = t[4] = false;
            for (int sidx = 0; sidx < rc.ks; ++sidx)
              for (int q = 0; q < nq; ++q) {
                const int bit = sidx * 5 + q;
                const bool nz = (static_cast<uint32_t>(hdr[H_MASK0 + bit / 32]) >> (bit % 32)) & 1u;
                if (!nz) continue;
                const bool second = sidx >= rc.ks / 2;
                if (q < 2 && second) t[1] = false;
                if (q >= 2 && !second) t[2] = false;
                if (q < 2 && !second) t[3] = false;
                if (q >= 2 && second) t[4] = false;
              }
            for (int k = 1; k <= 4; ++k)
              if (t[k]) { hdr[H_FLAGS] |= k << 2; break; }
            if (std::getenv("MHA_BP_PRINT")) {  // debugging aid: the zero-block mask of every unit
              fprintf(stderr, "unit ks %d ct0 %d nq %d:", rc.ks, pb.ct0, nq);
              for (int sidx = 0; sidx < rc.ks; ++sidx) {
                fprintf(stderr, " ");
                for (int q = 0; q < nq; ++q) {
                  const int bit = sidx * 5 + q;
                  fprintf(stderr, "%d", (static_cast<uint32_t>(hdr[H_MASK0 + bit / 32]) >> (bit % 32)) & 1u);
                }
              }
              fprintf(stderr, "\n");
            }
          }
        }
        pl.part_hdr.insert(pl.part_hdr.end(), hdr, hdr + kBpHdrInts);
        const size_t base = pl.part_lane.size();
        pl.part_lane.resize(base + static_cast<size_t>(kBpLaneRows) * 64, 0);
        int32_t *L = pl.part_lane.data() + base;
        const int ninst = static_cast<int>(rc.inst_row.size());
        const int zero_off = T * kBpRecDoubles;
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kk = lane >> 4;
          const int inst = pb.tile * 16 + i;
          for (int s = 0; s < kBpMaxKSteps; ++s) {
            const int kidx = 4 * s + kk;
            int off = zero_off;
            if (inst < ninst && !rc.fixed && kidx < rc.ni * ke) {
              const int kel = kidx / ke, m = kidx % ke;
              off = rc.inst_t[inst][kel] * kBpRecDoubles + m;
            }
            L[s * 64 + lane] = off;
          }
          for (int t = 0; t < 4; ++t) {
            const int r = kk + 4 * t;  // result register t of this lane belongs to tile row (lane >> 4) + 4 t
            const int instr = pb.tile * 16 + r;
            // run of consecutive rows the result row lies in and its CRS offset relative to the run's first row
            // (both part of the pattern), -1 = no row
            int packed = -1;
            if (instr < ninst) {
              const int o = rc.inst_row[instr];
              packed = (run_of[o] << 20) | (rowptr[rb.rows[r0 + o]] - rowptr[rb.rows[r0 + run_first[run_of[o]]]]);
            }
            L[(16 + t) * 64 + lane] = packed;
          }
          {  // the same for tile row (lane & 15): what the row-contiguous stores of the kernel consult
            const int instr = pb.tile * 16 + i;
            int packed = -1;
            if (instr < ninst) {
              const int o = rc.inst_row[instr];
              packed = (run_of[o] << 20) | (rowptr[rb.rows[r0 + o]] - rowptr[rb.rows[r0 + run_first[run_of[o]]]]);
            }
            L[20 * 64 + lane] = packed;
            L[21 * 64 + lane] = packed < 0 ? -1 : run_slot[packed >> 20] + (packed & 0xfffff);
          }
        }
      }
    }
    // wave 0 also fetches the element records, which the kernel builds only into its shapes of at most three chains:
    // give it such a unit (from the same SIMD if there is one: wavefront id mod 4)
    for (size_t k = 0; k < bin_fill.size(); ++k) {
      RoleBuild &role = roles[first_role + k];
      auto chains = [&](int wv) {  // of a wavefront that owns exactly one product unit, else 0
        if (role.wave_parts[wv].size() != 1) return 0;
        const int32_t *h = &pl.part_hdr[static_cast<size_t>(role.wave_parts[wv][0]) * kBpHdrInts];
        return (h[H_FLAGS] & 1) ? 0 : h[H_NTILE] + ((h[H_FLAGS] & 2) ? 1 : 0);
      };
      // ... or better none at all: a loader that stores nothing never waits for write acknowledgements (kernel:
      // run_loader_only), so an idle wavefront changes places with wave 0
      if (!role.wave_parts[0].empty()) {
        int idle = -1;
        for (int wv = kBpWaves - 1; wv >= 1 && idle < 0; --wv)
          if (role.wave_parts[wv].empty()) idle = wv;
        if (idle > 0 && (role.img_doubles > 0 || std::getenv("MHA_BP_LOADER_IDLE"))) { std::swap(role.wave_parts[0], role.wave_parts[idle]); continue; }
        if (role.img_doubles > 0) return fail("an image role has no free wavefront for the record loader");
      } else {
        continue;
      }
      if (chains(0) <= 3) continue;
      int best = -1;
      for (int wv = 1; wv < kBpWaves; ++wv) {
        const int nc = chains(wv);
        if (nc < 1 || nc > 3) continue;
        if (best < 0 || (wv % 4 == 0 && best % 4 != 0)) best = wv;
      }
      if (best > 0) std::swap(role.wave_parts[0], role.wave_parts[best]);
    }
    pl.num_classes += static_cast<int>(classes.size());
  }

  // ---- 3. (workgroups are cut after the block tables: step 5) ----
  pl.num_roles = static_cast<int>(roles.size());

  // ---- 4. role records, block-major tables ----
  pl.role.assign(static_cast<size_t>(pl.num_roles) * kBpRoleInts, 0);
  pl.part_ptr.assign(static_cast<size_t>(pl.num_roles) * (kBpWaves + 1), 0);
  // parts must be contiguous per (role, wave): renumber
  {
    std::vector<int32_t> new_hdr, new_lane;
    new_hdr.reserve(pl.part_hdr.size());
    new_lane.reserve(pl.part_lane.size());
    int next = 0;
    for (int k = 0; k < pl.num_roles; ++k)
      for (int wv = 0; wv <= kBpWaves; ++wv) {
        pl.part_ptr[static_cast<size_t>(k) * (kBpWaves + 1) + wv] = next;
        if (wv == kBpWaves) break;
        for (int pidx : roles[k].wave_parts[wv]) {
          new_hdr.insert(new_hdr.end(), pl.part_hdr.begin() + static_cast<size_t>(pidx) * kBpHdrInts,
                         pl.part_hdr.begin() + static_cast<size_t>(pidx + 1) * kBpHdrInts);
          new_lane.insert(new_lane.end(), pl.part_lane.begin() + static_cast<size_t>(pidx) * kBpLaneRows * 64,
                          pl.part_lane.begin() + static_cast<size_t>(pidx + 1) * kBpLaneRows * 64);
          ++next;
        }
      }
    pl.part_hdr.swap(new_hdr);
    pl.part_lane.swap(new_lane);
  }
  // ---- 5. persistent workgroups, exactly one per CU: the role-major block sequence is cut into num_cus contiguous
  //         pieces of equal cost; a piece that crosses a role boundary becomes several SEGMENTS (the workgroup reloads
  //         the LDS image between them), so small roles cost no extra workgroup and nobody waits for a free CU.
  //         Starting a role costs about three blocks (image load, pipeline fill), continuing it in a new segment one ----
  //         Roles with an LDS image run in a kernel of their own (the two families of unit code do not fit one register
  //         budget): two cuts, one segment array ----
  {
    const int nwg = std::max(1, num_cus);
    auto block_cost = [&](int k) { return static_cast<double>(std::max<int64_t>(1, roles[k].cost)); };
    pl.seg.clear();
    auto cut = [&](bool img, std::vector<int32_t> &wg_seg_ptr) {
      const size_t seg0 = pl.seg.size();
      double total_cost = 0.0;
      for (int k = 0; k < pl.num_roles; ++k)
        if ((roles[k].img_doubles > 0) == img) total_cost += (static_cast<double>(members[roles[k].pattern].size()) + 3.0) * block_cost(k);
      if (total_cost == 0.0) { wg_seg_ptr.assign(static_cast<size_t>(nwg) + 1, static_cast<int32_t>(seg0 / 4)); return; }
      for (double target = total_cost / nwg;; target *= 1.02) {
        pl.seg.resize(seg0);
        wg_seg_ptr.assign(1, static_cast<int32_t>(seg0 / 4));
        int wg = 0;
        double load = 0.0;  // of the current workgroup
        int last_role = -1;
        bool fits = true;
        for (int k = 0; k < pl.num_roles && fits; ++k) {
          if ((roles[k].img_doubles > 0) != img) continue;
          const double c = block_cost(k);
          const int nb = static_cast<int>(members[roles[k].pattern].size());
          int seg_cap = std::max(1, kBpSegInts / roles[k].nruns);
          if (seg_blocks > 0) seg_cap = std::min(seg_cap, seg_blocks);
          int first = 0;
          while (first < nb) {
            const double start = (last_role == k ? 1.0 : 3.0) * c;
            int take = static_cast<int>(std::floor((target - load - start) / c + 0.5));
            if (take < 1 && load > 0.0) {  // no room for another segment: next workgroup
              if (wg == nwg - 1) { fits = false; break; }
              wg_seg_ptr.push_back(static_cast<int32_t>(pl.seg.size() / 4));
              ++wg;
              load = 0.0;
              last_role = -1;
              continue;
            }
            take = std::max(1, std::min(std::min(take, nb - first), seg_cap));
            pl.seg.push_back(k);
            pl.seg.push_back(first);
            pl.seg.push_back(take);
            pl.seg.push_back(0);
            first += take;
            load += start + take * c;
            last_role = k;
          }
        }
        if (fits) {
          while (static_cast<int>(wg_seg_ptr.size()) < nwg + 1) wg_seg_ptr.push_back(static_cast<int32_t>(pl.seg.size() / 4));
          wg_seg_ptr.back() = static_cast<int32_t>(pl.seg.size() / 4);
          break;
        }
      }
    };
    cut(false, pl.wg_seg_ptr);
    cut(true, pl.wg_seg_ptr_img);
    pl.num_wgs = nwg;
  }
  // ---- 5b. order of a role's blocks inside its block-major tables.  A workgroup walks a contiguous range of them
  //          (its segment), one block at a time, and all workgroups run at about the same pace: the t-th blocks of all
  //          segments are written at about the same time.  They are made NEIGHBOURS IN THE CRS VALUE ARRAY (blocks sorted
  //          by the offset of their first row, dealt round robin over the role's segments): the ~256 x 64 KB in flight
  //          then cover a few contiguous megabytes instead of 4096 scattered row runs, which is what the memory's
  //          row buffers need (measured: the stores alone took 336 us for 1.08 GB with every workgroup in a region of
  //          its own).  MHA_BP_ORDER=morton keeps the blocks in partition order. ----
  std::vector<std::vector<int32_t>> role_blocks(pl.num_roles);
  {
    const char *env = std::getenv("MHA_BP_ORDER");
    const bool interleave = !(env && std::string(env) == "morton");
    std::vector<std::vector<std::pair<int, int>>> segs(pl.num_roles);  // per role: (first, blocks) of its segments
    for (size_t sg = 0; sg < pl.seg.size() / 4; ++sg) segs[pl.seg[4 * sg]].push_back({pl.seg[4 * sg + 1], pl.seg[4 * sg + 2]});
    for (int k = 0; k < pl.num_roles; ++k) {
      std::vector<int32_t> sorted = members[roles[k].pattern];
      if (interleave)
        std::stable_sort(sorted.begin(), sorted.end(), [&](int32_t x, int32_t y) { return rb.row_base[rb.row_ptr[x]] < rb.row_base[rb.row_ptr[y]]; });
      std::vector<int32_t> &out = role_blocks[k];
      out.assign(sorted.size(), -1);
      if (!interleave) { out = sorted; continue; }
      size_t next = 0;
      int longest = 0;
      for (const auto &sg : segs[k]) longest = std::max(longest, sg.second);
      for (int t = 0; t < longest; ++t)
        for (const auto &sg : segs[k])
          if (t < sg.second) out[static_cast<size_t>(sg.first) + t] = sorted[next++];
      if (next != sorted.size()) return fail("the segments of a role do not cover its blocks");
    }
  }
  for (int k = 0; k < pl.num_roles; ++k) {
    const RoleBuild &r = roles[k];
    const std::vector<int32_t> &blocks = role_blocks[k];
    int32_t *ro = &pl.role[static_cast<size_t>(k) * kBpRoleInts];
    const int64_t erec_base = static_cast<int64_t>(pl.erec_elem.size());
    const int64_t row_base = static_cast<int64_t>(pl.rowbase.size());
    ro[R_EREC_LO] = static_cast<int32_t>(erec_base & 0xffffffffll);
    ro[R_EREC_HI] = static_cast<int32_t>(erec_base >> 32);
    ro[R_ESTRIDE] = r.T + 1;
    ro[R_ROWB_LO] = static_cast<int32_t>(row_base & 0xffffffffll);
    ro[R_ROWB_HI] = static_cast<int32_t>(row_base >> 32);
    ro[R_NRUNS] = r.nruns;
    ro[R_NBLOCKS] = static_cast<int32_t>(blocks.size());
    ro[R_COST] = static_cast<int32_t>(std::min<int64_t>(r.cost, 0x7fffffff));
    ro[R_WOFF_LO] = static_cast<int32_t>(r.w_off & 0xffffffffll);
    ro[R_WOFF_HI] = static_cast<int32_t>(r.w_off >> 32);
    ro[R_WDOUBLES] = r.w_doubles;
    ro[R_PATTERN] = r.pattern;
    ro[R_NELEMS] = r.T;
    ro[R_IMG] = r.img_doubles;
    ro[R_WLDS] = r.img_doubles > 0 ? r.w_lds : r.w_doubles;
    ro[R_NCHUNK] = r.nchunk;
    ro[R_CHUNK_LO] = static_cast<int32_t>(r.chunk_off & 0xffffffffll);
    ro[R_CHUNK_HI] = static_cast<int32_t>(r.chunk_off >> 32);
    pl.role_runlen_off.push_back(static_cast<int64_t>(pl.runlen.size()));
    if (!blocks.empty()) {  // entries of every run, from the role's first block (the pattern fixes them)
      const int32_t b0 = blocks[0];
      for (int o = rb.row_ptr[b0]; o < rb.row_ptr[b0 + 1]; ++o) {
        if (o == rb.row_ptr[b0] || rb.rows[o - 1] + 1 != rb.rows[o]) pl.runlen.push_back(0);
        pl.runlen.back() += rowptr[rb.rows[o] + 1] - rowptr[rb.rows[o]];
      }
    }
    for (int32_t b : blocks) {
      for (int t = rb.elem_ptr[b]; t < rb.elem_ptr[b + 1]; ++t) pl.erec_elem.push_back(rb.elems[t]);
      pl.erec_elem.push_back(-1);
      for (int o = rb.row_ptr[b]; o < rb.row_ptr[b + 1]; ++o)  // CRS offset of the first row of every run
        if (o == rb.row_ptr[b] || rb.rows[o - 1] + 1 != rb.rows[o]) pl.rowbase.push_back(rb.row_base[o]);
    }
    int64_t mf = 0;
    for (int wv = 0; wv < kBpWaves; ++wv)
      for (int p = pl.part_ptr[static_cast<size_t>(k) * (kBpWaves + 1) + wv]; p < pl.part_ptr[static_cast<size_t>(k) * (kBpWaves + 1) + wv + 1]; ++p) {
        const int32_t *h = &pl.part_hdr[static_cast<size_t>(p) * kBpHdrInts];
        int bits = 0;
        for (int wd = 0; wd < 3; ++wd) bits += __builtin_popcount(static_cast<uint32_t>(h[H_MASK0 + wd]));
        mf += static_cast<int64_t>(bits) * static_cast<int64_t>(blocks.size());
      }
    pl.mfma_per_assembly += mf;
  }
  pl.usable = true;
  return pl;
}

void block_patterns_host_apply(const BlockPatternPlan &pl, const double *factors, double su, double st, bool overwrite,
                               double *vals) {
  MHA_REQUIRE(pl.usable, MHA_ERR_STATE, "block patterns not usable: " << pl.why);
  const int ke = pl.ke;
  std::vector<double> rec;
  for (int family = 0; family < 2; ++family)
  for (int wg = 0; wg < pl.num_wgs; ++wg)
  for (int sg = (family ? pl.wg_seg_ptr_img : pl.wg_seg_ptr)[wg]; sg < (family ? pl.wg_seg_ptr_img : pl.wg_seg_ptr)[wg + 1]; ++sg) {
    const int role = pl.seg[4 * sg], first = pl.seg[4 * sg + 1], nseg = pl.seg[4 * sg + 2];
    const int32_t *ro = &pl.role[static_cast<size_t>(role) * kBpRoleInts];
    const int64_t erec_base = (static_cast<int64_t>(ro[R_EREC_HI]) << 32) | static_cast<uint32_t>(ro[R_EREC_LO]);
    const int64_t row_base = (static_cast<int64_t>(ro[R_ROWB_HI]) << 32) | static_cast<uint32_t>(ro[R_ROWB_LO]);
    const int estride = ro[R_ESTRIDE], nruns = ro[R_NRUNS];
    MHA_REQUIRE(first >= 0 && first + nseg <= ro[R_NBLOCKS], MHA_ERR_STATE, "segment outside its role");
    const double *Wrole = pl.w.data() + ((static_cast<int64_t>(ro[R_WOFF_HI]) << 32) | static_cast<uint32_t>(ro[R_WOFF_LO]));
    for (int wv = 0; wv < kBpWaves; ++wv)
      for (int p = pl.part_ptr[static_cast<size_t>(role) * (kBpWaves + 1) + wv]; p < pl.part_ptr[static_cast<size_t>(role) * (kBpWaves + 1) + wv + 1]; ++p) {
        const int32_t *h = &pl.part_hdr[static_cast<size_t>(p) * kBpHdrInts];
        const int32_t *L = &pl.part_lane[static_cast<size_t>(p) * kBpLaneRows * 64];
        const int ks = h[H_KS], nct = h[H_NCT], len = h[H_LEN], stride = stride_for(nct);
        const bool fixed_class = h[H_FLAGS] & 1;
        if (fixed_class && !overwrite) continue;
        const double *W = Wrole + h[H_WOFF], *Wm = W + ro[R_WDOUBLES];
        const int c_begin = 16 * h[H_CT0], c_end = std::min(len, 16 * (h[H_CT0] + h[H_NTILE] + ((h[H_FLAGS] & 2) ? 1 : 0)));
        for (int i = 0; i < nseg; ++i) {
          const int64_t j = first + i;
          // the block's element records, as build_erec2_kernel lays them out
          rec.assign(static_cast<size_t>(estride) * kBpRecDoubles, 0.0);
          for (int t = 0; t < estride; ++t) {
            const int e = pl.erec_elem[static_cast<size_t>(erec_base + j * estride + t)];
            if (e >= 0)
              for (int m = 0; m < ke; ++m) rec[static_cast<size_t>(t) * kBpRecDoubles + m] = factors[static_cast<size_t>(e) * ke + m];
          }
          for (int row = 0; row < 16; ++row) {
            // result rows live in lanes (row & 3) << 4 .. with register t = row >> 2
            const int packed = L[(16 + (row >> 2)) * 64 + ((row & 3) << 4)];
            MHA_REQUIRE(packed == L[20 * 64 + row] && packed == L[20 * 64 + 48 + row], MHA_ERR_STATE, "row tables of a part disagree");
            if (packed < 0) continue;
            const int base = pl.rowbase[static_cast<size_t>(row_base + j * nruns + (packed >> 20))] + (packed & 0xfffff);
            for (int c = c_begin; c < c_end; ++c) {
              double v = 0.0;
              // chain of the unit that holds column c (bp_unit_col), then the products the kernel skips for it
              int qc = -1;
              for (int q = 0; q < h[H_NTILE] + ((h[H_FLAGS] & 2) ? 1 : 0) && qc < 0; ++q)
                for (int lc = 0; lc < 16; ++lc)
                  if (bp_unit_col(h[H_CT0], h[H_NTILE], q, lc) == c) { qc = q; break; }
              MHA_REQUIRE(qc >= 0, MHA_ERR_STATE, "a column of a unit belongs to none of its chains");
              const int trim = ((h[H_NTILE] == 4 && !(h[H_FLAGS] & 2)) || (h[H_NTILE] == 2 && (h[H_FLAGS] & 32))) ? (h[H_FLAGS] >> 2) & 7 : 0;
              for (int s = 0; s < ks; ++s) {
                const bool second = s >= ks / 2;
                if ((trim == 1 && qc < 2 && second) || (trim == 2 && qc >= 2 && !second) || (trim == 3 && qc < 2 && !second) ||
                    (trim == 4 && qc >= 2 && second))
                  continue;
                const int bit = s * 5 + qc;
                if (!((static_cast<uint32_t>(h[H_MASK0 + bit / 32]) >> (bit % 32)) & 1u)) continue;  // a zero block of W
                for (int kk = 0; kk < 4; ++kk) {
                  const int lane = kk * 16 + row;  // A operand: row = lane & 15, k = 4 s + (lane >> 4)
                  const size_t wi = static_cast<size_t>(4 * s + kk) * stride + c;
                  v += rec[L[s * 64 + lane]] * (su * W[wi] + st * Wm[wi]);
                }
              }
              if (overwrite) vals[base + c] = fixed_class ? 0.0 : v;
              else vals[base + c] += v;
            }
          }
        }
      }
  }
}

}  // namespace mha
