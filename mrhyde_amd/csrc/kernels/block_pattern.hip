// block_pattern.hip -- row-owner Jacobian of affine thermal elements on the fp64 matrix cores, one persistent
// 16-wave workgroup per CU, row blocks keyed by assembly pattern (block_pattern.hpp).
//
// Replaces the per-entry sumIntoValues of the reference's scatter (src/managers/assemblyManager.cpp:4031-4145, with
// thermal::volumeResidual src/physics/thermal.cpp:125-163 supplying res(e,i).dx(j)) for affine elements with
// element-wise constant coefficients:   vals[16 rows][row length] = G[16 rows][K] * W[K][row length],
// K = (elements around the dof) x (geometry components), on v_mfma_f64_16x16x4_f64 (operand maps: A[row = lane&15]
// [k = lane>>4], B[k = lane>>4][col = lane&15], D reg t: row = (lane>>4) + 4t, col = lane&15).
//
// Structure (what the previous forms of this kernel lacked):
//   * no dependent loads: a wavefront owns a PART (16 rows of one row class, every nphase-th block of its workgroup);
//     the offsets of its A operands inside a block's element records and the block-local indices of its result rows are
//     loop invariants in registers, the block's records / row offsets sit at  base + ordinal * stride;
//   * no barriers, no LDS accumulator, no atomics: W (all classes of the pattern, <= 150 KB) is read-only in LDS;
//   * A operands of the NEXT block are requested before the products of the current one; gfx950 counts loads and
//     stores in one in-order vmcnt, so the loads are issued from inline asm (hipcc does not track them) and retired with
//     a counted s_waitcnt vmcnt(4 * NCT) that leaves this block's stores in flight: a wavefront never waits for its own
//     stores, 16 wavefronts keep ~250 KB of stores in flight per CU;
//   * loads and stores go through raw buffer resources: 32-bit offsets (no 64-bit address arithmetic in the loops), and
//     the hardware drops lanes whose offset is out of range -- padding columns / rows are switched off that way, so
//     every store INSTRUCTION is always issued and the count behind the counted wait is exact.
// HBM traffic per assembly: the CRS values once (the compulsory write), 64 B per (block, touched element) of records.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_math.hpp"
#include "launch.hpp"
#include "../block_pattern.hpp"

namespace mha {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

enum { R_EREC_LO = 0, R_EREC_HI, R_ESTRIDE, R_ROWB_LO, R_ROWB_HI, R_NRUNS, R_NBLOCKS, R_COST, R_WOFF_LO, R_WOFF_HI,
       R_WDOUBLES, R_PATTERN, R_NELEMS };
enum { H_WOFF = 0, H_KS, H_NCT, H_LEN, H_PHASE, H_NPHASE, H_FLAGS, H_CLASS };

__global__ __launch_bounds__(256) void build_erec2_kernel(int64_t total, int nsym, const int32_t *__restrict__ erec_elem,
                                                          const double *__restrict__ geo, double *__restrict__ erec2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total * kBpRecDoubles; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / kBpRecDoubles;
    const int m = (int)(i - r * kBpRecDoubles);
    const int e = erec_elem[r];
    double v = 0.0;
    if (e >= 0) {
      if (m < nsym) v = geo[(size_t)e * kGeoRec + m];
      else if (m == nsym) v = geo[(size_t)e * kGeoRec + kGeoDet];
    }
    erec2[i] = v;
  }
}

// Raw buffer resource over [p, p + bytes): addresses are 32-bit offsets, lanes whose offset lies outside the range are
// dropped by the hardware.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4i make_rsrc(const void *p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));  // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;  // gfx9-family raw buffer, 32-bit data format
  return r;
}
constexpr unsigned kOutOfRange = 0x80000000u;  // offset of a lane that must not store
// the raw buffer store intrinsic with the descriptor as four ints (a store hipcc tracks like any other: it never waits for it)
__device__ void raw_buffer_store_b64(v2i data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");

// loads hipcc does not count (see the header): destination "=v", descriptor and block offset in scalar registers
__device__ __forceinline__ double asm_load_f64(const v4i &rsrc, unsigned lane_off, int block_off) {
  double v;
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(v) : "v"(lane_off), "s"(rsrc), "s"(block_off) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void asm_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct PartCtx {
  v4i erec;                // role's element records (buffer resource)
  v4i out;                 // the CRS values (buffer resource)
  const double *erec_ptr;  // the same records as a pointer (generic form)
  double *vals;
  const int *sbase;        // LDS: [block of the segment][run] CRS offset of the run's first row
  int nruns, estride, first, nblocks, phase, nphase, len;  // blocks first .. first + nblocks - 1 of the role
  const double *Wl;        // this lane's corner of the class's W image in LDS
  long long *tlog;         // profiling (DBG & 8): 8 wall-clock stamps of this wavefront
};

// 4 x 4 transpose of quarter waves across four registers: x[c] holds M[g][c] in quarter wave g, afterwards M[c][g]
// (gfx950's lane-swap instructions: v_permlane32_swap exchanges the upper half of its first operand with the lower half
// of its second, v_permlane16_swap the odd quarter waves of the first with the even ones of the second).
__device__ __forceinline__ void transpose_quarters(unsigned (&x)[4]) {
  auto swap32 = [](unsigned &a, unsigned &b) { const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; };
  auto swap16 = [](unsigned &a, unsigned &b) { const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; };
  swap32(x[0], x[2]);
  swap32(x[1], x[3]);
  swap16(x[0], x[1]);
  swap16(x[2], x[3]);
}

// One part: depth KS and NCT column tiles known at compile time, B blocks per loop iteration; overwrite mode only.
//
// A wavefront's loop iteration exposes one memory round trip (operands of iteration g + 1 are requested at the top of
// iteration g and awaited at its end), so the short row classes batch several blocks per iteration until an iteration
// carries enough matrix-core work to cover it.
//
// Stores: an MFMA result register holds 4 rows x 16 columns; stored as it stands that is four 128-byte pieces of four
// CRS rows per instruction, and the memory system takes such instructions at a third of the rate of contiguous ones
// (profiles/r2_probe.log: 0.53 -> 0.19 ms for the same bytes).  So the products of FOUR column tiles are transposed
// across the quarter waves (transpose_quarters: 32 lane-swap instructions per group) and every store instruction writes
// 64 consecutive entries of ONE row; the row's offset is wave-uniform and rides in the scalar offset of the buffer store.
// A last odd tile (NCT % 4 == 1) is stored as it stands; two or three left-over tiles fill a group with zero tiles.
//
// Register budget (168 at 12 waves per CU): operands of this and the next iteration 2 x 2 KS B, packed offsets KS / 2,
// one accumulator group 32 -- the per-iteration `opaque` statements keep hipcc from hoisting the unpacked offsets out of
// the loop, the scheduling barriers between the chains keep it from fetching the B operands of a whole group ahead
// (either way the asm-load destinations would be spilled: stored before they have landed).
template <int KS, int NCT, int B, int DBG>
__device__ __forceinline__ void run_part(const PartCtx &c, const int32_t *__restrict__ L, int lane) {
  constexpr int STRIDE = (NCT % 2 == 1) ? 16 * NCT : 16 * NCT + 16;
  constexpr int NPK = (KS + 1) / 2;
  constexpr int REM = NCT % 4;
  constexpr int NGRP = NCT / 4 + (REM >= 2 ? 1 : 0);      // groups of four tiles stored row-contiguous
  constexpr bool PLAIN_TAIL = REM == 1;                    // last tile stored as it stands
  constexpr int STORES = 16 * NGRP + (PLAIN_TAIL ? 4 : 0);
  static_assert(STORES * B < 64, "the counted wait must fit vmcnt");
  unsigned apk[NPK];  // byte offsets of the A operands inside a block's records, two per register
#pragma unroll
  for (int q = 0; q < NPK; ++q) {
    const unsigned lo = (unsigned)L[(2 * q) * 64 + lane] * 8u;
    const unsigned hi = (2 * q + 1 < KS) ? (unsigned)L[(2 * q + 1) * 64 + lane] * 8u : 0u;
    apk[q] = lo | (hi << 16);
  }
  const int relrow = L[20 * 64 + lane];  // tile row (lane & 15): run << 20 | CRS offset inside the run, -1 = no row
  auto aoff = [&](int s) { return (s & 1) ? (apk[s >> 1] >> 16) : (apk[s >> 1] & 0xffffu); };
  auto opaque = [&]() {
#pragma unroll
    for (int q = 0; q < NPK; ++q) asm volatile("" : "+v"(apk[q]));
  };
  // item b of iteration g is the part's block number phase + nphase (g B + b) inside the segment; past the end: the
  // last block again (loads stay in bounds and counted, stores are switched off)
  auto item = [&](int g, int b) { return c.phase + c.nphase * (g * B + b); };
  auto load_group = [&](int g, double (&dst)[B][KS]) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int i = min(item(g, b), c.nblocks - 1);
      const int boff = __builtin_amdgcn_readfirstlane((c.first + i) * c.estride * (kBpRecDoubles * 8));
#pragma unroll
      for (int s = 0; s < KS; ++s) dst[b][s] = asm_load_f64(c.erec, aoff(s), boff);
    }
  };
  auto chain = [&](const double (&A)[KS], int ct) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    if constexpr (!(DBG & 4)) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s], c.Wl[(4 * s) * STRIDE + 16 * ct], acc, 0, 0, 0);
    } else {
      acc[0] = A[0] + c.Wl[16 * ct];
    }
    return acc;
  };

  if (c.phase >= c.nblocks) return;
  double A[B][KS], An[B][KS];
  load_group(0, An);
  asm_wait_vmcnt<0>();
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int s = 0; s < KS; ++s) { asm volatile("" : "+v"(An[b][s])); A[b][s] = An[b][s]; }
  for (int g = 0;; ++g) {
    const bool more = item(g + 1, 0) < c.nblocks;
    opaque();
    if constexpr ((DBG & 8) != 0) { if (g == 2 && lane == 0) c.tlog[3] = wall_clock64(); }
    load_group(more ? g + 1 : g, An);
    if constexpr ((DBG & 8) != 0) { if (g == 2 && lane == 0) c.tlog[4] = wall_clock64(); }  // (the last iteration re-reads its own blocks: the count behind the wait stays exact)
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int i = item(g, b);
      const bool valid = i < c.nblocks && !(DBG & 2);
      // lane r < 16 (and its images in the other quarter waves): CRS offset of tile row r of this block, -1 = no store
      const int vrow = (valid && relrow >= 0) ? c.sbase[min(i, c.nblocks - 1) * c.nruns + (max(relrow, 0) >> 20)] + (relrow & 0xfffff) : -1;
#pragma unroll
      for (int grp = 0; grp < NGRP; ++grp) {
        v4d acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * grp + q < NCT) acc[q] = chain(A[b], 4 * grp + q);
          else acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
          __builtin_amdgcn_sched_barrier(0);
        }
        // lane l <- column 64 grp + l of rows q + 4 t
        const unsigned voff = (64 * grp + lane < c.len) ? (unsigned)(64 * grp + lane) * 8u : kOutOfRange;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          unsigned lo[4], hi[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { lo[q] = (unsigned)__double2loint(acc[q][t]); hi[q] = (unsigned)__double2hiint(acc[q][t]); }
          transpose_quarters(lo);
          transpose_quarters(hi);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rowoff = __builtin_amdgcn_readlane(vrow, q + 4 * t);  // wave-uniform: the row this instruction writes
            v2i bits;
            bits[0] = (int)lo[q];
            bits[1] = (int)hi[q];
            if constexpr ((DBG & 16) != 0)  // profiling: whole, aligned 128-byte lines only (clobbers the neighbouring rows)
              raw_buffer_store_b64(bits, c.out, (int)(rowoff >= 0 ? (unsigned)(64 * grp + lane) * 8u : kOutOfRange), (rowoff * 8) & ~127, 0);
            else
              raw_buffer_store_b64(bits, c.out, (int)(rowoff >= 0 ? voff : kOutOfRange), rowoff * 8, 0);
          }
        }
      }
      if constexpr (PLAIN_TAIL) {
        constexpr int ct = NCT - 1;
        const v4d acc = chain(A[b], ct);
        const bool in_row = 16 * ct + (lane & 15) < c.len;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // result register t of this lane belongs to tile row (lane >> 4) + 4 t
          const int rowoff = __builtin_amdgcn_ds_bpermute(((lane >> 4) + 4 * t) * 4, vrow);
          v2i bits;
          bits[0] = __double2loint(acc[t]);
          bits[1] = __double2hiint(acc[t]);
          raw_buffer_store_b64(bits, c.out, (int)((in_row && rowoff >= 0) ? (unsigned)(rowoff + 16 * ct + (lane & 15)) * 8u : kOutOfRange), 0, 0);
        }
      }
    }
    if constexpr ((DBG & 8) != 0) { if (g == 2 && lane == 0) c.tlog[5] = wall_clock64(); }
    asm_wait_vmcnt<STORES * B>();  // next operands are in; this iteration's stores stay in flight
    if constexpr ((DBG & 8) != 0) { if (g == 2 && lane == 0) c.tlog[6] = wall_clock64(); }
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int s = 0; s < KS; ++s) { asm volatile("" : "+v"(An[b][s])); A[b][s] = An[b][s]; }
    if (!more) break;
  }
}

// Any depth / width, store or accumulate, plain (compiler-counted) loads: parts the specialised forms do not cover,
// rows of fixed dofs (zeros when storing), and the accumulate mode.
__device__ void run_part_generic(const PartCtx &c, const int32_t *__restrict__ L, int lane, int ks, int nct, bool fixed_class,
                                 bool overwrite) {
  const int stride = (nct % 2 == 1) ? 16 * nct : 16 * nct + 16;
  const int l15 = lane & 15;
  int rel[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) rel[t] = L[(16 + t) * 64 + lane];
  for (int i = c.phase; i < c.nblocks; i += c.nphase) {
    const double *eb = c.erec_ptr + (size_t)(c.first + i) * c.estride * kBpRecDoubles;
    int base[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) base[t] = c.sbase[i * c.nruns + (max(rel[t], 0) >> 20)] + (rel[t] & 0xfffff);
    double A[kBpMaxKSteps];
#pragma unroll
    for (int s = 0; s < kBpMaxKSteps; ++s) A[s] = (s < ks) ? eb[L[s * 64 + lane]] : 0.0;
    for (int ct = 0; ct < nct; ++ct) {
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < kBpMaxKSteps; ++s)
        if (s < ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s], c.Wl[(4 * s) * stride + 16 * ct], acc, 0, 0, 0);
      const int col = 16 * ct + l15;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (rel[t] < 0 || col >= c.len) continue;
        double *p = c.vals + (size_t)base[t] + col;
        if (overwrite) *p = fixed_class ? 0.0 : acc[t];
        else if (!fixed_class) *p += acc[t];
      }
    }
  }
}

// DBG (profiling launches only, env MHA_BP_DBG): 1 plain-load form of every part, 2 no stores, 4 no products, 8 wall-clock
// stamps of every wavefront
template <int DBG>
__global__ __launch_bounds__(kBpWaves * 64) void block_pattern_jacobian_kernel(BlockPatternDev d, RowOut out, double su, double st) {
  extern __shared__ double W[];  // [max_w_doubles] the role's image, then [kBpSegInts] ints: the segment's run offsets
  int *sbase = reinterpret_cast<int *>(W + d.max_w_doubles);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  PartCtx c;
  c.vals = out.vals;
  c.out = make_rsrc(out.vals, (unsigned)(d.nnz * 8));
  long long *tlog = nullptr;
  int tslot = 0;
  constexpr bool TIMING = (DBG & 8) != 0;
  if constexpr (TIMING) tlog = d.timing + ((size_t)blockIdx.x * kBpWaves + wave) * 8;
  c.tlog = tlog;
  auto stamp = [&]() {  // slots 0 start, 1 image loaded, 2 first part done, 7 last part done; 3..6: iteration 2 of the first part
    if constexpr (TIMING) {
      const int slot = tslot < 3 ? tslot : 7;
      if (lane == 0) tlog[slot] = wall_clock64();
      ++tslot;
    }
  };
  stamp();
  const int sg_begin = d.wg_seg_ptr[blockIdx.x], sg_end = d.wg_seg_ptr[blockIdx.x + 1];
  int cur_role = -1;
  for (int sg = sg_begin; sg < sg_end; ++sg) {
    const int role = d.seg[4 * sg];
    const int32_t *ro = d.role + (size_t)role * kBpRoleInts;
    if (sg > sg_begin) __syncthreads();  // every wave is done with the previous segment (role image, run offsets)
    if (role != cur_role) {
      cur_role = role;
      // the role's image comes in two halves: rows of the stiffness components (x alpha_u kappa) and of the mass
      // component (x alpha_t rho cp); their sum is what the products read
      const long long woff = ((long long)ro[R_WOFF_HI] << 32) | (unsigned)ro[R_WOFF_LO];
      const double2 *srck = reinterpret_cast<const double2 *>(d.w + woff);
      const double2 *srcm = reinterpret_cast<const double2 *>(d.w + woff + ro[R_WDOUBLES]);
      double2 *dst = reinterpret_cast<double2 *>(W);
      const int n2 = ro[R_WDOUBLES] / 2;
      for (int i = tid; i < n2; i += kBpWaves * 64) {
        const double2 a = srck[i], b = srcm[i];
        dst[i] = make_double2(su * a.x + st * b.x, su * a.y + st * b.y);
      }
    }
    c.first = __builtin_amdgcn_readfirstlane(d.seg[4 * sg + 1]);
    c.nblocks = __builtin_amdgcn_readfirstlane(d.seg[4 * sg + 2]);
    {
      c.nruns = ro[R_NRUNS];
      const int32_t *rbase = d.rowbase + ((((long long)ro[R_ROWB_HI]) << 32) | (unsigned)ro[R_ROWB_LO]) + (size_t)c.first * c.nruns;
      for (int i = tid; i < c.nblocks * c.nruns; i += kBpWaves * 64) sbase[i] = rbase[i];
    }
    __syncthreads();
    stamp();
    c.sbase = sbase;
    c.erec_ptr = d.erec2 + ((((long long)ro[R_EREC_HI]) << 32) | (unsigned)ro[R_EREC_LO]) * kBpRecDoubles;
    c.erec = make_rsrc(c.erec_ptr, (unsigned)ro[R_NBLOCKS] * (unsigned)ro[R_ESTRIDE] * (kBpRecDoubles * 8u));
    c.estride = ro[R_ESTRIDE];
    const int32_t *pp = d.part_ptr + (size_t)role * (kBpWaves + 1);
    const int p_begin = pp[wave], p_end = pp[wave + 1];
    for (int p = p_begin; p < p_end; ++p) {
      const int32_t *h = d.part_hdr + (size_t)p * kBpHdrInts;
      const int32_t *L = d.part_lane + (size_t)p * kBpLaneRows * 64;
      const int ks = h[H_KS], nct = h[H_NCT];
      const bool fixed_class = (h[H_FLAGS] & 1) != 0;
      c.phase = __builtin_amdgcn_readfirstlane(h[H_PHASE]);
      c.nphase = __builtin_amdgcn_readfirstlane(h[H_NPHASE]);
      c.len = h[H_LEN];
      const int stride = (nct % 2 == 1) ? 16 * nct : 16 * nct + 16;
      c.Wl = W + h[H_WOFF] + (lane >> 4) * stride + (lane & 15);
      if (fixed_class && !out.overwrite) continue;
      const int key = (out.overwrite && !fixed_class && !(DBG & 1)) ? ks * 16 + nct : -1;
      switch (key) {
        // (depth, column tiles) -> blocks per iteration: as many as 16 operand registers and the 6-bit vmcnt allow
        // Q2 hexes: 8 / 4 / 2 / 1 elements around a vertex / edge / face / cell dof
        case 14 * 16 + 8: run_part<14, 8, 1, DBG>(c, L, lane); break;
        case 7 * 16 + 5: run_part<7, 5, 2, DBG>(c, L, lane); break;
        case 4 * 16 + 3: run_part<4, 3, 3, DBG>(c, L, lane); break;
        case 2 * 16 + 2: run_part<2, 2, 3, DBG>(c, L, lane); break;
        // Q1 hexes
        case 14 * 16 + 2: run_part<14, 2, 1, DBG>(c, L, lane); break;
        case 7 * 16 + 2: run_part<7, 2, 2, DBG>(c, L, lane); break;
        case 4 * 16 + 1: run_part<4, 1, 4, DBG>(c, L, lane); break;
        case 2 * 16 + 1: run_part<2, 1, 8, DBG>(c, L, lane); break;
        // quads (one k-step per element): Q2 vertex rows, Q1 rows, cell rows; Q4
        case 4 * 16 + 2: run_part<4, 2, 3, DBG>(c, L, lane); break;
        case 1 * 16 + 1: run_part<1, 1, 14, DBG>(c, L, lane); break;
        case 4 * 16 + 6: run_part<4, 6, 1, DBG>(c, L, lane); break;
        case 2 * 16 + 3: run_part<2, 3, 3, DBG>(c, L, lane); break;
        case 1 * 16 + 2: run_part<1, 2, 3, DBG>(c, L, lane); break;
        default: run_part_generic(c, L, lane, ks, nct, fixed_class, out.overwrite != 0); break;
      }
      stamp();
    }
  }
}

}  // namespace

void launch_build_erec2(int64_t total_records, int nsym, const int32_t *erec_elem, const double *geo, double *erec2,
                        hipStream_t stream) {
  if (total_records <= 0) return;
  const int64_t work = total_records * kBpRecDoubles;
  const int grid = (int)std::min<int64_t>((work + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(build_erec2_kernel, dim3(grid), dim3(256), 0, stream, total_records, nsym, erec_elem, geo, erec2);
  MHA_HIP(hipGetLastError());
}

void launch_block_pattern_jacobian(const BlockPatternDev &d, const RowOut &out, double su, double st, hipStream_t stream) {
  if (d.num_wgs <= 0 || !out.vals) return;
  const size_t lds = sizeof(double) * (size_t)d.max_w_doubles + sizeof(int) * kBpSegInts;
  MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "pattern matrices of " << lds << " B do not fit the LDS");
  auto go = [&](auto kern) {
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(d.num_wgs), dim3(kBpWaves * 64), lds, stream, d, out, su, st);
  };
  switch (d.dbg | (d.timing ? 8 : 0)) {
    case 0: go(block_pattern_jacobian_kernel<0>); break;
    case 1: go(block_pattern_jacobian_kernel<1>); break;
    case 2: go(block_pattern_jacobian_kernel<2>); break;
    case 4: go(block_pattern_jacobian_kernel<4>); break;
    case 6: go(block_pattern_jacobian_kernel<6>); break;
    case 8: go(block_pattern_jacobian_kernel<8>); break;
    case 16: go(block_pattern_jacobian_kernel<16>); break;
    case 24: go(block_pattern_jacobian_kernel<24>); break;
    case 10: go(block_pattern_jacobian_kernel<10>); break;
    case 12: go(block_pattern_jacobian_kernel<12>); break;
    case 14: go(block_pattern_jacobian_kernel<14>); break;
    default: MHA_REQUIRE(false, MHA_ERR_INVALID, "MHA_BP_DBG: unsupported combination");
  }
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
