// block_pattern.hip -- row-owner Jacobian of affine thermal elements on the fp64 matrix cores, one persistent
// 16-wave workgroup per CU, row blocks keyed by assembly pattern (block_pattern.hpp).
//
// Replaces the per-entry sumIntoValues of the reference's scatter (src/managers/assemblyManager.cpp:4031-4145, with
// thermal::volumeResidual src/physics/thermal.cpp:125-163 supplying res(e,i).dx(j)) for affine elements with
// element-wise constant coefficients:   vals[16 rows][row length] = G[16 rows][K] * W[K][row length],
// K = (elements around the dof) x (geometry components), on v_mfma_f64_16x16x4_f64 (operand maps: A[row = lane&15]
// [k = lane>>4], B[k = lane>>4][col = lane&15], D reg t: row = (lane>>4) + 4t, col = lane&15).
//
// Structure:
//   * a workgroup (12 wavefronts, one per CU, persistent) walks its blocks one at a time; every wavefront owns a UNIT
//     of the block's pattern -- up to four 16-column tiles of one 16-row class tile (block_pattern.hpp) -- and all
//     wavefronts meet at a barrier after every block.  The rows of a block are neighbours in the CRS value array:
//     written within a microsecond of each other they leave the L2 as whole DRAM pages.  (Free-running wavefronts, each
//     streaming "its" rows of many blocks, wrote the same bytes at a third of the rate: the L2 -> memory write requests
//     stalled, profiles/r2_pmc_mem.log.)
//   * no dependent loads: the offsets of a unit's A operands inside a block's element records and its result rows are
//     loop invariants in registers, the block's records sit at base + ordinal * stride, its row offsets in LDS;
//   * no LDS accumulator, no atomics: W (all classes of the pattern, <= 150 KB) is read-only in LDS;
//   * A operands of the next TWO blocks are in flight; gfx950 counts loads and stores in one in-order vmcnt, so the
//     loads are issued from inline asm (hipcc does not track them) and retired with a counted s_waitcnt that leaves
//     the stores of this and the previous block in flight: a wavefront never waits for its own stores;
//   * loads and stores go through raw buffer resources: 32-bit offsets, and the hardware drops lanes whose offset is out
//     of range -- padding columns / rows are switched off that way, so every store INSTRUCTION is always issued and the
//     count behind the counted wait is exact;
//   * an MFMA result register holds 4 rows x 16 columns; the products of a unit's four tiles are transposed across the
//     quarter waves (gfx950 lane swaps) so that every store writes 64 consecutive entries of ONE row.
// HBM traffic per assembly: the CRS values once (the compulsory write), 64 B per (block, touched element) of records.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device_math.hpp"
#include "launch.hpp"
#include "../block_pattern.hpp"

namespace mha {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

enum { R_EREC_LO = 0, R_EREC_HI, R_ESTRIDE, R_ROWB_LO, R_ROWB_HI, R_NRUNS, R_NBLOCKS, R_COST, R_WOFF_LO, R_WOFF_HI,
       R_WDOUBLES, R_PATTERN, R_NELEMS, R_IMG, R_WLDS, R_NCHUNK, R_CHUNK_LO, R_CHUNK_HI };
enum { H_WOFF = 0, H_KS, H_NCT, H_LEN, H_CT0, H_NTILE, H_FLAGS, H_CLASS, H_MASK0, H_MASK1, H_MASK2 };

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
#ifndef MHA_BP_STORE_AUX
#define MHA_BP_STORE_AUX 0  // cache policy bits of the row stores (experiments: 1 sc0, 2 nt, 16 sc1)
#endif
// the raw buffer store intrinsic with the descriptor as four ints (a store hipcc tracks like any other: it never waits for it)
__device__ void raw_buffer_store_b64(v2i data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");
__device__ void raw_buffer_store_b128(v4i data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");

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
// workgroup barrier that waits for LDS traffic only (a __syncthreads() would drain the stores in flight)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct UnitCtx {
  v4i erec;                // role's element records (buffer resource)
  v4i out;                 // the CRS values (buffer resource)
  const double *erec_ptr;  // the same records as a pointer (generic form)
  double *vals;
  const int *sbase;        // LDS: [block of the segment][run] CRS offset of the run's first row
  int nruns, estride, first, nblocks, len;  // blocks first .. first + nblocks - 1 of the role
  int ct0;                 // first column tile of the unit
  const double *Wl;        // this lane's corner of the class's W image in LDS
  unsigned mask[3];        // bit s * 5 + q: the W block of k-step s, tile q of the unit is not zero
  int trim;                // trim class of the unit (host: flags 4 / 8 / 16), wave-uniform
  double *recbuf;          // LDS: two images of a block's element records, recstride doubles apart
  int recstride;
  long long *tlog;         // profiling (DBG & 8): 8 wall-clock stamps of this wavefront
  // image roles (block_pattern.hpp)
  double *img;             // LDS: the block's image
  int trash;               // entry offset (from img) of 64 doubles nobody reads
  const int32_t *chunks;   // the role's chunk table [nchunk][kBpChunkInts]
  int nchunk, sidx;        // chunks of a block; this wavefront's index among the streaming ones (wave - 1)
  const double *wk, *wm;   // the unit's class in the role's global W image: stiffness rows, mass rows (REGW units)
  double su, st;
};

// Element records of a block: (T + 1) x 8 doubles, contiguous in memory.  One wavefront (the loader: wave 0) fetches
// them with a handful of coalesced loads two blocks ahead and puts them into a double-buffered LDS image; every unit
// reads its A operands from there.  (Gathered straight from memory by every unit they were ~100 scattered load
// instructions per block in front of the CU's one address pipeline, which the stores already saturate.)
constexpr int kRecLoads = 8;  // 64-lane loads of 8 bytes per block: (T + 1) * 8 doubles <= 512 (the host checks); DBG bit 32: <= 384, 6 loads

// One unit over all blocks of the segment: depth KS, NT column tiles and, with TAIL, one more tile; overwrite mode.
//
// Columns (bp_unit_col, block_pattern.hpp): the NT tiles come in pairs with interleaved columns -- lane (g, c) of chains
// 2p and 2p + 1 holds entries 32p + 2c and 32p + 2c + 1 of tile row g + 4t in result register t -- so the lane stores
// both with ONE 16-byte instruction, four 256-byte row pieces per instruction and no transposition; an odd last tile
// and the tail tile keep the MFMA layout (16 entries x 4 rows per register, 8-byte stores).  A CU's store path takes
// one instruction per ~40 cycles whatever its width: 16 x (8-byte, one row of 64 entries) stores per unit of four
// tiles became 8.  Where the row ends inside a pair (odd row length: every row of a hex mesh) the lane holding the
// last entry stores it with an 8-byte instruction of its own (STRADDLE, 4 more instructions; wave-uniform).
//
// Software pipeline, one block deep: while the products of block i accumulate, the results of block i - 1 are stored,
// a few store instructions after every k-step.  All wavefronts meet at a barrier once per block: the rows of a block
// reach the memory together.  The loader's record fetches are inline asm (hipcc does not track them; gfx950 counts
// loads and stores in one in-order vmcnt) retired with a counted s_waitcnt that leaves the current block's stores in
// flight (with STRADDLE stores it waits for the four oldest of them as well).
// TRIM = 4: products whose 4 x 16 block of W is zero are skipped by a scalar mask-bit test (host: H_MASK*).
// LOADER: this wavefront also fetches the element records (wave 0; only the shapes of at most three chains are built
// with it -- the record registers are what the widest shapes have no room for -- and the planner gives wave 0 such a
// unit; a wave 0 that holds a wider one runs the plain form).
template <int KS, int NT, bool TAIL, int DBG, int TRIM = 0, bool LOADER = false>
__device__ __forceinline__ void run_unit(const UnitCtx &c, const int32_t *__restrict__ L, int lane) {
  constexpr bool loader = LOADER;
  constexpr int NPK = (KS + 1) / 2;
  constexpr int NQ = NT + (TAIL ? 1 : 0);                 // accumulation chains
  constexpr int NP = NT / 2;                              // pairs of interleaved tiles
  constexpr int NL = NQ - 2 * NP;                         // tiles in MFMA layout: chains 2 NP .. NQ - 1
  constexpr int SPT = NP + NL, STORES = 4 * SPT;          // store instructions per result register / per block
  static_assert(STORES < 60, "the counted wait must fit vmcnt");
  unsigned apk[NPK];  // offsets (doubles) of the A operands inside a block's records, two per register
#pragma unroll
  for (int q = 0; q < NPK; ++q) {
    const unsigned lo = (unsigned)L[(2 * q) * 64 + lane];
    const unsigned hi = (2 * q + 1 < KS) ? (unsigned)L[(2 * q + 1) * 64 + lane] : 0u;
    apk[q] = lo | (hi << 16);
  }
  const int relrow = L[20 * 64 + lane];  // tile row (lane & 15): run << 20 | CRS offset inside the run, -1 = no row
  const int nct_all = (c.len + 15) / 16;
  const int stride = (nct_all % 2 == 1) ? 16 * nct_all : 16 * nct_all + 16;
  auto aoff = [&](int s) { return (s & 1) ? (apk[s >> 1] >> 16) : (apk[s >> 1] & 0xffffu); };
  const int l15 = lane & 15;
  const double *Wp = c.Wl + l15;  // pairs: lane c reads column 2c (+ 1) of the pair's 32
  // entries of this lane inside a row: pair p -> 16 ct0 + 32 p + 2 l15 (+ 1), plain tile u -> 16 (ct0 + 2 NP + u) + l15; the
  // limits below (wave-uniform) say which lanes have something to store
  const int cbase = __builtin_amdgcn_readfirstlane(16 * c.ct0), rlen = __builtin_amdgcn_readfirstlane(c.len - 16 * c.ct0);
  // the row ends inside the last pair on an odd entry: that lane's first value goes out alone
  const bool straddle = NP > 0 && NL == 0 && (rlen & 1) && rlen > 32 * (NP - 1) && rlen < 32 * NP;
  // ---- loader state: records of block i + 1 in registers, block i + 2 in flight ----
  const int rec_doubles = c.estride * kBpRecDoubles;
  // (the build with DBG bit 32 -- blocks of at most 47 touched elements, e.g. the 4 x 2 x 2 chunks of a hex mesh -- carries
  // six record registers instead of eight)
  constexpr int RL = !LOADER ? 1 : ((DBG & 32) ? 6 : kRecLoads);
  double Rc[RL];
  auto fetch = [&](int i, double (&dst)[RL]) {  // past the end: the last block again (in bounds, counted)
    const int boff = __builtin_amdgcn_readfirstlane((c.first + min(i, c.nblocks - 1)) * rec_doubles * 8);
#pragma unroll
    for (int k = 0; k < RL; ++k) dst[k] = asm_load_f64(c.erec, (unsigned)min(k * 64 + lane, rec_doubles - 1) * 8u, boff);
  };
  auto deposit = [&](double (&src)[RL], double *buf) {
#pragma unroll
    for (int k = 0; k < RL; ++k) {
      asm volatile("" : "+v"(src[k]));
      if (k * 64 + lane < rec_doubles) buf[k * 64 + lane] = src[k];
    }
  };
  if constexpr (loader) {
    fetch(0, Rc);
    asm_wait_vmcnt<0>();
    deposit(Rc, c.recbuf);  // block 0: visible after the first barrier
    fetch(1, Rc);
  }

  v4d accP[NQ];       // results of the previous block, stored during this one
  int rowP[4];        // byte offset of tile row (lane >> 4) + 4 t of the previous block in the CRS values, < 0 = no store
#pragma unroll
  for (int q = 0; q < NQ; ++q) accP[q] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 4; ++t) rowP[t] = -1;
  // store j (0 .. STORES - 1) of the previous block: result register t = j / SPT, then the pairs, then the plain tiles
  auto store_prev = [&](int j) {
    const int t = j / SPT, u = j % SPT;
    if (u < NP) {
      v4i bits;
      bits[0] = __double2loint(accP[2 * u][t]);
      bits[1] = __double2hiint(accP[2 * u][t]);
      bits[2] = __double2loint(accP[2 * u + 1][t]);
      bits[3] = __double2hiint(accP[2 * u + 1][t]);
      const bool on = rowP[t] >= 0 && 32 * u + 2 * l15 + 1 < rlen;
      // soffset stays the CONSTANT 0: with a scalar register there hipcc's hazard recognizer assumes that a 16-byte store
      // needs no wait state before a VALU write of its data registers (GCNHazardRecognizer::createsVALUHazard) and reuses
      // the first data register for the next address at once -- on gfx950 lanes 12..15 of every row then stored that
      // address in place of the low half of their first value (found as a 4e-7 relative error in 48 entries, first launch only)
      raw_buffer_store_b128(bits, c.out, (int)(on ? (unsigned)rowP[t] + (unsigned)(l15 * 16) + (unsigned)((cbase + 32 * u) * 8) : kOutOfRange), 0, MHA_BP_STORE_AUX);
    } else {
      const int q = 2 * NP + (u - NP);
      v2i bits;
      bits[0] = __double2loint(accP[q][t]);
      bits[1] = __double2hiint(accP[q][t]);
      const bool on = rowP[t] >= 0 && 16 * q + l15 < rlen;
      raw_buffer_store_b64(bits, c.out, (int)(on ? (unsigned)rowP[t] + (unsigned)(l15 * 8) : kOutOfRange), (cbase + 16 * q) * 8, MHA_BP_STORE_AUX);
    }
  };

  for (int i = 0; i <= c.nblocks; ++i) {  // iteration i: products of block i (discarded at i = nblocks), stores of block i - 1
    const int ic = min(i, c.nblocks - 1);
    const int vrowC = (relrow >= 0 && i < c.nblocks && !(DBG & 2)) ? (c.sbase[ic * c.nruns + (max(relrow, 0) >> 20)] + (relrow & 0xfffff)) * 8 : -1;
    lds_barrier();  // block i's records are in place; every wavefront has issued the stores of block i - 2
    asm volatile("" ::: "memory");  // (the LDS has changed: without this a wavefront that does not write it hoists all its W reads out of the loop, 112 registers)
    // markers for csrc/check_store_count.awk (run by the Makefile on the kernel's ISA): between the two comments of a
    // loader instantiation there must be exactly STORES (+ 4 with the straddle form) buffer_store instructions -- the
    // counted wait below retires the record loads by leaving that many younger operations in flight
    if constexpr (loader) asm volatile("; MHA_LOADER_ITER stores=%0" ::"n"(STORES));
    const double *rec = c.recbuf + (i & 1) * c.recstride;
    v4d acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    // TRIM 3 (units of two pairs): one pair of tiles sees only zero blocks of W during one half of the k-steps -- the
    // columns of a row are sorted by global id and so are a block's elements: around a vertex dof of a hex mesh four of
    // the eight elements reach only one end of the row.  The host classifies the unit (c.trim: 1..4 = which pair, which
    // half; block_pattern.cpp), wave-uniform; the whole k-loop exists once per class, branch-free inside (a branch per
    // product made hipcc copy the accumulators at every join: >1000 spilled registers).
    // TRIM 4: any pattern by mask bit (bit s * 5 + q of c.mask): one branch per product, for shapes with registers to spare.
    auto kloop = [&](auto cls_tag) {
      constexpr int CLS = decltype(cls_tag)::value;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double a = rec[aoff(s)];
        if constexpr (!(DBG & 4)) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const bool second = s >= KS / 2;
            if ((CLS == 1 && q < 2 && second) || (CLS == 2 && q >= 2 && !second) || (CLS == 3 && q < 2 && !second) ||
                (CLS == 4 && q >= 2 && second))
              continue;  // compile time
            const int mbit = s * 5 + q;  // a constant after unrolling: the mask word stays in a scalar register
            const unsigned mw = mbit < 32 ? c.mask[0] : (mbit < 64 ? c.mask[1] : c.mask[2]);
            const bool zero_block = TRIM == 4 && !((mw >> (mbit & 31)) & 1u);
            const double b = q < 2 * NP ? Wp[(4 * s) * stride + 32 * (q >> 1) + (q & 1)] : c.Wl[(4 * s) * stride + 16 * q];
            if (!zero_block) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
          }
        } else {
          acc[0][0] += a;
        }
#pragma unroll
        for (int j = (s * STORES) / KS; j < ((s + 1) * STORES) / KS; ++j) store_prev(j);
        if (s % 2 == 1) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (TRIM == 3) {
      switch (c.trim) {
        case 1: kloop(std::integral_constant<int, 1>{}); break;
        case 2: kloop(std::integral_constant<int, 2>{}); break;
        case 3: kloop(std::integral_constant<int, 3>{}); break;
        case 4: kloop(std::integral_constant<int, 4>{}); break;
        default: kloop(std::integral_constant<int, 0>{}); break;
      }
    } else {
      kloop(std::integral_constant<int, 0>{});
    }
    if (straddle) {  // wave-uniform
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v2i bits;
        bits[0] = __double2loint(accP[2 * (NP > 0 ? NP - 1 : 0)][t]);
        bits[1] = __double2hiint(accP[2 * (NP > 0 ? NP - 1 : 0)][t]);
        const bool on = rowP[t] >= 0 && 32 * (NP - 1) + 2 * l15 == rlen - 1;
        raw_buffer_store_b64(bits, c.out, (int)(on ? (unsigned)rowP[t] : kOutOfRange), (cbase + rlen - 1) * 8, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) accP[q] = acc[q];
#pragma unroll
    for (int t = 0; t < 4; ++t) rowP[t] = __builtin_amdgcn_ds_bpermute(((lane >> 4) + 4 * t) * 4, vrowC);
    if constexpr (loader) {
      // block i + 1's records (requested one iteration ago) into the other buffer -- last read during block i - 1, and
      // everybody has passed this iteration's barrier since; this iteration's stores stay in flight.  Then request
      // block i + 2's into the same registers.
      asm volatile("; MHA_LOADER_WAIT stores=%0" ::"n"(STORES));
      asm_wait_vmcnt<STORES>();
      deposit(Rc, c.recbuf + ((i + 1) & 1) * c.recstride);
      fetch(i + 2, Rc);
    }
  }
}

// The record loader as a wavefront of its own (wave 0 without a unit: what the planner arranges whenever a pattern has
// fewer units than wavefronts).  It issues no stores, so its loads never queue behind any in the in-order vmcnt: a
// loader that also stores waits for the write acknowledgements of its previous block before its records count as
// landed, and with it the whole workgroup at the next barrier -- the stores of consecutive blocks did not overlap
// (stores alone 314 us for 1.08 GB against 170-200 us for the same volume from a plain persistent store loop,
// profiles/micro/write_bw.hip).  Two register sets: the records of blocks i + 1 and i + 2 are in flight during block i.
template <int DBG>
__device__ __forceinline__ void run_loader_only(const UnitCtx &c, int lane) {
  const int rec_doubles = c.estride * kBpRecDoubles;
  constexpr int RL = kRecLoads;
  double Ra[RL], Rb[RL];
  auto fetch = [&](int i, double (&dst)[RL]) {  // past the end: the last block again (in bounds)
    const int boff = __builtin_amdgcn_readfirstlane((c.first + min(i, c.nblocks - 1)) * rec_doubles * 8);
#pragma unroll
    for (int k = 0; k < RL; ++k) dst[k] = asm_load_f64(c.erec, (unsigned)min(k * 64 + lane, rec_doubles - 1) * 8u, boff);
  };
  auto deposit = [&](double (&src)[RL], double *buf) {
#pragma unroll
    for (int k = 0; k < RL; ++k) {
      asm volatile("" : "+v"(src[k]));
      if (k * 64 + lane < rec_doubles) buf[k * 64 + lane] = src[k];
    }
  };
  fetch(0, Ra);
  asm_wait_vmcnt<0>();
  deposit(Ra, c.recbuf);  // block 0: visible after the first barrier
  fetch(1, Ra);
  fetch(2, Rb);
  for (int i = 0; i <= c.nblocks; i += 2) {
    lds_barrier();  // iteration i: the units read buffer i & 1
    asm_wait_vmcnt<RL>();  // block i + 1 has landed (i + 2 may still be in flight)
    deposit(Ra, c.recbuf + ((i + 1) & 1) * c.recstride);  // last read during block i - 1
    fetch(i + 3, Ra);
    if (i + 1 > c.nblocks) break;
    lds_barrier();  // iteration i + 1
    asm_wait_vmcnt<RL>();
    deposit(Rb, c.recbuf + (i & 1) * c.recstride);
    fetch(i + 4, Rb);
  }
  asm_wait_vmcnt<0>();  // nothing of this wavefront's may land after it has moved on
}

// ---------------------------------------------------------------------------------------------------------------------
// IMAGE ROLES (block_pattern.hpp): results go to an LDS image of the block's rows and leave in aligned 512-byte chunks.
// Per block two barriers: [products of block i  ||  stream-out of block i - 1] B1 [results of block i -> image] B2.
// ---------------------------------------------------------------------------------------------------------------------

// This wavefront's chunks m = 0 .. ncw - 1 (chunk sidx + kBpStreamWaves m of the block), descriptors one per lane.
// A chunk = 128 entries (1 KB) of a run, starting on a multiple of 128 entries from the run's ALIGNED start (its global
// offset rounded down to 16 entries): lane l moves entries 2l and 2l + 1 with one 16-byte store, whole 128-byte lines
// except where the run begins or ends; a lane that holds only one entry of the run (odd begin or end) stores it with
// an 8-byte instruction of its own, issued for those chunks only.  (A CU's store path takes about 40 cycles per
// instruction, masked lanes or not: 8-byte chunks alone kept the kernel at 109 us without any products or data.)
struct StreamCtx {
  int pk, lds0;    // lane m: chunk m -- run | chunk index inside the run << 8 | entries of the run << 16; LDS entry of its lane 0
  int vb, lo, hi;  // lane m, per block: byte offset of the chunk's first entry in the CRS values; entries lo <= e < hi of the chunk belong to the run
  int ncw;
};
__device__ __forceinline__ StreamCtx make_stream(const UnitCtx &c, int lane) {
  StreamCtx s;
  s.ncw = c.sidx < c.nchunk ? (c.nchunk - c.sidx + kBpStreamWaves - 1) / kBpStreamWaves : 0;
  const int j = min(c.sidx + kBpStreamWaves * lane, c.nchunk - 1);
  const int32_t *t = c.chunks + (size_t)j * kBpChunkInts;
  s.pk = t[0] | ((t[2] >> 7) << 8) | (t[3] << 16);  // (the host checks: < 256 runs, < 256 chunks per run, < 65536 entries)
  s.lds0 = t[1];
  s.vb = s.lo = s.hi = 0;
  return s;
}
// block ip is about to be streamed: where its runs lie in the CRS values (c.sbase), for all of this wavefront's chunks at once
__device__ __forceinline__ void stream_block(const UnitCtx &c, StreamCtx &s, int ip) {
  const unsigned pk = (unsigned)s.pk;
  const int run = pk & 0xff, e0 = ((pk >> 8) & 0xff) << 7, len = pk >> 16;
  const int goff = c.sbase[max(ip, 0) * c.nruns + run];
  const int a = goff & 15;
  s.vb = (goff - a + e0) * 8;
  s.lo = ip >= 0 ? a - e0 : 128;  // (no block: no entry)
  s.hi = a + len - e0;
}
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d stream_read(const UnitCtx &c, const StreamCtx &s, int m, int lane) {
  return *reinterpret_cast<const v2d *>(c.img + __builtin_amdgcn_readlane(s.lds0, m) + 2 * lane);
}
// chunk m of the block: 128 entries from the image (already read: v) to the CRS values
template <int DBG>
__device__ __forceinline__ void stream_store(const UnitCtx &c, const StreamCtx &s, int m, v2d v, int lane) {
  const int vb = __builtin_amdgcn_readlane(s.vb, m), lo = __builtin_amdgcn_readlane(s.lo, m), hi = __builtin_amdgcn_readlane(s.hi, m);
  const int e = 2 * lane;
  const bool on0 = e >= lo && e < hi && !(DBG & 2), on1 = e + 1 >= lo && e + 1 < hi && !(DBG & 2);
  v4i bits;
  bits[0] = __double2loint(v[0]);
  bits[1] = __double2hiint(v[0]);
  bits[2] = __double2loint(v[1]);
  bits[3] = __double2hiint(v[1]);
  raw_buffer_store_b128(bits, c.out, (int)((on0 && on1) ? (unsigned)(vb + lane * 16) : kOutOfRange), 0, 0);  // (soffset: the constant 0, see run_unit)
  if (((lo & 1) && lo > 0 && lo < 128) || ((hi & 1) && hi > 0 && hi < 128)) {  // wave-uniform: a lane with one entry of the run
    v2i b1;
    b1[0] = on0 ? bits[0] : bits[2];
    b1[1] = on0 ? bits[1] : bits[3];
    raw_buffer_store_b64(b1, c.out, (int)((on0 != on1) ? (unsigned)(vb + lane * 16 + (on0 ? 0 : 8)) : kOutOfRange), 0, 0);
  }
}
// chunks m0 .. m1 - 1, the image read of chunk m + 1 in flight while chunk m is stored
template <int DBG>
__device__ __forceinline__ void stream_range(const UnitCtx &c, const StreamCtx &s, int m0, int m1, int lane) {
  if (m0 >= m1) return;
  v2d v = stream_read(c, s, m0, lane);
  for (int m = m0; m < m1; ++m) {
    const v2d vn = stream_read(c, s, min(m + 1, m1 - 1), lane);
    stream_store<DBG>(c, s, m, v, lane);
    v = vn;
  }
}

// wave 0 of an image role: element records only (no stores: its loads never queue behind any)
template <int DBG>
__device__ __forceinline__ void run_loader_img(const UnitCtx &c, int lane) {
  const int rec_doubles = c.estride * kBpRecDoubles;
  constexpr int RL = kRecLoads;
  double Ra[RL], Rb[RL];
  auto fetch = [&](int i, double (&dst)[RL]) {  // past the end: the last block again (in bounds)
    const int boff = __builtin_amdgcn_readfirstlane((c.first + min(i, c.nblocks - 1)) * rec_doubles * 8);
#pragma unroll
    for (int k = 0; k < RL; ++k) dst[k] = asm_load_f64(c.erec, (unsigned)min(k * 64 + lane, rec_doubles - 1) * 8u, boff);
  };
  auto deposit = [&](double (&src)[RL], double *buf) {
#pragma unroll
    for (int k = 0; k < RL; ++k) {
      asm volatile("" : "+v"(src[k]));
      if (k * 64 + lane < rec_doubles) buf[k * 64 + lane] = src[k];
    }
  };
  fetch(0, Ra);
  asm_wait_vmcnt<0>();
  deposit(Ra, c.recbuf);
  fetch(1, Ra);
  fetch(2, Rb);
  lds_barrier();  // B0: block 0's records are in place
  for (int i = 0; i <= c.nblocks; i += 2) {
    lds_barrier();  // B1 of iteration i
    asm_wait_vmcnt<RL>();
    deposit(Ra, c.recbuf + ((i + 1) & 1) * c.recstride);  // block i + 1: that buffer was last read during iteration i - 1
    fetch(i + 3, Ra);
    lds_barrier();  // B2
    if (i + 1 > c.nblocks) break;
    lds_barrier();  // B1 of iteration i + 1
    asm_wait_vmcnt<RL>();
    deposit(Rb, c.recbuf + (i & 1) * c.recstride);
    fetch(i + 4, Rb);
    lds_barrier();  // B2
  }
  asm_wait_vmcnt<0>();
}

// a streaming wavefront without a unit
template <int DBG>
__device__ __forceinline__ void run_stream_only(const UnitCtx &c, int lane) {
  StreamCtx sc = make_stream(c, lane);
  lds_barrier();  // B0
  for (int i = 0; i <= c.nblocks; ++i) {
    stream_block(c, sc, i - 1);
    stream_range<DBG>(c, sc, 0, sc.ncw, lane);
    lds_barrier();  // B1
    lds_barrier();  // B2
  }
}

// One unit of an image role over all blocks of the segment.  REGW: the unit's blocks of W live in registers (loaded from
// the role's global image once per segment), else they are read from LDS as in run_unit.  TRIM 3 as in run_unit.
template <int KS, int NT, bool TAIL, int DBG, int TRIM, bool REGW>
__device__ __forceinline__ void run_unit_img(const UnitCtx &c, const int32_t *__restrict__ L, int lane) {
  constexpr int NPK = (KS + 1) / 2;
  constexpr int NQ = NT + (TAIL ? 1 : 0);
  constexpr int NP = NT / 2;
  static_assert(!REGW || (NT == 2 && !TAIL), "REGW units are one tile pair");
  unsigned apk[NPK];
#pragma unroll
  for (int q = 0; q < NPK; ++q) {
    const unsigned lo = (unsigned)L[(2 * q) * 64 + lane];
    const unsigned hi = (2 * q + 1 < KS) ? (unsigned)L[(2 * q + 1) * 64 + lane] : 0u;
    apk[q] = lo | (hi << 16);
  }
  const int relrow = L[20 * 64 + lane];  // tile row (lane & 15): run << 20 | CRS offset inside the run, -1 = no row
  const int imgrow = L[21 * 64 + lane];  // the same row's entry offset inside the image, before the run's alignment shift
  const int nct_all = (c.len + 15) / 16;
  const int stride = (nct_all % 2 == 1) ? 16 * nct_all : 16 * nct_all + 16;
  auto aoff = [&](int s) { return (s & 1) ? (apk[s >> 1] >> 16) : (apk[s >> 1] & 0xffffu); };
  const int l15 = lane & 15;
  const double *Wp = c.Wl + l15;
  const int cbase = __builtin_amdgcn_readfirstlane(16 * c.ct0), rlen = __builtin_amdgcn_readfirstlane(c.len - 16 * c.ct0);
  // REGW: B operands of the whole unit
  double Wr[REGW ? KS : 1][2];
  if constexpr (REGW) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const size_t idx = (size_t)(4 * s + (lane >> 4)) * stride + 16 * c.ct0 + 2 * l15 + q;
        Wr[s][q] = c.su * c.wk[idx] + c.st * c.wm[idx];
      }
      if (s % 2 == 1) __builtin_amdgcn_sched_barrier(0);  // (all 56 loads at once would need twice the registers)
    }
  }
  StreamCtx sc = make_stream(c, lane);
  // image entries of this lane inside a row, a trash slot for the lanes that have none (branch-free writes)
  const int trash = c.trash + lane;
  lds_barrier();  // B0
  for (int i = 0; i <= c.nblocks; ++i) {
    const int ic = min(i, c.nblocks - 1);
    stream_block(c, sc, i - 1);
    // image offset of tile row (lane & 15) of block i: the run's slot + its global offset mod 16
    const int irowC = (relrow >= 0 && i < c.nblocks) ? imgrow + (c.sbase[ic * c.nruns + (max(relrow, 0) >> 20)] & 15) : -1;
    asm volatile("" ::: "memory");  // (the LDS has changed)
    const double *rec = c.recbuf + (i & 1) * c.recstride;
    v4d acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
    auto kloop = [&](auto cls_tag) {
      constexpr int CLS = decltype(cls_tag)::value;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double a = rec[aoff(s)];
        if constexpr (!(DBG & 4)) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const bool second = s >= KS / 2;
            if ((CLS == 1 && q < 2 && second) || (CLS == 2 && q >= 2 && !second) || (CLS == 3 && q < 2 && !second) ||
                (CLS == 4 && q >= 2 && second))
              continue;  // compile time
            double b;
            if constexpr (REGW) b = Wr[s][q];
            else b = q < 2 * NP ? Wp[(4 * s) * stride + 32 * (q >> 1) + (q & 1)] : c.Wl[(4 * s) * stride + 16 * q];
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
          }
        } else {
          acc[0][0] += a;
        }
        // this wavefront's share of the previous block's image, spread over the k-steps
        stream_range<DBG>(c, sc, (s * sc.ncw) / KS, ((s + 1) * sc.ncw) / KS, lane);
      }
    };
    if constexpr (TRIM == 3) {
      switch (c.trim) {
        case 1: kloop(std::integral_constant<int, 1>{}); break;
        case 2: kloop(std::integral_constant<int, 2>{}); break;
        case 3: kloop(std::integral_constant<int, 3>{}); break;
        case 4: kloop(std::integral_constant<int, 4>{}); break;
        default: kloop(std::integral_constant<int, 0>{}); break;
      }
    } else {
      kloop(std::integral_constant<int, 0>{});
    }
    lds_barrier();  // B1: the image has been read out
    // results of block i -> image (every lane writes: those without an entry into the trash slot)
    int ro[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ro[t] = __builtin_amdgcn_ds_bpermute(((lane >> 4) + 4 * t) * 4, irowC);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rb = ro[t] + cbase;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = 32 * p + 2 * l15;
        c.img[(ro[t] >= 0 && col < rlen) ? rb + col : trash] = acc[2 * p][t];
        c.img[(ro[t] >= 0 && col + 1 < rlen) ? rb + col + 1 : trash] = acc[2 * p + 1][t];
      }
#pragma unroll
      for (int q = 2 * NP; q < NQ; ++q) {
        const int col = 16 * q + l15;
        c.img[(ro[t] >= 0 && col < rlen) ? rb + col : trash] = acc[q][t];
      }
    }
    lds_barrier();  // B2: the image is complete
  }
}

// Units of fixed dofs (isFixedDOF rows: the scatter skips them, assemblyManager.cpp:4075, 4120): zeros when storing.  A
// wavefront takes up to kMaxFixedUnits of them; no loads, one row per store instruction.
constexpr int kMaxFixedUnits = 6;
template <int DBG>
__device__ __forceinline__ void run_fixed_units(UnitCtx &c, const BlockPatternDev &d, int p_begin, int p_end, int lane) {
  const int nu = p_end - p_begin;
  int relrow[kMaxFixedUnits], col0[kMaxFixedUnits], ncol[kMaxFixedUnits];
#pragma unroll
  for (int u = 0; u < kMaxFixedUnits; ++u) {
    relrow[u] = -1; col0[u] = 0; ncol[u] = 0;
    if (u < nu) {
      const int32_t *h = d.part_hdr + (size_t)(p_begin + u) * kBpHdrInts;
      relrow[u] = d.part_lane[(size_t)(p_begin + u) * kBpLaneRows * 64 + 20 * 64 + lane];
      col0[u] = 16 * h[H_CT0];
      ncol[u] = min(h[H_LEN], 16 * (h[H_CT0] + h[H_NTILE] + ((h[H_FLAGS] & 2) ? 1 : 0))) - col0[u];  // <= 80
    }
  }
  const v2i zero = {0, 0};
  for (int i = 0; i <= c.nblocks; ++i) {  // (nblocks + 1 barriers per segment: the product units run one block behind with their stores)
    lds_barrier();
    if (i == c.nblocks) break;
#pragma unroll
    for (int u = 0; u < kMaxFixedUnits; ++u) {
      if (u >= nu) break;
      const int vrow = (relrow[u] >= 0 && !(DBG & 2)) ? c.sbase[i * c.nruns + (max(relrow[u], 0) >> 20)] + (relrow[u] & 0xfffff) : -1;
      for (int r = 0; r < 16; ++r) {
        const int rowoff = __builtin_amdgcn_readlane(vrow, r);
        if (rowoff < 0) continue;  // wave-uniform
        raw_buffer_store_b64(zero, c.out, (int)(lane < ncol[u] ? (unsigned)(col0[u] + lane) * 8u : kOutOfRange), rowoff * 8, 0);
        if (ncol[u] > 64) raw_buffer_store_b64(zero, c.out, (int)(lane + 64 < ncol[u] ? (unsigned)(col0[u] + 64 + lane) * 8u : kOutOfRange), rowoff * 8, 0);
      }
    }
  }
}

// One block of one unit, any depth / width, store or accumulate, plain (compiler-counted) loads: units the specialised
// form does not cover, rows of fixed dofs (zeros when storing), and the accumulate mode.
__device__ __forceinline__ void unit_generic(const UnitCtx &c, const int32_t *__restrict__ L, int lane, int i, int ks, int ntile,
                             bool fixed_class, bool overwrite) {
  const int nct_all = (c.len + 15) / 16;
  const int stride = (nct_all % 2 == 1) ? 16 * nct_all : 16 * nct_all + 16;
  const int l15 = lane & 15;
  int rel[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) rel[t] = L[(16 + t) * 64 + lane];
  const double *eb = c.erec_ptr + (size_t)(c.first + i) * c.estride * kBpRecDoubles;
  int base[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) base[t] = c.sbase[i * c.nruns + (max(rel[t], 0) >> 20)] + (rel[t] & 0xfffff);
  double A[kBpMaxKSteps];
#pragma unroll
  for (int s = 0; s < kBpMaxKSteps; ++s) A[s] = (s < ks) ? eb[L[s * 64 + lane]] : 0.0;
  for (int q = 0; q < ntile; ++q) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < kBpMaxKSteps; ++s)
      if (s < ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s], c.Wl[(4 * s) * stride + 16 * q], acc, 0, 0, 0);
    const int col = 16 * (c.ct0 + q) + l15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (rel[t] < 0 || col >= c.len) continue;
      double *p = c.vals + (size_t)base[t] + col;
      if (overwrite) *p = fixed_class ? 0.0 : acc[t];
      else if (!fixed_class) *p += acc[t];
    }
  }
}

// DBG (profiling launches only, env MHA_BP_DBG): 1 plain-load form of every unit, 2 no stores, 4 no products, 8 wall-clock
// stamps of every wavefront
// IMAGE: the launch for the roles with an LDS image (d.wg_seg_ptr_img); the two families of unit code do not fit one
// register budget, so each gets a kernel of its own.
template <int DBG, bool IMAGE = false>
__global__ __launch_bounds__(kBpWaves * 64) void block_pattern_jacobian_kernel(BlockPatternDev d, RowOut out, double su, double st) {
  extern __shared__ double W[];  // [max_w_doubles] the role's image, [kBpSegInts] ints: the segment's run offsets, 2 x element records of a block
  int *sbase = reinterpret_cast<int *>(W + d.max_w_doubles);
  double *recbuf = reinterpret_cast<double *>(sbase + kBpSegInts);  // [2][max_rec_doubles]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  UnitCtx c;
  c.vals = out.vals;
  c.out = make_rsrc(out.vals, (unsigned)(d.nnz * 8));
  long long *tlog = nullptr;
  int tslot = 0;
  constexpr bool TIMING = (DBG & 8) != 0;
  if constexpr (TIMING) tlog = d.timing + ((size_t)blockIdx.x * kBpWaves + wave) * 8;
  c.tlog = tlog;
  auto stamp = [&]() {  // slots 0 start, 1 image loaded, 2.. segments done, 7 end
    if constexpr (TIMING) {
      const int slot = tslot < 7 ? tslot : 7;
      if (lane == 0) tlog[slot] = wall_clock64();
      ++tslot;
    }
  };
  stamp();
  const int32_t *wsp = IMAGE ? d.wg_seg_ptr_img : d.wg_seg_ptr;
  const int sg_begin = wsp[blockIdx.x], sg_end = wsp[blockIdx.x + 1];
  int cur_role = -1;
  for (int sg = sg_begin; sg < sg_end; ++sg) {
    const int role = d.seg[4 * sg];
    const int32_t *ro = d.role + (size_t)role * kBpRoleInts;
    if (sg > sg_begin) lds_barrier();  // every wavefront is done with the previous segment's run offsets and image
    if (role != cur_role) {
      cur_role = role;
      // the role's image comes in two halves: rows of the stiffness components (x alpha_u kappa) and of the mass
      // component (x alpha_t rho cp); their sum is what the products read
      const long long woff = ((long long)ro[R_WOFF_HI] << 32) | (unsigned)ro[R_WOFF_LO];
      const double2 *srck = reinterpret_cast<const double2 *>(d.w + woff);
      const double2 *srcm = reinterpret_cast<const double2 *>(d.w + woff + ro[R_WDOUBLES]);
      double2 *dst = reinterpret_cast<double2 *>(W);
      const int n2 = (IMAGE ? ro[R_WLDS] : ro[R_WDOUBLES]) / 2;  // (image kernel: the REGW classes behind R_WLDS stay in memory)
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
    c.recbuf = recbuf;
    c.recstride = d.max_rec_doubles;
    const int32_t *pp = d.part_ptr + (size_t)role * (kBpWaves + 1);
    const int p_begin = pp[wave], p_end = pp[wave + 1];
    auto setup = [&](int p, int &ks, int &ntile, bool &fixed_class, bool &tail) -> const int32_t * {
      const int32_t *h = d.part_hdr + (size_t)p * kBpHdrInts;
      ks = h[H_KS];
      ntile = h[H_NTILE];
      fixed_class = (h[H_FLAGS] & 1) != 0;
      tail = (h[H_FLAGS] & 2) != 0;
      c.len = h[H_LEN];
      c.ct0 = __builtin_amdgcn_readfirstlane(h[H_CT0]);
#pragma unroll
      for (int wd = 0; wd < 3; ++wd) c.mask[wd] = (d.dbg & 16) ? 0xffffffffu : (unsigned)__builtin_amdgcn_readfirstlane(h[H_MASK0 + wd]);  // bit 16: no trim
      const int nct = h[H_NCT];
      const int stride = (nct % 2 == 1) ? 16 * nct : 16 * nct + 16;
      c.Wl = W + h[H_WOFF] + (lane >> 4) * stride + (lane & 15) + 16 * c.ct0;
      return d.part_lane + (size_t)p * kBpLaneRows * 64;
    };
    // fast form: this wavefront owns exactly one unit of a shape that is instantiated; it runs the whole segment (one
    // barrier per block inside).  Everything else goes block by block through the plain form.
    bool done = false;
    const bool loader = wave == 0;
    if constexpr (IMAGE) {
      // ---- image role: wave 0 loads records, every other wavefront runs its unit (if any) and streams ----
      c.img = W + ro[R_WLDS];
      c.trash = ro[R_IMG];  // (behind the image: the launcher adds the 512 bytes)
      c.chunks = d.chunk_tab + ((((long long)ro[R_CHUNK_HI]) << 32) | (unsigned)ro[R_CHUNK_LO]) * kBpChunkInts;
      c.nchunk = ro[R_NCHUNK];
      c.sidx = wave - 1;
      c.su = su;
      c.st = st;
      if (loader) {
        run_loader_img<DBG>(c, lane);
      } else if (p_end == p_begin) {
        run_stream_only<DBG>(c, lane);
      } else {
        int ks, ntile;
        bool fixed_class, tail;
        const int32_t *L = setup(p_begin, ks, ntile, fixed_class, tail);
        const int32_t *h = d.part_hdr + (size_t)p_begin * kBpHdrInts;
        const bool regw = (h[H_FLAGS] & 32) != 0;
        c.trim = __builtin_amdgcn_readfirstlane((d.dbg & 16) ? 0 : ((h[H_FLAGS] >> 2) & 7));
        const long long woff = ((long long)ro[R_WOFF_HI] << 32) | (unsigned)ro[R_WOFF_LO];
        c.wk = d.w + woff + h[H_WOFF];
        c.wm = c.wk + ro[R_WDOUBLES];
        const int key = ((ks * 8 + ntile) * 2 + (tail ? 1 : 0)) * 2 + (regw ? 1 : 0);
#define BP_IMG(KS_, NT_, TAIL_, TRIM_, REGW_)                                                \
  case ((KS_ * 8 + NT_) * 2 + (TAIL_ ? 1 : 0)) * 2 + (REGW_ ? 1 : 0):                        \
    run_unit_img<KS_, NT_, TAIL_, DBG, TRIM_, REGW_>(c, L, lane);                            \
    break;
        switch (key) {
          // Q2 hexes
          BP_IMG(14, 2, false, 0, true)
          BP_IMG(14, 4, false, 3, false)
          BP_IMG(7, 4, true, 0, false)
          BP_IMG(7, 4, false, 0, false)
          BP_IMG(4, 3, false, 0, false)
          BP_IMG(2, 2, false, 0, false)
          // Q1 hexes
          BP_IMG(14, 2, false, 0, false)
          BP_IMG(7, 2, false, 0, false)
          BP_IMG(4, 0, true, 0, false)
          BP_IMG(2, 0, true, 0, false)
          // quads
          BP_IMG(4, 2, false, 0, false)
          BP_IMG(4, 4, false, 0, false)
          BP_IMG(2, 3, false, 0, false)
          BP_IMG(1, 2, false, 0, false)
          BP_IMG(1, 0, true, 0, false)
          default: __builtin_trap();  // the planner only makes image roles of shapes listed here (kBpImageShapes)
        }
#undef BP_IMG
      }
      done = true;
    } else {
    if (!done && loader && p_end == p_begin && out.overwrite && !(DBG & 1)) {
      run_loader_only<DBG>(c, lane);
      done = true;
    }
    if (!done && p_end - p_begin == 1 && out.overwrite && !(DBG & 1)) {
      int ks, ntile;
      bool fixed_class, tail;
      const int32_t *L = setup(p_begin, ks, ntile, fixed_class, tail);
      c.trim = __builtin_amdgcn_readfirstlane((d.dbg & 16) ? 0 : ((d.part_hdr[(size_t)p_begin * kBpHdrInts + H_FLAGS] >> 2) & 7));  // host-computed trim class of the unit
      const int key = fixed_class ? -1 : (ks * 8 + ntile) * 2 + (tail ? 1 : 0);
      // zero-block pattern of the unit (bit s * 5 + q of c.mask set = the W block of k-step s, chain q is not zero)
      done = true;
      // (shapes of four or five chains have no LOADER form)
#define BP_UNIT(KS_, NT_, TAIL_, TRIM_)                                                       \
  case (KS_ * 8 + NT_) * 2 + (TAIL_ ? 1 : 0):                                                 \
    if constexpr (NT_ + (TAIL_ ? 1 : 0) <= 3) {                                               \
      if (loader) run_unit<KS_, NT_, TAIL_, DBG, TRIM_, true>(c, L, lane);                    \
      else run_unit<KS_, NT_, TAIL_, DBG, TRIM_, false>(c, L, lane);                          \
    } else {                                                                                  \
      if (loader) done = false;                                                               \
      else run_unit<KS_, NT_, TAIL_, DBG, TRIM_, false>(c, L, lane);                          \
    }                                                                                         \
    break;
      switch (key) {
        // Q2 hexes: 8 / 4 / 2 / 1 elements around a vertex / edge / face / cell dof
        BP_UNIT(14, 4, false, 3)
        BP_UNIT(7, 4, true, 0)
        BP_UNIT(7, 4, false, 0)
        BP_UNIT(4, 3, false, 0)
        BP_UNIT(2, 2, false, 0)
        // Q1 hexes
        BP_UNIT(14, 2, false, 0)
        BP_UNIT(7, 2, false, 0)
        BP_UNIT(4, 0, true, 0)
        BP_UNIT(2, 0, true, 0)
        // quads (one k-step per element)
        BP_UNIT(4, 2, false, 0)
        BP_UNIT(4, 4, false, 0)
        BP_UNIT(2, 3, false, 0)
        BP_UNIT(1, 2, false, 0)
        BP_UNIT(1, 0, true, 0)
        default: done = false; break;
      }
#undef BP_UNIT
    }
    if (!done && !loader && out.overwrite && !(DBG & 1) && p_end > p_begin && p_end - p_begin <= kMaxFixedUnits) {
      bool all_fixed = true;
      for (int p = p_begin; p < p_end; ++p) all_fixed = all_fixed && (d.part_hdr[(size_t)p * kBpHdrInts + H_FLAGS] & 1);
      if (all_fixed) {
        run_fixed_units<DBG>(c, d, p_begin, p_end, lane);
        done = true;
      }
    }
    if (!done) {
      for (int i = 0; i <= c.nblocks; ++i) {
        if (loader && i < c.nblocks) {  // plain form of the record fetch: this block's records, just in time
          const double *src = c.erec_ptr + (size_t)(c.first + i) * c.estride * kBpRecDoubles;
          double *dst = recbuf + (i & 1) * c.recstride;
          for (int k = lane; k < c.estride * kBpRecDoubles; k += 64) dst[k] = src[k];
        }
        lds_barrier();
        if (i == c.nblocks) break;
        for (int p = p_begin; p < p_end; ++p) {
          int ks, ntile;
          bool fixed_class, tail;
          const int32_t *L = setup(p, ks, ntile, fixed_class, tail);
          if (fixed_class && !out.overwrite) continue;
          unit_generic(c, L, lane, i, ks, ntile + (tail ? 1 : 0), fixed_class, out.overwrite != 0);
        }
      }
    }
    }
    stamp();
  }
  if constexpr (TIMING) { tslot = 7; stamp(); }
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

// ---- geometry-database mode: the rows of one block per pattern, replicated ------------------------------------------
// When every element of the block has the SAME geometry record (AssemblyManager: one shape in the geometry database)
// the CRS rows a row block produces depend on its assembly pattern only.  The Jacobian kernel then runs on ONE
// representative block per role and this kernel copies the representative's runs of consecutive rows to the same runs
// of every other block of the role.  The destination is walked in 1 KB chunks that start on 128-byte lines, 16 bytes
// per lane, so every store instruction but a run's first and last covers whole lines (the row pieces of the kernel
// above cannot: see DESIGN 3.1); the source comes out of the L2 (a pattern's rows are ~64 KB).  Bit-identical to the
// full kernel: the same inputs go through the same instructions, once instead of once per block.
// chunk = {destination / 16 bytes (a multiple of 8: the chunk starts on a 128-byte line), source entry of lane 0's first
// double (may be up to 15 below the run's first entry), first and one-past-last destination ENTRY of the run}: the list
// comes from the host, sorted by destination.  Four chunks per wavefront and pass, every load of the pass issued before
// its first store (one chunk at a time the kernel is a chain of dependent latencies: 326 us for the 1.07 GB of config 2).
template <int U, bool NT, bool NOLOAD = false>
__global__ __launch_bounds__(256) void replicate_runs_kernel(const int4 *__restrict__ chunks, int nchunks, double *vals) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (gridDim.x * 256) >> 6;
  for (int base = wave * U; base < nchunks; base += nwaves * U) {
    int4 t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = chunks[min(base + u, nchunks - 1)];  // wave-uniform: scalar loads
    double x0[U], x1[U];
    bool v0[U], v1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = (long long)t[u].x * 2 + 2 * lane;  // destination entry of this lane's first double
      v0[u] = base + u < nchunks && e >= t[u].z && e < t[u].w;
      v1[u] = base + u < nchunks && e + 1 >= t[u].z && e + 1 < t[u].w;
      const long long se = (long long)t[u].y + 2 * lane;
      x0[u] = (v0[u] && !NOLOAD) ? vals[se] : 1.0;  // (NOLOAD: profiling only -- what the stores alone cost)
      x1[u] = (v1[u] && !NOLOAD) ? vals[se + 1] : 1.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double *dst = vals + (long long)t[u].x * 2 + 2 * lane;
      if (v0[u] && v1[u]) {
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        v2d_t v = {x0[u], x1[u]};
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d_t *>(dst));
        else *reinterpret_cast<v2d_t *>(dst) = v;
      } else if (v0[u]) dst[0] = x0[u];
      else if (v1[u]) dst[1] = x1[u];
    }
  }
}

void launch_replicate_runs(const int32_t *chunks, int nchunks, double *vals, hipStream_t stream) {
  if (nchunks <= 0) return;
  MHA_REQUIRE((reinterpret_cast<uintptr_t>(vals) & 127u) == 0, MHA_ERR_INVALID, "geometry-database mode: CRS values must start on a 128-byte line");
  // nontemporal stores (the written lines are not read again before they leave the L2: 276 -> 249 us) and a grid that
  // gives every wavefront few passes (2048 workgroups 246 us, 8192: 225, 32768 with two chunks per pass: 220):
  // profiles/r3_k2_database_modes.sh.  MHA_REP_MODE: 0 four chunks per pass, plain stores; 1 four + nontemporal; 2 eight;
  // 3 eight + nontemporal; 4 (default) two + nontemporal; 5 one + nontemporal
  static const int mode = [] { const char *m = std::getenv("MHA_REP_MODE"); return m ? std::atoi(m) : 4; }();
  static const int wgs = [] { const char *m = std::getenv("MHA_REP_WGS"); return m ? std::atoi(m) : 32768; }();
  const int4 *ch = reinterpret_cast<const int4 *>(chunks);
  const int per_wg = 4 * (mode == 2 || mode == 3 ? 8 : mode == 4 ? 2 : mode == 5 ? 1 : 4);
  const int grid = std::min((nchunks + per_wg - 1) / per_wg, wgs);
  if (mode == 0) hipLaunchKernelGGL((replicate_runs_kernel<4, false>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else if (mode == 1) hipLaunchKernelGGL((replicate_runs_kernel<4, true>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else if (mode == 2) hipLaunchKernelGGL((replicate_runs_kernel<8, false>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else if (mode == 3) hipLaunchKernelGGL((replicate_runs_kernel<8, true>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else if (mode == 5) hipLaunchKernelGGL((replicate_runs_kernel<1, true>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else if (mode == 9) hipLaunchKernelGGL((replicate_runs_kernel<2, true, true>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else if (mode == 10) hipLaunchKernelGGL((replicate_runs_kernel<2, false, true>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  else hipLaunchKernelGGL((replicate_runs_kernel<2, true>), dim3(grid), dim3(256), 0, stream, ch, nchunks, vals);
  MHA_HIP(hipGetLastError());
}

void launch_block_pattern_jacobian(const BlockPatternDev &d, const RowOut &out, double su, double st, hipStream_t stream) {
  if (d.num_wgs <= 0 || !out.vals) return;
  const size_t lds = sizeof(double) * (size_t)d.max_w_doubles + sizeof(int) * kBpSegInts + 2 * sizeof(double) * (size_t)d.max_rec_doubles;
  MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "pattern matrices of " << lds << " B do not fit the LDS");
  auto go = [&](auto kern, const BlockPatternDev &dd) {
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(dd.num_wgs), dim3(kBpWaves * 64), lds, stream, dd, out, su, st);
  };
  // (bit 16 is a run-time switch: no k-step trim of the 14 x 4 units; 32 selects the small-record build)
  const int small_rec = (d.max_rec_doubles <= 384 && (d.dbg & 15) == 0 && !d.timing && !(d.dbg & 32)) ? 32 : 0;
  auto direct = [&](const BlockPatternDev &dd) {
    switch ((d.dbg & 15) | (d.timing ? 8 : 0) | small_rec) {
      case 32: go(block_pattern_jacobian_kernel<32>, dd); break;
      case 0: go(block_pattern_jacobian_kernel<0>, dd); break;
      case 1: go(block_pattern_jacobian_kernel<1>, dd); break;
      case 2: go(block_pattern_jacobian_kernel<2>, dd); break;
      case 4: go(block_pattern_jacobian_kernel<4>, dd); break;
      case 6: go(block_pattern_jacobian_kernel<6>, dd); break;
      case 8: go(block_pattern_jacobian_kernel<8>, dd); break;
      case 10: go(block_pattern_jacobian_kernel<10>, dd); break;
      case 12: go(block_pattern_jacobian_kernel<12>, dd); break;
      case 14: go(block_pattern_jacobian_kernel<14>, dd); break;
      default: MHA_REQUIRE(false, MHA_ERR_INVALID, "MHA_BP_DBG: unsupported combination");
    }
  };
  // the image roles first (most of a mesh: its interior), in the kernel of their own when rows are overwritten; when they
  // are accumulated (or on request, bit 1) through the plain form of the other kernel
  if (d.has_image) {
    if (out.overwrite && !(d.dbg & 1) && !d.timing) {
      switch (d.dbg & 6) {
        case 0: go(block_pattern_jacobian_kernel<0, true>, d); break;
        case 2: go(block_pattern_jacobian_kernel<2, true>, d); break;
        case 4: go(block_pattern_jacobian_kernel<4, true>, d); break;
        default: go(block_pattern_jacobian_kernel<6, true>, d); break;
      }
    } else {
      BlockPatternDev d2 = d;
      d2.wg_seg_ptr = d.wg_seg_ptr_img;
      direct(d2);
    }
  }
  if (d.has_direct) direct(d);
  MHA_HIP(hipGetLastError());
}

}  // namespace mha
