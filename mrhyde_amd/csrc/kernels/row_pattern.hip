// row_pattern.hip -- row-owner Jacobian of affine thermal elements as small GEMMs on the matrix cores.
//
// Replaces the per-entry sumIntoValues of the reference's scatter (src/managers/assemblyManager.cpp:4031-4145, with
// thermal::volumeResidual src/physics/thermal.cpp:125-163 supplying res(e,i).dx(j)) for affine elements and
// element-wise constant coefficients.  See row_pattern.hpp for the formulation: 16 CRS rows of one assembly pattern are
//   vals[16 rows][row length] = G[16 rows][K] * W[K][row length],   K = (incident elements) x (geometry components),
// G gathered from the per-element geometry factors (16 MB at 64^3, L2 resident), W constant per pattern and held in
// LDS by a persistent workgroup while it walks its range of the pattern-sorted row tiles.  One wavefront per 16 rows,
// v_mfma_f64_16x16x4_f64 (operand maps: A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15], D reg t:
// row = (lane>>4) + 4t, col = lane&15): register t of the 16 lanes of a quarter wave is 16 consecutive entries of one
// CRS row -- 128-byte stores.  No atomics, no accumulator in LDS, every value written once.
// HBM traffic: the CRS values once (the compulsory write), 8 B x K per row of gathers from L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device_math.hpp"
#include "launch.hpp"
#include "../row_pattern.hpp"

namespace mha {
namespace {

__global__ __launch_bounds__(256) void build_geok_kernel(int nelem, int nsym, int ke, const double *__restrict__ geo,
                                                         double *__restrict__ geok) {
  const int total = nelem * ke;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i / ke, c = i - e * ke;
    geok[i] = c < nsym ? geo[(size_t)e * kGeoRec + c] : (c == nsym ? geo[(size_t)e * kGeoRec + kGeoDet] : 0.0);
  }
}

constexpr int kMaxKSteps = 16;  // GEMM depth held in registers: 64 = 8 hexes x 8 or 16 quads x 4 (host: kMaxDepth)
constexpr int kMaxColTiles = 9;  // 144 columns: a Q2-hex vertex row (125) at any alignment (host checks)
constexpr int kRpThreads = kRowsPerSuperTile * 4;  // one wavefront per 16 rows of the super tile
#ifndef MHA_RP_ORDER
#define MHA_RP_ORDER 0  // 0: stores interleaved with the products; 1: products, next gather, wait, stores (spills: 2.4x slower)
#endif
#ifndef MHA_RP_MINW
#define MHA_RP_MINW 4  // waves per SIMD the register budget is cut for: 4 = two workgroups of 8 waves per CU
#endif

template <int KE>
__global__ __launch_bounds__(kRpThreads, MHA_RP_MINW) void row_pattern_jacobian_kernel(RowPatternDev rp, RowOut out, double su,
                                                                              double st) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int SPI = KE / 4;               // k-steps per incident element
  constexpr int MAXNI = kMaxKSteps / SPI;   // incident elements per row the registers hold
  extern __shared__ double W[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  auto lds_barrier = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  // scale of this lane's geometry component in k-step (q % SPI): alpha_u kappa on the stiffness components,
  // alpha_t rho cp on detJ (the mass term)
  double sc[SPI];
#pragma unroll
  for (int j = 0; j < SPI; ++j) {
    const int c = 4 * j + l4;
    sc[j] = c < rp.nsym ? su : (c == rp.nsym ? st : 0.0);
  }
  const int s_begin = rp.wg_ptr[blockIdx.x], s_end = rp.wg_ptr[blockIdx.x + 1];
  if (s_begin >= s_end) return;

  // Software pipeline over the workgroup's super tiles, three dependent loads deep: while tile s runs on the matrix
  // cores, the geometry factors of tile s+1 are being gathered (their element ids arrived one iteration ago), the row
  // record of tile s+2 is read (its descriptor arrived one iteration ago) and the descriptor of tile s+3 is requested.
  // Descriptors come through the vector memory path on purpose: a scalar load would share its counter with the LDS
  // reads of the product loop and stall them.
  struct Raw { int4 a, b; };  // as loaded: a = {pattern, ni, cols, stride}, b = {record offset lo, hi, W offset lo, hi}
  struct Desc { int pat, ni, shift, len, cols, stride, gstride; long long rec, woff; };  // wave-uniform (scalar registers)
  struct Rec { int base, meta, e[MAXNI]; };
  auto load_raw = [&](int s) {
    Raw d;
    d.a = make_int4(-1, 0, 0, 0);
    d.b = make_int4(0, 0, 0, 0);
    if (s < s_end) {
      const int4 *p = reinterpret_cast<const int4 *>(rp.st_desc) + 2 * (size_t)s;
      d.a = p[0];
      d.b = p[1];
    }
    return d;
  };
  auto uniform = [&](const Raw &r) {
    Desc d;
    d.pat = __builtin_amdgcn_readfirstlane(r.a.x);
    const int packed = __builtin_amdgcn_readfirstlane(r.a.y), strides = __builtin_amdgcn_readfirstlane(r.a.w);
    d.ni = packed & 0xff;
    d.shift = (packed >> 8) & 0xff;
    d.len = packed >> 16;
    d.cols = __builtin_amdgcn_readfirstlane(r.a.z);
    d.stride = strides & 0xffff;
    d.gstride = strides >> 16;
    d.rec = (long long)(unsigned)__builtin_amdgcn_readfirstlane(r.b.x) | ((long long)__builtin_amdgcn_readfirstlane(r.b.y) << 32);
    d.woff = (long long)(unsigned)__builtin_amdgcn_readfirstlane(r.b.z) | ((long long)__builtin_amdgcn_readfirstlane(r.b.w) << 32);
    return d;
  };
  auto load_rec = [&](const Desc &d) {
    Rec r;
    r.base = 0; r.meta = 0;
#pragma unroll
    for (int k = 0; k < MAXNI; ++k) r.e[k] = 0;
    if (d.ni > 0) {
      const int32_t *p = rp.st_rec + d.rec + (size_t)wave * (2 + d.ni) * 16 + l15;
      r.base = p[0];
      r.meta = p[16];
#pragma unroll
      for (int k = 0; k < MAXNI; ++k)
        if (k < d.ni) r.e[k] = p[(2 + k) * 16];
    }
    return r;
  };
  auto gather = [&](const Desc &d, const Rec &r, double *A) {  // G[row = l15][k = 4q + l4] = scale * geok[element q / SPI][4 (q % SPI) + l4]
#pragma unroll
    for (int q = 0; q < kMaxKSteps; ++q) {
      A[q] = 0.0;
      if (rp.dbg & 4) { A[q] = 1.0; continue; }
      if (q / SPI < d.ni && (r.meta & 0x3fffffff))
        A[q] = rp.geok[(size_t)r.e[q / SPI] * KE + 4 * (q % SPI) + l4] * sc[q % SPI];
    }
  };

  Desc d0 = uniform(load_raw(s_begin)), d1 = uniform(load_raw(s_begin + 1)), d2 = uniform(load_raw(s_begin + 2));
  Rec r0 = load_rec(d0), r1 = load_rec(d1);
  // at four waves per SIMD there are no registers for a second operand set: the next tile's factors are gathered
  // after this tile's products and the other wavefronts cover the latency
  constexpr bool kPrefetchA = MHA_RP_MINW < 4;
  double A_cur[kMaxKSteps], A_nxt[kPrefetchA ? kMaxKSteps : 1];
  gather(d0, r0, A_cur);
  int cur = -1;
  for (int s = s_begin; s < s_end; ++s) {
    const Raw raw3 = load_raw(s + 3);
    const Rec r2 = load_rec(d2);
    if constexpr (kPrefetchA) gather(d1, r1, A_nxt);
    const int pat = d0.pat, cols = d0.cols, stride = d0.stride;
    const int ks = d0.ni * SPI;
    if (pat != cur) {  // uniform in the workgroup: next pattern's W into LDS
      lds_barrier();   // every wave is done reading the old one
      cur = pat;
      // W[k][shift + c] = Wmem[k][c]: the row's columns start `shift` entries into their first 128-byte line, so the
      // 16-column tiles of the product are whole lines of the CRS values.  Columns outside [shift, shift + len) are
      // never stored and feed no other column: they may hold anything.
      const double *src = rp.w + d0.woff;
      const int len = d0.len, total = ks * 4 * len;
      for (int i = tid; i < total; i += kRpThreads) {
        const int k = i / len, c = i - k * len;
        W[k * stride + d0.shift + c] = src[(size_t)k * d0.gstride + c];
      }
      __syncthreads();
    }
    // destination rows of this lane's result registers: row l4 + 4t of the tile lives in lane l4 + 4t
    int obase[4], ometa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      obase[t] = __shfl(r0.base, l4 + 4 * t);
      ometa[t] = __shfl(r0.meta, l4 + 4 * t);
    }
    const int nct = cols >> 4;
    // branch-free chains for the depths that occur on quad / hex meshes (1, 2, 4, 8 elements around a dof): the LDS
    // reads of a chain are issued together and the MFMAs follow back to back
    auto chain = [&](auto kconst, const double *Bp) {
      constexpr int KS = decltype(kconst)::value;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < KS; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A_cur[q], Bp[(size_t)q * 4 * stride], acc, 0, 0, 0);
      return acc;
    };
    auto product = [&](int ct) {
      const double *Bp = W + l4 * stride + ct * 16 + l15;
      if (rp.dbg & 2) { v4d acc = {A_cur[0] + Bp[0], 0.0, 0.0, 0.0}; return acc; }
      switch (ks) {
        case 1: return chain(std::integral_constant<int, 1>(), Bp);
        case 2: return chain(std::integral_constant<int, 2>(), Bp);
        case 4: return chain(std::integral_constant<int, 4>(), Bp);
        case 8: return chain(std::integral_constant<int, 8>(), Bp);
        case 16: return chain(std::integral_constant<int, 16>(), Bp);
        default: break;
      }
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < kMaxKSteps; ++q)
        if (q < ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A_cur[q], Bp[(size_t)q * 4 * stride], acc, 0, 0, 0);
      return acc;
    };
    auto store_tile = [&](int ct, const v4d &acc) {
      const int col = ct * 16 + l15 - d0.shift;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (col < 0 || col >= (ometa[t] & 0x3fffffff)) continue;
        double *p = out.vals + (size_t)obase[t] + col;
        if (out.overwrite) {  // store: fixed rows (isFixedDOF, skipped by the scatter: assemblyManager.cpp:4075,4120) get zeros
          if (!((rp.dbg & 1) && acc[t] != 12345.678)) *p = (ometa[t] >> 30) ? 0.0 : acc[t];
        } else if (!(ometa[t] >> 30)) {  // accumulate: fixed rows stay untouched
          *p += acc[t];
        }
      }
    };
    if constexpr (kPrefetchA) {
      // Build variant (-DMHA_RP_MINW=2, one workgroup per CU): all column tiles first, then -- after the loads of the
      // next tiles have arrived -- the stores.  gfx950 counts loads and stores in one in-order vmcnt, so a load waited
      // for after a store waits for that store as well; in this order the loads run under the products and the stores
      // drain under the next tile's products.  Measured slower than 16 waves per CU without the prefetch
      // (profiles/README.md), kept for the record.
      v4d acc[kMaxColTiles];
#pragma unroll
      for (int ct = 0; ct < kMaxColTiles; ++ct)
        if (ct < nct) acc[ct] = product(ct);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): next tile's factors, records and descriptor are in
#pragma unroll
      for (int q = 0; q < kMaxKSteps; ++q) asm volatile("" : "+v"(A_nxt[q]));
#pragma unroll
      for (int ct = 0; ct < kMaxColTiles; ++ct)
        if (ct < nct) store_tile(ct, acc[ct]);
    } else {
#if MHA_RP_ORDER == 1
      // products of every column tile -> gather of the next tile into the (now dead) operand registers -> wait for it
      // -> stores.  gfx950 counts loads and stores in one in-order vmcnt: with the stores last, the only thing a wave
      // ever waits for is loads, and its stores drain under the next tile's products.
      v4d acc[kMaxColTiles];
#pragma unroll
      for (int ct = 0; ct < kMaxColTiles; ++ct)
        if (ct < nct) acc[ct] = product(ct);
      gather(d1, r1, A_cur);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
      for (int q = 0; q < kMaxKSteps; ++q) asm volatile("" : "+v"(A_cur[q]));
#pragma unroll
      for (int ct = 0; ct < kMaxColTiles; ++ct)
        if (ct < nct) store_tile(ct, acc[ct]);
#else
      for (int ct = 0; ct < nct; ++ct) store_tile(ct, product(ct));
#endif
    }
    if constexpr (kPrefetchA) {
#pragma unroll
      for (int q = 0; q < kMaxKSteps; ++q) A_cur[q] = A_nxt[q];
    } else {
#if MHA_RP_ORDER != 1
      gather(d1, r1, A_cur);
#endif
    }
    d0 = d1; d1 = d2; d2 = uniform(raw3);
    r0 = r1; r1 = r2;
  }
}

}  // namespace

void launch_build_geok(int nelem, int nsym, int ke, const double *geo, double *geok, hipStream_t stream) {
  if (nelem <= 0) return;
  const int grid = std::min((nelem * ke + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(build_geok_kernel, dim3(grid), dim3(256), 0, stream, nelem, nsym, ke, geo, geok);
  MHA_HIP(hipGetLastError());
}

void launch_row_pattern_jacobian(const RowPatternDev &rp, const RowOut &out, double su, double st, hipStream_t stream) {
  if (rp.num_wgs <= 0 || !out.vals) return;
  const size_t lds = sizeof(double) * (size_t)rp.max_w_doubles;
  MHA_REQUIRE(lds <= 80 * 1024, MHA_ERR_INVALID, "pattern matrix of " << lds << " B does not fit the LDS budget");
  auto go = [&](auto kern) {
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipLaunchKernelGGL(kern, dim3(rp.num_wgs), dim3(kRpThreads), lds, stream, rp, out, su, st);
    MHA_HIP(hipGetLastError());
  };
  if (rp.ke == 4) go(row_pattern_jacobian_kernel<4>);
  else if (rp.ke == 8) go(row_pattern_jacobian_kernel<8>);
  else MHA_REQUIRE(false, MHA_ERR_INVALID, "pattern kernel: unsupported depth per element " << rp.ke);
}

}  // namespace mha
