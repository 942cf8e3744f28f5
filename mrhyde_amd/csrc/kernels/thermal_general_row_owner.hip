// thermal_general_row_owner.hip -- thermal residual + Jacobian on GENERAL elements (non-affine geometry, coefficients
// that vary from point to point), row-owner form: no dense element matrices, no global atomics, no second pass.
//
// Replaces, for one thermal block of arbitrary hexes / quads, the whole volume part of AssemblyManager::assembleJacRes
//   gather + seeding      src/managers/assemblyManager.cpp:3598-3643, src/tools/workset.cpp:589-623, 836-847
//   geometry / basis      src/interfaces/discretizationInterface.cpp:732-776, 898-981 (recomputed on chip)
//   field evaluation      src/tools/workset.cpp:937-1062
//   thermal::volumeResidual   src/physics/thermal.cpp:71-165
//   scatter (fused)       src/managers/assemblyManager.cpp:4031-4145
// It is the general-element sibling of the affine row-owner pair (thermal_affine_residual.hip + block_pattern.hip) and
// takes over from thermal_general.hip + row_gather.hip, whose dense [E][n][n] round trip was 3.1 GB of the 4.3 GB the
// perturbed config 2 moved per assembly.
//
// Work unit = one ROW BLOCK of the host partition (row_blocks.hpp: the rows first touched by a Morton chunk of 2x2x2 /
// 4x4 elements, the <= 27 elements incident to them, and the (element, LID slot) PAIRS whose row the block owns).  One
// persistent workgroup of 8 wavefronts per CU walks the blocks.  Per block:
//   G1 fields of the touched elements on the matrix cores: [elements x dofs] x [dofs x (component, point)] gives the
//      reference gradient of the seeded solution (and u_t) at every point -- 16 small tile products shared by the waves
//   G2 one thread per (touched element, point): J, det, J^-1 from the vertices, the coefficient functions,
//      D_q = kappa w det J^-1 J^-T, the flux D_q grad u and the scalar residual data -> LDS.  Geometry is recomputed by
//      every block that touches the element (27/8 times on a hex mesh): ~150 flop per point, against the 2187
//      multiply-adds per pair of the product below
//   T  the Jacobian rows, 16 pairs at a time on the matrix cores (v_mfma_f64_16x16x4_f64):
//        K[pair][j] = sum_{(b,q)} P[pair][(b,q)] * Ghat[(b,q)][j],   P[pair][(b,q)] = alpha_u sum_a D_q^{ab}(elem) dhat_a N_i(q)
//      Only the rows a block owns are formed, so the product work is exactly that of the element-by-element sweep
//      (E n rows) however the rows are partitioned.  Ghat (the B operand) is the same for every pair of every element:
//      registers, re-read from LDS once per block.  A lane builds its three A values of a point group from six D entries
//      and three reference gradients read from LDS; the same reads give the pair's residual entry by quadrature
//      (thermal.cpp:125-163 restricted to the owned rows), reduced over the four lane groups and added to the row's sum.
//      The 16 x n results go into the block's CRS image in LDS with ds_add_f64 through the block-major one-byte slot
//      table (a row's elements overlap in most of its columns).
//      Meanwhile the fetching waves (0-1) start the phase by fetching the NEXT block's tables, vertices and seeded solution values
//      (global -> LDS, two dependent loads deep); the other wave of each SIMD runs its tiles under that latency
//   S  the finished rows stream out in contiguous runs (and are zeroed), the residual rows are written: every CRS
//      entry and every residual entry is stored exactly once, by its owner -- no atomics on global memory.
// HBM traffic per assembly = the CRS values + residual once, the slot table (one byte per (pair, column)), the block
// tables (row ids of the touched elements' dofs, pairs, runs), and vertices / solution of every element once per
// touching block (L2 absorbs most of the repeats).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device_math.hpp"
#include "launch.hpp"

namespace mha {
namespace {

constexpr int cpowr(int b, int e) { return e == 0 ? 1 : b * cpowr(b, e - 1); }
__host__ __device__ constexpr int sym_index(int dim, int a, int c) {  // upper triangle, row-major: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
  const int lo = a < c ? a : c, hi = a < c ? c : a;  // (no recursion: the optimiser must fold this inside unrolled loops)
  return lo * dim - lo * (lo - 1) / 2 + (hi - lo);
}

template <int DIM, int P, int NQ1, bool TR>
struct RG {
  static constexpr int M = P + 1;
  static constexpr int N = cpowr(M, DIM);
  static constexpr int NQ = cpowr(NQ1, DIM);
  static constexpr int NQ4 = (NQ + 3) / 4 * 4;      // points padded to whole k-steps
  static constexpr int NN = 1 << DIM;
  static constexpr int NSYM = DIM * (DIM + 1) / 2;
  static constexpr int QG = NQ4 / 4;                // k-steps per gradient component
  static constexpr int KSG = DIM * QG;              // k-steps of the stiffness product: k = (b, q)
  static constexpr int KSM = TR ? QG : 0;           // k-steps of the mass product
  static constexpr int CT = N > 16 ? 2 : 1;         // column tiles
  static constexpr int NT = 512, NW = NT / 64;
  // tables shared by all blocks (doubles)
  // dhat_a N_j(q) in the layout of the B operand: [(a, q/4)][q%4][j%16][j/16] -- a k-step's B values of a wavefront are
  // 64 consecutive entries (one ds_read per lane, both column tiles); entries with j >= N or q >= NQ are zero
  static constexpr int S_G = 0;
  static constexpr int G_SIZE = DIM * QG * 64 * CT;
  static constexpr int S_NV = S_G + G_SIZE;         // N_i(q)         [N][NQ4]
  static constexpr int S_XI = S_NV + N * NQ4;       // reference point coordinates [NQ][DIM]
  static constexpr int S_W = S_XI + NQ * DIM;       // reference weights [NQ]
  static constexpr int SHARED = (S_W + NQ + 1) / 2 * 2;
  // per touched element (doubles)
  static constexpr int E_D = NSYM * NQ4, E_M = NQ4, E_F = DIM * NQ4, E_S = NQ4;
  static constexpr int SCRATCH_PER_ELEM = NN * DIM + (TR ? 2 : 1) * N;  // vertices + seeded u (+ u_dot)
  static_assert(N <= 32, "two column tiles of 16");
};

// LDS carve (bytes).  A "table buffer" holds one block's small tables; there are two (current / next).
struct RgLds {
  int d, m, f, s, racc, acc, scratch, ids, slot, tab[2], total;
  int t_pairs, t_segs, t_rows, t_elems, t_pairoff, tab_bytes;  // offsets inside a table buffer
};
template <int DIM, int P, int NQ1, bool TR>
__host__ __device__ inline RgLds rg_layout(int T, int rows, int acc, int pairs, int segs) {
  using S = RG<DIM, P, NQ1, TR>;
  RgLds l;
  int o = S::SHARED * 8;
  l.d = o; o += T * S::E_D * 8;
  l.m = o; o += TR ? T * S::E_M * 8 : 0;
  l.f = o; o += T * S::E_F * 8;
  l.s = o; o += T * S::E_S * 8;
  l.racc = o; o += ((rows + 1) / 2 * 2) * 8;
  l.acc = o; o += ((acc + 1) / 2 * 2) * 8;
  l.scratch = o; o += T * S::SCRATCH_PER_ELEM * 8;
  l.ids = o; o += (T * S::N * 4 + 15) / 16 * 16;
  l.slot = o; o += (pairs * S::N + 15) / 16 * 16;
  int t = 0;
  l.t_pairs = t; t += pairs * 4;
  l.t_segs = t; t += segs * 12;
  l.t_rows = t; t += rows * 4;
  l.t_elems = t; t += T * 4;
  l.t_pairoff = t; t += pairs * 2;
  l.tab_bytes = (t + 15) / 16 * 16;
  l.tab[0] = o; o += l.tab_bytes;
  l.tab[1] = o; o += l.tab_bytes;
  l.total = o;
  return l;
}

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef MHA_RG_FETCH_WAVES
#define MHA_RG_FETCH_WAVES 2
#endif
constexpr int kRgSlotRegs = 2;  // uint4 per fetching thread: pairs * n <= 2 * 256 * 16 bytes

template <int DIM, int P, int NQ1, bool TR, bool EXPR, bool TIMING>
__global__ __launch_bounds__(512) void thermal_general_row_owner_kernel(BlockDev b, ThermalDev ph, RowBlocksDev rb,
                                                                        const uint8_t *__restrict__ slot8,
                                                                        const int32_t *__restrict__ blk_rows,
                                                                        const double *__restrict__ gp1d,
                                                                        const int32_t *__restrict__ blk_hdr, RowOut out,
                                                                        int dbg_arg, long long *timing) {
  const int dbg = TIMING ? dbg_arg : 0;  // the ablation switches exist in the profiling build only
  // dbg (env MHA_GRO_DBG, profiling only -- results are wrong): 1 no field / geometry phases, 2 no residual sums,
  // 4 no products, 8 no LDS adds of the tiles, 16 no CRS stores, 32 no fetch of the next block (stale data)
  using S = RG<DIM, P, NQ1, TR>;
  constexpr int N = S::N, NQ = S::NQ, NQ4 = S::NQ4, NN = S::NN, NSYM = S::NSYM, QG = S::QG, KSM = S::KSM;
  constexpr int CT = S::CT, NT = S::NT, NW = S::NW;
  extern __shared__ double smem[];
  char *base = reinterpret_cast<char *>(smem);
  const RgLds L = rg_layout<DIM, P, NQ1, TR>(rb.lds_elems, rb.lds_rows, rb.lds_acc, rb.lds_pairs, rb.lds_segs);
  double *sh = smem;
  double *s_D = reinterpret_cast<double *>(base + L.d), *s_M = reinterpret_cast<double *>(base + L.m);
  double *s_F = reinterpret_cast<double *>(base + L.f), *s_S = reinterpret_cast<double *>(base + L.s);
  double *racc = reinterpret_cast<double *>(base + L.racc), *acc = reinterpret_cast<double *>(base + L.acc);
  double *s_xn = reinterpret_cast<double *>(base + L.scratch);  // vertices, seeded u, u_dot of the touched elements
  int *s_ids = reinterpret_cast<int *>(base + L.ids);           // next block: global row of (touched element, dof)
  uint8_t *s_slot = reinterpret_cast<uint8_t *>(base + L.slot);
  __shared__ int s_offs[N], s_dofpos[N];
  __shared__ int s_tile, s_job;  // next tile / field job of the current block (taken by the waves as they become free)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
  const TimeDev &tm = ph.time;

  // ---- once per workgroup: zero the LDS (padding entries must stay finite), reference tables ----
  for (int i = tid; i < L.total / 8; i += NT) smem[i] = 0.0;
  __syncthreads();
  auto gidx = [](int a, int q, int j) { return ((a * QG + (q >> 2)) * 64 + (q & 3) * 16 + (j & 15)) * CT + (j >> 4); };
  for (int i = tid; i < N * NQ * DIM; i += NT) {
    const int j = i / (NQ * DIM), r = i - j * NQ * DIM, q = r / DIM, a = r - q * DIM;
    sh[S::S_G + gidx(a, q, j)] = b.ref_grad[i];
  }
  for (int i = tid; i < N * NQ; i += NT) sh[S::S_NV + (i / NQ) * NQ4 + i % NQ] = b.ref_basis[i];
  for (int i = tid; i < NQ * DIM; i += NT) {  // tensor cubature, x fastest (ref_tables.cpp)
    const int q = i / DIM, d = i - q * DIM;
    int qd = q;
    for (int k = 0; k < d; ++k) qd /= NQ1;
    sh[S::S_XI + i] = gp1d[qd % NQ1];
  }
  for (int i = tid; i < NQ; i += NT) sh[S::S_W + i] = b.ref_wts[i];
  for (int i = tid; i < N; i += NT) { s_offs[i] = b.offsets[i]; s_dofpos[b.offsets[i]] = i; }
  __syncthreads();
  auto lds_barrier = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  const bool jac = out.compute_jacobian != 0;

  // Fetch of a block's inputs by the threads of the fetching waves (waves 0..NF-1, NF = 2), in two stages so that nothing waits on a chain of
  // dependent loads.  Stage A (while the current block's fields are formed): the small tables and the slot table into
  // the other table buffer, the row ids of the touched elements' dofs into s_ids.  Stage B (once the scratch is free,
  // after G2, while the other waves start on the tiles): vertices and seeded solution values through those ids.
  // Every load of a stage is issued before its first store.
  constexpr int NF = MHA_RG_FETCH_WAVES;  // waves that load from global memory; the other NW - NF store to it
  constexpr int PT = NF * 64;
  constexpr int KA = (256 + PT - 1) / PT;                         // table entries per fetching thread (every table holds <= 256)
  constexpr int MAXT = (DIM == 3) ? 27 : 25;            // touched elements of a block (host caps)
  constexpr int UI = (MAXT * N + PT - 1) / PT;         // solution values per fetching thread
  constexpr int XI = (MAXT * NN * DIM + PT - 1) / PT;  // vertex coordinates per fetching thread
  int a_ur[UI], a_po[KA], a_sg0[KA], a_sg1[KA], a_sg2[KA], a_rw[KA], a_el = 0;  // stage A values in flight
  uint32_t a_pr[KA];
#pragma unroll
  for (int k = 0; k < KA; ++k) { a_po[k] = 0; a_sg0[k] = 0; a_sg1[k] = 0; a_sg2[k] = 0; a_rw[k] = 0; a_pr[k] = 0u; }
#define MHA_RG_STAGE_A_LOAD(h_)                                                                                     \
  {                                                                                                                 \
    const int t0 = (h_)[0], T_ = (h_)[1], p0 = (h_)[2], NP_ = (h_)[3], r0 = (h_)[4], NR_ = (h_)[5];                 \
    const int g0 = (h_)[6], NS_ = (h_)[7];                                                                          \
    _Pragma("unroll") for (int k = 0; k < UI; ++k) a_ur[k] = blk_rows[(size_t)t0 * N + min(tid + k * PT, T_ * N - 1)]; \
    _Pragma("unroll") for (int k = 0; k < KA; ++k) {                                                                \
      const int i = tid + k * PT;                                                                                   \
      if (i < NP_) { a_pr[k] = rb.pairs[p0 + i]; a_po[k] = rb.pair_off[p0 + i]; }                                   \
      if (i < NS_) { a_sg0[k] = rb.seg_acc[g0 + i]; a_sg1[k] = rb.seg_base[g0 + i]; a_sg2[k] = rb.seg_len[g0 + i]; } \
      if (i < NR_) a_rw[k] = rb.row_len[r0 + i] < 0 ? ~rb.rows[r0 + i] : rb.rows[r0 + i];                           \
    }                                                                                                               \
    if (tid < T_) a_el = rb.elems[t0 + tid];                                                                        \
  }
#define MHA_RG_STAGE_A_STORE(h_, tb_)                                                                               \
  {                                                                                                                 \
    const int T_ = (h_)[1], NP_ = (h_)[3], NR_ = (h_)[5], NS_ = (h_)[7];                                            \
    char *tbn = (tb_);                                                                                              \
    _Pragma("unroll") for (int k = 0; k < UI; ++k) if (tid + k * PT < T_ * N) s_ids[tid + k * PT] = a_ur[k];        \
    _Pragma("unroll") for (int k = 0; k < KA; ++k) {                                                                \
      const int i = tid + k * PT;                                                                                   \
      if (i < NP_) {                                                                                                \
        reinterpret_cast<uint32_t *>(tbn + L.t_pairs)[i] = a_pr[k];                                                 \
        reinterpret_cast<uint16_t *>(tbn + L.t_pairoff)[i] = (uint16_t)a_po[k];                                     \
      }                                                                                                             \
      if (i < NS_) {                                                                                                \
        int *t_segs = reinterpret_cast<int *>(tbn + L.t_segs);                                                      \
        t_segs[i] = a_sg0[k];                                                                                       \
        t_segs[rb.lds_segs + i] = a_sg1[k];                                                                         \
        t_segs[2 * rb.lds_segs + i] = a_sg2[k];                                                                     \
      }                                                                                                             \
      if (i < NR_) reinterpret_cast<int *>(tbn + L.t_rows)[i] = a_rw[k];                                            \
    }                                                                                                               \
    if (tid < T_) reinterpret_cast<int *>(tbn + L.t_elems)[tid] = a_el;                                             \
  }
#define MHA_RG_STAGE_B(h_, tb_)                                                                                     \
  {                                                                                                                 \
    const int T_ = (h_)[1];                                                                                         \
    const int *t_el = reinterpret_cast<const int *>((tb_) + L.t_elems);                                             \
    double *n_ue = s_xn + T_ * NN * DIM, *n_ud = n_ue + T_ * N;                                                     \
    double ue[UI], ud[UI], xv[XI];                                                                                  \
    _Pragma("unroll") for (int k = 0; k < UI; ++k) {                                                                \
      const int row = s_ids[min(tid + k * PT, T_ * N - 1)];                                                         \
      const double cu = tm.u[row];                                                                                  \
      ue[k] = cu;                                                                                                   \
      ud[k] = 0.0;                                                                                                  \
      if constexpr (TR) { /* Workset::computeSolnTransientSeeded, value parts (workset.cpp:589-623) */              \
        const double *cp = tm.u_prev + (size_t)row * tm.nsteps, *cs = tm.u_stage + (size_t)row * tm.nstages;        \
        double beta_u = (1.0 - tm.alpha_u) * cp[0];                                                                 \
        for (int s = 0; s < tm.stage; ++s) beta_u += tm.stage_ratio[s] * (cs[s] - cp[0]);                           \
        double beta_t = 0.0;                                                                                        \
        for (int s = 1; s < tm.nsteps + 1; ++s) beta_t += tm.bdf[s] * cp[s - 1];                                    \
        beta_t *= tm.timewt;                                                                                        \
        ue[k] = tm.alpha_u * cu + beta_u;                                                                           \
        ud[k] = tm.alpha_t * cu + beta_t;                                                                           \
      }                                                                                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < XI; ++k) {                                                                \
      const int i = min(tid + k * PT, T_ * NN * DIM - 1), t = i / (NN * DIM);                                       \
      xv[k] = b.nodes[(size_t)t_el[t] * NN * DIM + (i - t * NN * DIM)];                                             \
    }                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < UI; ++k) if (tid + k * PT < T_ * N) {                                     \
      n_ue[tid + k * PT] = ue[k];                                                                                   \
      if constexpr (TR) n_ud[tid + k * PT] = ud[k];                                                                 \
    }                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < XI; ++k) if (tid + k * PT < T_ * NN * DIM) s_xn[tid + k * PT] = xv[k];    \
  }

  // Block headers {first touched element, T, first pair, NP, first row, NR, first run, NS, slot offset / 16, slot
  // uint4s} travel one block ahead in LDS: no wave ever waits on a scalar load chain at the top of a block.
  constexpr int HW = 12;
  __shared__ int s_hdr[2][HW];
  // the slot table of the next block: requested by the fetching waves when their stage B is issued, held in registers
  // under their tiles, stored once the current block's accumulation is over (after the barrier that ends T)
  constexpr int KS4 = (kRgSlotRegs * 256 + PT - 1) / PT;
  uint4 slr[KS4];
#pragma unroll
  for (int k = 0; k < KS4; ++k) slr[k] = uint4{0u, 0u, 0u, 0u};
#define MHA_RG_SLOT_LOAD(h_)                                                                   \
  {                                                                                            \
    const uint4 *ssrc = reinterpret_cast<const uint4 *>(slot8) + (h_)[8];                      \
    const int n16 = (h_)[9];                                                                   \
    _Pragma("unroll") for (int k = 0; k < KS4; ++k) if (tid + k * PT < n16) slr[k] = ssrc[tid + k * PT]; \
  }
#define MHA_RG_SLOT_STORE(h_)                                                                  \
  {                                                                                            \
    uint4 *dsl = reinterpret_cast<uint4 *>(s_slot);                                            \
    const int n16 = (h_)[9];                                                                   \
    _Pragma("unroll") for (int k = 0; k < KS4; ++k) if (tid + k * PT < n16) dsl[tid + k * PT] = slr[k]; \
  }
  int cur = 0;  // table buffer / header slot of the current block
  const bool fetcher = wave < NF;
  if (tid == 0) { s_job = 0; s_tile = 0; }
  if ((int)blockIdx.x < rb.num_blocks) {
    if (tid < HW) {
      s_hdr[0][tid] = blk_hdr[(size_t)blockIdx.x * HW + tid];
      if ((int)(blockIdx.x + gridDim.x) < rb.num_blocks) s_hdr[1][tid] = blk_hdr[(size_t)(blockIdx.x + gridDim.x) * HW + tid];
    }
    lds_barrier();
    if (fetcher) {
      MHA_RG_SLOT_LOAD(s_hdr[0])
      MHA_RG_STAGE_A_LOAD(s_hdr[0])
      MHA_RG_STAGE_A_STORE(s_hdr[0], base + L.tab[0])
      MHA_RG_SLOT_STORE(s_hdr[0])
    }
    lds_barrier();
    if (fetcher) MHA_RG_STAGE_B(s_hdr[0], base + L.tab[0])
  }
  lds_barrier();
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define MHA_RG_STAMP(k_) if constexpr (TIMING) { const long long now = __builtin_readcyclecounter(); tacc[k_] += now - tprev; tprev = now; }

  for (int blk = blockIdx.x; blk < rb.num_blocks; blk += gridDim.x) {
    if constexpr (TIMING) tprev = __builtin_readcyclecounter();
    int hc[HW], hn[HW];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      hc[k] = __builtin_amdgcn_readfirstlane(s_hdr[cur][k]);
      hn[k] = __builtin_amdgcn_readfirstlane(s_hdr[cur ^ 1][k]);
    }
    const int T = hc[1], NP = hc[3], NR = hc[5], NS = hc[7];
    const int next = blk + gridDim.x;
    const bool prefetch = fetcher && next < rb.num_blocks && !(dbg & 32);
    char *tb = base + L.tab[cur], *tbnext = base + L.tab[cur ^ 1];
    const uint32_t *s_pairs = reinterpret_cast<const uint32_t *>(tb + L.t_pairs);
    const int *s_segs = reinterpret_cast<const int *>(tb + L.t_segs), *s_rows = reinterpret_cast<const int *>(tb + L.t_rows);
    const int *s_elems = reinterpret_cast<const int *>(tb + L.t_elems);
    const uint16_t *s_pairoff = reinterpret_cast<const uint16_t *>(tb + L.t_pairoff);
    const double *s_ue = s_xn + T * NN * DIM, *s_ud = s_ue + T * N;
    int hdr2 = 0;  // header of the block after the next one (lanes 0..HW-1 of wave 0), stored after the first barrier
    if (tid < HW && next + (int)gridDim.x < rb.num_blocks) hdr2 = blk_hdr[(size_t)(next + gridDim.x) * HW + tid];

    // ---- G1 (all waves, the fetching ones join late). fields on the matrix cores: gu[t][(a,q)] = sum_j ue[t][j] dhat_a N_j(q) -> s_F,
    //      tt[t][q] = sum_j ud[t][j] N_j(q) -> s_S; the fetching waves issue stage A of the next block first
    if (prefetch) MHA_RG_STAGE_A_LOAD(hn)
    if (!(dbg & 1)) {
      constexpr int KJ = (N + 3) / 4, CG = (DIM * NQ4 + 15) / 16, CS = TR ? (NQ4 + 15) / 16 : 0;
      const int RT = (T + 15) / 16;
      for (;;) {  // jobs are taken as the waves become free (the fetching waves join after issuing stage A)
        int job = 0;
        if (lane == 0) job = atomicAdd(&s_job, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= RT * (CG + CS)) break;
        const int rt = job / (CG + CS), cj = job - rt * (CG + CS);
        const bool grad = cj < CG;
        const int c = (grad ? cj : cj - CG) * 16 + l15;   // output column of this lane's B values
        const int t = rt * 16 + l15;                       // element of this lane's A values
        const double *ua = (grad ? s_ue : s_ud) + min(t, T - 1) * N;
        const bool cok = grad ? c < DIM * NQ4 : c < NQ4;
        const int ac = min(c / NQ4, DIM - 1), qc = min(c - (c / NQ4) * NQ4, NQ4 - 1);  // clamped: loads are unconditional
        const int bbase = grad ? S::S_G + gidx(ac, qc, 0) : S::S_NV + qc;  // + offset of basis function j
        double av[KJ], bv[KJ];
#pragma unroll
        for (int ks = 0; ks < KJ; ++ks) {  // all operands first: one LDS latency per job, not one per k-step
          const int j = 4 * ks + g4, jc = min(j, N - 1);
          const double ar = ua[jc], br = sh[bbase + (grad ? (jc & 15) * CT + (jc >> 4) : jc * NQ4)];
          av[ks] = (j < N && t < T) ? ar : 0.0;
          bv[ks] = (j < N && cok) ? br : 0.0;
        }
        v4d d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KJ; ++ks) d = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], d, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int te = rt * 16 + g4 + 4 * u;
          if (te < T && cok) (grad ? s_F + te * DIM * NQ4 : s_S + te * NQ4)[c] = d[u];
        }
      }
    }
    lds_barrier();
    // Every wave is past the previous block's store phase and has read the current header: the other table buffer and
    // the header slot may be replaced now (this is what lets a block end without a barrier of its own).
    if (prefetch) MHA_RG_STAGE_A_STORE(hn, tbnext)
    if (tid < HW) s_hdr[cur][tid] = hdr2;
    if (tid == 0) { s_job = 0; s_tile = 0; }   // fields done, tiles not started: both counters are idle here
    MHA_RG_STAMP(0)

    // ---- G2. geometry, coefficients, point-wise residual data at (touched element, point) ----
    for (int item = tid; item < ((dbg & 1) ? 0 : T * NQ); item += NT) {
      const int t = item / NQ, q = item - t * NQ;
      const int e = s_elems[t];
      const double *xn = s_xn + t * NN * DIM;
      // C1 geometry basis at the point, from the reference coordinates (shards vertex order: the (-,-) (+,-) (+,+) (-,+)
      // loop at z = -1, then at z = +1)
      double hm[DIM], hp[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        const double xi = sh[S::S_XI + q * DIM + d];
        hm[d] = 0.5 * (1.0 - xi);
        hp[d] = 0.5 * (1.0 + xi);
      }
      double J[DIM * DIM], Ji[DIM * DIM], det, x[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < DIM * DIM; ++i) J[i] = 0.0;
#pragma unroll
      for (int v = 0; v < NN; ++v) {
        const int cc = v & 3;
        const bool px = cc == 1 || cc == 2, py = cc >= 2, pz = v >= 4;
        const double fx = px ? hp[0] : hm[0], fy = py ? hp[1] : hm[1], fz = (DIM == 3) ? (pz ? hp[DIM - 1] : hm[DIM - 1]) : 1.0;
        double gv[DIM];  // gradient of the vertex function, value
        gv[0] = (px ? 0.5 : -0.5) * fy * fz;
        gv[1] = (py ? 0.5 : -0.5) * fx * fz;
        if constexpr (DIM == 3) gv[DIM - 1] = (pz ? 0.5 : -0.5) * fx * fy;
        const double val = fx * fy * fz;
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          const double xr = xn[v * DIM + r];
          x[r] += xr * val;
#pragma unroll
          for (int c = 0; c < DIM; ++c) J[r * DIM + c] += xr * gv[c];
        }
      }
      invert<DIM>(J, Ji, det);
      const double w = sh[S::S_W + q] * det;
      const double kap = eval_func<DIM, EXPR>(ph.diff, e, q, NQ, x);
      const double rc = eval_func<DIM, EXPR>(ph.rho, e, q, NQ, x) * eval_func<DIM, EXPR>(ph.cp, e, q, NQ, x);
      const double f = eval_func<DIM, EXPR>(ph.source, e, q, NQ, x);
      double D[DIM][DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a)
#pragma unroll
        for (int c = a; c < DIM; ++c) {
          double s = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) s += Ji[a * DIM + d] * Ji[c * DIM + d];
          D[a][c] = D[c][a] = kap * w * s;
          s_D[(t * NSYM + sym_index(DIM, a, c)) * NQ4 + q] = tm.alpha_u * D[a][c];
        }
      double gh[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) gh[a] = s_F[(t * DIM + a) * NQ4 + q];
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) s += D[a][c] * gh[c];
        s_F[(t * DIM + a) * NQ4 + q] = s;  // own (t, q) entries only: no other thread reads them in this phase
      }
      const double tt = TR ? s_S[t * NQ4 + q] : 0.0;
      s_S[t * NQ4 + q] = (rc * tt - f) * w;
      if constexpr (TR) s_M[t * NQ4 + q] = tm.alpha_t * rc * w;
    }
    lds_barrier();  // point data complete; the scratch (vertices, seeded values) is dead
    MHA_RG_STAMP(1)

    // ---- T. stage B of the next block (fetching waves), Jacobian rows + residual entries of the pairs ----
    if (prefetch) {
      MHA_RG_SLOT_LOAD(hn)
      MHA_RG_STAGE_B(hn, tbnext)
    }
    MHA_RG_STAMP(2)
    auto tiles = [&](auto jac_c) {
      constexpr bool JAC = decltype(jac_c)::value;
      // B[k = lane>>4 (+4s)][col = lane&15] of k-step s = (b, four points) is read from the Ghat table for every product (one
      // 16-byte read per lane covers both column tiles): held in registers for the whole tile loop (84 VGPRs) it left no
      // room to request a point group's operands ahead of the previous group's products.  The mass columns (N_j(q)) are
      // few enough to stay in registers.
      double Bm[KSM > 0 ? KSM : 1][CT];
      if constexpr (JAC) {
#pragma unroll
        for (int s = 0; s < KSM; ++s)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const int j = ct * 16 + l15;
            const double v = sh[S::S_NV + min(j, N - 1) * NQ4 + 4 * s + g4];
            Bm[s][ct] = (j < N) ? v : 0.0;
          }
      }
      const double *bB = sh + S::S_G + lane * CT;  // + (b * QG + qg) * 64 * CT
      // tiles are taken as the waves become free: the fetching waves join late
      for (;;) {
        int tile = 0;
        if (lane == 0) tile = atomicAdd(&s_tile, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile * 16 >= NP) break;
        const int pmine = tile * 16 + l15;
        const uint32_t pk = s_pairs[min(pmine, NP - 1)];
        const int o = pk >> 16, t = (pk >> 8) & 0xff, i = s_dofpos[pk & 0xff];
        const double *aD = s_D + t * NSYM * NQ4 + g4;           // + sym(b,a) * NQ4 + 4 qg
        const double *aG = sh + S::S_G + (g4 * 16 + (i & 15)) * CT + (i >> 4);  // + (a * QG + qg) * 64 * CT
        const double *aF = s_F + t * DIM * NQ4 + g4, *aS = s_S + t * NQ4 + g4;
        const double *aM = s_M + t * NQ4 + g4, *aN = sh + S::S_NV + i * NQ4 + g4;
        v4d c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
        double rp = 0.0;
        if (!(dbg & 4)) {
          // operands of point group qg+1 are requested before the products of group qg are issued: two register sets.
          // The fences keep that order (left alone, the scheduler hoists every read of the tile to the top: scratch)
          double G3[2][DIM], D6[2][NSYM], F3[2][DIM], S1[2], NV[2], M1[2], BB[2][DIM][2];
#define MHA_RG_LOAD(qg_, s_)                                                                 \
  {                                                                                          \
    _Pragma("unroll") for (int aa = 0; aa < DIM; ++aa) G3[s_][aa] = aG[(aa * QG + (qg_)) * 64 * CT]; \
    if constexpr (JAC) { _Pragma("unroll") for (int bb = 0; bb < DIM; ++bb) {                           \
      if constexpr (CT == 2) { const v2d bv = *reinterpret_cast<const v2d *>(bB + (bb * QG + (qg_)) * 64 * CT); BB[s_][bb][0] = bv[0]; BB[s_][bb][1] = bv[1]; } \
      else BB[s_][bb][0] = bB[(bb * QG + (qg_)) * 64 * CT]; } }                               \
    NV[s_] = aN[4 * (qg_)];                                                                  \
    S1[s_] = aS[4 * (qg_)];                                                                  \
    _Pragma("unroll") for (int aa = 0; aa < DIM; ++aa) F3[s_][aa] = aF[aa * NQ4 + 4 * (qg_)]; \
    if constexpr (JAC) {                                                                               \
      _Pragma("unroll") for (int c = 0; c < NSYM; ++c) D6[s_][c] = aD[c * NQ4 + 4 * (qg_)];  \
      if constexpr (TR) M1[s_] = aM[4 * (qg_)];                                              \
    }                                                                                        \
  }
          MHA_RG_LOAD(0, 0)
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) {
            const int sb = qg & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (qg + 1 < QG) MHA_RG_LOAD(qg + 1, (qg + 1) & 1)
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 2)) {
              rp += S1[sb] * NV[sb];
#pragma unroll
              for (int aa = 0; aa < DIM; ++aa) rp += F3[sb][aa] * G3[sb][aa];
            }
            if constexpr (JAC) {
#pragma unroll
              for (int bb = 0; bb < DIM; ++bb) {
                double a = 0.0;
#pragma unroll
                for (int aa = 0; aa < DIM; ++aa) a += D6[sb][sym_index(DIM, bb, aa)] * G3[sb][aa];
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, BB[sb][bb][0], c0, 0, 0, 0);
                if constexpr (CT == 2) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, BB[sb][bb][1], c1, 0, 0, 0);
              }
              if constexpr (TR) {
                const double a = M1[sb] * NV[sb];
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bm[qg][0], c0, 0, 0, 0);
                if constexpr (CT == 2) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bm[qg][CT - 1], c1, 0, 0, 0);
              }
            }
          }
#undef MHA_RG_LOAD
        }
        // residual entry of the pair: sum over the four lane groups; the global vector receives -res.val() (scatterRes)
        rp += __shfl_xor(rp, 16);
        rp += __shfl_xor(rp, 32);
        if (g4 == 0 && pmine < NP && !(dbg & 2)) {
          if (!JAC && out.ordered) acc[pmine] = -rp;  // deterministic mode: the row's owner sums its pairs in pair order below
          else atomicAdd(&racc[o], -rp);
        }
        // D register u: row (lane>>4) + 4u, column lane&15 -> entry slot[pair][LID position of column dof]
        if constexpr (JAC) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int pp = tile * 16 + g4 + 4 * u;
            if (pp < NP && !(dbg & 8)) {
              const int po = s_pairoff[pp];
              const uint8_t *sl = s_slot + pp * N;
              if (l15 < N) atomicAdd(&acc[po + sl[s_offs[l15]]], c0[u]);
              if constexpr (CT == 2) {
                if (16 + l15 < N) atomicAdd(&acc[po + sl[s_offs[16 + l15]]], c1[u]);
              }
            }
          }
        }
      }
    };
    if (jac) tiles(std::true_type()); else tiles(std::false_type());
    MHA_RG_STAMP(3)
    lds_barrier();
    MHA_RG_STAMP(4)
    if (prefetch) MHA_RG_SLOT_STORE(hn)  // nobody reads the slot table again before the next block's T

    // ---- S. stream the finished rows in contiguous runs; what has been read is zeroed for the next block ----
    // Only waves NF..7 store to global memory, only waves 0..NF-1 load from it (measured: 2 fetching waves 2.12 ms,
    // 3: 2.13 ms, 4: 2.17 ms on perturbed config 2, profiles/r2_fetch_waves.log): on gfx950 a wave's loads and stores retire
    // through one in-order counter (vmcnt), so a wave that has just streamed out CRS rows would wait for the last of
    // those writes to be acknowledged before it could use the first value it loads for the next block.
    if (jac && !fetcher) {
      for (int sg = wave - NF; sg < ((dbg & 16) ? 0 : NS); sg += NW - NF) {
        int len = s_segs[2 * rb.lds_segs + sg];
        const bool fixed_run = len < 0;  // run of fixed rows: zeros when storing, untouched when accumulating
        if (fixed_run) len = -len;
        if (fixed_run && !out.overwrite) continue;
        double *dst = out.vals + s_segs[rb.lds_segs + sg];
        double *src = acc + s_segs[sg];
        for (int k0 = 0; k0 < len; k0 += 256) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 64 + lane;
            v[u] = (k < len) ? src[k] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 64 + lane;
            if (k < len) {
              if (out.overwrite) dst[k] = v[u]; else dst[k] += v[u];
              src[k] = 0.0;
            }
          }
        }
      }
    }
    for (int i = tid - PT; i >= 0 && i < NR; i += NT - PT) {
      const int rr = s_rows[i];
      double v = racc[i];
      racc[i] = 0.0;
      if (!jac && out.ordered) {  // pair order is fixed by the host partition: the sum is bit-reproducible
        v = 0.0;
        for (int p = 0; p < NP; ++p)
          if ((int)(s_pairs[p] >> 16) == i) v += acc[p];
      }
      if (rr < 0) { if (out.overwrite) out.res[~rr] = 0.0; }  // fixed rows are skipped by the scatter
      else out.res[rr] = out.overwrite ? v : out.res[rr] + v;
    }
    // no barrier here: the next block's field phase touches neither the accumulators nor this block's tables, and the
    // first barrier of the next block orders this store phase before anything that does
    MHA_RG_STAMP(5)
    cur ^= 1;
  }
  if constexpr (TIMING)
    if (lane == 0)
      for (int k = 0; k < 6; ++k) timing[((size_t)blockIdx.x * NW + wave) * 8 + k] = tacc[k];
#undef MHA_RG_STAMP
#undef MHA_RG_STAGE_A_LOAD
#undef MHA_RG_STAGE_A_STORE
#undef MHA_RG_SLOT_LOAD
#undef MHA_RG_SLOT_STORE
#undef MHA_RG_STAGE_B
}

template <int DIM, int P, int NQ1>
void launch_rg(const BlockDev &b, const ThermalDev &ph, RowBlocksDev rb, const uint8_t *slot8, const int32_t *blk_rows,
               const double *gp1d, const int32_t *blk_hdr, long long *timing, const RowOut &out, int num_cus,
               hipStream_t stream) {
  const bool tr = ph.time.transient != 0, expr = has_expression(ph);
  rb.lds_acc = (rb.lds_acc + 1) / 2 * 2;
  constexpr size_t n_dofs = RG<DIM, P, NQ1, true>::N;
  MHA_REQUIRE(rb.lds_pairs * n_dofs <= size_t(kRgSlotRegs) * 256 * 16 && rb.lds_elems <= 32 && rb.lds_pairs <= 256 &&
                  rb.lds_rows <= 256 && rb.lds_segs <= 256, MHA_ERR_INVALID,
              "general row-owner kernel: row block exceeds the kernel's caps");
  auto go = [&](auto kern, size_t lds) {
    MHA_REQUIRE(lds <= 160 * 1024, MHA_ERR_INVALID, "general row-owner kernel needs " << lds << " B of LDS (> 160 KiB)");
    MHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const int dbg = [] { const char *m = std::getenv("MHA_GRO_DBG"); return m ? std::atoi(m) : 0; }();
    hipLaunchKernelGGL(kern, dim3(std::min(rb.num_blocks, num_cus)), dim3(512), lds, stream, b, ph, rb, slot8, blk_rows, gp1d, blk_hdr, out, dbg, timing);
  };
  if (timing) {  // profiling build of the two plain-coefficient variants only
    if (tr) go(thermal_general_row_owner_kernel<DIM, P, NQ1, true, false, true>,
               rg_layout<DIM, P, NQ1, true>(rb.lds_elems, rb.lds_rows, rb.lds_acc, rb.lds_pairs, rb.lds_segs).total);
    else go(thermal_general_row_owner_kernel<DIM, P, NQ1, false, false, true>,
            rg_layout<DIM, P, NQ1, false>(rb.lds_elems, rb.lds_rows, rb.lds_acc, rb.lds_pairs, rb.lds_segs).total);
  } else if (tr) {
    const size_t lds = rg_layout<DIM, P, NQ1, true>(rb.lds_elems, rb.lds_rows, rb.lds_acc, rb.lds_pairs, rb.lds_segs).total;
    if (expr) go(thermal_general_row_owner_kernel<DIM, P, NQ1, true, true, false>, lds);
    else go(thermal_general_row_owner_kernel<DIM, P, NQ1, true, false, false>, lds);
  } else {
    const size_t lds = rg_layout<DIM, P, NQ1, false>(rb.lds_elems, rb.lds_rows, rb.lds_acc, rb.lds_pairs, rb.lds_segs).total;
    if (expr) go(thermal_general_row_owner_kernel<DIM, P, NQ1, false, true, false>, lds);
    else go(thermal_general_row_owner_kernel<DIM, P, NQ1, false, false, false>, lds);
  }
  MHA_HIP(hipGetLastError());
}

}  // namespace

bool thermal_general_row_owner_supported(int dim, int order, int nq1) {
  return (dim == 2 && ((order == 1 && nq1 == 2) || (order == 2 && nq1 == 3) || (order == 3 && nq1 == 4) ||
                       (order == 4 && nq1 == 5))) ||
         (dim == 3 && ((order == 1 && nq1 == 2) || (order == 2 && nq1 == 3)));
}

size_t thermal_general_row_owner_lds(int dim, int order, int nq1, const RowBlocksDev &rb0) {
  RowBlocksDev rb = rb0;
#define MHA_RG_LDS(D_, P_, Q_)                                                                                  \
  if (dim == D_ && order == P_ && nq1 == Q_)                                                                    \
    return rg_layout<D_, P_, Q_, true>(rb.lds_elems, rb.lds_rows, rb.lds_acc, rb.lds_pairs, rb.lds_segs).total;
  MHA_RG_LDS(2, 1, 2) MHA_RG_LDS(2, 2, 3) MHA_RG_LDS(2, 3, 4) MHA_RG_LDS(2, 4, 5) MHA_RG_LDS(3, 1, 2) MHA_RG_LDS(3, 2, 3)
#undef MHA_RG_LDS
  return ~size_t(0);
}

void launch_thermal_general_row_owner(int dim, int order, int nq1, const BlockDev &b, const ThermalDev &ph,
                                      const RowBlocksDev &rb, const uint8_t *slot8, const int32_t *blk_rows,
                                      const double *gp1d, const int32_t *blk_hdr, long long *timing, const RowOut &out, int num_cus,
               hipStream_t stream) {
  if (rb.num_blocks <= 0) return;
  MHA_REQUIRE(rb.lds_acc < 65536, MHA_ERR_INVALID, "general row-owner kernel: accumulator offsets must fit 16 bits");
#define MHA_RG_GO(D_, P_, Q_) \
  if (dim == D_ && order == P_ && nq1 == Q_) return launch_rg<D_, P_, Q_>(b, ph, rb, slot8, blk_rows, gp1d, blk_hdr, timing, out, num_cus, stream);
  MHA_RG_GO(2, 1, 2) MHA_RG_GO(2, 2, 3) MHA_RG_GO(2, 3, 4) MHA_RG_GO(2, 4, 5) MHA_RG_GO(3, 1, 2) MHA_RG_GO(3, 2, 3)
#undef MHA_RG_GO
  MHA_REQUIRE(false, MHA_ERR_INVALID, "general row-owner kernel: unsupported (dim, order, points/dir)");
}

}  // namespace mha
