// mfma_f64_rate.hip -- what does one v_mfma_f64_16x16x4_f64 cost on an MI355X box, and does it share a pipe with the
// fp64 vector FMAs?  The floors quoted for the matrix phases of the block-pattern kernel, the general row-owner kernel
// and the point engine assume 64 cycles per instruction per SIMD (78.6 TFLOP/s on 256 CUs x 4 SIMDs at 2.4 GHz).
//   * mfma: W waves per SIMD, each a loop of MFMAs over C independent accumulators (C = 1: a dependent chain)
//   * fma:  the same with v_fma_f64 (64 lanes, 128 flops per instruction)
//   * mixed: W waves per SIMD, even waves MFMA, odd waves FMA -- do the two overlap?
// Cycles come from s_memtime (100 MHz constant clock on gfx9: use wall time) -> reported as ns and as cycles at the
// clock rocm-smi / hipDeviceProp report.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(1024) void mfma_loop(double *out, int iters, double a0, double b0) {
  v4d acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 1.2345e300) out[0] = s;
}

template <int C>
__global__ __launch_bounds__(1024) void fma_loop(double *out, int iters, double a0, double b0) {
  double acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = threadIdx.x * 1e-9 + c;
  double a = a0, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = __builtin_fma(acc[c], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += acc[c];
  if (s == 1.2345e300) out[0] = s;
}

// waves 0-3 of the workgroup (one per SIMD): MFMA chains (4 accumulators) when mode & 1; waves 4-7 (the second wave of each
// SIMD): FMA chains (8 accumulators) when mode & 2; other waves leave at once.  mode 3 against modes 1 and 2: the time
// of the longer one = the two kinds overlap, their sum = they share the pipe.
__global__ __launch_bounds__(512) void mixed_loop(double *out, int n_mfma, int n_fma, double a0, double b0, int mode) {
  const int wave = threadIdx.x >> 6;
  double s = 0.0;
  if (wave < 4) {
    if (!(mode & 1)) return;
    v4d acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = {0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < n_mfma / 4; ++it) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  } else {
    if (!(mode & 2)) return;
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = threadIdx.x * 1e-9 + c;
    for (int it = 0; it < n_fma / 8; ++it) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(acc[c], a0, b0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) s += acc[c];
  }
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("%s: %d CUs, clockRate %.3f GHz\n", prop.name, cus, ghz);
  double *out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  auto time = [&](auto fn) {
    fn(); fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)ms / 5 * 1e6;  // ns per launch
  };
  // one workgroup per CU, W waves per SIMD = 4 W waves per workgroup
  for (int W : {1, 2, 4}) {
    auto report = [&](const char *name, int C, double ns, double flops_per_instr) {
      const double instr_per_simd = (double)iters * C * W;
      const double ns_per = ns / instr_per_simd;
      printf("%-6s W=%d waves/SIMD  C=%d chains: %7.2f ns per instruction per SIMD = %6.1f cycles at %.2f GHz  -> %6.1f TFLOP/s on %d CUs\n",
             name, W, C, ns_per, ns_per * ghz, ghz, flops_per_instr / ns_per * 4 * cus * 1e-3, cus);
    };
    report("mfma", 1, time([&] { mfma_loop<1><<<cus, 256 * W>>>(out, iters, 1.0, 1e-3); }), 2048);
    report("mfma", 2, time([&] { mfma_loop<2><<<cus, 256 * W>>>(out, iters, 1.0, 1e-3); }), 2048);
    report("mfma", 4, time([&] { mfma_loop<4><<<cus, 256 * W>>>(out, iters, 1.0, 1e-3); }), 2048);
    report("fma", 1, time([&] { fma_loop<1><<<cus, 256 * W>>>(out, iters, 1.0, 1e-3); }), 128);
    report("fma", 4, time([&] { fma_loop<4><<<cus, 256 * W>>>(out, iters, 1.0, 1e-3); }), 128);
    report("fma", 8, time([&] { fma_loop<8><<<cus, 256 * W>>>(out, iters, 1.0, 1e-3); }), 128);
  }
  // co-issue of the matrix and the vector fp64 instructions from two waves of one SIMD
  for (int fpm : {8, 16, 32}) {
    const int nm = 40000, nf = nm * fpm;
    const double t_m = time([&] { mixed_loop<<<cus, 512>>>(out, nm, nf, 1.0, 1e-3, 1); });
    const double t_f = time([&] { mixed_loop<<<cus, 512>>>(out, nm, nf, 1.0, 1e-3, 2); });
    const double t_b = time([&] { mixed_loop<<<cus, 512>>>(out, nm, nf, 1.0, 1e-3, 3); });
    printf("per SIMD: %d MFMAs alone %.1f us; %d FMAs alone %.1f us; both (two waves) %.1f us  (sum %.1f, max %.1f)\n", nm, t_m * 1e-3,
           nf, t_f * 1e-3, t_b * 1e-3, (t_m + t_f) * 1e-3, (t_m > t_f ? t_m : t_f) * 1e-3);
  }
  return 0;
}
