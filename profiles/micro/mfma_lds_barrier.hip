// mfma_lds_barrier.hip -- the point loop of the q-streamed point engine in isolation: 8 waves of one workgroup per CU,
// per iteration every wave reads NREAD doubles per lane from LDS, issues NM independent v_mfma_f64_16x16x4_f64 and
// (optionally) writes 4 doubles per lane back and meets the others at an LDS-only barrier.  How many cycles does an
// iteration take against NM x 64 x (waves per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NM, bool LDS, bool WRITE, bool BARRIER>
__global__ __launch_bounds__(512) void loop(double *out, int iters) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int k = tid; k < 2 * 8 * 256; k += 512) sm[k] = 1e-3 * k;
  __syncthreads();
  v4d acc[NM];
#pragma unroll
  for (int p = 0; p < NM; ++p) acc[p] = {0.0, 0.0, 0.0, 0.0};
  double b = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
    const double *buf = sm + (it & 1) * 2048 + lane;
    double av[NM];
#pragma unroll
    for (int p = 0; p < NM; ++p) av[p] = LDS ? buf[(p % 8) * 256] : b + p;
#pragma unroll
    for (int p = 0; p < NM; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p], b, acc[p], 0, 0, 0);
    if (WRITE) {
      double *wb = sm + ((it + 1) & 1) * 2048 + wv * 256 + lane;
      wb[0] = acc[0][0]; wb[64] = acc[0][1]; wb[128] = acc[0][2]; wb[192] = acc[0][3];
    }
    if (BARRIER) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
  }
  double s = 0.0;
#pragma unroll
  for (int p = 0; p < NM; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  double *out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  auto run = [&](const char *name, auto kern, int nm, int threads) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 40960, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 40960, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms / 3 * 1e6 / iters;
    printf("%-58s %d threads: %7.1f ns = %7.0f cycles per iteration (MFMA floor %d)\n", name, threads, ns, ns * ghz, nm * 64 * (threads / 256));
  };
  for (int threads : {256, 512}) {
    run("8 MFMAs, operands in registers, no barrier", loop<8, false, false, false>, 8, threads);
    run("8 MFMAs, operands in registers, barrier", loop<8, false, false, true>, 8, threads);
    run("8 MFMAs, operands from LDS, no barrier", loop<8, true, false, false>, 8, threads);
    run("8 MFMAs, operands from LDS, barrier", loop<8, true, false, true>, 8, threads);
    run("8 MFMAs, operands from LDS, LDS writes, barrier", loop<8, true, true, true>, 8, threads);
    run("16 MFMAs, operands from LDS, LDS writes, barrier", loop<16, true, true, true>, 16, threads);
    run("4 MFMAs, operands from LDS, LDS writes, barrier", loop<4, true, true, true>, 4, threads);
  }
  return 0;
}
