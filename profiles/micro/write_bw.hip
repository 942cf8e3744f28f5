// write_bw.hip -- what store rates does an MI355X box sustain for the block-pattern kernel's volume (1.08 GB)?
// Patterns: memset; a free grid of 16-byte stores; 256 persistent workgroups of 768 threads writing 64 KB "blocks"
// (a) each in a region of its own (b) block t of all workgroups adjacent (c) as (b) but every block cut into 16 runs
// of 4 KB that lie `stride` apart (the row runs of a 4x2x2-element chunk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void grid_store(double2 *p, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}

// mode 0: workgroup w writes blocks [w*nb, (w+1)*nb); mode 1: block t of workgroup w is block t*gridDim+w
// runs: the block's 64 KB are `runs` pieces `run_stride` bytes apart (runs = 1: contiguous)
__global__ __launch_bounds__(768) void persistent_store(char *base, int nb, int mode, int runs, size_t run_stride, size_t block_stride, int barrier) {
  const int tid = threadIdx.x;
  const size_t run_bytes = 65536 / runs;
  for (int t = 0; t < nb; ++t) {
    const size_t b = mode == 0 ? (size_t)blockIdx.x * nb + t : (size_t)t * gridDim.x + blockIdx.x;
    // block_stride 0: chunk-like layout -- 16 blocks share a 1 MB window, block s of the window owns the 4 KB slots
    // s, s + 16, ... (runs 64 KB apart)
    char *blk = block_stride ? base + b * block_stride : base + (b / 16) * (size_t)(1 << 20) + (b % 16) * 4096;
    for (size_t off = (size_t)tid * 16; off < 65536; off += 768 * 16) {
      const size_t r = off / run_bytes, o = off % run_bytes;
      *reinterpret_cast<double2 *>(blk + r * run_stride + o) = make_double2(1.0, (double)t);
    }
    if (barrier) __syncthreads();
  }
}

int main() {
  const size_t bytes = (size_t)16384 * 65536;  // 1.07 GB
  char *buf;
  CK(hipMalloc(&buf, bytes + (64 << 20)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char *name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s %8.1f us  %6.2f TB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
  };
  time("hipMemsetAsync", [&] { hipMemsetAsync(buf, 0, bytes, 0); });
  time("grid of 16-byte stores (4096 x 256)", [&] { grid_store<<<4096, 256>>>((double2 *)buf, bytes / 16); });
  time("grid of 16-byte stores (65536 x 256)", [&] { grid_store<<<65536, 256>>>((double2 *)buf, bytes / 16); });
  for (int barrier = 0; barrier < 2; ++barrier) {
    char nm[128];
    snprintf(nm, 128, "persistent 256x768, own region, contiguous blocks, barrier %d", barrier);
    time(nm, [&] { persistent_store<<<256, 768>>>(buf, 64, 0, 1, 0, 65536, barrier); });
    snprintf(nm, 128, "persistent 256x768, interleaved, contiguous blocks, barrier %d", barrier);
    time(nm, [&] { persistent_store<<<256, 768>>>(buf, 64, 1, 1, 0, 65536, barrier); });
  }
  // chunk-like: a block = 16 runs of 4 KB; runs 4 per z plane 64 KB apart, planes 8 MB apart is approximated by a run
  // stride of 64 KB with blocks 4 KB apart inside a 1 MB window  (16 blocks x 16 runs x 4 KB = 1 MB)
  for (int mode = 0; mode < 2; ++mode) {
    char nm[128];
    snprintf(nm, 128, "persistent, %s, 16 runs of 4 KB 64 KB apart, barrier 1", mode ? "interleaved" : "own region");
    // block b -> window b / 16 (1 MB), slot b % 16 (4 KB): emulate with block_stride 4 KB inside windows: needs b-dependent base; approximate
    // with block_stride = 65536 + 0 and run_stride 65536*16: blocks of a 16-group interleave their runs
    time(nm, [&] { persistent_store<<<256, 768>>>(buf, 64, mode, 16, 65536, 0, 1); });
  }
  return 0;
}
