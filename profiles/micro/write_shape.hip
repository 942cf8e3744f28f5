// write_shape.hip -- store shapes of the block-pattern kernel in a plain persistent loop (256 workgroups x 768 threads,
// 64 KB "blocks", barrier per block, block t of all workgroups adjacent): what does each shape cost?
//   shape 0: one instruction = 1 KB contiguous, 16 B per lane (the reference: write_bw.hip)
//   shape 1: one instruction = 4 pieces of 256 B, 1 KB apart (rows), 16 B per lane; 4 instructions fill 4 rows
//   shape 2: shape 1 with every row starting on an odd multiple of 8 B (misaligned 16-byte lanes)
//   shape 3: 8 B per lane: one instruction = 4 pieces of 128 B, 1 KB apart; 8 instructions fill 4 rows
//   shape 4: shape 2 with rows of 1000 B (125 entries) packed back to back: pieces of neighbouring rows share lines
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768) void k(char *base, int nb, int shape) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  for (int t = 0; t < nb; ++t) {
    char *blk = base + ((size_t)t * gridDim.x + blockIdx.x) * 65536;
    if (shape == 0) {
      for (int j = wave; j < 64; j += 12) *reinterpret_cast<double2 *>(blk + j * 1024 + lane * 16) = make_double2(1.0, t);
    } else if (shape == 1 || shape == 2) {
      // 64 rows of 1 KB; group of 4 rows r0..r0+3; instruction p writes piece p (256 B) of each of the 4 rows
      const int mis = shape == 2 ? 8 : 0;
      for (int j = wave; j < 64; j += 12) {
        const int rg = j >> 2, p = j & 3;
        char *a = blk + (rg * 4 + g) * 1024 + p * 256 + c * 16 + mis;
        if (!(mis && rg == 15 && g == 3 && p == 3 && c == 15)) *reinterpret_cast<double2 *>(a) = make_double2(1.0, t);
      }
    } else if (shape == 3) {
      for (int j = wave; j < 128; j += 12) {
        const int rg = j >> 3, p = j & 7;
        *reinterpret_cast<double *>(blk + (rg * 4 + g) * 1024 + p * 128 + c * 8) = 1.0;
      }
    } else {
      // rows of 1000 B back to back (65 rows = 65000 B): 4 pieces of 256 B per row, the last one 232 B (lanes masked)
      for (int j = wave; j < 68; j += 12) {
        const int rg = j >> 2, p = j & 3;
        const int row = rg * 4 + g;
        const int off = p * 256 + c * 16;
        if (row < 65 && off + 16 <= 1000) *reinterpret_cast<double2 *>(blk + row * 1000 + off) = make_double2(1.0, t);
      }
    }
    __syncthreads();
  }
}
int main() {
  const size_t bytes = (size_t)16384 * 65536;
  char *buf;
  if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int shape = 0; shape < 5; ++shape) {
    for (int i = 0; i < 3; ++i) k<<<256, 768>>>(buf, 64, shape);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<<<256, 768>>>(buf, 64, shape);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("shape %d: %8.1f us per 1.07 GB  %6.2f TB/s\n", shape, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
