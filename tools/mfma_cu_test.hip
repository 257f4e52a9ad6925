// FP64 MFMA throughput of ONE CU as a function of the wavefronts that issue it (1024-thread
// workgroup, wavefront w on SIMD w % 4): cycles per v_mfma_f64_16x16x4_f64 of a wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(long long *cyc, double *out, unsigned mask, int slot, int nblk) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (mask >> wave & 1) {
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + lane * 1e-3, b = 0.5;
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < 128; ++it)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    asm volatile("" : "+v"(acc));
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) {
      cyc[slot * 32 + wave] = t1 - t0;
      cyc[slot * 32 + 16 + wave] = w1 - w0;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0];
  }
}
int main() {
  long long *cyc, h[32 * 8];
  double *out;
  hipMalloc(&cyc, sizeof(h));
  hipMalloc(&out, 1024 * 1024 * 8);
  hipMemset(cyc, 0, sizeof(h));
  const unsigned masks[] = {0x0001, 0x1111, 0x000f, 0x00ff, 0xffff, 0xffff, 0xffff};
  const int blocks[] = {1, 1, 1, 1, 1, 256, 1024};
  for (int i = 0; i < 7; ++i) hipLaunchKernelGGL(k, dim3(blocks[i]), dim3(1024), 0, 0, cyc, out, masks[i], i, blocks[i]);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 7; ++i) {
    printf("mask 0x%04x, %4d workgroups: cycles per MFMA of wave", masks[i], blocks[i]);
    int nw = 0;
    double wsum = 0;
    for (int w = 0; w < 16; ++w)
      if (masks[i] >> w & 1) {
        printf(" %.0f", h[i * 32 + w] / 1024.0);
        wsum += h[i * 32 + 16 + w] * 10.0 / 1024.0;  // ns per MFMA
        ++nw;
      }
    printf("   (%.1f ns per MFMA of a wave => CU rate %.1f GFLOP/s)\n", wsum / nw, nw * 2048.0 / (wsum / nw));
  }
  return 0;
}
