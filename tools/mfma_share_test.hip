// Which wavefronts' FP64 MFMAs slow down wavefront 0's FP64 vector FMAs?  (SIMD mapping of the
// wavefronts of a 1024-thread workgroup, and whether the FP64 pipe is shared across SIMDs.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(long long *cyc, double *out, unsigned mask, int slot, int kind) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ volatile int stop;
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  if (wave == 0) {
    double x[8];
    float y[8];
    for (int j = 0; j < 8; ++j) { x[j] = 1.0 + lane + j; y[j] = 1.0f + lane + j; }
    __builtin_amdgcn_s_sleep(20);
    const long long t0 = clock64();
    if (kind == 0) {
      for (int i = 0; i < 256; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = fma(x[j], 0.999, 1e-3);
    } else {
      for (int i = 0; i < 256; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fmaf(y[j], 0.999f, 1e-3f);
    }
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]), "+v"(y[j]));
    const long long t1 = clock64();
    if (lane == 0) cyc[slot] = t1 - t0;
    out[lane] = x[0] + x[7] + y[0] + y[7];
    stop = 1;
  } else if (mask >> wave & 1) {
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + lane * 1e-3, b = 0.5;
    for (int it = 0; it < 4096 && !stop; ++it)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    out[64 + threadIdx.x] = acc[0];
  }
}
int main() {
  long long *cyc, h[64];
  double *out;
  hipMalloc(&cyc, sizeof(h));
  hipMalloc(&out, 2048 * 8);
  int slot = 0;
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, cyc, out, 0u, slot++, 0);
  for (int w = 1; w < 16; ++w) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, cyc, out, 1u << w, slot++, 0);
  const unsigned masks[] = {0x0110, 0x1110, 0x000e, 0x00fe, 0xfffe};
  for (unsigned m : masks) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, cyc, out, m, slot++, 0);
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, cyc, out, 0u, slot++, 1);
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, cyc, out, 0xfffeu, slot++, 1);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("wave 0, 2048 independent fp64 FMAs: cycles per FMA\n  alone: %.1f\n", h[0] / 2048.0);
  for (int w = 1; w < 16; ++w) printf("  MFMA on wave %2d: %.1f\n", w, h[w] / 2048.0);
  for (int i = 0; i < 5; ++i) printf("  MFMA on waves mask 0x%04x: %.1f\n", masks[i], h[16 + i] / 2048.0);
  printf("fp32 FMAs alone: %.1f   beside MFMA f64 on waves 1..15: %.1f\n", h[21] / 2048.0, h[22] / 2048.0);
  return 0;
}
