// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (FP64 matrix peak).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters, const char *tag) {
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * NACC * 2048.0;
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * NACC * (blocks / 256.0 > 1 ? blocks / 256.0 : 1));
  printf("%s blocks=%d nacc=%d: %.3f ms  %.2f TFLOP/s  (~%.1f cyc/MFMA/SIMD at 2.4GHz)\n", tag, blocks,
         NACC, ms, flops / ms / 1e9, cyc);
  hipFree(out);
}

int main() {
  run<1>(256, 20000, "dep-chain  ");
  run<4>(256, 20000, "4 acc      ");
  run<16>(256, 5000, "16 acc     ");
  run<16>(512, 5000, "16 acc 2wg ");
  run<16>(1024, 5000, "16 acc 4wg ");
  return 0;
}
