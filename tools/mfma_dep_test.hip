// Latency of a DEPENDENT chain of v_mfma_f64_16x16x4_f64 (same accumulator) against independent
// chains, one wavefront on a CU; and the same with a second wavefront running DPP FMACs on the
// same SIMD (does FP64 MFMA share the vector FP64 pipe?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NCH>
__device__ __forceinline__ long long run_mfma(int n, double a, double b, double4_t *out) {
  double4_t acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(acc[c]));
  const long long t1 = clock64();
#pragma unroll
  for (int c = 0; c < NCH; ++c) out[c] = acc[c];
  return t1 - t0;
}
__global__ __launch_bounds__(1024) void k(long long *cyc, double4_t *out, int mode) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double a = 1.0 + lane * 1e-3, b = 0.5;
  long long c = 0;
  const int n = 256;
  if (mode == 0 && wave == 0) c = run_mfma<1>(n, a, b, out + lane * 8);
  if (mode == 1 && wave == 0) c = run_mfma<2>(n, a, b, out + lane * 8);
  if (mode == 2 && wave == 0) c = run_mfma<4>(n, a, b, out + lane * 8);
  if (mode == 3 && (wave & 3) == 0) c = run_mfma<1>(n, a, b, out + threadIdx.x * 8);  // 4 waves, SIMD 0
  if (mode == 4) {  // wave 0: dependent fp64 FMA chain; waves 4, 8, 12 (same SIMD): MFMA chains
    if (wave == 0) {
      double x = a;
      const long long t0 = clock64();
      for (int i = 0; i < 1024; ++i) x = fma(x, 0.999, 1e-3);
      asm volatile("" : "+v"(x));
      c = clock64() - t0;
      out[lane][0] = x;
    } else if ((wave & 3) == 0) {
      run_mfma<1>(4 * n, a, b, out + threadIdx.x * 8);
    }
  }
  if (mode == 5 && wave == 0) {  // the FMA chain alone
    double x = a;
    const long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) x = fma(x, 0.999, 1e-3);
    asm volatile("" : "+v"(x));
    c = clock64() - t0;
    out[lane][0] = x;
  }
  if (mode == 6) {  // wave 0: independent fp64 FMAs (8 chains); waves 4, 8, 12: MFMA
    if (wave == 0) {
      double x[8];
      for (int j = 0; j < 8; ++j) x[j] = a + j;
      const long long t0 = clock64();
      for (int i = 0; i < 256; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = fma(x[j], 0.999, 1e-3);
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
      c = clock64() - t0;
      out[lane][0] = x[0] + x[7];
    } else if ((wave & 3) == 0) {
      run_mfma<1>(4 * n, a, b, out + threadIdx.x * 8);
    }
  }
  if (mode == 7 && wave == 0) {
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = a + j;
    const long long t0 = clock64();
    for (int i = 0; i < 256; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = fma(x[j], 0.999, 1e-3);
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
    c = clock64() - t0;
    out[lane][0] = x[0] + x[7];
  }
  if (lane == 0 && wave == 0) cyc[mode] = c;
}
int main() {
  long long *cyc, h[8];
  double4_t *out;
  hipMalloc(&cyc, 64);
  hipMalloc(&out, 1024 * 8 * sizeof(double4_t));
  for (int m = 0; m < 8; ++m) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, cyc, out, m);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("dependent MFMA chain, 1 wave:          %.1f cycles per MFMA\n", h[0] / 256.0);
  printf("2 interleaved chains, 1 wave:          %.1f cycles per MFMA\n", h[1] / 512.0);
  printf("4 interleaved chains, 1 wave:          %.1f cycles per MFMA\n", h[2] / 1024.0);
  printf("4 waves on one SIMD, 1 chain each:     %.1f cycles per MFMA of a wave\n", h[3] / 256.0);
  printf("dependent FMA chain alone:             %.1f cycles per FMA\n", h[5] / 1024.0);
  printf("dependent FMA chain beside 3 MFMA waves on its SIMD: %.1f cycles per FMA\n", h[4] / 1024.0);
  printf("8 independent FMA chains alone:        %.1f cycles per FMA\n", h[7] / 2048.0);
  printf("8 independent FMA chains beside 3 MFMA waves:        %.1f cycles per FMA\n", h[6] / 2048.0);
  return 0;
}
