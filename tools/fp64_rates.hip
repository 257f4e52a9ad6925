// Microbenchmark: FP64 MFMA vs FP64 VALU FMA rates and shader clock on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

// mode 0: MFMA only; 1: VALU fma only; 2: waves 0-3 MFMA, waves 4-7 VALU (8 waves / WG)
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *stamps, int iters,
                                         double a0, double b0) {
  const int wave = threadIdx.x >> 6;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  const bool do_mfma = (MODE == 0) || (MODE == 2 && wave < 4);
  if (do_mfma) {
    double4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = a0 + i;
    double a = 1.0 + threadIdx.x * 1e-12, b = b0 * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
    }
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
  }
}

template <int MODE>
void run(int threads, int iters, const char *tag) {
  const int blocks = 256;
  double *out;
  unsigned long long *st;
  hipMalloc(&out, sizeof(double) * blocks * 512);
  hipMalloc(&st, sizeof(unsigned long long) * blocks * 16);
  hipMemset(st, 0, sizeof(unsigned long long) * blocks * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) k<MODE><<<blocks, threads>>>(out, st, iters, 1.0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, st, iters, 1.0, 0.5);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), st, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost);
  const int waves = threads / 64;
  double cyc = 0, rt = 0;
  for (int w = 0; w < waves; ++w) { cyc += h[w * 2]; rt += h[w * 2 + 1]; }
  cyc /= waves; rt /= waves;
  const double clk_ghz = cyc / (rt * 10.0);  // realtime ticks are 100 MHz -> 10 ns
  double mfma_waves = (MODE == 0) ? waves : (MODE == 2 ? 4 : 0);
  double valu_waves = (MODE == 1) ? waves : (MODE == 2 ? waves - 4 : 0);
  double fl_mfma = blocks * mfma_waves * (double)iters * 8 * 2048.0;
  double fl_valu = blocks * valu_waves * (double)iters * 64 * 64 * 2.0;
  printf("%s threads=%d: %.3f ms  clock %.2f GHz  mfma %.1f TF  valu %.1f TF  total %.1f TF", tag, threads, ms,
         clk_ghz, fl_mfma / ms / 1e9, fl_valu / ms / 1e9, (fl_mfma + fl_valu) / ms / 1e9);
  if (MODE == 0) printf("  cycles/MFMA/SIMD %.1f", cyc / ((double)iters * 8 * (waves / 4.0)));
  if (MODE == 1) printf("  cycles/DFMA(wave64)/SIMD %.2f", cyc / ((double)iters * 64 * (waves / 4.0)));
  printf("\n");
  hipFree(out);
  hipFree(st);
}

int main() {
  run<0>(256, 4000, "MFMA 1 wave/SIMD ");
  run<0>(256, 20000, "MFMA long 4.5ms  ");
  run<0>(256, 80000, "MFMA long 18ms   ");
  run<0>(512, 4000, "MFMA 2 wave/SIMD ");
  run<1>(256, 4000, "VALU 1 wave/SIMD ");
  run<1>(512, 4000, "VALU 2 wave/SIMD ");
  run<2>(512, 4000, "MFMA+VALU        ");
  return 0;
}
