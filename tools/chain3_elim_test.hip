// c3_eliminate (pgf_chain3.h) alone on a CU: cycles per 16 x 16 pivot tile.
#include "pgf_chain3.h"
#include <cstdio>
__global__ __launch_bounds__(1024) void k(const double *A, long long *cyc, double *out, int reps, int mode) {
  __shared__ double P[16 * C3_PLD], Dl[64];
  __shared__ double XX[16 * 256];
  __shared__ int flag;
  if (threadIdx.x == 0) flag = 0;
  for (int i = threadIdx.x; i < 4096; i += 1024) XX[i] = 1e-3 * i;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 256) P[(threadIdx.x >> 4) * C3_PLD + (threadIdx.x & 15)] = A[threadIdx.x];
  __syncthreads();
  if (wave == 0) {
    bool bad = false;
    int neg = 0;
    double nl[16], e[16];
    long long t0 = 0;
    for (int rep = 0; rep < reps; ++rep) {
      if (rep == 1) t0 = clock64();
      c3_eliminate(P, nl, e, Dl, Dl + 32, bad, neg, lane & 15, 16, lane == 0);
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(nl[j]), "+v"(e[j]));
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[mode] = (t1 - t0) / (reps - 1);
    __hip_atomic_store(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += nl[j] + e[j];
    out[lane] = s + neg + bad;
  } else {
    // what the other wavefronts do meanwhile
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    double y = lane;
    for (int it = 0; it < (1 << 22); ++it) {
      if (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      if (mode == 1) __builtin_amdgcn_s_sleep(1);                          // polling LDS
      if (mode == 2 && (wave & 3) != 0)                                       // MFMA on SIMDs 1-3, operands from LDS
        for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(XX[s2 * 64 + lane], XX[1024 + s2 * 64 + lane], acc, 0, 0, 0);
      if (mode == 3)                                                          // MFMA everywhere
        for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(XX[s2 * 64 + lane], XX[1024 + s2 * 64 + lane], acc, 0, 0, 0);
      if (mode == 4 && (wave & 3) != 0)                                       // fp64 VALU on SIMDs 1-3
        for (int s2 = 0; s2 < 16; ++s2) y = fma(y, 0.999, 1e-3);
      if (mode == 5) __builtin_amdgcn_s_sleep(8);
    }
    out[64 + threadIdx.x] = acc[0] + y;
  }
}
int main() {
  double hA[256], *dA, *out;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      hA[i * 16 + j] = (i == j) ? 4.0 + 0.01 * i : 0.3 / (1.0 + ((hi * 7 + lo * 13) % 11));
    }
  long long *cyc, h;
  hipMalloc(&dA, sizeof(hA));
  hipMalloc(&out, 2048 * 8);
  hipMalloc(&cyc, 64);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  const char *names[] = {"others busy-poll LDS", "others poll LDS with s_sleep 1", "MFMA (LDS operands) on SIMDs 1-3", "MFMA on all SIMDs", "fp64 FMAs on SIMDs 1-3", "others poll with s_sleep 8"};
  long long hh[8];
  for (int m = 0; m < 6; ++m) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, dA, cyc, out, 65, m);
  hipDeviceSynchronize();
  hipMemcpy(hh, cyc, 48, hipMemcpyDeviceToHost);
  for (int m = 0; m < 6; ++m) printf("c3_eliminate, %-36s %lld shader cycles per tile\n", names[m], hh[m]);
  (void)h;
  return 0;
}
