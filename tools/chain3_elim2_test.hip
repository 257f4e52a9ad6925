// c3_eliminate + the LDS stores that follow it in chain3_body, with the LDS tiles at a LOW or a
// HIGH offset of a large LDS allocation, other wavefronts waiting at a barrier.
#include "pgf_chain3.h"
#include <cstdio>
template <int BIG>
__global__ __launch_bounds__(1024) void k(const double *A, long long *cyc, double *out, int reps, int off, int slot) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BIG];
  double *P = reinterpret_cast<double *>(smem + off);
  double *EI = P + 16 * C3_PLD, *LK = EI + 16 * C3_PLD, *Dl = LK + 16 * C3_PLD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15;
  if (threadIdx.x < 256) P[(threadIdx.x >> 4) * C3_PLD + (threadIdx.x & 15)] = A[threadIdx.x];
  __syncthreads();
  if (wave == 0) {
    bool bad = false;
    int neg = 0;
    double nl[16], e[16];
    long long t0 = 0, ta = 0, tb = 0;
    for (int rep = 0; rep < reps; ++rep) {
      if (rep == 1) t0 = clock64();
      const long long a0 = clock64();
      c3_eliminate(P, nl, e, Dl, Dl + 32, bad, neg, l15, 16, lane == 0);
      const long long a1 = clock64();
      if (lane < 16) {
        const double dmine = Dl[l15];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          double2_t v, w;
          v.x = e[j];
          v.y = e[j + 1];
          *reinterpret_cast<double2_t *>(EI + l15 * C3_PLD + j) = v;
          w.x = (j == l15) ? dmine : -nl[j];
          w.y = (j + 1 == l15) ? dmine : -nl[j + 1];
          *reinterpret_cast<double2_t *>(LK + l15 * C3_PLD + j) = w;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long a2 = clock64();
      ta += a1 - a0;
      tb += a2 - a1;
    }
    const long long t1 = clock64();
    if (lane == 0) {
      cyc[3 * slot] = (t1 - t0) / (reps - 1);
      cyc[3 * slot + 1] = ta / reps;
      cyc[3 * slot + 2] = tb / reps;
    }
    out[lane] = neg + bad;
  }
  __syncthreads();
}
int main() {
  double hA[256], *dA, *out;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      hA[i * 16 + j] = (i == j) ? 4.0 + 0.01 * i : 0.3 / (1.0 + ((hi * 7 + lo * 13) % 11));
    }
  long long *cyc, h[12];
  hipMalloc(&dA, sizeof(hA));
  hipMalloc(&out, 2048 * 8);
  hipMalloc(&cyc, sizeof(h));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<16384>, dim3(1), dim3(1024), 0, 0, dA, cyc, out, 33, 0, 0);
  hipLaunchKernelGGL(k<147456>, dim3(1), dim3(1024), 0, 0, dA, cyc, out, 33, 0, 1);
  hipLaunchKernelGGL(k<147456>, dim3(1), dim3(1024), 0, 0, dA, cyc, out, 33, 131072, 2);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[] = {"16 KB of LDS", "144 KB, tiles at offset 0", "144 KB, tiles at offset 128 KB"};
  for (int i = 0; i < 3; ++i) printf("%-32s per tile %lld cycles: elimination %lld, stores %lld\n", nm[i], h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  return 0;
}
