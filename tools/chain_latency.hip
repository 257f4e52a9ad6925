// Latency of the dependent operations on the factorisation's pivot chain, ONE wavefront:
// cycles (s_memtime, shader clock) per link of a chain of N dependent operations.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b) {
  __shared__ double sh[64];
  double x = a + threadIdx.x * 1e-9;
  const long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) x = fma(x, b, a);                                   // dependent fp64 FMA
    if (MODE == 1) x = __builtin_amdgcn_rcp(x) + 1.0;                  // rcp + add
    if (MODE == 2) x = fma(bcast(x, i & 15), b, a);                    // lane -> sgpr -> all lanes + FMA
    if (MODE == 3) {                                                   // through LDS: write, broadcast read, FMA
      sh[threadIdx.x] = x;
      x = fma(sh[i & 15], b, a);
    }
    if (MODE == 4) x = fma(__shfl(x, i & 15), b, a);                   // ds_bpermute + FMA
    if (MODE == 5) {                                                   // the chain's column: bcast, rcp, 2 Newton FMAs, FMA
      const double d = bcast(x, i & 15);
      double r = __builtin_amdgcn_rcp(d);
      r = fma(r, fma(-d, r, 1.0), r);
      x = fma(-b, r, x);
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  double *out;
  long long *cyc, h[8];
  hipMalloc(&out, 64 * 8);
  hipMalloc(&cyc, 8 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 0.999);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 0.999);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 0.999);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 0.999);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 0.999);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 0.999);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char *name[] = {"fp64 FMA", "v_rcp_f64 + add", "v_readlane x2 + FMA", "LDS write + broadcast read + FMA",
                        "ds_bpermute x2 + FMA", "column: readlane, rcp, 2 Newton FMAs, FMA"};
  for (int m = 0; m < 6; ++m) printf("%-44s %6.1f cycles per link\n", name[m], (double)h[m] / N);
  return 0;
}
