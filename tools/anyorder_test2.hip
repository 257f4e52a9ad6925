// Any-order launch next to a ONE-workgroup kernel with a large LDS footprint (the shape of
// k_diag_chain beside k_ldlt_update).  Prints wall time of [K1; K2] with and without the flag.
// hipcc --offload-arch=gfx950 -O2 tools/anyorder_test2.hip -o /tmp/ao2 && /tmp/ao2
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
template <int LDSB>
__global__ void k_one(double *out, int spin) {
  __shared__ double buf[LDSB / 8];
  double acc = threadIdx.x;
  buf[threadIdx.x] = acc;
  __syncthreads();
  for (int i = 0; i < spin; ++i) acc = fma(acc, 1.0000001, buf[(threadIdx.x + i) & 1023]);
  if (acc == 123.456) out[0] = acc;
}
__global__ void k_many(double *out, int spin) {
  double acc = threadIdx.x + blockIdx.x;
  for (int i = 0; i < spin; ++i) acc = fma(acc, 1.0000001, 1e-9);
  if (acc == 123.456) out[1] = acc;
}
template <int LDSB>
void run(hipStream_t s, double *d, int threads) {
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int it = 0; it < 20; ++it) {
        hipLaunchKernelGGL(k_one<LDSB>, dim3(1), dim3(threads), 0, s, d, 20000);
        hipExtLaunchKernelGGL(k_many, dim3(4096), dim3(256), 0, s, nullptr, nullptr,
                              flag ? hipExtAnyOrderLaunch : 0, d, 4000);
        hipLaunchKernelGGL(k_many, dim3(64), dim3(256), 0, s, d, 100);
      }
      hipStreamSynchronize(s);
      auto t1 = std::chrono::steady_clock::now();
      printf("LDS %d threads %d flag=%d: %.1f us per (one; many; small) triple\n", LDSB, threads, flag,
             std::chrono::duration<double, std::micro>(t1 - t0).count() / 20);
    }
  }
}
int main() {
  double *d;
  hipMalloc(&d, 16);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  run<8192>(s, d, 1024);
  run<145424>(s, d, 1024);
  run<145424>(s, d, 512);
  return 0;
}
