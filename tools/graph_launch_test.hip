// Does a captured hipGraph retire a chain of small dependent kernels faster than stream launches?
// 30 kernels of ~3 us each (dependent through one buffer), 200 repetitions; wall time per chain.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(double *p, int n, double a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0;
}
int main() {
  const int n = 150000, NK = 30, REP = 200;
  double *d;
  hipMalloc(&d, n * sizeof(double));
  hipMemset(d, 0, n * sizeof(double));
  double *hpin;
  hipHostMalloc(&hpin, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  auto chain = [&]() {
    for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, s, d, n, 0.5);
    hipMemcpyAsync(hpin, d, 8, hipMemcpyDeviceToHost, s);
  };
  for (int w = 0; w < 20; ++w) chain();
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < REP; ++r) {
    chain();
    hipStreamSynchronize(s);
  }
  auto t1 = std::chrono::steady_clock::now();
  printf("stream launches: %.1f us per chain of %d kernels + copy\n",
         std::chrono::duration<double, std::micro>(t1 - t0).count() / REP, NK);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  chain();
  hipStreamEndCapture(s, &g);
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  printf("instantiate: %s\n", hipGetErrorString(e));
  for (int w = 0; w < 20; ++w) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < REP; ++r) {
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
  }
  t1 = std::chrono::steady_clock::now();
  printf("graph launch:    %.1f us per chain\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / REP);
  return 0;
}
