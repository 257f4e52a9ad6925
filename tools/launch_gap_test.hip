// Back-to-back latency of dependent tiny kernels in one stream: normal launches vs
// hipExtAnyOrderLaunch.  Also checks whether any-order launches still see the previous
// kernel's writes (each kernel increments a counter in memory through a different XCD).
// hipcc --offload-arch=gfx950 -O2 tools/launch_gap_test.hip -o /tmp/lg && /tmp/lg
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_inc(long long *p, int which) {
  // the workgroup with blockIdx.x == which does the increment: consecutive launches use
  // different XCDs (workgroup id % 8), so a stale L2 would lose increments
  if ((int)blockIdx.x == which && threadIdx.x == 0) p[0] = p[0] + 1;
}
int main() {
  long long *d, h = 0;
  hipMalloc(&d, sizeof(long long));
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int n = 2000;
  for (int flag = 0; flag < 2; ++flag) {
    hipMemset(d, 0, sizeof(long long));
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i)
      hipExtLaunchKernelGGL(k_inc, dim3(8), dim3(64), 0, s, nullptr, nullptr,
                            flag ? hipExtAnyOrderLaunch : 0, d, i % 8);
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("flag=%d: %.2f us per launch, counter %lld of %d\n", flag,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n, h, n);
  }
  return 0;
}
