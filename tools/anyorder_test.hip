// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this GPU?
// K1 spins ~1 ms and records when it ends, K2 (launched right after, any-order flag) records
// when it starts.  hipcc --offload-arch=gfx950 -O2 tools/anyorder_test.hip -o /tmp/ao && /tmp/ao
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_long(long long *t, int spin) {
  double acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = fma(acc, 1.0000001, 1e-9);
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = wall_clock64() + (long long)(acc * 0.0);
}
__global__ void k_short(long long *t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
int main() {
  long long *d, h[2];
  hipMalloc(&d, 2 * sizeof(long long));
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_long, dim3(64), dim3(256), 0, s, d, 400000);
      hipExtLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, s, nullptr, nullptr,
                            flag ? hipExtAnyOrderLaunch : 0, d);
      hipStreamSynchronize(s);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("flag=%d: short kernel started %lld ticks %s the long kernel ended\n", flag,
             h[1] > h[0] ? h[1] - h[0] : h[0] - h[1], h[1] > h[0] ? "AFTER" : "BEFORE");
    }
  }
  return 0;
}
