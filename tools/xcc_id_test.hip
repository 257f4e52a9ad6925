// Which XCD does workgroup i of a 1-D grid land on?  (s_getreg HW_REG_XCC_ID, id 20, bits 3:0)
// hipcc --offload-arch=gfx950 -O2 tools/xcc_id_test.hip -o /tmp/xcc && /tmp/xcc
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(6164) & 15);
}
int main() {
  const int n = 64;
  int *d, h[n];
  hipMalloc(&d, n * sizeof(int));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("launch %d:", rep);
    for (int i = 0; i < n; ++i) printf(" %d", h[i]);
    printf("\n");
  }
  return 0;
}
