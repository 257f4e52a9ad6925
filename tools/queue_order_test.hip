// Does the in-order guarantee of ONE HIP stream hold while a second stream keeps the GPU busy?
// Stream A: K1 writes a buffer (late workgroups are slow), K2 checks it, repeatedly.
// Stream B: an unrelated long kernel, back to back.  Prints the number of stale reads.
// hipcc --offload-arch=gfx950 -O2 tools/queue_order_test.hip -o /tmp/qot && /tmp/qot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_write(double *x, int n, double val, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // later workgroups take longer, like the tail of a trailing update
  double acc = val;
  const int reps = spin * (1 + (int)(blockIdx.x % 8));
  for (int r = 0; r < reps; ++r) acc = fma(acc, 1.0000001, 1e-9);
  if (i < n) x[i] = val + 0.0 * acc;
}

__global__ void k_check(const double *x, int n, double val, unsigned long long *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && x[i] != val) atomicAdd(err, 1ull);
}

__global__ void k_busy(double *y, int n, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = (double)i;
  for (int r = 0; r < reps; ++r) acc = fma(acc, 0.9999999, 1e-3);
  if (i < n) y[i] = acc;
}

int main(int argc, char **argv) {
  const int with_b = argc > 1 ? atoi(argv[1]) : 1;
  const int iters = argc > 2 ? atoi(argv[2]) : 300;
  const int n = 1 << 22;
  double *x, *y;
  unsigned long long *err, herr = 0;
  hipMalloc(&x, n * sizeof(double));
  hipMalloc(&y, n * sizeof(double));
  hipMalloc(&err, sizeof(*err));
  hipMemset(err, 0, sizeof(*err));
  hipMemset(x, 0, n * sizeof(double));
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  for (int it = 1; it <= iters; ++it) {
    if (with_b) hipLaunchKernelGGL(k_busy, dim3(n / 256), dim3(256), 0, b, y, n, 2000);
    hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, a, x, n, (double)it, 50);
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, a, x, n, (double)it, err);
  }
  hipDeviceSynchronize();
  hipMemcpy(&herr, err, sizeof(herr), hipMemcpyDeviceToHost);
  printf("second stream %s: %d iterations, stale reads = %llu\n", with_b ? "busy" : "idle", iters,
         herr);
  return 0;
}
