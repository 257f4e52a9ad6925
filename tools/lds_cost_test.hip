// Issue cost of LDS stores / loads of one wavefront (cycles per instruction, back to back, then
// s_waitcnt): ds_write_b64 / b128, 16 or 64 active lanes, row stride 144 B; ds_read_b64 / b128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(long long *cyc, double *out) {
  __shared__ __attribute__((aligned(16))) double S[64 * 18 + 64 * 16];
  const int lane = threadIdx.x;
  double v[16];
  for (int j = 0; j < 16; ++j) v[j] = lane + j;
  for (int i = lane; i < 64 * 18; i += 64) S[i] = i;
  __syncthreads();
  long long t0 = 0, t1 = 0;
  double acc = 0.0;
  for (int rep = 0; rep < 9; ++rep) {
    if (rep == 1) t0 = clock64();
    if (MODE == 0) {  // 8 x ds_write_b128, 16 lanes
      if (lane < 16)
#pragma unroll
        for (int j = 0; j < 16; j += 2) *reinterpret_cast<double2_t *>(&S[lane * 18 + j]) = (double2_t){v[j], v[j + 1]};
    } else if (MODE == 1) {  // 16 x ds_write_b64, 16 lanes
      if (lane < 16)
#pragma unroll
        for (int j = 0; j < 16; ++j) S[lane * 18 + j] = v[j];
    } else if (MODE == 2) {  // 8 x ds_write_b128, 64 lanes (distinct rows)
#pragma unroll
      for (int j = 0; j < 16; j += 2) *reinterpret_cast<double2_t *>(&S[lane * 18 + j]) = (double2_t){v[j], v[j + 1]};
    } else if (MODE == 3) {  // 16 x ds_write_b64, 64 lanes, [j][lane] (conflict-free)
#pragma unroll
      for (int j = 0; j < 16; ++j) S[j * 64 + lane] = v[j];
    } else if (MODE == 4) {  // 8 x ds_read_b128 of a row, 64 lanes (16 distinct rows)
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const double2_t r = *reinterpret_cast<const double2_t *>(&S[(lane & 15) * 18 + j]);
        acc += r.x + r.y;
      }
    } else if (MODE == 5) {  // 16 x ds_read_b64 [j][lane]
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += S[j * 64 + lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(v[j]));
  }
  t1 = clock64();
  if (lane == 0) cyc[MODE] = (t1 - t0) / 8;
  out[lane] = acc + S[lane];
}
int main() {
  long long *cyc, h[8];
  double *out;
  hipMalloc(&cyc, 64);
  hipMalloc(&out, 64 * 8);
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, cyc, out);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, cyc, out);
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, cyc, out);
  hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, cyc, out);
  hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, cyc, out);
  hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, cyc, out);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const char *nm[] = {"8 x ds_write_b128, 16 lanes, rows of 144 B", "16 x ds_write_b64, 16 lanes, rows of 144 B",
                      "8 x ds_write_b128, 64 lanes, rows of 144 B", "16 x ds_write_b64, 64 lanes, [j][lane]",
                      "8 x ds_read_b128 of a row (16 distinct rows)", "16 x ds_read_b64 [j][lane]"};
  for (int i = 0; i < 6; ++i) printf("%-48s %lld cycles\n", nm[i], h[i]);
  return 0;
}
