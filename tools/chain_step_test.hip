// chain_a_plus / chain_b_own of pgf_factor2.hip AS THEY WERE before chain_b_own took its
// multipliers from LDS (a snapshot, copied in) in isolation: ONE wavefront eliminating a
// 16-column step of a 64 x 64 tile in LDS, and one wavefront per 64 rows solving the rows below
// -- how long do they take with the CU to themselves?  (MI355X: a+ 1.65 us per step, b 2.05 us:
// the measurement that sent chain_b_own to LDS broadcast reads.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
#define C_LD 66
#define C_WLD 18
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_recip(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  return r;
}
__device__ __forceinline__ void chain_a_plus(double (*M)[C_LD], double (*Wt)[C_WLD], double *dD,
                                             double *dI, int &s_bad, int lane, int sb, int ncol) {
  const int cb = sb * 16;
  double a[16], w[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t *>(&M[lane][cb + k]);
    a[k] = v.x;
    a[k + 1] = v.y;
  }
  // The serial chain of the whole factorisation runs through this loop: pivot -> reciprocal ->
  // multiplier column -> the ONE entry the next pivot needs -> next pivot.  Written software-
  // pipelined, with a scheduling barrier per column: left alone, the compiler's list scheduler
  // turns the right-looking updates into a lazy (left-looking) order in which column j waits
  // for a chain of j dependent FMAs right before its pivot -- 3.5 us per 16 columns instead
  // of about one.  Per column: the next pivot's entry is updated first and its reciprocal
  // chain started, the other 14 - j updates (independent FMAs, two v_readlane each) fill in.
  // classes flagged bad: sNaN, qNaN, -inf, -0, +0, +inf
  // Dependent fp64 operations cost ~30 cycles each on a lone wavefront, so the chain carries
  // as few as possible: reciprocal seed + ONE Newton step (v_rcp_f64 delivers > 26 bits; the
  // pivots only enter through products and the 1e-10 bar leaves five digits), and the product
  // of the next pivot's two factors is formed while the reciprocal is still in flight:
  //   a[j+1] -= (w[j] * c) * (1 / d)   instead of   a[j+1] -= (w[j] / d) * c.
  double d = lane_bcast(a[0], cb);
  bool bad_any = __builtin_amdgcn_class(d, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200) && cb < ncol;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    w[j] = a[j];
    double r = __builtin_amdgcn_rcp(d);
    double pc = 0.0;
    if (j + 1 < 16) pc = a[j] * lane_bcast(a[j], cb + j + 1);  // beside the reciprocal
    r = fma(r, fma(-d, r, 1.0), r);
    if (j + 1 < 16) {
      a[j + 1] = fma(-pc, r, a[j + 1]);
      d = lane_bcast(a[j + 1], cb + j + 1);
      bad_any |= __builtin_amdgcn_class(d, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200) &&
                 (cb + j + 1) < ncol;
    }
    const double l = a[j] * r;
#pragma unroll
    for (int k = j + 2; k < 16; ++k) a[k] = fma(-l, lane_bcast(w[j], cb + k), a[k]);
    a[j] = l;
    __builtin_amdgcn_sched_barrier(0);
  }
  const int tr = lane - cb;  // row inside the 16 x 16 tile
  if (tr >= 0 && tr < 16) {
    double d_mine = 1.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < tr) M[lane][cb + k] = a[k];
      if (k == tr) d_mine = w[k];
    }
    const bool ok = !__builtin_amdgcn_class(d_mine, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200);
    M[lane][lane] = d_mine;
    dD[lane] = d_mine;
    dI[lane] = ok ? fast_recip(d_mine) : 0.0;
    if (tr == 0 && bad_any) s_bad = 1;
  } else if (tr >= 16) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      double2_t wv, lv;
      wv.x = w[k];
      wv.y = w[k + 1];
      lv.x = a[k];
      lv.y = a[k + 1];
      *reinterpret_cast<double2_t *>(&Wt[lane][k]) = wv;
      *reinterpret_cast<double2_t *>(&M[lane][cb + k]) = lv;
    }
  }
}

__device__ __forceinline__ void chain_b_own(double (*M)[C_LD], int row, int sbp, int lane) {
  const int cb = sbp * 16;
  double x[16], tl[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t *>(&M[row][cb + k]);
    x[k] = v.x;
    x[k + 1] = v.y;
    const double2_t u = *reinterpret_cast<const double2_t *>(&M[cb + (lane & 15)][cb + k]);
    tl[k] = u.x;
    tl[k + 1] = u.y;
  }
#pragma unroll
  for (int t = 0; t < 15; ++t) {
    const double xt = x[t];
#pragma unroll
    for (int j = t + 1; j < 16; ++j) x[j] = fma(-xt, lane_bcast(tl[t], j), x[j]);  // L_bb[j][t]
    __builtin_amdgcn_sched_barrier(0);  // eager (right-looking) order, see chain_a_plus
  }
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    double2_t wv;
    wv.x = x[k];
    wv.y = x[k + 1];
    *reinterpret_cast<double2_t *>(&M[row][cb + k]) = wv;
  }
}


__global__ __launch_bounds__(1024) void k_test(long long *cyc, double *out, int mode) {
  __shared__ double M[256][C_LD];
  __shared__ double Wt[2][64][C_WLD];
  __shared__ double dD[64], dI[64];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int p = tid; p < 256 * 64; p += blockDim.x) {
    const int r = p >> 6, c = p & 63;
    M[r][c] = (r == c) ? 4.0 + 0.01 * r : 0.3 / (1.0 + ((r * 7 + c * 13) % 11));
  }
  __syncthreads();
  long long t0 = 0, t1 = 0;
  for (int rep = 0; rep < 2; ++rep) {
    __syncthreads();
    t0 = wall_clock64();
    if (wave == 0 && (mode & 1)) {
      for (int sb = 0; sb < 4; ++sb) chain_a_plus(M, Wt[sb & 1], dD, dI, s_bad, lane, sb, 64);
    }
    if (wave >= 1 && wave <= 3 && (mode & 2)) {
      for (int sb = 0; sb < 4; ++sb) chain_b_own(M, 64 * wave + lane, sb, lane);
    }
    __syncthreads();
    t1 = wall_clock64();
  }
  if (tid == 0) cyc[mode] = t1 - t0;
  out[tid & 63] = M[tid & 63][3];
}
int main() {
  long long *cyc, h[4];
  double *out;
  hipMalloc(&cyc, 32);
  hipMalloc(&out, 64 * 8);
  for (int mode = 1; mode <= 3; ++mode) hipLaunchKernelGGL(k_test, dim3(1), dim3(1024), 0, 0, cyc, out, mode);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("four 16-column steps (no barriers between): a+ alone %.2f us, b alone %.2f us, both %.2f us (100 MHz clock)\n",
         h[1] * 0.01, h[2] * 0.01, h[3] * 0.01);
  return 0;
}
