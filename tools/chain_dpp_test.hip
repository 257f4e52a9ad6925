// Elimination of a 16 x 16 pivot tile with DPP row broadcasts instead of v_readlane:
//   v_fmac_f64_dpp acc, src row_newbcast:k, mult     acc += (lane k's src) * mult
// (gfx90a+ "DP ALU DPP": 64-bit DPP operands with row_newbcast only.)  Lane i of every row of 16
// lanes holds row i of the tile (all four DPP rows hold the same tile), so a rank-1 update is
// ONE instruction per target column instead of two v_readlane + one FMA.
//   mode 0: the v_readlane elimination of pgf_factor2.hip's chain_a_plus, pivot tile only
//   mode 1: DPP, pivot tile only (F16)
//   mode 2: DPP, pivot tile + one riding row per lane (64 rows below ride along: F16 + 64)
//   mode 3: as 1 with "s_nop 1" in front of every DPP instruction (hazard margin)
// Prints the time per 16-column step (lone wavefront) and the error against a host LDL^T.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int L>
__device__ __forceinline__ double row_bcast(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xf, 0xf, false);
}
// acc += (lane L's src) * mult
template <int L, bool NOP>
__device__ __forceinline__ void fmac_bcast(double &acc, double src, double mult) {
  if (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mult), "n"(L));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mult), "n"(L));
}

// column C of the elimination; a[] = the lane's row (lower triangle valid), l[] = multipliers,
// r = 1 / pivot C on entry, 1 / pivot C + 1 on exit; b[] = the riding row (RIDE)
template <int C, bool NOP, bool RIDE>
__device__ __forceinline__ void dpp_col(double (&a)[16], double (&l)[16], double (&b)[16],
                                        double (&m)[16], double &r, double (&dd)[16]) {
  const double nl = -a[C] * r;  // -L[i][C]
  double rn = 0.0;
  if (C + 1 < 16) {
    // next pivot's column first, its reciprocal started, the other updates fill in behind
    fmac_bcast<(C + 1) & 15, NOP>(a[(C + 1) & 15], a[C], nl);
    const double d = row_bcast<(C + 1) & 15>(a[(C + 1) & 15]);
    dd[(C + 1) & 15] = d;
    rn = __builtin_amdgcn_rcp(d);
  }
  l[C] = nl;
  if (RIDE) m[C] = -b[C] * r;
#define UPD(K)                                                    \
  if (K > C + 1) {                                                \
    fmac_bcast<K, NOP>(a[K], a[C], nl);                           \
  }                                                               \
  if (RIDE && K > C) {                                            \
    fmac_bcast<K, NOP>(b[K], a[C], m[C]);                         \
  }
  UPD(1) UPD(2) UPD(3) UPD(4) UPD(5) UPD(6) UPD(7) UPD(8) UPD(9) UPD(10) UPD(11) UPD(12) UPD(13)
  UPD(14) UPD(15)
#undef UPD
  if (C + 1 < 16) {
    const double d = dd[(C + 1) & 15];
    rn = fma(rn, fma(-d, rn, 1.0), rn);
    r = rn;
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <bool NOP, bool RIDE>
__device__ __forceinline__ void f16_dpp(double (&a)[16], double (&l)[16], double (&b)[16],
                                        double (&m)[16], double (&dd)[16]) {
  const double d0 = row_bcast<0>(a[0]);
  dd[0] = d0;
  double r = __builtin_amdgcn_rcp(d0);
  r = fma(r, fma(-d0, r, 1.0), r);
  dpp_col<0, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<1, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<2, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<3, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<4, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<5, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<6, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<7, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<8, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<9, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<10, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<11, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<12, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<13, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<14, NOP, RIDE>(a, l, b, m, r, dd);
  dpp_col<15, NOP, RIDE>(a, l, b, m, r, dd);
}

// the v_readlane scheme of chain_a_plus, restricted to the 16 x 16 tile (lanes 0..15)
__device__ __forceinline__ void f16_readlane(double (&a)[16], double (&w)[16]) {
  double d = lane_bcast(a[0], 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    w[j] = a[j];
    double r = __builtin_amdgcn_rcp(d);
    double pc = 0.0;
    if (j + 1 < 16) pc = a[j] * lane_bcast(a[j], j + 1);
    r = fma(r, fma(-d, r, 1.0), r);
    if (j + 1 < 16) {
      a[j + 1] = fma(-pc, r, a[j + 1]);
      d = lane_bcast(a[j + 1], j + 1);
    }
    const double l = a[j] * r;
#pragma unroll
    for (int k = j + 2; k < 16; ++k) a[k] = fma(-l, lane_bcast(w[j], k), a[k]);
    a[j] = l;
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(64) void k_test(const double *A, const double *B, double *Lout,
                                             double *Dout, double *Xout, long long *cyc, int mode,
                                             int reps) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  double a0[16], b0[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    a0[j] = A[i * 16 + j];
    b0[j] = B[lane * 16 + j];
  }
  double a[16], l[16], b[16], m[16], dd[16];
  long long t0 = 0, t1 = 0, c0 = 0, c1 = 0;
  for (int rep = 0; rep < reps; ++rep) {
    if (rep == 1) {
      t0 = wall_clock64();
      c0 = clock64();
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a[j] = a0[j];
      b[j] = b0[j];
      l[j] = 0.0;
      m[j] = 0.0;
      dd[j] = 0.0;
    }
    // (keeps the compiler from hoisting or merging repetitions)
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]), "+v"(b[j]));
    if (mode == 0) {
      f16_readlane(a, l);  // l = w here
    } else if (mode == 1) {
      f16_dpp<false, false>(a, l, b, m, dd);
    } else if (mode == 2) {
      f16_dpp<false, true>(a, l, b, m, dd);
    } else {
      f16_dpp<true, false>(a, l, b, m, dd);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]), "+v"(b[j]), "+v"(l[j]), "+v"(m[j]));
  }
  t1 = wall_clock64();
  c1 = clock64();
  if (lane == 0) {
    cyc[2 * mode] = t1 - t0;
    cyc[2 * mode + 1] = c1 - c0;
  }
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // mode 0 leaves L in a[] (w in l[]); the DPP modes leave -L in l[] and W = L D in a[]
      double v = (mode == 0) ? a[j] : -l[j];
      Lout[(mode * 16 + i) * 16 + j] = v;
    }
    double dv = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j == i) dv = (mode == 0) ? l[j] : dd[j];
    Dout[mode * 16 + i] = dv;
  }
  if (mode == 2) {
#pragma unroll
    for (int j = 0; j < 16; ++j) Xout[lane * 16 + j] = b[j];
  }
}

int main() {
  const int n = 16;
  std::vector<double> A(n * n), B(64 * n), L(n * n, 0.0), D(n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const int lo = std::min(i, j), hi = std::max(i, j);
      A[i * n + j] = (i == j) ? ((i % 3 == 2) ? -(3.0 + 0.1 * i) : 4.0 + 0.01 * i)
                              : 0.3 / (1.0 + ((hi * 7 + lo * 13) % 11));
    }
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < n; ++j) B[i * n + j] = 0.2 + 0.01 * ((i * 5 + j * 3) % 17);
  // host LDL^T
  std::vector<double> M = A;
  for (int c = 0; c < n; ++c) {
    D[c] = M[c * n + c];
    for (int i2 = c + 1; i2 < n; ++i2) L[i2 * n + c] = M[i2 * n + c] / D[c];
    for (int i2 = c + 1; i2 < n; ++i2)
      for (int j = c + 1; j <= i2; ++j) M[i2 * n + j] -= L[i2 * n + c] * M[j * n + c];
  }
  // riding rows: X L^T = B  (X = rows of W-form: x[j] = b[j] - sum_{t<j} x[t] L[j][t])
  std::vector<double> X(64 * n);
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < n; ++j) {
      double s = B[i * n + j];
      for (int t = 0; t < j; ++t) s -= X[i * n + t] * L[j * n + t];
      X[i * n + j] = s;
    }
  double *dA, *dB, *dL, *dD, *dX;
  long long *dc, hc[8];
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&dB, B.size() * 8);
  hipMalloc(&dL, 4 * n * n * 8);
  hipMalloc(&dD, 4 * n * 8);
  hipMalloc(&dX, 64 * n * 8);
  hipMalloc(&dc, sizeof(hc));
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipMemset(dL, 0, 4 * n * n * 8);
  const int reps = 65;
  for (int mode = 0; mode < 4; ++mode)
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dL, dD, dX, dc, mode, reps);
  hipDeviceSynchronize();
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  std::vector<double> hL(4 * n * n), hD(4 * n), hX(64 * n);
  hipMemcpy(hL.data(), dL, hL.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hD.data(), dD, hD.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hX.data(), dX, hX.size() * 8, hipMemcpyDeviceToHost);
  const char *names[4] = {"v_readlane, tile only", "DPP, tile only", "DPP, tile + 64 riding rows",
                          "DPP + s_nop, tile only"};
  for (int mode = 0; mode < 4; ++mode) {
    double eL = 0.0, eD = 0.0;
    for (int i = 0; i < n; ++i) {
      eD = std::max(eD, std::fabs(hD[mode * n + i] - D[i]));
      for (int j = 0; j < i; ++j) eL = std::max(eL, std::fabs(hL[(mode * n + i) * n + j] - L[i * n + j]));
    }
    printf("%-28s %.3f us per step, %6.0f shader cycles   err L %.1e D %.1e\n", names[mode],
           hc[2 * mode] * 0.01 / (reps - 1), (double)hc[2 * mode + 1] / (reps - 1), eL, eD);
  }
  double eX = 0.0;
  for (size_t q = 0; q < X.size(); ++q) eX = std::max(eX, std::fabs(hX[q] - X[q]));
  printf("riding rows: err X %.1e\n", eX);
  return 0;
}
