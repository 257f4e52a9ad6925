// Dense LDL^T factorisation and triangular solves for the reduced KKT matrix
// (replaces SuperLU gstrf/gstrs reached through scipy.sparse.linalg.splu at
// reference pygradflow/linear_solver/lu_solver.py:14,21).
//
// K is symmetric quasi-definite ([[H_II + lamb I, J_I'],[J_I, -delta I]]), so an
// unpivoted LDL^T exists for the natural order; the number of negative pivots is
// the inertia the reference asks its linear solver for (num_neg_eigvals).
//
// Layout: lower triangle, row-major, row stride ldk (multiple of 16 doubles, so
// every row starts on a 128-byte line).  Right-looking blocked algorithm:
//   per 64-column panel   k_ldlt_diag   one workgroup, block in LDS
//                         k_ldlt_trsm   one lane per panel row, row in VGPRs
//                         k_ldlt_update 128x128 tiles, v_mfma_f64_16x16x4_f64
// The trailing update is the FP64-MFMA-bound kernel the roofline is quoted on.
#include "pgf_internal.h"

#include <algorithm>

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ diag block
// Unblocked right-looking LDL^T of one NB x NB diagonal block held in LDS.
// One barrier per column: column j is only scaled at the very end, the rank-1
// update uses the unscaled column and 1/d_j.
template <int NB>
__global__ __launch_bounds__(256) void k_ldlt_diag(double *__restrict__ K, int64_t ldk, int N,
                                                    int c0, double *__restrict__ dvec,
                                                    double *__restrict__ dinv,
                                                    int *__restrict__ flags) {
  __shared__ double Ad[NB][NB + 1];
  __shared__ double dI[NB];
  const int tid = threadIdx.x;
  const int nb = min(NB, N - c0);
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j <= i) v = K[(int64_t)(c0 + i) * ldk + c0 + j];
    Ad[i][j] = v;
  }
  __syncthreads();
  const int ti = tid >> 4, tk = tid & 15;
  for (int j = 0; j < nb; ++j) {
    const double d = Ad[j][j];
    const bool bad = (d == 0.0) || !(fabs(d) <= 1.79e308);
    const double di = bad ? 0.0 : 1.0 / d;
    if (tid == 0) {
      dI[j] = di;
      if (bad) atomicOr(&flags[0], 1);
    }
    // trailing update: A[i][k] -= A[i][j] * A[k][j] / d   for j < k <= i < nb
    for (int i = j + 1 + ti; i < nb; i += 16) {
      const double li = Ad[i][j] * di;
      for (int k = j + 1 + tk; k <= i; k += 16) Ad[i][k] = fma(-li, Ad[k][j], Ad[i][k]);
    }
    __syncthreads();
  }
  // scale columns, write back L (strictly lower), D and 1/D
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    if (i < nb && j < i) K[(int64_t)(c0 + i) * ldk + c0 + j] = Ad[i][j] * dI[j];
  }
  if (tid < nb) {
    const double d = Ad[tid][tid];
    dvec[c0 + tid] = d;
    dinv[c0 + tid] = dI[tid];
    K[(int64_t)(c0 + tid) * ldk + c0 + tid] = d;
  }
  // inertia: count negative pivots of this block
  if (tid < 64) {
    int neg = 0;
    for (int j = tid; j < nb; j += 64) neg += (Ad[j][j] < 0.0) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) neg += __shfl_down(neg, off);
    if (tid == 0 && neg) atomicAdd(&flags[1], neg);
  }
}

// ------------------------------------------------------------------ panel TRSM
// Rows below the diagonal block: X L_kk^T = A_ik  (X = L_ik D_kk).  One lane owns
// one row (NB doubles in VGPRs, fully unrolled substitution); L_kk is broadcast
// from LDS.  Writes W = X (workspace, row stride NB) and L = X D^-1 in place.
template <int NB>
__global__ __launch_bounds__(64) void k_ldlt_trsm(double *__restrict__ K, int64_t ldk,
                                                   double *__restrict__ W, int N, int nrows,
                                                   int c0, const double *__restrict__ dinv) {
  __shared__ double Ls[NB][NB];
  __shared__ double dis[NB];
  const int tid = threadIdx.x;
  const int nb = min(NB, N - c0);
  for (int idx = tid; idx < NB * NB; idx += 64) {
    const int i = idx / NB, j = idx % NB;
    Ls[i][j] = (i < nb && j < i) ? K[(int64_t)(c0 + i) * ldk + c0 + j] : 0.0;
  }
  dis[tid] = (tid < nb) ? dinv[c0 + tid] : 0.0;
  __syncthreads();
  // rows below the block start at c0 + nb (a partial last block is followed only by
  // carried right-hand-side rows)
  const int r = c0 + nb + blockIdx.x * 64 + tid;
  if (r >= nrows) return;
  double *rowp = K + (int64_t)r * ldk + c0;
  double x[NB];
  if (nb == NB) {
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
      const double2_t v = *reinterpret_cast<const double2_t *>(rowp + j);
      x[j] = v.x;
      x[j + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j) x[j] = (j < nb) ? rowp[j] : 0.0;
  }
#pragma unroll
  for (int j = 1; j < NB; ++j) {
    double s = x[j];
#pragma unroll
    for (int t = 0; t < j; ++t) s = fma(-x[t], Ls[j][t], s);
    x[j] = s;
  }
  double *wp = W + (int64_t)r * NB;
#pragma unroll
  for (int j = 0; j < NB; j += 2) {
    double2_t w;
    w.x = x[j];
    w.y = x[j + 1];
    *reinterpret_cast<double2_t *>(wp + j) = w;
  }
  if (nb == NB) {
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
      double2_t l;
      l.x = x[j] * dis[j];
      l.y = x[j + 1] * dis[j + 1];
      *reinterpret_cast<double2_t *>(rowp + j) = l;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j < nb) rowp[j] = x[j] * dis[j];
  }
}

// ------------------------------------------------------------------ trailing update
// C[i][j] -= sum_k W[i][k] * L[j][k]   for row0 <= i < nrows, col0 <= j < colEnd, j <= i
// (rows >= N are carried right-hand sides: every column < N is "below" them).
// 128 x 128 tile per workgroup, 4 wavefronts as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles
// of v_mfma_f64_16x16x4_f64 (A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15],
// C/D: row = (l>>4) + 4*reg, col = l&15).  K-chunks of 16 staged through LDS with the
// next chunk prefetched into registers; LDS rows padded to 18 doubles (bank-conflict
// free ds_read_b64 for the fragment pattern, 16-byte aligned ds_write_b128).
#define UPD_BM 128
#define UPD_BK 16
#define UPD_LDS 18

__global__ __launch_bounds__(256, 2) void k_ldlt_update(double *__restrict__ K, int64_t ldk,
                                                        const double *__restrict__ W,
                                                        int64_t ldw, int N, int nrows, int row0,
                                                        int col0, int colEnd, int kc0, int KB) {
  const int i0 = row0 + blockIdx.y * UPD_BM;
  const int j0 = col0 + blockIdx.x * UPD_BM;
  if (j0 > i0 + UPD_BM - 1) return;  // tile entirely above the diagonal
  __shared__ __attribute__((aligned(16))) double As[UPD_BM][UPD_LDS];
  __shared__ __attribute__((aligned(16))) double Bs[UPD_BM][UPD_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // staging map: piece p = q*256 + tid -> row p>>3, two doubles at column (p&7)*2
  double2_t pa[4], pb[4];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = q * 256 + tid;
      const int row = p >> 3, kofs = (p & 7) * 2;
      const int gi = i0 + row, gj = j0 + row;
      double2_t va = (double2_t){0.0, 0.0}, vb = (double2_t){0.0, 0.0};
      if (gi < nrows) va = *reinterpret_cast<const double2_t *>(W + (int64_t)gi * ldw + kk + kofs);
      if (gj < colEnd)
        vb = *reinterpret_cast<const double2_t *>(K + (int64_t)gj * ldk + kc0 + kk + kofs);
      pa[q] = va;
      pb[q] = vb;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = q * 256 + tid;
      const int row = p >> 3, kofs = (p & 7) * 2;
      *reinterpret_cast<double2_t *>(&As[row][kofs]) = pa[q];
      *reinterpret_cast<double2_t *>(&Bs[row][kofs]) = pb[q];
    }
  };

  fetch(0);
  for (int kk = 0; kk < KB; kk += UPD_BK) {
    __syncthreads();  // previous chunk's fragment reads are done
    stage();
    __syncthreads();
    if (kk + UPD_BK < KB) fetch(kk + UPD_BK);
#pragma unroll
    for (int ks = 0; ks < UPD_BK; ks += 4) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = As[wr * 64 + t * 16 + l15][ks + l4];
        b[t] = Bs[wc * 64 + t * 16 + l15][ks + l4];
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
  }

  // epilogue: C -= acc on the lower triangle of the region
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      const int j = j0 + wc * 64 + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * 64 + mi * 16 + l4 + 4 * r;
        if (i < nrows && j < colEnd && j <= i) {
          double *cp = K + (int64_t)i * ldk + j;
          *cp = *cp - acc[mi][nj][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ triangular solves
// One launch per 64-row block.  Every workgroup (one wavefront) redundantly solves the
// 64 x 64 diagonal system with wavefront shuffles, then updates its own 64 entries of the
// work vector; workgroup 0 also publishes the block of the solution.  `z` is the work
// vector (updated in place below/above the block), `x` receives the solution.
template <int NB>
__global__ __launch_bounds__(64) void k_trsv_fwd(const double *__restrict__ K, int64_t ldk,
                                                  double *__restrict__ z, double *__restrict__ x,
                                                  int N, int c0) {
  const int lane = threadIdx.x;
  const int nb = min(NB, N - c0);
  // L_kk row of this lane (strictly lower part)
  double lrow[NB];
  const double *lp = K + (int64_t)(c0 + lane) * ldk + c0;
#pragma unroll
  for (int j = 0; j < NB; ++j) lrow[j] = (lane < nb && j < lane) ? lp[j] : 0.0;
  double xv = (lane < nb) ? z[c0 + lane] : 0.0;
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) {
    const double xj = __shfl(xv, j);
    xv = fma(-lrow[j], xj, xv);  // lrow[j] == 0 for lanes <= j
  }
  if (blockIdx.x == 0 && lane < nb) x[c0 + lane] = xv;
  // update rows below the block
  const int r = c0 + NB + blockIdx.x * 64 + lane;
  const bool live = r < N;
  const double *rp = K + (int64_t)(live ? r : 0) * ldk + c0;
  double s = live ? z[r] : 0.0;
#pragma unroll
  for (int j = 0; j < NB; j += 2) {
    double2_t v = (double2_t){0.0, 0.0};
    if (live) v = *reinterpret_cast<const double2_t *>(rp + j);
    s = fma(-v.x, __shfl(xv, j), s);
    s = fma(-v.y, __shfl(xv, j + 1), s);
  }
  if (live) z[r] = s;
}

// Backward: L^T s = w.  Block solved with the transposed diagonal block; columns to the
// left are updated with the block row L[c0.., t]^T (coalesced across lanes).
template <int NB>
__global__ __launch_bounds__(64) void k_trsv_bwd(const double *__restrict__ K, int64_t ldk,
                                                  double *__restrict__ z, double *__restrict__ x,
                                                  int N, int c0) {
  const int lane = threadIdx.x;
  const int nb = min(NB, N - c0);
  double lcol[NB];  // L[c0 + j][c0 + lane] for j > lane
#pragma unroll
  for (int j = 0; j < NB; ++j)
    lcol[j] = (j < nb && j > lane) ? K[(int64_t)(c0 + j) * ldk + c0 + lane] : 0.0;
  double xv = (lane < nb) ? z[c0 + lane] : 0.0;
#pragma unroll
  for (int j = NB - 1; j > 0; --j) {
    const double xj = __shfl(xv, j);
    xv = fma(-lcol[j], xj, xv);  // lcol[j] == 0 for lanes >= j
  }
  if (blockIdx.x == 0 && lane < nb) x[c0 + lane] = xv;
  // update entries left of the block: z[t] -= sum_j L[c0 + j][t] * x_j
  const int t = blockIdx.x * 64 + lane;
  if (c0 == 0) return;
  const bool live = t < c0;
  double s = live ? z[t] : 0.0;
  const double *cp = K + (int64_t)c0 * ldk + (live ? t : 0);
#pragma unroll 8
  for (int j = 0; j < NB; ++j) {
    const double xj = __shfl(xv, j);
    const double l = (live && j < nb) ? cp[(int64_t)j * ldk] : 0.0;
    s = fma(-l, xj, s);
  }
  if (live) z[t] = s;
}

__global__ void k_vec_scale(double *__restrict__ z, const double *__restrict__ dinv, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) z[i] *= dinv[i];
}

__global__ void k_vec_copy_strided(double *__restrict__ dst, const double *__restrict__ src,
                                   int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) dst[i] = src[i];
}

// ------------------------------------------------------------------ host side
static inline int64_t pick_ldk(int Nmax) {
  int64_t ld = ((int64_t)Nmax + 1 + 15) / 16 * 16;
  if (ld % 512 == 0) ld += 16;  // keep row starts off one HBM channel
  return ld;
}

hipError_t ldlt_alloc(DenseLdlt &f, int Nmax, hipStream_t stream) {
  f.Nmax = Nmax;
  f.ldk = pick_ldk(Nmax);
  f.stream = stream;
  hipError_t e;
  const size_t rows = (size_t)Nmax + 1 + PGF_NB;
  if ((e = hipMalloc(&f.K, rows * f.ldk * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.W, rows * PGF_NB * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.dvec, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.dinv, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.zwork, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.flags, 4 * sizeof(int))) != hipSuccess) return e;
  if ((e = hipHostMalloc(&f.h_flags, 4 * sizeof(int))) != hipSuccess) return e;
  return hipSuccess;
}

void ldlt_free(DenseLdlt &f) {
  if (f.K) (void)hipFree(f.K);
  if (f.W) (void)hipFree(f.W);
  if (f.dvec) (void)hipFree(f.dvec);
  if (f.dinv) (void)hipFree(f.dinv);
  if (f.zwork) (void)hipFree(f.zwork);
  if (f.flags) (void)hipFree(f.flags);
  if (f.h_flags) (void)hipHostFree(f.h_flags);
  f = DenseLdlt();
}

static hipEvent_t prof_event(PgfProfile *p) {
  if (!p->pool.empty()) {
    hipEvent_t e = p->pool.back();
    p->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

hipError_t ldlt_factor_async(DenseLdlt &f, int N, int nrows) {
  f.N = N;
  f.factored = false;
  hipStream_t s = f.stream;
  hipError_t e = hipMemsetAsync(f.flags, 0, 4 * sizeof(int), s);
  if (e != hipSuccess) return e;
  PgfProfile *p = (f.prof && f.prof->enabled) ? f.prof : nullptr;
  if (p) {
    p->factor_span.first = prof_event(p);
    p->factor_span.second = prof_event(p);
    p->factor_open = true;
    (void)hipEventRecord(p->factor_span.first, s);
  }
  for (int c0 = 0; c0 < N; c0 += PGF_NB) {
    hipLaunchKernelGGL(k_ldlt_diag<PGF_NB>, dim3(1), dim3(256), 0, s, f.K, f.ldk, N, c0, f.dvec,
                       f.dinv, f.flags);
    const int below = nrows - std::min(c0 + PGF_NB, N);
    if (below > 0) {
      hipLaunchKernelGGL(k_ldlt_trsm<PGF_NB>, dim3((below + 63) / 64), dim3(64), 0, s, f.K, f.ldk,
                         f.W, N, nrows, c0, f.dinv);
      const int c1 = c0 + PGF_NB;
      if (c1 < N) {
        const int tr = (nrows - c1 + UPD_BM - 1) / UPD_BM;
        const int tc = (N - c1 + UPD_BM - 1) / UPD_BM;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (p) {
          e0 = prof_event(p);
          e1 = prof_event(p);
          (void)hipEventRecord(e0, s);
        }
        hipLaunchKernelGGL(k_ldlt_update, dim3(tc, tr), dim3(256), 0, s, f.K, f.ldk, f.W,
                           (int64_t)PGF_NB, N, nrows, c1, c1, N, c0, PGF_NB);
        if (p) {
          (void)hipEventRecord(e1, s);
          p->update_spans.emplace_back(e0, e1);
          const double t = (double)(N - c1);
          // algorithmic flops of this launch: lower triangle of the trailing block
          // (+ carried rows), 2 flops per multiply-add, K-depth NB
          p->update_flops.push_back((t * (t + 1.0) + 2.0 * t * (nrows - N)) * PGF_NB);
        }
      }
    }
  }
  if (p) (void)hipEventRecord(p->factor_span.second, s);
  e = hipMemcpyAsync(f.h_flags, f.flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

int ldlt_finish(DenseLdlt &f, hipError_t *err) {
  hipError_t e = hipStreamSynchronize(f.stream);
  if (err) *err = e;
  if (e != hipSuccess) return -1;
  f.n_neg = f.h_flags[1];
  f.factored = (f.h_flags[0] == 0);
  return f.h_flags[0] ? 1 : 0;
}

hipError_t ldlt_backsolve_async(DenseLdlt &f, const double *w, double *sol) {
  const int N = f.N;
  hipStream_t s = f.stream;
  if (N == 0) return hipSuccess;
  hipLaunchKernelGGL(k_vec_copy_strided, dim3((N + 255) / 256), dim3(256), 0, s, f.zwork, w, N);
  const int last = ((N - 1) / PGF_NB) * PGF_NB;
  for (int c0 = last; c0 >= 0; c0 -= PGF_NB) {
    const int g = c0 > 0 ? (c0 + 63) / 64 : 1;
    hipLaunchKernelGGL(k_trsv_bwd<PGF_NB>, dim3(g), dim3(64), 0, s, f.K, f.ldk, f.zwork, sol, N,
                       c0);
  }
  return hipGetLastError();
}

hipError_t ldlt_solve_async(DenseLdlt &f, const double *rhs, double *sol) {
  const int N = f.N;
  hipStream_t s = f.stream;
  if (N == 0) return hipSuccess;
  // forward: L y = rhs  (work in zwork, y lands in sol)
  hipLaunchKernelGGL(k_vec_copy_strided, dim3((N + 255) / 256), dim3(256), 0, s, f.zwork, rhs, N);
  for (int c0 = 0; c0 < N; c0 += PGF_NB) {
    const int below = N - (c0 + PGF_NB);
    const int g = below > 0 ? (below + 63) / 64 : 1;
    hipLaunchKernelGGL(k_trsv_fwd<PGF_NB>, dim3(g), dim3(64), 0, s, f.K, f.ldk, f.zwork, sol, N,
                       c0);
  }
  // diagonal: y <- D^-1 y
  hipLaunchKernelGGL(k_vec_scale, dim3((N + 255) / 256), dim3(256), 0, s, sol, f.dinv, N);
  // backward: L^T s = y
  return ldlt_backsolve_async(f, sol, sol);
}
