// Dense LU factorisation with partial (row) pivoting,  P A = L U,  and the solves with A and
// A^T: the unsymmetric counterpart of pgf_ldlt.hip / pgf_factor2.hip.
//
// The reference's default linear solver is LU for EVERY matrix (scipy.sparse.linalg.splu at
// pygradflow/linear_solver/lu_solver.py:9-21; its `symmetric` flag is ignored there), and three
// of its four step-solver formulations hand it an unsymmetric (n + m)^2 Newton matrix
// (step/solver/standard_step_solver.py:40-92, extended_step_solver.py:39-112,
// asymmetric_step_solver.py:38-173).  This file is the linear solver behind those
// formulations (HipLinearSolver(symmetric=False)); it is not on the headline path (the
// symmetric reduced KKT system goes through the LDL^T kernels) and is written for
// robustness first: classical right-looking blocked LU, panel width 32,
//   k_lu_panel    ONE workgroup on a column-major copy of the panel (k_lu_panel_load / _store):
//                 per column pivot search (max |a|, smallest row on ties), interchange inside
//                 the panel, scaling, rank-1 update; k_lu_swap_rows applies the panel's
//                 interchanges to the rest of the rows afterwards
//   k_lu_trsm     U12 = L11^-1 A12, one lane per column
//   k_lu_update   A22 -= L21 U12, 64 x 64 tiles, register-blocked fp64 FMAs (on gfx950 the
//                 vector fp64 FMA rate equals the MFMA rate; at K-depth 32 the update is
//                 bound by its C traffic anyway)
// and blocked triangular solves (64-row blocks: k_tri_diag + k_tri_gemv), with a TRANS
// switch for A^T x = b (LinearSolver.solve(rhs, trans=True), reference
// linear_solver/linear_solver.py:23-25, used by step/cond_estimate.py:82).
// A zero or non-finite pivot sets flags[0] -> PGF_SINGULAR -> LinearSolverError, like the
// RuntimeError of splu (lu_solver.py:13-17).
#include "pgf_internal.h"
#include "pgf_ldlt_dev.h"

#include <algorithm>
#include <cmath>
#include <vector>

#define LU_PB 32

// The panel rows [c0, N) x columns [c0, c0 + pb) as a column-major copy PT[j][r] (and back):
// in A a column is strided by the row length, one cache line per entry -- the pivot search and
// the rank-1 updates of a 32-column panel took 1.85 ms at N = 5120 (95 % of the factorisation).
__global__ __launch_bounds__(256) void k_lu_panel_load(const double *__restrict__ A, int64_t ld,
                                                       int N, int c0, int pb,
                                                       double *__restrict__ PT, int64_t ldp) {
  const int r = c0 + blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const double *row = A + (int64_t)r * ld + c0;
  for (int j = 0; j < pb; ++j) PT[(int64_t)j * ldp + r] = row[j];
}
__global__ __launch_bounds__(256) void k_lu_panel_store(double *__restrict__ A, int64_t ld, int N,
                                                        int c0, int pb,
                                                        const double *__restrict__ PT, int64_t ldp) {
  const int r = c0 + blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  double *row = A + (int64_t)r * ld + c0;
  for (int j = 0; j < pb; ++j) row[j] = PT[(int64_t)j * ldp + r];
}

// ONE workgroup factorises the panel in PT: per column the pivot (max |a|, smallest row on
// ties; a NaN wins: it must be reported), the interchange of the two rows INSIDE the panel,
// the multipliers and the rank-1 update of the panel's remaining columns.  The interchanges
// of the rest of the two rows are k_lu_swap_rows' (after the panel, all 32 at once).
__global__ __launch_bounds__(1024) void k_lu_panel(double *__restrict__ PT, int64_t ldp, int N, int c0,
                                                   int pb, int *__restrict__ piv,
                                                   int *__restrict__ flags) {
  __shared__ double sval[16];
  __shared__ int sidx[16];
  __shared__ double prow[LU_PB];
  __shared__ int s_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto better = [](double v, int vi, double w, int wi) {  // (v, vi) beats (w, wi)
    return (v != v && w == w) || (w == w && (v > w || (v == w && vi < wi)));
  };
  for (int j = 0; j < pb; ++j) {
    const int col = c0 + j;
    double *cj = PT + (int64_t)j * ldp;
    double best = -1.0;
    int bi = N;
    for (int r = col + tid; r < N; r += 1024) {
      const double v = fabs(cj[r]);
      if (better(v, r, best, bi)) {
        best = v;
        bi = r;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double v = __shfl_down(best, off);
      const int vi = __shfl_down(bi, off);
      if (better(v, vi, best, bi)) {
        best = v;
        bi = vi;
      }
    }
    if (lane == 0) {
      sval[wave] = best;
      sidx[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      double v = sval[0];
      int vi = sidx[0];
      for (int w = 1; w < 16; ++w)
        if (better(sval[w], sidx[w], v, vi)) {
          v = sval[w];
          vi = sidx[w];
        }
      const int p = min(vi, N - 1);
      s_p = p;
      piv[col] = p;
      if (!(v > 0.0) || !(v <= 1.79e308)) atomicOr(&flags[0], 1);
    }
    __syncthreads();
    const int p = s_p;
    // interchange inside the panel, then the pivot row's entries of the columns to the right
    if (tid < pb) {
      double *ct = PT + (int64_t)tid * ldp;
      const double a = ct[col], b = ct[p];
      ct[col] = b;
      ct[p] = a;
      prow[tid] = b;
    }
    __syncthreads();
    const double rinv = 1.0 / prow[j];
    for (int r = col + 1 + tid; r < N; r += 1024) {
      const double l = cj[r] * rinv;
      cj[r] = l;
      for (int k = j + 1; k < pb; ++k) {
        double *ck = PT + (int64_t)k * ldp;
        ck[r] = fma(-l, prow[k], ck[r]);
      }
    }
    __syncthreads();
  }
}

// The same panel factorisation with the rows in REGISTERS (round 3).  k_lu_panel streams the
// remaining panel through one CU for every column (rank-1 update: 32 x 16 x 40 KB = 21 MB per
// panel at N = 5120, 0.44 ms, 70 of the factorisation's 88 ms).  Here thread t owns the rows
// c0 + t + 1024 i (i < R) and the panel is taken in groups of eight columns, left-looking:
//   (1) the group's slice of the owned rows goes into registers (R x 8 doubles);
//   (2) U12 = L11^-1 A12 for the pivot rows the panel already has (wavefront 0, <= 24 x 8), then
//       every other row takes  s -= L21 U12  from its own multipliers (one read of the panel's
//       earlier columns per group instead of one rank-1 pass over the panel per column);
//   (3) the eight columns are eliminated in registers: pivot search over the owned rows, the two
//       rows exchanged (registers through LDS for the group, global memory for the panel's other
//       columns), the pivot row broadcast through LDS, scaling and rank-1 update in registers;
//   (4) the slice goes back.
// Same pivots as k_lu_panel (max |a|, smallest row on ties, a NaN wins), same piv / flags.
#define LU_G 8
template <int R>
__global__ __launch_bounds__(1024) void k_lu_panel_reg(double *__restrict__ PT, int64_t ldp, int N, int c0,
                                                       int pb, int *__restrict__ piv,
                                                       int *__restrict__ flags) {
  __shared__ double sval[16];
  __shared__ int sidx[16];
  __shared__ double prow[LU_G];
  __shared__ double xrow[2][LU_G];
  __shared__ double u12[LU_PB][LU_G];
  __shared__ double l11[LU_PB - LU_G][LU_PB - LU_G + 1];
  __shared__ int s_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto better = [](double v, int vi, double w, int wi) {  // (v, vi) beats (w, wi)
    return (v != v && w == w) || (w == w && (v > w || (v == w && vi < wi)));
  };
  double s[R][LU_G];
  for (int g0 = 0; g0 < pb; g0 += LU_G) {
    const int gw = min(LU_G, pb - g0);  // columns of this group
    // (1) the owned rows' entries of the group's columns
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int r = c0 + tid + 1024 * i;
#pragma unroll
      for (int c = 0; c < LU_G; ++c) s[i][c] = (r < N && c < gw) ? PT[(int64_t)(g0 + c) * ldp + r] : 0.0;
    }
    // (2) the panel's earlier columns: U12 = L11^-1 A12 on the g0 pivot rows, by wavefront 0 with
    // lane <-> column (g0 <= 24 sequential steps of g0 FMAs), then the update of the rows below
    if (g0 > 0) {
      // L11 (unit lower, g0 x g0) and A12 (g0 x gw) into LDS
      for (int e = tid; e < g0 * g0; e += 1024) {
        const int i2 = e / g0, q = e - i2 * g0;
        l11[i2][q] = (q < i2) ? PT[(int64_t)q * ldp + c0 + i2] : 0.0;
      }
      for (int e = tid; e < g0 * LU_G; e += 1024) {
        const int q = e / LU_G, c = e - q * LU_G;
        u12[q][c] = (c < gw) ? PT[(int64_t)(g0 + c) * ldp + c0 + q] : 0.0;
      }
      __syncthreads();
      if (wave == 0) {  // forward substitution, lane = (column, row slot); LDS is in order within a wavefront
        const int c = lane & 7, slot = lane >> 3;
        for (int q = 0; q + 1 < g0; ++q) {
          const double xq = u12[q][c];
          for (int i2 = q + 1 + slot; i2 < g0; i2 += 8) u12[i2][c] = fma(-l11[i2][q], xq, u12[i2][c]);
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      for (int e = tid; e < g0 * LU_G; e += 1024) {
        const int q = e / LU_G, c = e - q * LU_G;
        if (c < gw) PT[(int64_t)(g0 + c) * ldp + c0 + q] = u12[q][c];  // final entries of U
      }
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int r = c0 + tid + 1024 * i;
        if (r >= c0 + g0 && r < N) {
          for (int q = 0; q < g0; ++q) {
            const double l = PT[(int64_t)q * ldp + r];
#pragma unroll
            for (int c = 0; c < LU_G; ++c) s[i][c] = fma(-l, u12[q][c], s[i][c]);
          }
        }
      }
    }
    // (3) the group's columns (unrolled: c is a compile-time register index)
#pragma unroll
    for (int c = 0; c < LU_G; ++c) {
      if (c >= gw) break;
      const int col = c0 + g0 + c;
      double best = -1.0;
      int bi = N;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int r = c0 + tid + 1024 * i;
        if (r >= col && r < N) {
          const double v = fabs(s[i][c]);
          if (better(v, r, best, bi)) {
            best = v;
            bi = r;
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double v = __shfl_down(best, off);
        const int vi = __shfl_down(bi, off);
        if (better(v, vi, best, bi)) {
          best = v;
          bi = vi;
        }
      }
      if (lane == 0) {
        sval[wave] = best;
        sidx[wave] = bi;
      }
      __syncthreads();
      if (tid == 0) {
        double v = sval[0];
        int vi = sidx[0];
        for (int w = 1; w < 16; ++w)
          if (better(sval[w], sidx[w], v, vi)) {
            v = sval[w];
            vi = sidx[w];
          }
        const int p = min(vi, N - 1);
        s_p = p;
        piv[col] = p;
        if (!(v > 0.0) || !(v <= 1.79e308)) atomicOr(&flags[0], 1);
      }
      __syncthreads();
      const int p = s_p;
      // the two rows' entries of the group (registers, through LDS) ...
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int r = c0 + tid + 1024 * i;
        if (r == col || r == p) {
          const int slot = (r == col) ? 0 : 1;
#pragma unroll
          for (int cc = 0; cc < LU_G; ++cc) xrow[slot][cc] = s[i][cc];
        }
      }
      // ... and of the panel's other columns (global memory: earlier multipliers, later raw entries)
      if (p != col && tid < pb && (tid < g0 || tid >= g0 + gw)) {
        double *ct = PT + (int64_t)tid * ldp;
        const double a = ct[col], b = ct[p];
        ct[col] = b;
        ct[p] = a;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int r = c0 + tid + 1024 * i;
        if (p != col && (r == col || r == p)) {
          const int slot = (r == col) ? 1 : 0;  // the other row's entries
#pragma unroll
          for (int cc = 0; cc < LU_G; ++cc) s[i][cc] = xrow[slot][cc];
        }
      }
      // pivot row (now at `col'): xrow[1] held row p's entries before the exchange
      if (tid < LU_G) prow[tid] = (p != col) ? xrow[1][tid] : xrow[0][tid];
      __syncthreads();
      const double rinv = 1.0 / prow[c];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int r = c0 + tid + 1024 * i;
        if (r > col && r < N) {
          const double l = s[i][c] * rinv;
          s[i][c] = l;
#pragma unroll
          for (int cc = c + 1; cc < LU_G; ++cc) s[i][cc] = fma(-l, prow[cc], s[i][cc]);
        }
      }
      __syncthreads();  // (sval / xrow / prow are rewritten by the next column)
    }
    // (4) (the pivot rows of earlier groups hold U12, which step 2 has already stored)
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int r = c0 + tid + 1024 * i;
      if (r >= c0 + g0 && r < N)
#pragma unroll
        for (int c = 0; c < LU_G; ++c)
          if (c < gw) PT[(int64_t)(g0 + c) * ldp + r] = s[i][c];
    }
    __syncthreads();  // the next group reads these columns as multipliers
  }
}

// the panel's row interchanges applied to the columns outside it (L to the left included, as
// LAPACK's dgetrf does): thread <-> column, the pb interchanges in their order
__global__ __launch_bounds__(256) void k_lu_swap_rows(double *__restrict__ A, int64_t ld, int N, int c0,
                                                      int pb, const int *__restrict__ piv) {
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= c0) c += pb;  // skip the panel's own columns
  if (c >= N) return;
  for (int j = 0; j < pb; ++j) {
    const int r = c0 + j, p = piv[r];
    if (p != r) {
      const double a = A[(int64_t)r * ld + c], b = A[(int64_t)p * ld + c];
      A[(int64_t)r * ld + c] = b;
      A[(int64_t)p * ld + c] = a;
    }
  }
}

// U12 = L11^-1 A12: lane <-> column of A12.  The panel's unit-lower L11 is wavefront-uniform
// data: lane r < 32 of every wavefront holds ROW r of it in registers and the multipliers are
// broadcast with v_readlane (scalar operands of the FMAs).  Read as LDS broadcasts the 496
// entries were all hoisted to the top of the kernel -- ~1000 registers, 550 of them spilled
// to scratch -- whatever scheduling barriers or volatile qualifiers stood in between.
// A narrower last panel is padded with zeros.
__global__ __launch_bounds__(256) void k_lu_trsm(double *A, int64_t ld, int N, int c0, int pb) {
  const int tid = threadIdx.x, r = tid & 31;
  double lrow[LU_PB];
#pragma unroll
  for (int k = 0; k < LU_PB; ++k)
    lrow[k] = (r < pb && k < pb && k < r) ? A[(int64_t)(c0 + r) * ld + c0 + k] : 0.0;
  const int j = c0 + pb + blockIdx.x * 256 + tid;
  const bool live = j < N;  // (no early return: every lane serves v_readlane)
  double x[LU_PB];
#pragma unroll
  for (int k = 0; k < LU_PB; ++k) x[k] = (live && k < pb) ? A[(int64_t)(c0 + k) * ld + j] : 0.0;
#pragma unroll
  for (int k = 0; k < LU_PB; ++k) {
#pragma unroll
    for (int i = k + 1; i < LU_PB; ++i) x[i] = fma(-lane_bcast(lrow[k], i), x[k], x[i]);  // L11[i][k]
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < LU_PB; ++k)
    if (live && k < pb) A[(int64_t)(c0 + k) * ld + j] = x[k];
}

// A22 -= L21 U12 over rows / columns [c1, N), K-depth pb <= 32 (L21 = columns [c0, c0 + pb)
// of the rows, U12 = rows [c0, c0 + pb) of the columns).  64 x 64 tile per workgroup of 256
// threads, 4 x 4 entries per thread.
__global__ __launch_bounds__(256) void k_lu_update(double *A, int64_t ld, int N, int c0, int pb,
                                                   int c1) {
  __shared__ double As[64][LU_PB + 1];
  __shared__ double Bs[LU_PB][64 + 1];
  const int tid = threadIdx.x;
  const int i0 = c1 + blockIdx.y * 64, j0 = c1 + blockIdx.x * 64;
  for (int idx = tid; idx < 64 * LU_PB; idx += 256) {
    const int r = idx / LU_PB, k = idx % LU_PB;
    As[r][k] = (i0 + r < N && k < pb) ? A[(int64_t)(i0 + r) * ld + c0 + k] : 0.0;
  }
  for (int idx = tid; idx < LU_PB * 64; idx += 256) {
    const int k = idx / 64, c = idx % 64;
    Bs[k][c] = (j0 + c < N && k < pb) ? A[(int64_t)(c0 + k) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;  // rows ty + 16 a, columns tx + 16 b
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 8
  for (int k = 0; k < LU_PB; ++k) {
    double av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = As[ty + 16 * a][k];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = Bs[k][tx + 16 * b];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ty + 16 * a;
    if (i >= N) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + tx + 16 * b;
      if (j < N) A[(int64_t)i * ld + j] -= acc[a][b];
    }
  }
}

// ---------------------------------------------------------------- triangular solves
// T(i, k): entry of the triangular factor the solve works with (TRANS reads the stored
// matrix transposed: U^T is lower, L^T is upper).
template <bool TRANS>
__device__ __forceinline__ double tri_at(const double *A, int64_t ld, int i, int k) {
  return TRANS ? A[(int64_t)k * ld + i] : A[(int64_t)i * ld + k];
}

// solve the 64-row diagonal block b0: x[b0 .. b0+nb) in place.  One workgroup of 64 lanes;
// LOWER: forward inside the block, else backward.
template <bool LOWER, bool UNIT, bool TRANS>
__global__ __launch_bounds__(64) void k_tri_diag(const double *__restrict__ A, int64_t ld, int N,
                                                 int b0, double *__restrict__ x) {
  __shared__ double T[64][65];
  __shared__ double xs[64];
  const int tid = threadIdx.x;
  const int nb = min(64, N - b0);
  for (int idx = tid; idx < 64 * 64; idx += 64) {
    const int i = idx >> 6, k = idx & 63;
    T[i][k] = (i < nb && k < nb) ? tri_at<TRANS>(A, ld, b0 + i, b0 + k) : (i == k ? 1.0 : 0.0);
  }
  xs[tid] = (tid < nb) ? x[b0 + tid] : 0.0;
  __syncthreads();
  double mine = xs[tid];
  for (int s = 0; s < nb; ++s) {
    const int k = LOWER ? s : nb - 1 - s;
    if (tid == k) {
      if (!UNIT) mine /= T[k][k];
      xs[k] = mine;
    }
    __syncthreads();
    const bool later = LOWER ? (tid > k) : (tid < k);
    if (later && tid < nb) mine = fma(-T[tid][k], xs[k], mine);
  }
  if (tid < nb) x[b0 + tid] = mine;
}

// after block b0 is solved: x[r] -= sum_k T(r, b0 + k) x[b0 + k] for the rows still open
// (LOWER: r >= b0 + nb, else r < b0); one lane per row
template <bool LOWER, bool TRANS>
__global__ __launch_bounds__(256) void k_tri_gemv(const double *__restrict__ A, int64_t ld, int N,
                                                  int b0, double *__restrict__ x) {
  __shared__ double xs[64];
  const int tid = threadIdx.x;
  const int nb = min(64, N - b0);
  if (tid < 64) xs[tid] = (tid < nb) ? x[b0 + tid] : 0.0;
  __syncthreads();
  const int r = (LOWER ? b0 + nb : 0) + blockIdx.x * 256 + tid;
  const int rend = LOWER ? N : b0;
  if (r >= rend) return;
  double s0 = 0.0, s1 = 0.0;
  for (int k = 0; k + 1 < nb; k += 2) {
    s0 = fma(tri_at<TRANS>(A, ld, r, b0 + k), xs[k], s0);
    s1 = fma(tri_at<TRANS>(A, ld, r, b0 + k + 1), xs[k + 1], s1);
  }
  if (nb & 1) s0 = fma(tri_at<TRANS>(A, ld, r, b0 + nb - 1), xs[nb - 1], s0);
  x[r] -= s0 + s1;
}

// out[i] = in[perm[i]] (gather) or out[perm[i]] = in[i] (scatter)
__global__ void k_permute(const double *__restrict__ in, const int *__restrict__ perm,
                          double *__restrict__ out, int N, int scatter) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (scatter) out[perm[i]] = in[i];
  else out[i] = in[perm[i]];
}

// ---------------------------------------------------------------- host side
hipError_t lu_alloc(DenseLu &f, int N, hipStream_t stream) {
  f.N = N;
  f.ld = ((int64_t)N + 15) / 16 * 16;
  if (f.ld % 512 == 0) f.ld += 16;
  f.stream = stream;
  hipError_t e;
  const size_t rows = (size_t)std::max(N, 1);
  if ((e = hipMalloc((void **)&f.A, rows * f.ld * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc((void **)&f.piv, rows * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMalloc((void **)&f.perm, rows * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMalloc((void **)&f.work, rows * sizeof(double))) != hipSuccess) return e;
  f.ldp = ((int64_t)rows + 15) / 16 * 16 + 16;  // (+16: panel columns do not share channels)
  if ((e = hipMalloc((void **)&f.PT, (size_t)LU_PB * f.ldp * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc((void **)&f.flags, 4 * sizeof(int))) != hipSuccess) return e;
  return hipSuccess;
}

void lu_free(DenseLu &f) {
  if (f.A) (void)hipFree(f.A);
  if (f.piv) (void)hipFree(f.piv);
  if (f.perm) (void)hipFree(f.perm);
  if (f.work) (void)hipFree(f.work);
  if (f.PT) (void)hipFree(f.PT);
  if (f.flags) (void)hipFree(f.flags);
  f = DenseLu();
}

// factorise f.A in place; returns 0 ok / 1 singular / -1 HIP error (*err)
int lu_factor(DenseLu &f, hipError_t *err) {
  const int N = f.N;
  hipStream_t s = f.stream;
  hipError_t e = hipMemsetAsync(f.flags, 0, 4 * sizeof(int), s);
  for (int c0 = 0; c0 < N && e == hipSuccess; c0 += LU_PB) {
    const int pb = std::min(LU_PB, N - c0);
    const int gr = (N - c0 + 255) / 256;
    hipLaunchKernelGGL(k_lu_panel_load, dim3(gr), dim3(256), 0, s, f.A, f.ld, N, c0, pb, f.PT, f.ldp);
    // rows in registers while the panel is at most 5 x 1024 rows tall (PGF_LU_PANEL=1: always the
    // streaming kernel)
    static const bool regs = !(getenv("PGF_LU_PANEL") && atoi(getenv("PGF_LU_PANEL")) == 1);
    const int R = (N - c0 + 1023) / 1024;
    if (regs && R <= 5) {
      switch (R) {
        case 1: hipLaunchKernelGGL(k_lu_panel_reg<1>, dim3(1), dim3(1024), 0, s, f.PT, f.ldp, N, c0, pb, f.piv, f.flags); break;
        case 2: hipLaunchKernelGGL(k_lu_panel_reg<2>, dim3(1), dim3(1024), 0, s, f.PT, f.ldp, N, c0, pb, f.piv, f.flags); break;
        case 3: hipLaunchKernelGGL(k_lu_panel_reg<3>, dim3(1), dim3(1024), 0, s, f.PT, f.ldp, N, c0, pb, f.piv, f.flags); break;
        case 4: hipLaunchKernelGGL(k_lu_panel_reg<4>, dim3(1), dim3(1024), 0, s, f.PT, f.ldp, N, c0, pb, f.piv, f.flags); break;
        default: hipLaunchKernelGGL(k_lu_panel_reg<5>, dim3(1), dim3(1024), 0, s, f.PT, f.ldp, N, c0, pb, f.piv, f.flags); break;
      }
    } else {
      hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(1024), 0, s, f.PT, f.ldp, N, c0, pb, f.piv, f.flags);
    }
    hipLaunchKernelGGL(k_lu_panel_store, dim3(gr), dim3(256), 0, s, f.A, f.ld, N, c0, pb, f.PT, f.ldp);
    if (N > pb)
      hipLaunchKernelGGL(k_lu_swap_rows, dim3((N - pb + 255) / 256), dim3(256), 0, s, f.A, f.ld, N, c0,
                         pb, f.piv);
    const int c1 = c0 + pb;
    if (c1 < N) {
      hipLaunchKernelGGL(k_lu_trsm, dim3((N - c1 + 255) / 256), dim3(256), 0, s, f.A, f.ld, N, c0, pb);
      const int t = (N - c1 + 63) / 64;
      hipLaunchKernelGGL(k_lu_update, dim3(t, t), dim3(256), 0, s, f.A, f.ld, N, c0, pb, c1);
    }
    e = hipGetLastError();
  }
  std::vector<int> piv((size_t)std::max(N, 1)), perm((size_t)std::max(N, 1));
  int h_flags[4] = {0, 0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h_flags, f.flags, sizeof(h_flags), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && N)
    e = hipMemcpyAsync(piv.data(), f.piv, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (err) *err = e;
  if (e != hipSuccess) return -1;
  // the sequence of row interchanges as one permutation: row i of P A is row perm[i] of A
  for (int i = 0; i < N; ++i) perm[i] = i;
  for (int i = 0; i < N; ++i) std::swap(perm[i], perm[std::min(std::max(piv[i], 0), N - 1)]);
  if (N) {
    e = hipMemcpyAsync(f.perm, perm.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (err) *err = e;
    if (e != hipSuccess) return -1;
  }
  f.factored = (h_flags[0] == 0);
  return h_flags[0] ? 1 : 0;
}

template <bool LOWER, bool UNIT, bool TRANS>
static void tri_solve(DenseLu &f, double *x) {
  const int N = f.N;
  hipStream_t s = f.stream;
  const int nblk = (N + 63) / 64;
  for (int bi = 0; bi < nblk; ++bi) {
    const int b = LOWER ? bi : nblk - 1 - bi;
    const int b0 = b * 64;
    hipLaunchKernelGGL((k_tri_diag<LOWER, UNIT, TRANS>), dim3(1), dim3(64), 0, s, f.A, f.ld, N, b0, x);
    const int open = LOWER ? N - std::min(N, b0 + 64) : b0;
    if (open > 0)
      hipLaunchKernelGGL((k_tri_gemv<LOWER, TRANS>), dim3((open + 255) / 256), dim3(256), 0, s, f.A,
                         f.ld, N, b0, x);
  }
}

// sol <- A^-1 rhs (trans == 0) or A^-T rhs (device vectors of length N; rhs is preserved)
hipError_t lu_solve_async(DenseLu &f, const double *rhs, double *sol, int trans) {
  const int N = f.N;
  if (N == 0) return hipSuccess;
  hipStream_t s = f.stream;
  const dim3 g((N + 255) / 256), b(256);
  if (!trans) {
    // P A = L U:  A x = b  <=>  L U x = P b
    hipLaunchKernelGGL(k_permute, g, b, 0, s, rhs, f.perm, sol, N, 0);
    tri_solve<true, true, false>(f, sol);
    tri_solve<false, false, false>(f, sol);
  } else {
    // A^T = U^T L^T P:  U^T z = b (lower),  L^T w = z (upper, unit),  x = P^T w
    hipError_t e = hipMemcpyAsync(f.work, rhs, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    tri_solve<true, false, true>(f, f.work);
    tri_solve<false, true, true>(f, f.work);
    hipLaunchKernelGGL(k_permute, g, b, 0, s, f.work, f.perm, sol, N, 1);
  }
  return hipGetLastError();
}
