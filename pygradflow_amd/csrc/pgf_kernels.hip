// Elementwise, compaction, gather-assembly and matrix-vector kernels of the Newton step.
// Compiled with -ffp-contract=off: the active-set mask must be bit-exact with the
// reference's numpy expressions, so no multiply-add may be fused here unless written
// as an explicit fma() (only inside dot products, whose summation order differs from
// scipy's anyway and is covered by the 1e-10 iterate tolerance).
#include "pgf_kernels.h"

#include <algorithm>

#include "pgf_internal.h"

#define ACTIVE_EPS 1e-8  // reference implicit_func.py:44

// ---------------------------------------------------------------- bounds (a2)
__device__ __forceinline__ void b_scale_bounds(int n, double lamb, const double *__restrict__ lb,
                               const double *__restrict__ ub, double *__restrict__ slb,
                               double *__restrict__ sub) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    slb[i] = lamb * lb[i];
    sub[i] = lamb * ub[i];
  }
}

// ---------------------------------------------------------------- p and mask (a3, a4)
// tau form: (f_x * x + f_x0 * x_hat) - f_d * g, evaluated left to right as numpy does
// (implicit_func.py:237-244); plain form: lamb * x_hat - g (:246).
__device__ __forceinline__ void b_active_set(int n, int use_tau, double lamb, double f_x,
    double f_x0, double f_d,
                             const double *__restrict__ xhat, const double *__restrict__ x,
                             const double *__restrict__ g, const double *__restrict__ slb,
                             const double *__restrict__ sub, uint8_t *__restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p;
  if (use_tau) {
    const double a = f_x * x[i];
    const double b = f_x0 * xhat[i];
    const double c = f_d * g[i];
    p = (a + b) - c;
  } else {
    p = lamb * xhat[i] - g[i];
  }
  const double lo = slb[i] - ACTIVE_EPS;
  const double hi = sub[i] + ACTIVE_EPS;
  mask[i] = (p < lo || p > hi) ? 1 : 0;
}

// ---------------------------------------------------------------- compaction (k2)
// One workgroup of 1024 lanes walks the mask in chunks; wavefront ballots + a scan of the
// 16 wave totals give stable (ascending) index lists of the inactive and active sets.
// pos[j] = rank of j inside its own list.
__device__ __forceinline__ void b_compact(int n, const uint8_t *__restrict__ mask,
                                                  int *__restrict__ idxI, int *__restrict__ idxA,
                                                  int *__restrict__ pos, int *__restrict__ counts) {
  __shared__ int wtot[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int baseI = 0;
  for (int start = 0; start < n; start += 1024) {
    const int j = start + tid;
    const bool valid = j < n;
    const bool inact = valid && (mask[j] == 0);
    const unsigned long long bal = __ballot(inact);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      const int t = wtot[w];
      if (w < wave) woff += t;
      tot += t;
    }
    if (valid) {
      if (inact) {
        const int r = baseI + woff + before;
        idxI[r] = j;
        pos[j] = r;
      } else {
        const int r = j - (baseI + woff + before);  // active rank = j - #inactive before j
        idxA[r] = j;
        pos[j] = r;
      }
    }
    baseI += tot;
    __syncthreads();
  }
  if (tid == 0) {
    counts[0] = baseI;
    counts[1] = n - baseI;
    base_s = baseI;
  }
  (void)base_s;
}

// ---------------------------------------------------------------- residual (a5, a6)
// F = [lamb x - P(p) ; -(lamb y - (lamb y_hat + c))], P clips only masked entries
// (np.clip == min(max(p, lo), hi)).  Also emits b0full = mask ? dt * F_x : 0 (a8).
__device__ __forceinline__ void b_residual(int n, int m, double lamb, double dt,
    const double *__restrict__ xhat,
                           const double *__restrict__ yhat, const double *__restrict__ x,
                           const double *__restrict__ y, const double *__restrict__ g,
                           const double *__restrict__ c, const double *__restrict__ slb,
                           const double *__restrict__ sub, const uint8_t *__restrict__ mask,
                           double *__restrict__ F, double *__restrict__ b0full) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double p = lamb * xhat[i] - g[i];
    const bool act = mask[i] != 0;
    if (act) p = fmin(fmax(p, slb[i]), sub[i]);
    const double f = lamb * x[i] - p;
    F[i] = f;
    if (b0full) b0full[i] = act ? dt * f : 0.0;
  } else if (i < n + m) {
    const int r = i - n;
    const double t = lamb * yhat[r] + c[r];
    F[i] = -(lamb * y[r] - t);
  }
}

// ---------------------------------------------------------------- dot-product rows
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

// Fixed-order dot product of one matrix row with a vector by one wavefront.
__device__ __forceinline__ double row_dot(const double *__restrict__ row,
                                          const double *__restrict__ v, int cols, int lane) {
  double acc = 0.0;
  int j = lane * 2;
  if ((((uintptr_t)row) & 15) == 0) {
    for (; j + 1 < cols; j += 128) {
      const double2 a = *reinterpret_cast<const double2 *>(row + j);
      const double2 b = *reinterpret_cast<const double2 *>(v + j);
      acc = fma(a.x, b.x, acc);
      acc = fma(a.y, b.y, acc);
    }
    if (j < cols) acc = fma(row[j], v[j], acc);
  } else {
    for (j = lane; j < cols; j += 64) acc = fma(row[j], v[j], acc);
  }
  return wave_sum(acc);
}

// reduced right-hand side (a9 part, a11):
//   i <  nI : rhs[i] = F[I[i]]            - H[I[i], :] . b0full
//   i >= nI : rhs[i] = fact * F[n + r]    - J[r, :]    . b0full      (r = i - nI)
// b0full is zero on the inactive set, so the full-row dot equals the reference's
// H_lamb[I, A] b0 / J[:, A] b0 (symmetric_step_solver.py:87-91); skipped when |A| = 0.
// The H part reads the ACTIVE rows instead of the inactive ones (H is symmetric:
// H[I, A] b0 = (H[A, I])^T b0): |A| n instead of |I| n entries -- b_active_rows_partial leaves
// partial[p][j] = sum over the active rows a of chunk p of H[a][j] b0[a], fixed chunks, and the
// rows i < nI pick their column out of it.  (One row-dot per inactive row read 90 % of every
// instance's H in a batch with 10 % active variables: 53 us of a 1.3 ms batched step.)
__device__ __forceinline__ void b_active_rows_partial(int n, int nA, const int *__restrict__ idxA,
                                                      const double *__restrict__ H, int64_t ldh,
                                                      const double *__restrict__ b0full, int nparts,
                                                      double *__restrict__ partial) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n || nA <= 0) return;
  const int chunk = (nA + nparts - 1) / nparts;
  const int a0 = blockIdx.y * chunk, a1 = min(nA, a0 + chunk);
  double acc = 0.0;
  for (int a = a0; a < a1; ++a) {
    const int ga = idxA[a];
    acc = fma(H[(int64_t)ga * ldh + j], b0full[ga], acc);
  }
  partial[(int64_t)blockIdx.y * n + j] = acc;
}
__device__ __forceinline__ void b_reduced_rhs(int n, int m, int nI, int nA, double fact,
                                                     const double *__restrict__ F,
                                                     const int *__restrict__ idxI,
                                                     const double *__restrict__ J, int64_t ldj,
                                                     const double *__restrict__ b0full,
                                                     const double *__restrict__ partial, int nparts,
                                                     double *__restrict__ rhs) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nI + m) return;
  if (i < nI) {
    const int gi = idxI[i];
    double corr = 0.0;
    if (nA > 0) {  // lane p takes chunk p (nparts <= 64), summed over the wavefront in a fixed order
      corr = lane < nparts ? partial[(int64_t)lane * n + gi] : 0.0;
      corr = wave_sum(corr);
    }
    if (lane == 0) rhs[i] = F[gi] - corr;
    return;
  }
  const int r = i - nI;
  double corr = 0.0;
  if (nA > 0) corr = row_dot(J + (int64_t)r * ldj, b0full, n, lane);
  if (lane == 0) rhs[i] = fact * F[n + r] - corr;
}

// ---------------------------------------------------------------- K assembly (a10, a12)
// Lower triangle of K = [[H[I,I] + lamb I, .],[J[:,I], -delta I]] gathered from the
// device-resident H, J.  A workgroup covers ASM_ROWS rows x 256 columns (lanes run along
// columns: coalesced stores, loads coalesced whenever I is contiguous); the column's index
// in H / J is looked up once per lane.  One row per workgroup was dispatch-bound in the
// batched step (1.6 M workgroups).
#define ASM_ROWS 8
template <int ROWS = ASM_ROWS>
__device__ __forceinline__ void b_assemble_kkt(double *__restrict__ K, int64_t ldk,
                                               const double *__restrict__ H, int64_t ldh,
                                               const double *__restrict__ J, int64_t ldj,
                                               const int *__restrict__ idxI, int nI, int m,
                                               double lamb, double delta) {
  const int N = nI + m;
  const int i0 = blockIdx.y * ROWS;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if ((int)blockIdx.x * 256 >= N || (int)blockIdx.x * 256 > i0 + ROWS - 1) return;  // (whole workgroups only:
                                                             // every lane may be asked for its row index)
  const int gj = (j < nI) ? idxI[j] : 0;
  // The rows' indices in H come from ONE vector load (lane u holds row i0 + u's) and reach the
  // scalar unit through v_readlane: looked up row by row (`idxI[i]', a scalar load and its wait in
  // front of every row's global load) the rows ran one after the other -- 0.9 ms for a batch of
  // 256 x 1024^2, 2.4 TB/s.  Eight rows at a time: all loads of the group, then the stores.
  static_assert(ROWS <= 64, "one lane per row of the workgroup");
  const int lane = threadIdx.x & 63;
  const int rowidx = (lane < ROWS && i0 + lane < nI) ? idxI[i0 + lane] : 0;
  constexpr int G = ROWS < 8 ? ROWS : 8;
#pragma unroll
  for (int r0 = 0; r0 < ROWS; r0 += G) {
    double v[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int i = i0 + r0 + u;
      const int gi = __builtin_amdgcn_readlane(rowidx, r0 + u);
      v[u] = 0.0;
      if (i < N && j <= i && j < N) {
        if (i < nI) {
          v[u] = H[(int64_t)gi * ldh + gj];
        } else if (j < nI) {
          v[u] = J[(int64_t)(i - nI) * ldj + gj];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int i = i0 + r0 + u;
      if (i < N && j <= i && j < N) {
        double w = v[u];
        if (i == j) w = (i < nI) ? w + lamb : -delta;
        K[(int64_t)i * ldk + j] = w;
      }
    }
  }
}

__device__ __forceinline__ void b_copy(double *__restrict__ dst, const double *__restrict__ src,
    int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__device__ __forceinline__ void b_copy_u8(uint8_t *__restrict__ dst,
    const uint8_t *__restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// count positions where two masks differ (ActiveSet policy, newton.py:210)
__device__ __forceinline__ void b_mask_diff(int n, const uint8_t *__restrict__ a,
    const uint8_t *__restrict__ b,
                            int *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool d = (i < n) && (a[i] != b[i]);
  const unsigned long long bal = __ballot(d);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(out, __popcll(bal));
}

// ---------------------------------------------------------------- step update (a9, a15, a16)
// dx[I] = s[:nI], dx[A] = b0;  dy = fact * (s[nI:] - rho * b2)
// xn = clip(x - dx, lb, ub) with dx rewritten where clipped; yn = y - dy
// per-block partial sums of dx^2 + dy^2 in fixed order -> red[blockIdx.x]
__device__ __forceinline__ void b_step_update(
    int n, int m, int nI, double fact, double rho, const double *__restrict__ x,
    const double *__restrict__ y, const double *__restrict__ lb, const double *__restrict__ ub,
    const uint8_t *__restrict__ mask, const int *__restrict__ pos,
    const double *__restrict__ b0full, const double *__restrict__ F,
    const double *__restrict__ sol, double *__restrict__ dx, double *__restrict__ dy,
    double *__restrict__ xn, double *__restrict__ yn, double *__restrict__ red) {
  __shared__ double part[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  if (i < n) {
    double d = mask[i] ? b0full[i] : sol[pos[i]];
    const double xi = x[i];
    double v = xi - d;
    const double lo = lb[i], hi = ub[i];
    if (v < lo) {
      v = lo;
      d = xi - lo;
    }
    if (v > hi) {
      v = hi;
      d = xi - hi;
    }
    dx[i] = d;
    xn[i] = v;
    sq = d * d;
  } else if (i < n + m) {
    const int r = i - n;
    const double t = rho * F[n + r];
    const double d = fact * (sol[nI + r] - t);
    dy[r] = d;
    yn[r] = y[r] - d;
    sq = d * d;
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) red[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out[0] = sqrt(sum red[0..cnt)) (or the plain sum when take_sqrt == 0), fixed order
__device__ __forceinline__ void b_final_reduce(const double *__restrict__ red, int cnt,
                                                      double *__restrict__ out, int take_sqrt) {
  __shared__ double part[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < cnt; i += 256) s += red[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (part[0] + part[1]) + (part[2] + part[3]);
    out[0] = take_sqrt ? sqrt(t) : t;
  }
}

// ---------------------------------------------------------------- linear-quadratic evaluation
// out[r] = M[r, :] . v + sgn * add[r]     (c = A x - b ;  g = Q x + (q + A'(rho c + y)))
__device__ __forceinline__ void b_gemv_rows(int rows, int cols,
                                                   const double *__restrict__ M, int64_t ld,
                                                   const double *__restrict__ v,
                                                   const double *__restrict__ add, double sgn,
                                                   double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const double d = row_dot(M + (int64_t)r * ld, v, cols, lane);
  if (lane == 0) out[r] = d + sgn * add[r];
}

// partial[rb][j] = sum_{r in chunk rb} M[r][j] * w[r]   (transposed product, fixed order)
__device__ __forceinline__ void b_gemvT_partial(int rows, int cols,
                                                       const double *__restrict__ M, int64_t ld,
                                                       const double *__restrict__ w, int chunk,
                                                       double *__restrict__ partial) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  const int r0 = blockIdx.y * chunk;
  const int r1 = min(rows, r0 + chunk);
  double acc = 0.0;
  for (int r = r0; r < r1; ++r) acc = fma(M[(int64_t)r * ld + j], w[r], acc);
  partial[(int64_t)blockIdx.y * cols + j] = acc;
}

// out[j] = base[j] + sum_rb partial[rb][j]
__device__ __forceinline__ void b_sum_partials(int cols, int nparts,
    const double *__restrict__ partial,
                               const double *__restrict__ base, double *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += partial[(int64_t)p * cols + j];
  out[j] = base[j] + s;
}

// w = rho * c + y
__device__ __forceinline__ void b_mult_vec(int m, double rho, const double *__restrict__ c,
                           const double *__restrict__ y, double *__restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) w[i] = rho * c[i] + y[i];
}

// squared entries of the UNSCALED residual with its own mask (ImplicitFunc.value_at,
// implicit_func.py:131-161): per-block partial sums -> red
__device__ __forceinline__ void b_unscaled_res_sq(
    int n, int m, double dt, const double *__restrict__ xhat, const double *__restrict__ yhat,
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ g,
    const double *__restrict__ c, const double *__restrict__ lb, const double *__restrict__ ub,
    double *__restrict__ red) {
  __shared__ double part[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  if (i < n) {
    double p = xhat[i] - dt * g[i];
    const bool act = (p < lb[i] - ACTIVE_EPS) || (p > ub[i] + ACTIVE_EPS);
    if (act) p = fmin(fmax(p, lb[i]), ub[i]);
    const double f = x[i] - p;
    sq = f * f;
  } else if (i < n + m) {
    const int r = i - n;
    const double f = y[r] - (yhat[r] + dt * c[r]);
    sq = f * f;
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) red[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// ---------------------------------------------------------------- termination measures
// Iterate-level residuals the outer loop tests between Newton calls (SURVEY.md 8f rank 4;
// reference iterate.py:136-181, active_set.py:4-29), per-block maxima of
//   [0] | r + d |   r = obj_grad + J'y (given), d = bounds_dual        -> stat_res
//   [1] | c |                                                         -> cons_violation
//   [2] max(lb - x, 0), max(x - ub, 0)                                -> bound_violation
//   [3] | y |                                                         -> DualNormUpdate
// written to red[4 * blockIdx.x + k].  np.maximum / np.linalg.norm(inf) propagate NaN;
// fmax does not, so a NaN entry is forwarded explicitly.
__device__ __forceinline__ double nan_max(double a, double b) {
  return (a != a || b != b) ? (a + b) : fmax(a, b);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = nan_max(v, __shfl_down(v, off));
  return v;
}

__device__ __forceinline__ void b_measures(int n, int m, double active_tol,
                                           const double *__restrict__ x,
                                           const double *__restrict__ y,
                                           const double *__restrict__ r,
                                           const double *__restrict__ c,
                                           const double *__restrict__ lb,
                                           const double *__restrict__ ub,
                                           double *__restrict__ red) {
  __shared__ double part[4][4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  if (i < n) {
    const double xi = x[i], lo = lb[i], hi = ub[i], ri = r[i];
    const bool al = fabs(xi - lo) <= active_tol;
    const bool au = fabs(hi - xi) <= active_tol;
    const double nr = -ri;
    double d = 0.0;
    if (al && au)
      d = nr;
    else if (au)
      d = (nr != nr) ? nr : fmax(nr, 0.0);
    else if (al)
      d = (nr != nr) ? nr : fmin(nr, 0.0);
    v0 = fabs(ri + d);
    const double bl = lo - xi, bu = xi - hi;
    v2 = nan_max((bl != bl) ? bl : fmax(bl, 0.0), (bu != bu) ? bu : fmax(bu, 0.0));
  } else if (i < n + m) {
    v1 = fabs(c[i - n]);
    v3 = fabs(y[i - n]);
  }
  v0 = wave_max(v0);
  v1 = wave_max(v1);
  v2 = wave_max(v2);
  v3 = wave_max(v3);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    part[w][0] = v0;
    part[w][1] = v1;
    part[w][2] = v2;
    part[w][3] = v3;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    red[4 * blockIdx.x + k] =
        nan_max(nan_max(part[0][k], part[1][k]), nan_max(part[2][k], part[3][k]));
  }
}

// out[k] = max over blocks of red[4 b + k]
__device__ __forceinline__ void b_measures_final(const double *__restrict__ red, int nb,
                                                 double *__restrict__ out) {
  __shared__ double part[4][4];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = nan_max(v[k], red[4 * b + k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_max(v[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) part[w][k] = v[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    out[k] = nan_max(nan_max(part[0][k], part[1][k]), nan_max(part[2][k], part[3][k]));
  }
}

// ---------------------------------------------------------------- single-instance kernels
__global__ void k_scale_bounds(int n, double lamb, const double *__restrict__ lb,
    const double *__restrict__ ub, double *__restrict__ slb, double *__restrict__ sub) {
  b_scale_bounds(n, lamb, lb, ub, slb, sub);
}

__global__ void k_active_set(int n, int use_tau, double lamb, double f_x, double f_x0, double f_d,
    const double *__restrict__ xhat, const double *__restrict__ x, const double *__restrict__ g,
    const double *__restrict__ slb, const double *__restrict__ sub, uint8_t *__restrict__ mask) {
  b_active_set(n, use_tau, lamb, f_x, f_x0, f_d, xhat, x, g, slb, sub, mask);
}

__global__ __launch_bounds__(1024) void k_compact(int n, const uint8_t *__restrict__ mask,
    int *__restrict__ idxI, int *__restrict__ idxA, int *__restrict__ pos,
    int *__restrict__ counts) {
  b_compact(n, mask, idxI, idxA, pos, counts);
}

__global__ void k_residual(int n, int m, double lamb, double dt, const double *__restrict__ xhat,
    const double *__restrict__ yhat, const double *__restrict__ x, const double *__restrict__ y,
    const double *__restrict__ g, const double *__restrict__ c, const double *__restrict__ slb,
    const double *__restrict__ sub, const uint8_t *__restrict__ mask, double *__restrict__ F,
    double *__restrict__ b0full) {
  b_residual(n, m, lamb, dt, xhat, yhat, x, y, g, c, slb, sub, mask, F, b0full);
}

__global__ __launch_bounds__(256) void k_active_rows_partial(int n, int nA, const int *__restrict__ idxA,
    const double *__restrict__ H, int64_t ldh, const double *__restrict__ b0full, int nparts,
    double *__restrict__ partial) {
  b_active_rows_partial(n, nA, idxA, H, ldh, b0full, nparts, partial);
}
__global__ __launch_bounds__(256) void k_reduced_rhs(int n, int m, int nI, int nA, double fact,
    const double *__restrict__ F, const int *__restrict__ idxI, const double *__restrict__ J,
    int64_t ldj, const double *__restrict__ b0full, const double *__restrict__ partial, int nparts,
    double *__restrict__ rhs) {
  b_reduced_rhs(n, m, nI, nA, fact, F, idxI, J, ldj, b0full, partial, nparts, rhs);
}

__global__ __launch_bounds__(256) void k_assemble_kkt(double *__restrict__ K, int64_t ldk,
    const double *__restrict__ H, int64_t ldh, const double *__restrict__ J, int64_t ldj,
    const int *__restrict__ idxI, int nI, int m, double lamb, double delta) {
  b_assemble_kkt(K, ldk, H, ldh, J, ldj, idxI, nI, m, lamb, delta);
}

__global__ void k_copy(double *__restrict__ dst, const double *__restrict__ src, int n) {
  b_copy(dst, src, n);
}

__global__ void k_copy_u8(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int n) {
  b_copy_u8(dst, src, n);
}

__global__ void k_mask_diff(int n, const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
    int *__restrict__ out) {
  b_mask_diff(n, a, b, out);
}

__global__ __launch_bounds__(256) void k_step_update(int n, int m, int nI, double fact,
    double rho, const double *__restrict__ x, const double *__restrict__ y,
    const double *__restrict__ lb, const double *__restrict__ ub,
    const uint8_t *__restrict__ mask, const int *__restrict__ pos,
    const double *__restrict__ b0full, const double *__restrict__ F,
    const double *__restrict__ sol, double *__restrict__ dx, double *__restrict__ dy,
    double *__restrict__ xn, double *__restrict__ yn, double *__restrict__ red) {
  b_step_update(n, m, nI, fact, rho, x, y, lb, ub, mask, pos, b0full, F, sol, dx, dy, xn, yn, red);
}

__global__ __launch_bounds__(256) void k_final_reduce(const double *__restrict__ red, int cnt,
    double *__restrict__ out, int take_sqrt) {
  b_final_reduce(red, cnt, out, take_sqrt);
}

__global__ __launch_bounds__(256) void k_gemv_rows(int rows, int cols,
    const double *__restrict__ M, int64_t ld, const double *__restrict__ v,
    const double *__restrict__ add, double sgn, double *__restrict__ out) {
  b_gemv_rows(rows, cols, M, ld, v, add, sgn, out);
}

__global__ __launch_bounds__(256) void k_gemvT_partial(int rows, int cols,
    const double *__restrict__ M, int64_t ld, const double *__restrict__ w, int chunk,
    double *__restrict__ partial) {
  b_gemvT_partial(rows, cols, M, ld, w, chunk, partial);
}

__global__ void k_sum_partials(int cols, int nparts, const double *__restrict__ partial,
    const double *__restrict__ base, double *__restrict__ out) {
  b_sum_partials(cols, nparts, partial, base, out);
}

__global__ void k_mult_vec(int m, double rho, const double *__restrict__ c,
    const double *__restrict__ y, double *__restrict__ w) {
  b_mult_vec(m, rho, c, y, w);
}

__global__ __launch_bounds__(256) void k_unscaled_res_sq(int n, int m, double dt,
    const double *__restrict__ xhat, const double *__restrict__ yhat,
    const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ g,
    const double *__restrict__ c, const double *__restrict__ lb, const double *__restrict__ ub,
    double *__restrict__ red) {
  b_unscaled_res_sq(n, m, dt, xhat, yhat, x, y, g, c, lb, ub, red);
}

// ---------------------------------------------------------------- launch wrappers
static inline dim3 g1(int n, int b = 256) { return dim3((n + b - 1) / b); }

void launch_scale_bounds(hipStream_t s, int n, double lamb, const double *lb, const double *ub,
                         double *slb, double *sub) {
  if (n) hipLaunchKernelGGL(k_scale_bounds, g1(n), dim3(256), 0, s, n, lamb, lb, ub, slb, sub);
}

void launch_active_set(hipStream_t s, int n, int use_tau, double lamb, double f_x, double f_x0,
                       double f_d, const double *xhat, const double *x, const double *g,
                       const double *slb, const double *sub, uint8_t *mask) {
  if (n)
    hipLaunchKernelGGL(k_active_set, g1(n), dim3(256), 0, s, n, use_tau, lamb, f_x, f_x0, f_d,
                       xhat, x, g, slb, sub, mask);
}

void launch_compact(hipStream_t s, int n, const uint8_t *mask, int *idxI, int *idxA, int *pos,
                    int *counts) {
  hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, n, mask, idxI, idxA, pos, counts);
}

void launch_residual(hipStream_t s, int n, int m, double lamb, double dt, const double *xhat,
                     const double *yhat, const double *x, const double *y, const double *g,
                     const double *c, const double *slb, const double *sub, const uint8_t *mask,
                     double *F, double *b0full) {
  if (n + m)
    hipLaunchKernelGGL(k_residual, g1(n + m), dim3(256), 0, s, n, m, lamb, dt, xhat, yhat, x, y,
                       g, c, slb, sub, mask, F, b0full);
}

void launch_reduced_rhs(hipStream_t s, int n, int m, int nI, int nA, double fact, const double *F,
                        const int *idxI, const int *idxA, const double *H, int64_t ldh, const double *J,
                        int64_t ldj, const double *b0full, double *partial, int nparts, double *rhs) {
  const int N = nI + m;
  if (!N) return;
  if (nA > 0 && nI > 0)
    hipLaunchKernelGGL(k_active_rows_partial, dim3((n + 255) / 256, nparts), dim3(256), 0, s, n, nA, idxA,
                       H, ldh, b0full, nparts, partial);
  hipLaunchKernelGGL(k_reduced_rhs, dim3((N + 3) / 4), dim3(256), 0, s, n, m, nI, nA, fact, F, idxI, J,
                     ldj, b0full, partial, nparts, rhs);
}

void launch_assemble_kkt(hipStream_t s, double *K, int64_t ldk, const double *H, int64_t ldh,
                         const double *J, int64_t ldj, const int *idxI, int nI, int m,
                         double lamb, double delta) {
  const int N = nI + m;
  if (N)
    hipLaunchKernelGGL(k_assemble_kkt, dim3((N + 255) / 256, (N + ASM_ROWS - 1) / ASM_ROWS),
                       dim3(256), 0, s, K, ldk, H, ldh, J, ldj, idxI, nI, m, lamb, delta);
}

void launch_copy(hipStream_t s, double *dst, const double *src, int n) {
  if (n) hipLaunchKernelGGL(k_copy, g1(n), dim3(256), 0, s, dst, src, n);
}

void launch_copy_u8(hipStream_t s, uint8_t *dst, const uint8_t *src, int n) {
  if (n) hipLaunchKernelGGL(k_copy_u8, g1(n), dim3(256), 0, s, dst, src, n);
}

void launch_mask_diff(hipStream_t s, int n, const uint8_t *a, const uint8_t *b, int *out) {
  if (n) hipLaunchKernelGGL(k_mask_diff, g1(n), dim3(256), 0, s, n, a, b, out);
}

int step_update_blocks(int n, int m) { return (n + m + 255) / 256; }

void launch_step_update(hipStream_t s, int n, int m, int nI, double fact, double rho,
                        const double *x, const double *y, const double *lb, const double *ub,
                        const uint8_t *mask, const int *pos, const double *b0full,
                        const double *F, const double *sol, double *dx, double *dy, double *xn,
                        double *yn, double *red, double *diff_out) {
  const int nb = step_update_blocks(n, m);
  if (nb)
    hipLaunchKernelGGL(k_step_update, dim3(nb), dim3(256), 0, s, n, m, nI, fact, rho, x, y, lb, ub,
                       mask, pos, b0full, F, sol, dx, dy, xn, yn, red);
  hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(256), 0, s, red, nb, diff_out, 1);
}

void launch_gemv_rows(hipStream_t s, int rows, int cols, const double *M, int64_t ld,
                      const double *v, const double *add, double sgn, double *out) {
  if (rows)
    hipLaunchKernelGGL(k_gemv_rows, dim3((rows + 3) / 4), dim3(256), 0, s, rows, cols, M, ld, v,
                       add, sgn, out);
}

void launch_gemvT(hipStream_t s, int rows, int cols, const double *M, int64_t ld,
                  const double *w, const double *base, double *partial, int nparts, double *out) {
  // out = base + M' w, in nparts fixed row chunks
  if (!cols) return;
  if (rows == 0) {
    launch_copy(s, out, base, cols);
    return;
  }
  const int chunk = (rows + nparts - 1) / nparts;
  const int used = (rows + chunk - 1) / chunk;
  hipLaunchKernelGGL(k_gemvT_partial, dim3((cols + 255) / 256, used), dim3(256), 0, s, rows, cols,
                     M, ld, w, chunk, partial);
  hipLaunchKernelGGL(k_sum_partials, g1(cols), dim3(256), 0, s, cols, used, partial, base, out);
}

void launch_mult_vec(hipStream_t s, int m, double rho, const double *c, const double *y,
                     double *w) {
  if (m) hipLaunchKernelGGL(k_mult_vec, g1(m), dim3(256), 0, s, m, rho, c, y, w);
}

void launch_unscaled_res_norm(hipStream_t s, int n, int m, double dt, const double *xhat,
                              const double *yhat, const double *x, const double *y,
                              const double *g, const double *c, const double *lb,
                              const double *ub, double *red, double *out) {
  const int nb = (n + m + 255) / 256;
  if (nb)
    hipLaunchKernelGGL(k_unscaled_res_sq, dim3(nb), dim3(256), 0, s, n, m, dt, xhat, yhat, x, y, g,
                       c, lb, ub, red);
  hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(256), 0, s, red, nb, out, 1);
}

void launch_final_reduce(hipStream_t s, const double *red, int cnt, double *out, int take_sqrt) {
  hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(256), 0, s, red, cnt, out, take_sqrt);
}

__global__ __launch_bounds__(256) void k_measures(int n, int m, double active_tol,
                                                  const double *__restrict__ x,
                                                  const double *__restrict__ y,
                                                  const double *__restrict__ r,
                                                  const double *__restrict__ c,
                                                  const double *__restrict__ lb,
                                                  const double *__restrict__ ub,
                                                  double *__restrict__ red) {
  b_measures(n, m, active_tol, x, y, r, c, lb, ub, red);
}

__global__ __launch_bounds__(256) void k_measures_final(const double *__restrict__ red, int nb,
                                                        double *__restrict__ out) {
  b_measures_final(red, nb, out);
}

void launch_measures(hipStream_t s, int n, int m, double active_tol, const double *x,
                     const double *y, const double *r, const double *c, const double *lb,
                     const double *ub, double *red, double *out) {
  const int nb = (n + m + 255) / 256;
  if (nb)
    hipLaunchKernelGGL(k_measures, dim3(nb), dim3(256), 0, s, n, m, active_tol, x, y, r, c, lb, ub,
                       red);
  hipLaunchKernelGGL(k_measures_final, dim3(1), dim3(256), 0, s, red, nb, out);
}

// ---------------------------------------------------------------- CSR -> dense (plugin path)
// dst (rows x ld, zeroed beforehand) += CSR entries; one wavefront per row.  Duplicate entries
// of a row are summed in storage order by lane 0 ... in practice scipy's canonical CSR has
// none, and a plain store per entry would drop them silently, so entries are accumulated.
__global__ __launch_bounds__(256) void k_csr_to_dense(int rows, const int *__restrict__ ptr,
                                                      const int *__restrict__ idx,
                                                      const double *__restrict__ val,
                                                      double *__restrict__ dst, int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int p0 = ptr[r], p1 = ptr[r + 1];
  double *row = dst + (int64_t)r * ld;
  for (int p = p0 + lane; p < p1; p += 64) atomicAdd(row + idx[p], val[p]);
}

void launch_csr_to_dense(hipStream_t s, int rows, const int *ptr, const int *idx, const double *val,
                         double *dst, int64_t ld) {
  if (rows)
    hipLaunchKernelGGL(k_csr_to_dense, dim3((rows + 3) / 4), dim3(256), 0, s, rows, ptr, idx, val,
                       dst, ld);
}

// ---------------------------------------------------------------- residual of the reduced system
// v <- the reduced solution's variable part scattered to the full index set (zero on the
// active set), lv <- lambda v
__global__ void k_expand_sol(int n, int nI, const int *__restrict__ pos, const uint8_t *__restrict__ mask,
                             const double *__restrict__ sol, double lamb, double *__restrict__ v,
                             double *__restrict__ lv) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double val = mask[j] ? 0.0 : sol[pos[j]];
  v[j] = val;
  lv[j] = lamb * val;
}

// r = rhs - K s from u = H v + lambda v + J' s_y (full length n) and wy = J v - delta s_y;
// red[0..2] <- max |r|, max |rhs|, max |s| (bit patterns of non-negative doubles order like
// integers; red zeroed by the caller)
__global__ __launch_bounds__(256) void k_kkt_residual(int N, int nI, const int *__restrict__ idxI,
                                                      const double *rhs,  // may alias r
                                                      const double *__restrict__ sol,
                                                      const double *__restrict__ u,
                                                      const double *__restrict__ wy, double *r,
                                                      unsigned long long *__restrict__ red) {
  __shared__ double sh[3][256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double ar = 0.0, ab = 0.0, as = 0.0;
  if (i < N) {
    const double ks = i < nI ? u[idxI[i]] : wy[i - nI];
    const double ri = rhs[i] - ks;
    r[i] = ri;
    ar = fabs(ri);
    ab = fabs(rhs[i]);
    as = fabs(sol[i]);
    if (ar != ar) ar = __builtin_inf();  // a NaN must not vanish in the max
  }
  sh[0][threadIdx.x] = ar;
  sh[1][threadIdx.x] = ab;
  sh[2][threadIdx.x] = as;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q)
        sh[q][threadIdx.x] = fmax(sh[q][threadIdx.x], sh[q][threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x < 3) atomicMax(&red[threadIdx.x], (unsigned long long)__double_as_longlong(sh[threadIdx.x][0]));
}

__global__ void k_axpy1(int N, const double *__restrict__ d, double *__restrict__ s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) s[i] += d[i];
}

// full symmetric matrix from its lower triangle (LU fallback of the reduced KKT system)
__global__ void k_symmetrize(double *__restrict__ A, int64_t ld, int N) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (i < N && j < N && j > i) A[(int64_t)i * ld + j] = A[(int64_t)j * ld + i];
}

void launch_kkt_residual(hipStream_t s, int n, int m, int nI, double lamb, double delta,
                         const double *H, int64_t ldh, const double *J, int64_t ldj,
                         const int *idxI, const int *pos, const uint8_t *mask, const double *rhs,
                         const double *sol, double *v, double *lv, double *u, double *wy,
                         double *partial, int nparts, double *r, double *red3) {
  const int N = nI + m;
  (void)hipMemsetAsync(red3, 0, 3 * sizeof(double), s);
  if (N == 0) return;
  if (n) hipLaunchKernelGGL(k_expand_sol, g1(n), dim3(256), 0, s, n, nI, pos, mask, sol, lamb, v, lv);
  // u = H v + (lambda v + J' s_y)
  launch_gemvT(s, m, n, J, ldj, sol + nI, lv, partial, nparts, u);
  launch_gemv_rows(s, n, n, H, ldh, v, u, 1.0, lv);  // lv is free again: it receives H v + u
  // wy = J v - delta s_y
  launch_gemv_rows(s, m, n, J, ldj, v, sol + nI, -delta, wy);
  hipLaunchKernelGGL(k_kkt_residual, g1(N), dim3(256), 0, s, N, nI, idxI, rhs, sol, lv, wy, r,
                     reinterpret_cast<unsigned long long *>(red3));
}

// ---- residual check of this step's solve AND g, c at the new point in one pass over H and J ----
// The check applies K from H, J and the mask to the solution; the next step's evaluation applies
// the same matrices to the new point: streamed separately they read H once and J twice EACH
// (336 MB at config 2).  Two right-hand vectors per matrix pass instead: every dot product is the
// row_dot / b_gemvT_partial arithmetic of the separate kernels, in the same order (bit-identical
// g, c and residual), the matrices are read once.
__device__ __forceinline__ void row_dot2(const double *__restrict__ row, const double *__restrict__ v1,
                                         const double *__restrict__ v2, int cols, int lane, double &d1,
                                         double &d2) {
  double a1 = 0.0, a2 = 0.0;
  int j = lane * 2;
  if ((((uintptr_t)row) & 15) == 0) {
    for (; j + 1 < cols; j += 128) {
      const double2 a = *reinterpret_cast<const double2 *>(row + j);
      const double2 b1 = *reinterpret_cast<const double2 *>(v1 + j);
      const double2 b2 = *reinterpret_cast<const double2 *>(v2 + j);
      a1 = fma(a.x, b1.x, a1);
      a1 = fma(a.y, b1.y, a1);
      a2 = fma(a.x, b2.x, a2);
      a2 = fma(a.y, b2.y, a2);
    }
    if (j < cols) {
      a1 = fma(row[j], v1[j], a1);
      a2 = fma(row[j], v2[j], a2);
    }
  } else {
    for (j = lane; j < cols; j += 64) {
      a1 = fma(row[j], v1[j], a1);
      a2 = fma(row[j], v2[j], a2);
    }
  }
  d1 = wave_sum(a1);
  d2 = wave_sum(a2);
}
// out1 = M v1 + sgn1 add1, out2 = M v2 + sgn2 add2 (b_gemv_rows twice, one read of M)
__global__ __launch_bounds__(256) void k_gemv_rows2(int rows, int cols, const double *__restrict__ M, int64_t ld,
                                                    const double *__restrict__ v1, const double *__restrict__ add1,
                                                    double sgn1, double *__restrict__ out1,
                                                    const double *__restrict__ v2, const double *__restrict__ add2,
                                                    double sgn2, double *__restrict__ out2) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  double d1, d2;
  row_dot2(M + (int64_t)r * ld, v1, v2, cols, lane, d1, d2);
  if (lane == 0) {
    out1[r] = d1 + sgn1 * add1[r];
    out2[r] = d2 + sgn2 * add2[r];
  }
}
// partial1[rb][j] = sum_{r in chunk rb} M[r][j] w1[r], partial2 likewise with w2
__global__ __launch_bounds__(256) void k_gemvT_partial2(int rows, int cols, const double *__restrict__ M,
                                                        int64_t ld, const double *__restrict__ w1,
                                                        const double *__restrict__ w2, int chunk,
                                                        double *__restrict__ partial1,
                                                        double *__restrict__ partial2) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  const int r0 = blockIdx.y * chunk;
  const int r1 = min(rows, r0 + chunk);
  double a1 = 0.0, a2 = 0.0;
  for (int r = r0; r < r1; ++r) {
    const double mv = M[(int64_t)r * ld + j];
    a1 = fma(mv, w1[r], a1);
    a2 = fma(mv, w2[r], a2);
  }
  partial1[(int64_t)blockIdx.y * cols + j] = a1;
  partial2[(int64_t)blockIdx.y * cols + j] = a2;
}
__global__ void k_sum_partials2(int cols, int nparts, const double *__restrict__ partial1,
                                const double *__restrict__ base1, double *__restrict__ out1,
                                const double *__restrict__ partial2, const double *__restrict__ base2,
                                double *__restrict__ out2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  double s1 = 0.0, s2 = 0.0;
  for (int p = 0; p < nparts; ++p) {
    s1 += partial1[(int64_t)p * cols + j];
    s2 += partial2[(int64_t)p * cols + j];
  }
  out1[j] = base1[j] + s1;
  out2[j] = base2[j] + s2;
}

// launch_kkt_residual (solution sol of the system rhs; r, red3 as there) and the evaluation
//   c = J xn - b ; w = rho c + yn ; tmpn = q + J' w ; g = H xn + tmpn
// at the point (xn, yn) the step update has just produced.  partial: 2 * nparts * n doubles.
void launch_residual_and_eval(hipStream_t s, int n, int m, int nI, double lamb, double delta, const double *H,
                              int64_t ldh, const double *J, int64_t ldj, const int *idxI, const int *pos,
                              const uint8_t *mask, const double *rhs, const double *sol, double *v, double *lv,
                              double *u, double *wy, double *partial, int nparts, double *r, double *red3,
                              const double *xn, const double *yn, const double *b, const double *q, double rho,
                              double *c, double *w, double *tmpn, double *g) {
  const int N = nI + m;
  (void)hipMemsetAsync(red3, 0, 3 * sizeof(double), s);
  if (n) hipLaunchKernelGGL(k_expand_sol, g1(n), dim3(256), 0, s, n, nI, pos, mask, sol, lamb, v, lv);
  // one pass over J: c = J xn - b, wy = J v - delta s_y
  if (m)
    hipLaunchKernelGGL(k_gemv_rows2, dim3((m + 3) / 4), dim3(256), 0, s, m, n, J, ldj, xn, b, -1.0, c, v,
                       sol + nI, -delta, wy);
  launch_mult_vec(s, m, rho, c, yn, w);
  // one pass over J (transposed): tmpn = q + J' w, u = lambda v + J' s_y
  if (n) {
    if (m == 0) {
      launch_copy(s, tmpn, q, n);
      launch_copy(s, u, lv, n);
    } else {
      const int chunk = (m + nparts - 1) / nparts;
      const int used = (m + chunk - 1) / chunk;
      double *p2 = partial + (size_t)nparts * n;
      hipLaunchKernelGGL(k_gemvT_partial2, dim3((n + 255) / 256, used), dim3(256), 0, s, m, n, J, ldj, w,
                         sol + nI, chunk, partial, p2);
      hipLaunchKernelGGL(k_sum_partials2, g1(n), dim3(256), 0, s, n, used, partial, q, tmpn, p2, lv, u);
    }
    // one pass over H: g = H xn + tmpn, lv <- H v + u
    hipLaunchKernelGGL(k_gemv_rows2, dim3((n + 3) / 4), dim3(256), 0, s, n, n, H, ldh, xn, tmpn, 1.0, g, v, u,
                       1.0, lv);
  }
  if (N)
    hipLaunchKernelGGL(k_kkt_residual, g1(N), dim3(256), 0, s, N, nI, idxI, rhs, sol, lv, wy, r,
                       reinterpret_cast<unsigned long long *>(red3));
}

// ---- matrix norms for the normwise backward error of the residual guard (pgf_api.hip)
// out[0] = max_i sum_j |A_ij| (rows x cols, row-major): one workgroup per row; non-negative
// doubles order like their bit patterns, so atomicMax on the 64-bit words does the reduction
__global__ __launch_bounds__(256) void k_abs_rowsum_max(int cols, const double *__restrict__ A, int64_t ld,
                                                        unsigned long long *__restrict__ out) {
  __shared__ double part[4];
  const double *row = A + (int64_t)blockIdx.x * ld;
  double acc = 0.0;
  for (int j = threadIdx.x; j < cols; j += 256) acc += fabs(row[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = part[0] + part[1] + part[2] + part[3];
    if (t == t) atomicMax(out, (unsigned long long)__double_as_longlong(t));
  }
}
// out[0] = max_j sum_i |A_ij|: one thread per column (coalesced across the threads of a row)
__global__ __launch_bounds__(256) void k_abs_colsum_max(int rows, int cols, const double *__restrict__ A,
                                                        int64_t ld, unsigned long long *__restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (j < cols)
    for (int i = 0; i < rows; ++i) acc += fabs(A[(int64_t)i * ld + j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc = fmax(acc, __shfl_down(acc, off));
  if ((threadIdx.x & 63) == 0 && acc == acc) atomicMax(out, (unsigned long long)__double_as_longlong(acc));
}
// norms3: ||H||_inf, ||J||_inf, ||J||_1 (zeroed here)
void launch_matrix_norms(hipStream_t s, int n, int m, const double *H, int64_t ldh, const double *J,
                         int64_t ldj, double *norms3) {
  (void)hipMemsetAsync(norms3, 0, 3 * sizeof(double), s);
  unsigned long long *o = reinterpret_cast<unsigned long long *>(norms3);
  if (n && H) hipLaunchKernelGGL(k_abs_rowsum_max, dim3(n), dim3(256), 0, s, n, H, ldh, o);
  if (n && m && J) {
    hipLaunchKernelGGL(k_abs_rowsum_max, dim3(m), dim3(256), 0, s, n, J, ldj, o + 1);
    hipLaunchKernelGGL(k_abs_colsum_max, g1(n), dim3(256), 0, s, m, n, J, ldj, o + 2);
  }
}

void launch_axpy1(hipStream_t s, int N, const double *d, double *x) {
  if (N) hipLaunchKernelGGL(k_axpy1, g1(N), dim3(256), 0, s, N, d, x);
}

void launch_symmetrize(hipStream_t s, double *A, int64_t ld, int N) {
  if (N) hipLaunchKernelGGL(k_symmetrize, dim3((N + 255) / 256, N), dim3(256), 0, s, A, ld, N);
}

// ================================================================ batched kernels
// Same bodies as above, one instance per blockIdx.z, sizes read on the device (BInst in
// pgf_internal.h).  Nothing here needs a host round trip.
__global__ void kb_advance(const BInst *__restrict__ tab, int n, int m,
                           const uint8_t *__restrict__ accept) {
  const BInst &I = tab[blockIdx.z];
  const double lamb = I.ps[BPS_LAMB];
  const bool acc = accept[blockIdx.z] != 0;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    if (acc)
      I.xhat[i] = I.x[i];
    else
      I.x[i] = I.xhat[i];
    I.slb[i] = lamb * I.lb[i];
    I.sub[i] = lamb * I.ub[i];
  } else if (i < n + m) {
    if (acc)
      I.yhat[i - n] = I.y[i - n];
    else
      I.y[i - n] = I.yhat[i - n];
  }
  if (i == 0) {
    I.ctl[0] = 0;
    I.ctl[1] = 0;
    I.ctl[2] = 0;
    I.ctl[3] = 0;
  }
}

__global__ void kb_set_frozen(const BInst *__restrict__ tab, int B,
                              const uint8_t *__restrict__ frozen) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) tab[i].ctl[3] = frozen[i] ? 1 : 0;
}

// which 0: c = J x - b ; which 1: g = H x + tmpn
__global__ __launch_bounds__(256) void kb_gemv_rows(const BInst *__restrict__ tab, int which,
                                                    int n, int m) {
  const BInst &I = tab[blockIdx.z];
  if (which == 0)
    b_gemv_rows(m, n, I.J, I.ldj, I.x, I.b, -1.0, I.c);
  else
    b_gemv_rows(n, n, I.H, I.ldh, I.x, I.tmpn, 1.0, I.g);
}

__global__ void kb_mult_vec(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  b_mult_vec(m, I.ps[BPS_RHO], I.c, I.y, I.w);
}

__global__ __launch_bounds__(256) void kb_gemvT_partial(const BInst *__restrict__ tab, int n, int m,
                                                        int chunk) {
  const BInst &I = tab[blockIdx.z];
  b_gemvT_partial(m, n, I.J, I.ldj, I.w, chunk, I.partial);
}

__global__ void kb_sum_partials(const BInst *__restrict__ tab, int n, int used) {
  const BInst &I = tab[blockIdx.z];
  b_sum_partials(n, used, I.partial, I.q, I.tmpn);
}

// tau NaN = None; the factors of the tau form exactly as the host computes them for a single
// instance (implicit_func.py:237-244)
__global__ void kb_active_set(const BInst *__restrict__ tab, int n, double tau) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  const double lamb = I.ps[BPS_LAMB];
  const int use_tau = (tau != tau) ? 0 : 1;
  const double f_x = use_tau ? lamb * (1 - tau * lamb) : 0.0;
  const double f_x0 = use_tau ? tau * lamb * lamb : 0.0;
  const double f_d = use_tau ? tau * lamb : 0.0;
  b_active_set(n, use_tau, lamb, f_x, f_x0, f_d, I.xhat, I.x, I.g, I.slb, I.sub, I.mask_new);
}

// Adopt mask_new (mode 2: always; mode 1: when it differs elementwise from the current mask
// or there is none yet -- newton.py:203-215; mode 0: keep), rebuild the index lists when
// adopted, and decide whether this step factorises: ctl[0] = !factor_valid.
__global__ __launch_bounds__(1024) void kb_mask_adopt(const BInst *__restrict__ tab, int n,
                                                      int mode) {
  const BInst &I = tab[blockIdx.z];
  const int tid = threadIdx.x;
  if (I.ctl[3]) {  // frozen: nothing of this instance moves in this step
    if (tid == 0) I.ctl[0] = 0;
    return;
  }
  int diff = 0;
  if (mode == 2 || (mode == 1 && I.ctl[2] == 0)) {
    diff = 1;
  } else if (mode == 1) {
    for (int j = tid; j < n; j += 1024) diff |= (I.mask[j] != I.mask_new[j]) ? 1 : 0;
  }
  const int changed = __syncthreads_or(diff);
  if (changed) {
    for (int j = tid; j < n; j += 1024) I.mask[j] = I.mask_new[j];
    b_compact(n, I.mask_new, I.idxI, I.idxA, I.pos, I.counts);
  }
  if (tid == 0) {
    if (changed) {
      I.ctl[2] = 1;
      I.ctl[1] = 0;
    }
    I.ctl[0] = (changed || I.ctl[1] == 0) ? 1 : 0;
  }
}

__global__ void kb_residual(const BInst *__restrict__ tab, int n, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  const double lamb = I.ps[BPS_LAMB], dt = I.ps[BPS_DT];
  b_residual(n, m, lamb, dt, I.xhat, I.yhat, I.x, I.y, I.g, I.c, I.slb, I.sub, I.mask, I.F,
             I.b0full);
}

__global__ __launch_bounds__(256) void kb_active_rows_partial(const BInst *__restrict__ tab, int n,
                                                              int nparts) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  b_active_rows_partial(n, I.counts[1], I.idxA, I.H, I.ldh, I.b0full, nparts, I.partial);
}
__global__ __launch_bounds__(256) void kb_reduced_rhs(const BInst *__restrict__ tab, int n,
                                                      int m, int nparts) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  b_reduced_rhs(n, m, I.counts[0], I.counts[1], I.ps[BPS_FACT], I.F, I.idxI, I.J, I.ldj, I.b0full,
                I.partial, nparts, I.rhs);
}

// K (lower triangle) + the right-hand side in row N, only for instances that factorise
// (KB_ASM_ROWS rows per workgroup: with 8, a batch of 256 instances of N = 1280 is 206 000
// workgroups and the launch is bound by their dispatch)
#define KB_ASM_ROWS 32
__global__ __launch_bounds__(256) void kb_assemble(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] == 0) return;
  const double lamb = I.ps[BPS_LAMB], delta = I.ps[BPS_DELTA];
  const int nI = I.counts[0], N = nI + m;
  const int i0 = blockIdx.y * KB_ASM_ROWS;
  if (i0 > N) return;
  if (i0 == 0 && blockIdx.x == 0 && threadIdx.x < 4) I.flags[threadIdx.x] = 0;
  if (i0 <= N && N < i0 + KB_ASM_ROWS) {  // this row block also holds row N: the rhs
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < N) I.K[(int64_t)N * I.ldk + j] = I.rhs[j];
  }
  b_assemble_kkt<KB_ASM_ROWS>(I.K, I.ldk, I.H, I.ldh, I.J, I.ldj, I.idxI, nI, m, lamb, delta);
}

__global__ __launch_bounds__(256) void kb_step_update(const BInst *__restrict__ tab, int n,
                                                      int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  b_step_update(n, m, I.counts[0], I.ps[BPS_FACT], I.ps[BPS_RHO], I.x, I.y, I.lb, I.ub, I.mask, I.pos, I.b0full, I.F,
                I.sol, I.dx, I.dy, I.xn, I.yn, I.red);
  // the new point replaces the current one in place (each lane re-reads its own entry) -- unless
  // the sampled residual of the solve failed: the host repairs such an instance from the point it
  // started at (batch_repair_instance, pgf_api.hip)
  if (I.flags[3]) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n)
    I.x[i] = I.xn[i];
  else if (i < n + m)
    I.y[i - n] = I.yn[i - n];
}

// Accuracy guard of the batched path.  The factor has overwritten the assembled matrix and a
// full residual would read H and J of every instance again (3 GB for 256 instances, 7 % of the
// step); element growth in an unpivoted LDL^T spoils the whole solution vector, so a SAMPLE of
// the rows of r = rhs - K s tells just as well: KB_NSAMPLE rows spread over the reduced system,
// K applied from H, J, the index list and the instance's lambda / delta, one wavefront per row.
// max |r_sampled| > KB_RES_TOL max |rhs| sets flags[3]: the step is reported as failed
// (kb_step_final), which every controller answers with a rejected step and a doubled lambda --
// the reference's own recovery path (step_control.py:80-107) and what makes the matrix
// quasi-definite again.
#define KB_NSAMPLE 32
#define KB_RES_TOL 1e-8
__global__ __launch_bounds__(256) void kb_sample_residual(const BInst *__restrict__ tab, int m) {
  // grid (KB_NSAMPLE / 4, 1, B): one sampled row per wavefront (a row is a chain of dependent
  // gathers: 32 rows one after the other in one workgroup cost 0.1 ms per batched step)
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  __shared__ double bmax[4];
  const int nI = I.counts[0], N = nI + m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double lamb = I.ps[BPS_LAMB], delta = I.ps[BPS_DELTA];
  double rb = 0.0;
  for (int j = tid; j < N; j += 256) rb = fmax(rb, fabs(I.rhs[j]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) rb = fmax(rb, __shfl_down(rb, off));
  if (lane == 0) bmax[wave] = rb;
  __syncthreads();
  const double b = fmax(fmax(bmax[0], bmax[1]), fmax(bmax[2], bmax[3]));
  const int ns = min(KB_NSAMPLE, N);
  const int k = blockIdx.x * 4 + wave;
  if (k >= ns) return;
  const int i = (int)(((long long)k * N) / ns);
  double acc = 0.0;
  if (i < nI) {
    const double *hrow = I.H + (int64_t)I.idxI[i] * I.ldh;
    for (int j = lane; j < nI; j += 64) acc = fma(hrow[I.idxI[j]], I.sol[j], acc);
    const double *jcol = I.J + I.idxI[i];
    for (int r = lane; r < m; r += 64) acc = fma(jcol[(int64_t)r * I.ldj], I.sol[nI + r], acc);
    if (lane == 0) acc = fma(lamb, I.sol[i], acc);
  } else {
    const double *jrow = I.J + (int64_t)(i - nI) * I.ldj;
    for (int j = lane; j < nI; j += 64) acc = fma(jrow[I.idxI[j]], I.sol[j], acc);
    if (lane == 0) acc = fma(-delta, I.sol[i], acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) {
    const double r = fabs(I.rhs[i] - acc);
    // (flags[3] was cleared by kb_solve_prep_bwd of this step; set only on failure: no
    // contention in the normal case)
    if (!(r <= KB_RES_TOL * (b > 0.0 ? b : 1.0))) atomicOr(&I.flags[3], 1);
  }
}

__global__ __launch_bounds__(256) void kb_step_final(const BInst *__restrict__ tab, int nb,
                                                     double *__restrict__ diff_out,
                                                     int *__restrict__ flags_out) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) {  // frozen: no step was taken
    if (threadIdx.x == 0) {
      diff_out[blockIdx.z] = 0.0;
      flags_out[3 * blockIdx.z] = 0;
      flags_out[3 * blockIdx.z + 1] = I.flags[1];
      flags_out[3 * blockIdx.z + 2] = I.counts[0];
    }
    return;
  }
  b_final_reduce(I.red, nb, diff_out + blockIdx.z, 1);
  if (threadIdx.x == 0) {
    // bit 0: zero / non-finite pivot; bit 1: the chain's helper workgroups failed a check
    // (a chained solve of the instance that failed its own checks counts like the helpers)
    // bit 2: the sampled residual of the solve is too large (kb_sample_residual)
    const int bad = (I.flags[0] ? 1 : 0) | ((I.flags[2] || I.cctl[1]) ? 2 : 0) | (I.flags[3] ? 4 : 0);
    I.cctl[1] = 0;
    // (sticky, behind the per-instance triples: a device-resident controller loop overwrites the
    // triples step after step; pgf_batch_ctl_read must still learn that a hand-over failed)
    if (bad & 2) atomicOr(flags_out + 3 * gridDim.z, 1);
    flags_out[3 * blockIdx.z] = bad;
    flags_out[3 * blockIdx.z + 1] = I.flags[1];
    flags_out[3 * blockIdx.z + 2] = I.counts[0];
    if (I.ctl[0]) I.ctl[1] = (bad == 0) ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void kb_unscaled_res_sq(const BInst *__restrict__ tab, int n,
                                                          int m) {
  const BInst &I = tab[blockIdx.z];
  b_unscaled_res_sq(n, m, I.ps[BPS_DT], I.xhat, I.yhat, I.x, I.y, I.g, I.c, I.lb, I.ub, I.red);
}

__global__ __launch_bounds__(256) void kb_norm_final(const BInst *__restrict__ tab, int nb,
                                                     double *__restrict__ norm_out) {
  const BInst &I = tab[blockIdx.z];
  b_final_reduce(I.red, nb, norm_out + blockIdx.z, 1);
}

static inline dim3 gb(int cnt, int per, int B) { return dim3((cnt + per - 1) / per, 1, B); }

// ---------------------------------------------------------------- device-resident step controller
// The reference's default DistanceRatioController (step/distance_ratio_control.py:12-78, PI law of
// controller.py:54-77) for every instance of a batch WITHOUT host round trips: three tiny kernels
// around the two batched Newton steps of an outer iteration keep lambda, the PI integral, the
// accept flag and the early exits on the device.  cs: DCS_STRIDE doubles per instance; cp: the
// controller's constants; log: 3 doubles per (iteration, instance).
__global__ void kb_dctl_begin(int B, const double *__restrict__ cs, const double *__restrict__ cp,
                              double *__restrict__ ps, uint8_t *__restrict__ accept) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const double *c = cs + (size_t)DCS_STRIDE * i;
  // the host controller hands over dt = 1 / lambda and the library recomputes lambda = 1 / dt
  // (pgf_batch_advance_outer_each): same two roundings here
  const double dt = 1.0 / c[DCS_LAMB];
  const double lamb = 1.0 / dt, rho = cp[DCP_RHO];
  double *p = ps + (size_t)BPS_STRIDE * i;
  p[BPS_DT] = dt;
  p[BPS_LAMB] = lamb;
  p[BPS_RHO] = rho;
  p[BPS_FACT] = 1.0 / (1.0 + lamb * rho);
  p[BPS_DELTA] = lamb / (1.0 + lamb * rho);
  accept[i] = c[DCS_ACCEPTED] != 0.0 ? 1 : 0;
}

// after the first Newton step: failed factorisation -> reject, 2 lambda; converged residual ->
// accept, lambda * lamb_red; zero step -> accept; all three freeze the instance for step two
__global__ void kb_dctl_mid(const BInst *__restrict__ tab, int B, double *__restrict__ cs,
                            const double *__restrict__ cp, const double *__restrict__ diff,
                            const int *__restrict__ flags, const double *__restrict__ norm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double *c = cs + (size_t)DCS_STRIDE * i;
  const double lamb = c[DCS_LAMB];
  c[DCS_USED] = lamb;
  double next = lamb, acc = 1.0, done = 1.0;
  if (flags[3 * i] != 0) {
    next = 2.0 * lamb;
    acc = 0.0;
  } else if (norm[i] <= cp[DCP_NEWTON_TOL]) {
    next = fmax(lamb * cp[DCP_LAMB_RED], cp[DCP_LAMB_MIN]);
  } else if (diff[i] == 0.0) {
    next = lamb;
  } else {
    done = 0.0;
  }
  c[DCS_NEXT] = next;
  c[DCS_ACCEPTED] = acc;
  c[DCS_DONE] = done;
  c[DCS_FIRST] = diff[i];
  tab[i].ctl[3] = done != 0.0 ? 1 : 0;
}

// after the second step: theta test + PI update on the log scale for the instances still in play
__global__ void kb_dctl_end(int B, double *__restrict__ cs, const double *__restrict__ cp,
                            const double *__restrict__ diff, const int *__restrict__ flags,
                            double *__restrict__ log3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double *c = cs + (size_t)DCS_STRIDE * i;
  const double lamb = c[DCS_LAMB];
  double next = c[DCS_NEXT], acc = c[DCS_ACCEPTED];
  if (c[DCS_DONE] == 0.0) {
    const double second = diff[i];
    if (flags[3 * i] != 0) {
      next = 2.0 * lamb;
      acc = 0.0;
    } else if (second == 0.0) {
      next = lamb;
      acc = 1.0;
    } else {
      const double theta = second / c[DCS_FIRST];
      if (theta <= cp[DCP_THETA_MAX]) {
        const double e = cp[DCP_LOG_THETA_REF] - log(theta);
        const double integral = c[DCS_INTEGRAL] + e;
        c[DCS_INTEGRAL] = integral;
        const double u = cp[DCP_K_P] * e + cp[DCP_K_I] * integral;
        next = fmax(cp[DCP_LAMB_MIN], lamb / exp(u));
        acc = 1.0;
      } else {
        next = lamb * cp[DCP_LAMB_INC];
        acc = 0.0;
      }
    }
  }
  c[DCS_LAMB] = next;
  c[DCS_ACCEPTED] = acc;
  if (log3) {
    log3[3 * i] = lamb;
    log3[3 * i + 1] = next;
    log3[3 * i + 2] = acc;
  }
}

void batch_launch_dctl_begin(hipStream_t s, int B, const double *cs, const double *cp, double *ps,
                             uint8_t *accept) {
  hipLaunchKernelGGL(kb_dctl_begin, dim3((B + 255) / 256), dim3(256), 0, s, B, cs, cp, ps, accept);
}
void batch_launch_dctl_mid(hipStream_t s, const BInst *tab, int B, double *cs, const double *cp,
                           const double *diff, const int *flags, const double *norm) {
  hipLaunchKernelGGL(kb_dctl_mid, dim3((B + 255) / 256), dim3(256), 0, s, tab, B, cs, cp, diff, flags,
                     norm);
}
void batch_launch_dctl_end(hipStream_t s, int B, double *cs, const double *cp, const double *diff,
                           const int *flags, double *log3) {
  hipLaunchKernelGGL(kb_dctl_end, dim3((B + 255) / 256), dim3(256), 0, s, B, cs, cp, diff, flags, log3);
}

void batch_launch_advance(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                          const uint8_t *accept) {
  hipLaunchKernelGGL(kb_advance, gb(std::max(1, sc.n + sc.m), 256, B), dim3(256), 0, s, tab, sc.n,
                     sc.m, accept);
}

void batch_launch_set_frozen(hipStream_t s, const BInst *tab, int B, const uint8_t *frozen) {
  hipLaunchKernelGGL(kb_set_frozen, dim3((B + 255) / 256), dim3(256), 0, s, tab, B, frozen);
}

// c = A x - b ; w = rho c + y ; tmpn = q + A' w ; g = Q x + tmpn   for every instance
void batch_launch_eval(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc, int nparts) {
  const int n = sc.n, m = sc.m;
  if (!n) return;
  int used = 0;
  if (m) {
    hipLaunchKernelGGL(kb_gemv_rows, gb(m, 4, B), dim3(256), 0, s, tab, 0, n, m);
    hipLaunchKernelGGL(kb_mult_vec, gb(m, 256, B), dim3(256), 0, s, tab, m);
    const int chunk = (m + nparts - 1) / nparts;
    used = (m + chunk - 1) / chunk;
    hipLaunchKernelGGL(kb_gemvT_partial, dim3((n + 255) / 256, used, B), dim3(256), 0, s, tab, n, m,
                       chunk);
  }
  hipLaunchKernelGGL(kb_sum_partials, gb(n, 256, B), dim3(256), 0, s, tab, n, used);
  hipLaunchKernelGGL(kb_gemv_rows, gb(n, 4, B), dim3(256), 0, s, tab, 1, n, m);
}

void batch_launch_mask(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc, int mode,
                       double tau) {
  if (mode != 0 && sc.n)
    hipLaunchKernelGGL(kb_active_set, gb(sc.n, 256, B), dim3(256), 0, s, tab, sc.n, tau);
  hipLaunchKernelGGL(kb_mask_adopt, dim3(1, 1, B), dim3(1024), 0, s, tab, sc.n, mode);
}

// ---- condensed order, batched (single-instance twins: k_cond_* at the end of this file) -------
// V[i][r] = J[r][idxI[i]], zero beyond m; instances that factorise this step only
__global__ __launch_bounds__(256) void kb_cond_panel(const BInst *__restrict__ tab, int m, int mp) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] == 0) return;
  const int nI = I.counts[0];
  const int i0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  if (i0 >= nI) return;
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i = i0 + tx;
  const int col = i < nI ? I.idxI[i] : 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + ty + 8 * p;
    tile[ty + 8 * p][tx] = (r < m && i < nI) ? I.J[(int64_t)r * I.ldj + col] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ii = i0 + ty + 8 * p, r = r0 + tx;
    if (ii < nI && r < mp) I.V[(int64_t)ii * I.ldv + r] = tile[tx][ty + 8 * p];
  }
}
// row nI of V <- b_y, vd <- -1 / delta
__global__ void kb_cond_tail(const BInst *__restrict__ tab, int m, int mp) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] == 0) return;
  const int nI = I.counts[0];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mp) return;
  I.V[(int64_t)nI * I.ldv + r] = (r < m) ? I.rhs[nI + r] : 0.0;
  I.vd[r] = -1.0 / I.ps[BPS_DELTA];
}
// zwork[i] = b_x[i] + dot(V[i][0:m], b_y) / delta: the forward solve's input when the factor is reused
__global__ __launch_bounds__(256) void kb_cond_prep_fwd(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] != 0 || I.ctl[3]) return;
  const int nI = I.counts[0];
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nI) return;
  const double d = row_dot(I.V + (int64_t)i * I.ldv, I.rhs + nI, m, lane);
  if (lane == 0) I.zwork[i] = I.rhs[i] + d / I.ps[BPS_DELTA];
}
// sol_y[r] = (sum_i V[i][r] sol_x[i] - b_y[r]) / delta: 64 columns per workgroup, sixteen lane
// groups take the rows i = g, g + 16, ... (four loads in flight each) and are summed in a fixed order
__global__ __launch_bounds__(1024) void kb_cond_y(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  __shared__ double part[16][64];
  const int nI = I.counts[0];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + c;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (r < m) {
    const double *__restrict__ v = I.V + r;
    const double *__restrict__ x = I.sol;
    const int64_t ld = I.ldv;
    int i = g;
    for (; i + 48 < nI; i += 64) {
      s0 = fma(v[(int64_t)i * ld], x[i], s0);
      s1 = fma(v[(int64_t)(i + 16) * ld], x[i + 16], s1);
      s2 = fma(v[(int64_t)(i + 32) * ld], x[i + 32], s2);
      s3 = fma(v[(int64_t)(i + 48) * ld], x[i + 48], s3);
    }
    for (; i < nI; i += 16) s0 = fma(v[(int64_t)i * ld], x[i], s0);
  }
  part[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && r < m) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][c];
    I.sol[nI + r] = (s - I.rhs[nI + r]) / I.ps[BPS_DELTA];
  }
}

void batch_launch_rhs_assemble(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc, int cond_mp) {
  const int n = sc.n, m = sc.m, Nmax = n + m;
  if (!Nmax) return;
  hipLaunchKernelGGL(kb_residual, gb(Nmax, 256, B), dim3(256), 0, s, tab, n, m);
  // (the scratch `partial' holds 32 n doubles per instance: PGF_GEMVT_PARTS in pgf_api.hip)
  if (n) hipLaunchKernelGGL(kb_active_rows_partial, dim3((n + 255) / 256, 32, B), dim3(256), 0, s, tab, n, 32);
  hipLaunchKernelGGL(kb_reduced_rhs, gb(Nmax, 4, B), dim3(256), 0, s, tab, n, m, 32);
  if (cond_mp > 0) {
    // A and b_x (the assembly kernel with no constraint rows), V, b_y
    hipLaunchKernelGGL(kb_assemble, dim3((n + 255) / 256, n / KB_ASM_ROWS + 1, B), dim3(256), 0, s, tab, 0);
    hipLaunchKernelGGL(kb_cond_panel, dim3((n + 31) / 32, cond_mp / 32, B), dim3(256), 0, s, tab, m, cond_mp);
    hipLaunchKernelGGL(kb_cond_tail, gb(cond_mp, 256, B), dim3(256), 0, s, tab, m, cond_mp);
    return;
  }
  hipLaunchKernelGGL(kb_assemble, dim3((Nmax + 255) / 256, Nmax / KB_ASM_ROWS + 1, B), dim3(256), 0,
                     s, tab, m);
}
void batch_launch_cond_prep_fwd(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc) {
  if (sc.n) hipLaunchKernelGGL(kb_cond_prep_fwd, gb(sc.n, 4, B), dim3(256), 0, s, tab, sc.m);
}
void batch_launch_cond_y(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc) {
  if (sc.m) hipLaunchKernelGGL(kb_cond_y, dim3((sc.m + 63) / 64, 1, B), dim3(1024), 0, s, tab, sc.m);
}

void batch_launch_step_update(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                              double *diff_out, int *flags_out) {
  const int nb = step_update_blocks(sc.n, sc.m);
  hipLaunchKernelGGL(kb_sample_residual, dim3(KB_NSAMPLE / 4, 1, B), dim3(256), 0, s, tab, sc.m);
  if (nb)
    hipLaunchKernelGGL(kb_step_update, dim3(nb, 1, B), dim3(256), 0, s, tab, sc.n, sc.m);
  hipLaunchKernelGGL(kb_step_final, dim3(1, 1, B), dim3(256), 0, s, tab, nb, diff_out, flags_out);
}

void batch_launch_res_norm(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                           double *norm_out) {
  const int nb = (sc.n + sc.m + 255) / 256;
  if (nb)
    hipLaunchKernelGGL(kb_unscaled_res_sq, dim3(nb, 1, B), dim3(256), 0, s, tab, sc.n, sc.m);
  hipLaunchKernelGGL(kb_norm_final, dim3(1, 1, B), dim3(256), 0, s, tab, nb, norm_out);
}

// r = Q x + (q + A'y) into tmpn / g is NOT touched: the measures use their own vector (F is
// free between steps).  which 2: F[0:n] = H x + tmpn (after kb_sum_partials with w = y)
__global__ void kb_copy_y_to_w(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) I.w[i] = I.y[i];
}

__global__ __launch_bounds__(256) void kb_gemv_stat(const BInst *__restrict__ tab, int n) {
  const BInst &I = tab[blockIdx.z];
  b_gemv_rows(n, n, I.H, I.ldh, I.x, I.tmpn, 1.0, I.F);
}

__global__ __launch_bounds__(256) void kb_measures(const BInst *__restrict__ tab, int n, int m,
                                                   double active_tol, double *__restrict__ red4,
                                                   int nb) {
  const BInst &I = tab[blockIdx.z];
  b_measures(n, m, active_tol, I.x, I.y, I.F, I.c, I.lb, I.ub, red4 + (size_t)blockIdx.z * 4 * nb);
}

__global__ __launch_bounds__(256) void kb_measures_final(const double *__restrict__ red4, int nb,
                                                         double *__restrict__ out) {
  b_measures_final(red4 + (size_t)blockIdx.z * 4 * nb, nb, out + 4 * blockIdx.z);
}

// c must be fresh (batch_launch_eval).  Leaves tmpn / w overwritten: the caller marks the
// evaluation stale.
void batch_launch_measures(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                           int nparts, double active_tol, double *red4, double *out) {
  const int n = sc.n, m = sc.m;
  const int nb = (n + m + 255) / 256;
  if (n) {
    int used = 0;
    if (m) {
      hipLaunchKernelGGL(kb_copy_y_to_w, gb(m, 256, B), dim3(256), 0, s, tab, m);
      const int chunk = (m + nparts - 1) / nparts;
      used = (m + chunk - 1) / chunk;
      hipLaunchKernelGGL(kb_gemvT_partial, dim3((n + 255) / 256, used, B), dim3(256), 0, s, tab, n,
                         m, chunk);
    }
    hipLaunchKernelGGL(kb_sum_partials, gb(n, 256, B), dim3(256), 0, s, tab, n, used);
    hipLaunchKernelGGL(kb_gemv_stat, gb(n, 4, B), dim3(256), 0, s, tab, n);
  }
  if (nb)
    hipLaunchKernelGGL(kb_measures, dim3(nb, 1, B), dim3(256), 0, s, tab, n, m, active_tol, red4,
                       nb);
  hipLaunchKernelGGL(kb_measures_final, dim3(1, 1, B), dim3(256), 0, s, red4, nb, out);
}

// ---------------------------------------------------------------- condensed KKT system
// With the constraint block -delta I eliminated first (block elimination of
// [[H[I,I] + lamb I, J_I^T], [J_I, -delta I]], the reference's SymmetricStepSolver matrix,
// step/solver/symmetric_step_solver.py:35-76) what is factorised is the nI x nI Schur complement
//     S = H[I,I] + lamb I + J_I^T J_I / delta,
// applied by the factorisation itself from the panel V = J_I^T (nI x m, row-major, zero-padded to
// a multiple of 32 columns) with D-scaling vd = -1 / delta (DenseLdlt::V, pgf_factor2.hip).
// Row nI of V carries the constraint rows of the right-hand side, so that the elimination turns
// row nI of K (b_x) into b_x + J_I^T b_y / delta on the way.
//   S s_x = b_x + J_I^T b_y / delta,      s_y = (J_I s_x - b_y) / delta.

// V[i][r] = J[r][idxI[i]] (r < m), 0 for the padding columns: 32 x 32 tiles through LDS, reads
// run along i (contiguous whenever I is), writes along r
__global__ __launch_bounds__(256) void k_cond_panel(double *__restrict__ V, int64_t ldv, int mp,
                                                    const double *__restrict__ J, int64_t ldj,
                                                    const int *__restrict__ idxI, int nI, int m) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int i = i0 + tx;
  const int col = i < nI ? idxI[i] : 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + ty + 8 * p;
    tile[ty + 8 * p][tx] = (r < m && i < nI) ? J[(int64_t)r * ldj + col] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int ii = i0 + ty + 8 * p, r = r0 + tx;
    if (ii < nI && r < mp) V[(int64_t)ii * ldv + r] = tile[tx][ty + 8 * p];
  }
}

// row nI of V <- the constraint part of the right-hand side (or zeros), vd <- -1 / delta
__global__ void k_cond_tail(double *__restrict__ Vrow, double *__restrict__ vd, int mp, int m,
                            const double *__restrict__ rhs_y, double delta) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mp) return;
  Vrow[r] = (rhs_y && r < m) ? rhs_y[r] : 0.0;
  vd[r] = -1.0 / delta;
}

// out[i] = rhs[i] + dot(V[i][0:m], rhs[nI:nI+m]) / delta
__global__ __launch_bounds__(256) void k_cond_rhs(int nI, int m, const double *__restrict__ V, int64_t ldv,
                                                  const double *__restrict__ rhs, double delta,
                                                  double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nI) return;
  const double d = row_dot(V + (int64_t)i * ldv, rhs + nI, m, lane);
  if (lane == 0) out[i] = rhs[i] + d / delta;
}

// sol_y[r] = (sum_p partial[p][r] - rhs_y[r]) / delta   (partial: V^T s_x in fixed row chunks);
// 64 columns per workgroup, four groups of lanes take the chunks p = g, g + 4, ... and are summed
// in a fixed order
__global__ __launch_bounds__(256) void k_cond_y(int m, int nparts, const double *__restrict__ partial,
                                                const double *__restrict__ rhs_y, double delta,
                                                double *__restrict__ sol_y) {
  __shared__ double part[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + c;
  double s = 0.0;
  if (r < m)
    for (int p = g; p < nparts; p += 4) s += partial[(int64_t)p * m + r];
  part[g][c] = s;
  __syncthreads();
  if (g == 0 && r < m) sol_y[r] = (((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) - rhs_y[r]) / delta;
}

void launch_cond_panel(hipStream_t s, double *V, int64_t ldv, int mp, double *vd, const double *J,
                       int64_t ldj, const int *idxI, int nI, int m, double delta, const double *rhs_y) {
  if (nI > 0 && mp > 0)
    hipLaunchKernelGGL(k_cond_panel, dim3((nI + 31) / 32, mp / 32), dim3(256), 0, s, V, ldv, mp, J, ldj,
                       idxI, nI, m);
  if (mp > 0)
    hipLaunchKernelGGL(k_cond_tail, g1(mp), dim3(256), 0, s, V + (int64_t)nI * ldv, vd, mp, m, rhs_y, delta);
}

void launch_cond_rhs(hipStream_t s, int nI, int m, const double *V, int64_t ldv, const double *rhs,
                     double delta, double *out) {
  if (nI) hipLaunchKernelGGL(k_cond_rhs, dim3((nI + 3) / 4), dim3(256), 0, s, nI, m, V, ldv, rhs, delta, out);
}

void launch_cond_y(hipStream_t s, int nI, int m, const double *V, int64_t ldv, const double *solx,
                   const double *rhs_y, double delta, double *partial, size_t partial_cap, double *sol_y) {
  if (!m) return;
  // as many row chunks as the scratch holds (up to 128): the product reads V once, 8 m nI bytes
  const int nparts = (int)std::max<size_t>(1, std::min<size_t>(128, partial_cap / (size_t)m));
  const int chunk = (std::max(nI, 1) + nparts - 1) / nparts;
  const int used = nI > 0 ? (nI + chunk - 1) / chunk : 0;
  if (used)
    hipLaunchKernelGGL(k_gemvT_partial, dim3((m + 255) / 256, used), dim3(256), 0, s, nI, m, V, ldv, solx,
                       chunk, partial);
  hipLaunchKernelGGL(k_cond_y, dim3((m + 63) / 64), dim3(256), 0, s, m, used, partial, rhs_y, delta, sol_y);
}
