// Elementwise, compaction, gather-assembly and matrix-vector kernels of the Newton step.
// Compiled with -ffp-contract=off: the active-set mask must be bit-exact with the
// reference's numpy expressions, so no multiply-add may be fused here unless written
// as an explicit fma() (only inside dot products, whose summation order differs from
// scipy's anyway and is covered by the 1e-10 iterate tolerance).
#include "pgf_kernels.h"

#define ACTIVE_EPS 1e-8  // reference implicit_func.py:44

// ---------------------------------------------------------------- bounds (a2)
__global__ void k_scale_bounds(int n, double lamb, const double *__restrict__ lb,
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
__global__ void k_active_set(int n, int use_tau, double lamb, double f_x, double f_x0, double f_d,
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
__global__ __launch_bounds__(1024) void k_compact(int n, const uint8_t *__restrict__ mask,
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
__global__ void k_residual(int n, int m, double lamb, double dt, const double *__restrict__ xhat,
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
__global__ __launch_bounds__(256) void k_reduced_rhs(int n, int m, int nI, int nA, double fact,
                                                     const double *__restrict__ F,
                                                     const int *__restrict__ idxI,
                                                     const double *__restrict__ H, int64_t ldh,
                                                     const double *__restrict__ J, int64_t ldj,
                                                     const double *__restrict__ b0full,
                                                     double *__restrict__ rhs) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nI + m) return;
  double base;
  const double *row;
  if (i < nI) {
    const int gi = idxI[i];
    base = F[gi];
    row = H + (int64_t)gi * ldh;
  } else {
    const int r = i - nI;
    base = fact * F[n + r];
    row = J + (int64_t)r * ldj;
  }
  double corr = 0.0;
  if (nA > 0) corr = row_dot(row, b0full, n, lane);
  if (lane == 0) rhs[i] = base - corr;
}

// ---------------------------------------------------------------- K assembly (a10, a12)
// Lower triangle of K = [[H[I,I] + lamb I, .],[J[:,I], -delta I]] gathered from the
// device-resident H, J.  blockIdx.y = row of K, lanes run along columns (coalesced
// stores; loads coalesced whenever I is contiguous).
__global__ __launch_bounds__(256) void k_assemble_kkt(double *__restrict__ K, int64_t ldk,
                                                      const double *__restrict__ H, int64_t ldh,
                                                      const double *__restrict__ J, int64_t ldj,
                                                      const int *__restrict__ idxI, int nI, int m,
                                                      double lamb, double delta) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > i) return;
  double v;
  if (i < nI) {
    v = H[(int64_t)idxI[i] * ldh + idxI[j]];
    if (i == j) v = v + lamb;
  } else {
    const int r = i - nI;
    if (j < nI)
      v = J[(int64_t)r * ldj + idxI[j]];
    else
      v = (j == i) ? -delta : 0.0;
  }
  K[(int64_t)i * ldk + j] = v;
}

__global__ void k_copy(double *__restrict__ dst, const double *__restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void k_copy_u8(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// count positions where two masks differ (ActiveSet policy, newton.py:210)
__global__ void k_mask_diff(int n, const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
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
__global__ __launch_bounds__(256) void k_step_update(
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
__global__ __launch_bounds__(256) void k_final_reduce(const double *__restrict__ red, int cnt,
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
__global__ __launch_bounds__(256) void k_gemv_rows(int rows, int cols,
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
__global__ __launch_bounds__(256) void k_gemvT_partial(int rows, int cols,
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
__global__ void k_sum_partials(int cols, int nparts, const double *__restrict__ partial,
                               const double *__restrict__ base, double *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += partial[(int64_t)p * cols + j];
  out[j] = base[j] + s;
}

// w = rho * c + y
__global__ void k_mult_vec(int m, double rho, const double *__restrict__ c,
                           const double *__restrict__ y, double *__restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) w[i] = rho * c[i] + y[i];
}

// squared entries of the UNSCALED residual with its own mask (ImplicitFunc.value_at,
// implicit_func.py:131-161): per-block partial sums -> red
__global__ __launch_bounds__(256) void k_unscaled_res_sq(
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
                        const int *idxI, const double *H, int64_t ldh, const double *J,
                        int64_t ldj, const double *b0full, double *rhs) {
  const int N = nI + m;
  if (N)
    hipLaunchKernelGGL(k_reduced_rhs, dim3((N + 3) / 4), dim3(256), 0, s, n, m, nI, nA, fact, F,
                       idxI, H, ldh, J, ldj, b0full, rhs);
}

void launch_assemble_kkt(hipStream_t s, double *K, int64_t ldk, const double *H, int64_t ldh,
                         const double *J, int64_t ldj, const int *idxI, int nI, int m,
                         double lamb, double delta) {
  const int N = nI + m;
  if (N)
    hipLaunchKernelGGL(k_assemble_kkt, dim3((N + 255) / 256, N), dim3(256), 0, s, K, ldk, H, ldh,
                       J, ldj, idxI, nI, m, lamb, delta);
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
