// Sparse (banded) variant of the Newton/KKT step: CSR Hessian / Jacobian resident in HBM,
// KKT matrix assembled into a symmetric band (after a host-computed bandwidth-reducing
// permutation of the FIXED full pattern), banded LDL^T and banded triangular solves.
//
// The reduced system of the reference (symmetric_step_solver.py:49-94) drops the active
// rows / columns; here the system keeps its full size n + m and an active variable a
// becomes an identity row / column with right-hand side b0[a].  The two systems have the
// same solution on the inactive set, s[a] = b0[a] is what the reference scatters into
// dx[A] anyway (symmetric_step_solver.py:115-121), the added unit pivots are positive so
// the inertia count is unchanged, and -- the point -- the sparsity pattern, permutation
// and band layout never change while the mask churns.
//
// Band layout: row i of the permuted matrix stores K[i][i - d] at band[i * ldb + d],
// d = 0 .. bw (lower band, row-major); ldb = bw + 1 rounded up to even.
//
// Compiled with -ffp-contract=off like pgf_kernels.hip (explicit fma() only).
#include "pgf_sparse.h"

#define ACTIVE_EPS 1e-8

// ---------------------------------------------------------------- CSR products
// y[r] = sum_k val[k] * x[col[k]]  (+ sgn * add[r]);  rows are short: one lane per row
__global__ void k_csr_spmv(int rows, const int *__restrict__ ptr, const int *__restrict__ col,
                           const double *__restrict__ val, const double *__restrict__ x,
                           const double *__restrict__ add, double sgn, double *__restrict__ y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  double acc = 0.0;
  for (int k = ptr[r]; k < ptr[r + 1]; ++k) acc = fma(val[k], x[col[k]], acc);
  y[r] = add ? acc + sgn * add[r] : acc;
}

// out[j] = base[j] + sum_k val[map[k]] * w[row[k]] over the entries of column j
// (transposed product through the column-ordered copy of the pattern: no atomics)
__global__ void k_csc_spmvT(int cols, const int *__restrict__ tptr, const int *__restrict__ trow,
                            const int *__restrict__ tmap, const double *__restrict__ val,
                            const double *__restrict__ w, const double *__restrict__ base,
                            double *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  double acc = 0.0;
  for (int k = tptr[j]; k < tptr[j + 1]; ++k) acc = fma(val[tmap[k]], w[trow[k]], acc);
  out[j] = base[j] + acc;
}

// The evaluation of the device-resident sparse mode in two launches instead of four (the banded
// path is bound by the number of its small dependent launches, not by their arithmetic):
//   c = J x - b ; w = rho c + y                  (one lane per constraint row)
//   g = H x + (q + J' w)                         (one lane per variable: row of H, column of J)
// Operation order as in the separate kernels (k_csr_spmv, k_mult_vec, k_csc_spmvT): the active-set
// mask computed from g is compared bit for bit.
__global__ void k_sp_eval_c(int m, const int *__restrict__ ptr, const int *__restrict__ col,
                            const double *__restrict__ val, const double *__restrict__ x,
                            const double *__restrict__ b, double rho, const double *__restrict__ y,
                            double *__restrict__ c, double *__restrict__ w) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  double acc = 0.0;
  for (int k = ptr[r]; k < ptr[r + 1]; ++k) acc = fma(val[k], x[col[k]], acc);
  const double cr = acc + -1.0 * b[r];
  c[r] = cr;
  w[r] = rho * cr + y[r];
}

__global__ void k_sp_eval_g(int n, const int *__restrict__ hptr, const int *__restrict__ hcol,
                            const double *__restrict__ hval, const int *__restrict__ tptr,
                            const int *__restrict__ trow, const int *__restrict__ tmap,
                            const double *__restrict__ jval, const double *__restrict__ x,
                            const double *__restrict__ w, const double *__restrict__ q,
                            double *__restrict__ g) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double at = 0.0;
  for (int k = tptr[j]; k < tptr[j + 1]; ++k) at = fma(jval[tmap[k]], w[trow[k]], at);
  const double base = q[j] + at;  // = tmpn[j] of the separate kernels
  double acc = 0.0;
  for (int k = hptr[j]; k < hptr[j + 1]; ++k) acc = fma(hval[k], x[hcol[k]], acc);
  g[j] = acc + 1.0 * base;
}

// permuted right-hand side with the products H b0, J b0 formed on the fly (one launch for
// k_csr_spmv x 2 + k_band_rhs)
__global__ void k_band_rhs_fused(int n, int m, const uint8_t *__restrict__ mask,
                                 const double *__restrict__ F, const double *__restrict__ b0full,
                                 const int *__restrict__ hptr, const int *__restrict__ hcol,
                                 const double *__restrict__ hval, const int *__restrict__ jptr,
                                 const int *__restrict__ jcol, const double *__restrict__ jval,
                                 double fact, const int *__restrict__ pos, double *__restrict__ brhs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + m) return;
  double v;
  if (i < n) {
    double acc = 0.0;
    for (int k = hptr[i]; k < hptr[i + 1]; ++k) acc = fma(hval[k], b0full[hcol[k]], acc);
    v = mask[i] ? b0full[i] : F[i] - acc;
  } else {
    const int r = i - n;
    double acc = 0.0;
    for (int k = jptr[r]; k < jptr[r + 1]; ++k) acc = fma(jval[k], b0full[jcol[k]], acc);
    v = fact * F[i] - acc;
  }
  brhs[pos[i]] = v;
}

// ---------------------------------------------------------------- band assembly
// diagonal: lamb (inactive variable), 1 (active variable), -delta (constraint)
__global__ void k_band_set_diag(int n, int m, const int *__restrict__ pos,
                                const uint8_t *__restrict__ mask, double lamb, double delta,
                                double *__restrict__ band, int ldb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + m) return;
  double v;
  if (i < n)
    v = mask[i] ? 1.0 : lamb;
  else
    v = -delta;
  band[(int64_t)pos[i] * ldb] = v;
}

// H entries (lower part in permuted order; slot < 0 marks the mirrored duplicates)
__global__ void k_band_scatter_H(int nnz, const int *__restrict__ row, const int *__restrict__ col,
                                 const double *__restrict__ val, const int *__restrict__ slot,
                                 const uint8_t *__restrict__ mask, double *__restrict__ band) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int s = slot[k];
  if (s < 0) return;
  if (mask[row[k]] || mask[col[k]]) return;
  band[s] += val[k];  // slots are unique per entry; the diagonal was set before this launch
}

__global__ void k_band_scatter_J(int nnz, const int *__restrict__ col,
                                 const double *__restrict__ val, const int *__restrict__ slot,
                                 const uint8_t *__restrict__ mask, double *__restrict__ band) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  if (mask[col[k]]) return;
  band[slot[k]] = val[k];
}

// full-size right-hand side in permuted order:
//   variable i inactive: F_i - (H b0)_i     active: b0_i = dt F_i
//   constraint r       : fact F_{n+r} - (J b0)_r
__global__ void k_band_rhs(int n, int m, const uint8_t *__restrict__ mask,
                           const double *__restrict__ F, const double *__restrict__ b0full,
                           const double *__restrict__ Hb0, const double *__restrict__ Jb0,
                           double fact, const int *__restrict__ pos, double *__restrict__ brhs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + m) return;
  double v;
  if (i < n)
    v = mask[i] ? b0full[i] : F[i] - Hb0[i];
  else
    v = fact * F[i] - Jb0[i - n];
  brhs[pos[i]] = v;
}

// ---------------------------------------------------------------- banded LDL^T + forward solve
__device__ __forceinline__ double recip2(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
}

#define BAND_LDS_DOUBLES 15360  // 120 KB panel of band rows + right-hand side

// One wavefront walks the band.  Panels of P rows are staged in LDS (coalesced loads),
// columns are eliminated one after the other inside the panel (lane <-> pair (r, k) of
// the (bw+1)^2 / 2 window update), consecutive panels overlap by bw rows through global
// memory.  The right-hand side rides along (forward substitution for free).  The chain
// pivot -> reciprocal -> update -> next pivot is inherently serial: this kernel is
// latency-bound by design; the planned successor partitions the band into independent
// segments with a small dense Schur complement.
__global__ __launch_bounds__(64) void k_band_factor(double *__restrict__ band, int ldb, int bw,
                                                    double *__restrict__ rhs, int N,
                                                    int *__restrict__ flags) {
  __shared__ double sm[BAND_LDS_DOUBLES];
  const int lane = threadIdx.x;
  const int P = BAND_LDS_DOUBLES / (ldb + 1);
  double *Bp = sm;            // [P][ldb]
  double *rp = sm + P * ldb;  // [P]
  // lane -> pair (r, k), 1 <= k <= r <= bw
  const int npairs = bw * (bw + 1) / 2;
  int r = 0, k = 0;
  if (lane < npairs) {
    int t = lane;
    r = 1;
    while (t >= r) {
      t -= r;
      ++r;
    }
    k = t + 1;
  }
  const bool act = lane < npairs;
  int neg = 0, bad = 0;
  const int stepP = P - bw;
  for (int s0 = 0; s0 < N; s0 += stepP) {
    const int rows = min(P, N - s0);
    // stage rows [s0, s0 + rows) (+ zero padding up to P)
    for (int idx = lane; idx < P * ldb; idx += 64) {
      const int rr = idx / ldb;
      Bp[idx] = (rr < rows) ? band[(int64_t)s0 * ldb + idx] : 0.0;
    }
    for (int idx = lane; idx < P; idx += 64) rp[idx] = (idx < rows) ? rhs[s0 + idx] : 0.0;
    __syncthreads();
    // columns whose whole window lies in the panel (all remaining ones in the last panel)
    const bool last = (s0 + P >= N);
    const int ncols = last ? rows : stepP;
    for (int j = 0; j < ncols; ++j) {
      const double d = Bp[j * ldb];
      const bool isbad = (d == 0.0) || !(fabs(d) <= 1.79e308);
      const double di = isbad ? 0.0 : recip2(d);
      bad |= isbad ? 1 : 0;
      neg += (d < 0.0) ? 1 : 0;
      if (act && j + r < P) {
        const double cr = Bp[(j + r) * ldb + r];
        const double ck = Bp[(j + k) * ldb + k];
        const double l = cr * di;
        double t = Bp[(j + r) * ldb + (r - k)];
        t = fma(-l, ck, t);
        double rr = 0.0;
        if (k == r) rr = fma(-l, rp[j], rp[j + r]);
        Bp[(j + r) * ldb + (r - k)] = t;
        if (k == r) rp[j + r] = rr;
        if (k == 1) Bp[(j + r) * ldb + r] = l;  // L entry (column read by all lanes above)
      }
    }
    __syncthreads();
    for (int idx = lane; idx < rows * ldb; idx += 64) band[(int64_t)s0 * ldb + idx] = Bp[idx];
    for (int idx = lane; idx < rows; idx += 64) rhs[s0 + idx] = rp[idx];
    __threadfence();  // the next panel re-reads the bw overlapping rows from memory
    __syncthreads();
    if (last) break;
  }
  if (lane == 0) {
    if (bad) atomicOr(&flags[0], 1);
    if (neg) atomicAdd(&flags[1], neg);
  }
}

// y <- L^-1 z with the stored factor (back-solve steps that reuse a factorisation):
// mirror image of k_band_backsolve, walking forward.
__global__ __launch_bounds__(64) void k_band_fwdsolve(const double *__restrict__ band, int ldb,
                                                      int bw, double *__restrict__ z, int N) {
  __shared__ double sm[BAND_LDS_DOUBLES];
  const int lane = threadIdx.x;
  const int P = BAND_LDS_DOUBLES / (ldb + 1);
  double *Bp = sm;
  double *zp = sm + P * ldb;
  const int rr = lane + 1;
  const int stepP = P - bw;
  for (int s0 = 0; s0 < N; s0 += stepP) {
    const int rows = min(P, N - s0);
    for (int idx = lane; idx < rows * ldb; idx += 64) Bp[idx] = band[(int64_t)s0 * ldb + idx];
    for (int idx = lane; idx < rows; idx += 64) zp[idx] = z[s0 + idx];
    __syncthreads();
    const bool last = (s0 + P >= N);
    const int ncols = last ? rows : stepP;
    for (int j = 0; j < ncols; ++j) {
      const double yj = zp[j];
      if (rr <= bw && j + rr < rows) zp[j + rr] = fma(-Bp[(j + rr) * ldb + rr], yj, zp[j + rr]);
    }
    __syncthreads();
    for (int idx = lane; idx < rows; idx += 64) z[s0 + idx] = zp[idx];
    __threadfence();
    __syncthreads();
    if (last) break;
  }
}

// z <- D^-1 z
__global__ void k_band_scale(double *__restrict__ z, const double *__restrict__ band, int ldb,
                             int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) z[i] = z[i] / band[(int64_t)i * ldb];
}

// x <- L^-T z, sequential from the last row (right-looking: once x_i is final it is folded
// into the bw entries above it, lane <-> distance r).  Panels of P rows in LDS, walked from
// the end; a panel finalises and propagates its rows [bw, P) and hands its top bw rows
// (complete, but not yet propagated upwards) to the next panel through global memory.
__global__ __launch_bounds__(64) void k_band_backsolve(const double *__restrict__ band, int ldb,
                                                       int bw, double *__restrict__ z, int N) {
  __shared__ double sm[BAND_LDS_DOUBLES];
  const int lane = threadIdx.x;
  const int P = BAND_LDS_DOUBLES / (ldb + 1);
  double *Bp = sm;
  double *zp = sm + P * ldb;
  const int rr = lane + 1;
  int e0 = N;
  while (e0 > 0) {
    const int s0 = max(0, e0 - P);
    const int rows = e0 - s0;
    for (int idx = lane; idx < rows * ldb; idx += 64) Bp[idx] = band[(int64_t)s0 * ldb + idx];
    for (int idx = lane; idx < rows; idx += 64) zp[idx] = z[s0 + idx];
    __syncthreads();
    const int lo = (s0 == 0) ? 0 : bw;
    for (int i = rows - 1; i >= lo; --i) {
      const double xi = zp[i];
      if (rr <= bw && i - rr >= 0) zp[i - rr] = fma(-Bp[i * ldb + rr], xi, zp[i - rr]);
    }
    __syncthreads();
    for (int idx = lane; idx < rows; idx += 64) z[s0 + idx] = zp[idx];
    __threadfence();
    __syncthreads();
    if (s0 == 0) break;
    e0 = s0 + bw;
  }
}

// full-length vector <-> permuted band order (LinearSolver.solve against the banded factor)
__global__ void k_band_permute(int N, const int *__restrict__ pos, const double *__restrict__ in,
                               double *__restrict__ out, int gather) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (gather) out[i] = in[pos[i]];
  else out[pos[i]] = in[i];
}

// ---------------------------------------------------------------- step update (a9, a15, a16)
__global__ __launch_bounds__(256) void k_band_step_update(
    int n, int m, const int *__restrict__ pos, const double *__restrict__ sol, double fact,
    double rho, const double *__restrict__ x, const double *__restrict__ y,
    const double *__restrict__ lb, const double *__restrict__ ub, const double *__restrict__ F,
    double *__restrict__ dx, double *__restrict__ dy, double *__restrict__ xn,
    double *__restrict__ yn, double *__restrict__ red) {
  __shared__ double part[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  if (i < n) {
    double d = sol[pos[i]];
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
    const double t = rho * F[i];
    const double d = fact * (sol[pos[i]] - t);
    dy[r] = d;
    yn[r] = y[r] - d;
    sq = d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) red[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// ---------------------------------------------------------------- launch wrappers
static inline dim3 g1(int n, int b = 256) { return dim3((n + b - 1) / b); }

void sp_launch_spmv(hipStream_t s, int rows, const int *ptr, const int *col, const double *val,
                    const double *x, const double *add, double sgn, double *y) {
  if (rows) hipLaunchKernelGGL(k_csr_spmv, g1(rows), dim3(256), 0, s, rows, ptr, col, val, x, add, sgn, y);
}

void sp_launch_spmvT(hipStream_t s, int cols, const int *tptr, const int *trow, const int *tmap,
                     const double *val, const double *w, const double *base, double *out) {
  if (cols)
    hipLaunchKernelGGL(k_csc_spmvT, g1(cols), dim3(256), 0, s, cols, tptr, trow, tmap, val, w, base, out);
}

void sp_launch_eval(hipStream_t s, const SparseDev &sp, int n, int m, const double *x, const double *y,
                    const double *b, const double *q, double rho, double *c, double *w, double *g) {
  if (m) hipLaunchKernelGGL(k_sp_eval_c, g1(m), dim3(256), 0, s, m, sp.Jptr, sp.Jcol, sp.Jval, x, b, rho, y, c, w);
  if (n)
    hipLaunchKernelGGL(k_sp_eval_g, g1(n), dim3(256), 0, s, n, sp.Hptr, sp.Hcol, sp.Hval, sp.JTptr, sp.JTrow,
                       sp.JTmap, sp.Jval, x, w, q, g);
}

void sp_launch_assemble(hipStream_t s, const SparseDev &sp, int n, int m, const uint8_t *mask,
                        double lamb, double delta) {
  const int N = n + m;
  (void)hipMemsetAsync(sp.band, 0, (size_t)(N + 1) * sp.ldb * sizeof(double), s);
  hipLaunchKernelGGL(k_band_set_diag, g1(N), dim3(256), 0, s, n, m, sp.pos, mask, lamb, delta,
                     sp.band, sp.ldb);
  if (sp.nnzH)
    hipLaunchKernelGGL(k_band_scatter_H, g1(sp.nnzH), dim3(256), 0, s, sp.nnzH, sp.Hrow, sp.Hcol,
                       sp.Hval, sp.Hslot, mask, sp.band);
  if (sp.nnzJ)
    hipLaunchKernelGGL(k_band_scatter_J, g1(sp.nnzJ), dim3(256), 0, s, sp.nnzJ, sp.Jcol, sp.Jval,
                       sp.Jslot, mask, sp.band);
}

void sp_launch_rhs(hipStream_t s, const SparseDev &sp, int n, int m, const uint8_t *mask,
                   const double *F, const double *b0full, double fact, double *Hb0, double *Jb0) {
  (void)Hb0;
  (void)Jb0;
  if (n + m)
    hipLaunchKernelGGL(k_band_rhs_fused, g1(n + m), dim3(256), 0, s, n, m, mask, F, b0full, sp.Hptr, sp.Hcol,
                       sp.Hval, sp.Jptr, sp.Jcol, sp.Jval, fact, sp.pos, sp.brhs);
}

void sp_launch_permute(hipStream_t s, const SparseDev &sp, int N, const double *in, double *out,
                       int gather) {
  if (N) hipLaunchKernelGGL(k_band_permute, g1(N), dim3(256), 0, s, N, sp.pos, in, out, gather);
}

void sp_launch_factor(hipStream_t s, const SparseDev &sp, int N, int *flags) {
  (void)hipMemsetAsync(flags, 0, 4 * sizeof(int), s);
  hipLaunchKernelGGL(k_band_factor, dim3(1), dim3(64), 0, s, sp.band, sp.ldb, sp.bw, sp.brhs, N, flags);
}

void sp_launch_fwdsolve(hipStream_t s, const SparseDev &sp, int N) {
  if (N == 0) return;
  hipLaunchKernelGGL(k_band_fwdsolve, dim3(1), dim3(64), 0, s, sp.band, sp.ldb, sp.bw, sp.brhs, N);
}

void sp_launch_backsolve(hipStream_t s, const SparseDev &sp, int N) {
  if (N == 0) return;
  hipLaunchKernelGGL(k_band_scale, g1(N), dim3(256), 0, s, sp.brhs, sp.band, sp.ldb, N);
  hipLaunchKernelGGL(k_band_backsolve, dim3(1), dim3(64), 0, s, sp.band, sp.ldb, sp.bw, sp.brhs, N);
}

void sp_launch_step_update(hipStream_t s, const SparseDev &sp, int n, int m, double fact,
                           double rho, const double *x, const double *y, const double *lb,
                           const double *ub, const double *F, double *dx, double *dy, double *xn,
                           double *yn, double *red) {
  const int nb = (n + m + 255) / 256;
  if (nb)
    hipLaunchKernelGGL(k_band_step_update, dim3(nb), dim3(256), 0, s, n, m, sp.pos, sp.brhs, fact,
                       rho, x, y, lb, ub, F, dx, dy, xn, yn, red);
}

// ================================================================= block cyclic reduction
// The permuted KKT matrix has half-bandwidth bw <= 8, i.e. it is block tridiagonal with
// 8 x 8 blocks: D_i (diagonal), L_i (coupling to the left neighbour), U_i = L_{i+1}^T.
// Cyclic reduction eliminates every other block row per level (all of them in parallel):
//   kept block i, eliminated neighbours i-s, i+s:
//     alpha = L_i inv(D_{i-s}),  gamma = U_i inv(D_{i+s})
//     D_i -= alpha U_{i-s} + gamma L_{i+s};  f_i -= alpha f_{i-s} + gamma f_{i+s}
//     L_i <- -alpha L_{i-s}  (now couples to i-2s);  U_i <- -gamma U_{i+s}
// log2(N/8) levels instead of N sequential pivots; back-substitution walks the levels in
// reverse.  Every principal block and Schur complement of a symmetric quasi-definite matrix
// is again quasi-definite, so the 8 x 8 pivots D_i are invertible without pivoting, and by
// Haynsworth's inertia additivity the number of negative eigenvalues of K is the sum of the
// negative pivots met while inverting the D_i.  One wavefront per block, lane <-> (row, col).
#define BCR_B 8

// block extraction from the band (+ identity padding of the last block)
__global__ __launch_bounds__(64) void k_bcr_extract(const double *__restrict__ band, int ldb, int bw,
                                                    const double *__restrict__ rhs, int N, int nb,
                                                    double *__restrict__ D, double *__restrict__ L,
                                                    double *__restrict__ U, double *__restrict__ F,
                                                    double *__restrict__ rhs0,
                                                    int *__restrict__ flags) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i == 0 && lane < 4) flags[lane] = 0;  // (every kernel that sets them runs after this one)
  // accuracy guard (k_band_residual): the right-hand side survives the solve in rhs0
  if (rhs0 && lane < 8 && i * 8 + lane < N) rhs0[i * 8 + lane] = rhs[i * 8 + lane];
  const int r = lane >> 3, c = lane & 7;
  const int gr = i * 8 + r, gc = i * 8 + c;
  // diagonal block, symmetric fill
  double d = (r == c) ? 1.0 : 0.0;
  if (gr < N && gc < N) {
    const int hi = max(gr, gc), lo = min(gr, gc);
    d = (hi - lo <= bw) ? band[(int64_t)hi * ldb + (hi - lo)] : 0.0;
  }
  D[(int64_t)i * 64 + lane] = d;
  // L_i[r][c] = K[8i + r][8(i-1) + c]
  double l = 0.0;
  if (i > 0 && gr < N) {
    const int cc = (i - 1) * 8 + c;
    const int dist = gr - cc;
    if (dist <= bw) l = band[(int64_t)gr * ldb + dist];
  }
  L[(int64_t)i * 64 + lane] = l;
  // U_i[r][c] = K[8i + r][8(i+1) + c] = K[8(i+1) + c][8i + r]
  double u = 0.0;
  if (i + 1 < nb && gr < N) {
    const int rr = (i + 1) * 8 + c;
    const int dist = rr - gr;
    if (rr < N && dist <= bw) u = band[(int64_t)rr * ldb + dist];
  }
  U[(int64_t)i * 64 + lane] = u;
  if (lane < 8) F[(int64_t)i * 8 + lane] = (i * 8 + lane < N) ? rhs[i * 8 + lane] : 0.0;
}

// in-place Gauss-Jordan inverse of the 8 x 8 block in LDS (one wavefront, lane = (r, c));
// returns the number of negative pivots, sets *bad on a zero / non-finite pivot
__device__ __forceinline__ int gj_inverse8(double *M, int lane, int *bad) {
  const int r = lane >> 3, c = lane & 7;
  int neg = 0;
  double mrc = M[lane];  // own entry: stays in a register between the steps
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double p = M[k * 8 + k];
    const bool isbad = (p == 0.0) || !(fabs(p) <= 1.79e308);
    *bad |= isbad ? 1 : 0;
    neg += (p < 0.0) ? 1 : 0;
    // v_rcp_f64 + two Newton steps (full precision) instead of the ~30-instruction IEEE
    // division: eight of them sat on every block operation's dependent chain
    const double d = isbad ? 0.0 : recip2(p);
    const double mrk = M[r * 8 + k], mkc = M[k * 8 + c];
    double v;
    if (r == k && c == k)
      v = d;
    else if (r == k)
      v = mkc * d;
    else if (c == k)
      v = -mrk * d;
    else
      v = fma(-mrk * d, mkc, mrc);
    M[lane] = v;  // all lanes have read before any lane writes (one wavefront, lockstep)
    mrc = v;
  }
  return neg;
}

// ---- per-block work of one wavefront (lane = threadIdx.x & 63); `sm` is the wavefront's own
// LDS scratch of BCR_SCRATCH doubles.  Used by the one-block-per-workgroup kernels of the
// large levels and by the fused tail kernel below.
#define BCR_SCRATCH (3 * 64 + 8)

// The negative-pivot count goes to negcnt[gi] (gi = the block's global index), NOT to one
// atomic counter: in a KKT matrix nearly every block has negative pivots, and 9 375
// workgroups adding to the same word serialised in L2 (110 us for the first level alone).
__device__ __forceinline__ void bcr_invert_block(const double *__restrict__ D,
                                                 double *__restrict__ Dinv, int i, int lane,
                                                 double *sm, int *__restrict__ flags,
                                                 int *__restrict__ negcnt, int gi) {
  double *M = sm;
  M[lane] = D[(int64_t)i * 64 + lane];
  int bad = 0;
  const int neg = gj_inverse8(M, lane, &bad);
  Dinv[(int64_t)i * 64 + lane] = M[lane];
  if (lane == 0) {
    if (bad) atomicOr(&flags[0], 1);
    negcnt[gi] = neg;
  }
}

// 8 x 8 product helper: out[r][c] = sum_k A[r][k] B[k][c], operands in LDS
__device__ __forceinline__ double mm8(const double *A, const double *B, int r, int c) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = fma(A[r * 8 + k], B[k * 8 + c], acc);
  return acc;
}

// kept block i of the level with stride s (neighbours i - s, i + s are eliminated)
__device__ __forceinline__ void bcr_reduce_block(double *__restrict__ D, double *__restrict__ L,
                                                 double *__restrict__ U, double *__restrict__ F,
                                                 const double *__restrict__ Dinv, int nb, int s,
                                                 int i, int lane, double *sm) {
  double *A = sm, *B = sm + 64, *T = sm + 128, *fs = sm + 192;
  const int r = lane >> 3, c = lane & 7;
  double dv = D[(int64_t)i * 64 + lane];
  double fv = (lane < 8) ? F[(int64_t)i * 8 + lane] : 0.0;
  double lnew = 0.0, unew = 0.0;
  const int le = i - s, ri = i + s;
  if (le >= 0) {
    A[lane] = L[(int64_t)i * 64 + lane];
    B[lane] = Dinv[(int64_t)le * 64 + lane];
    const double al = mm8(A, B, r, c);  // alpha = L_i inv(D_left)
    T[lane] = al;
    B[lane] = U[(int64_t)le * 64 + lane];
    if (lane < 8) fs[lane] = F[(int64_t)le * 8 + lane];
    dv -= mm8(T, B, r, c);              // D_i -= alpha U_left
    if (lane < 8) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fma(T[lane * 8 + k], fs[k], acc);
      fv -= acc;                        // f_i -= alpha f_left
    }
    B[lane] = L[(int64_t)le * 64 + lane];
    lnew = -mm8(T, B, r, c);            // couples i to i - 2s
  }
  if (ri < nb) {
    A[lane] = U[(int64_t)i * 64 + lane];
    B[lane] = Dinv[(int64_t)ri * 64 + lane];
    const double ga = mm8(A, B, r, c);  // gamma = U_i inv(D_right)
    T[lane] = ga;
    B[lane] = L[(int64_t)ri * 64 + lane];
    if (lane < 8) fs[lane] = F[(int64_t)ri * 8 + lane];
    dv -= mm8(T, B, r, c);              // D_i -= gamma L_right
    if (lane < 8) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fma(T[lane * 8 + k], fs[k], acc);
      fv -= acc;
    }
    B[lane] = U[(int64_t)ri * 64 + lane];
    unew = -mm8(T, B, r, c);            // couples i to i + 2s
  }
  D[(int64_t)i * 64 + lane] = dv;
  L[(int64_t)i * 64 + lane] = lnew;
  U[(int64_t)i * 64 + lane] = unew;
  if (lane < 8) F[(int64_t)i * 8 + lane] = fv;
}

// One level in ONE launch: kept block i inverts its two eliminated neighbours itself (each
// eliminated block is inverted twice, by its left and by its right kept neighbour -- 2 x 1.5 us
// of redundant work against a dependent launch of ~5 us per level) and then reduces.  The
// left kept neighbour of an eliminated block always exists, so IT stores the inverse for the
// back-substitution and reports the block's pivots.
// Result to (Do, Lo, Uo, Fo)[io] -- the block itself for the one-level launches, an LDS slot
// or the other block set for the two-level ones; `report`: store the right neighbour's inverse
// and pivot count (exactly one caller per eliminated block does).  (io0, gr): index of the
// right neighbour in Dinv / negcnt (differs from ri when the inputs are LDS slots).
__device__ __forceinline__ void bcr_level_block(const double *D, const double *L, const double *U,
                                                const double *F, double *__restrict__ Dinv, int nb,
                                                int s, int i, int lane, double *sm,
                                                int *__restrict__ flags, int *__restrict__ negcnt,
                                                double *Do, double *Lo, double *Uo, double *Fo, int io,
                                                bool report, int gr) {
  double *A = sm, *B = sm + 64, *T = sm + 128, *fs = sm + 192;
  const int r = lane >> 3, c = lane & 7;
  double dv = D[(int64_t)i * 64 + lane];
  double fv = (lane < 8) ? F[(int64_t)i * 8 + lane] : 0.0;
  double lnew = 0.0, unew = 0.0;
  const int le = i - s, ri = i + s;
  if (le >= 0) {
    A[lane] = L[(int64_t)i * 64 + lane];
    B[lane] = D[(int64_t)le * 64 + lane];
    int bad = 0;
    (void)gj_inverse8(B, lane, &bad);   // reported by the kept block to the left of `le`
    const double al = mm8(A, B, r, c);  // alpha = L_i inv(D_left)
    T[lane] = al;
    B[lane] = U[(int64_t)le * 64 + lane];
    if (lane < 8) fs[lane] = F[(int64_t)le * 8 + lane];
    dv -= mm8(T, B, r, c);              // D_i -= alpha U_left
    if (lane < 8) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fma(T[lane * 8 + k], fs[k], acc);
      fv -= acc;                        // f_i -= alpha f_left
    }
    B[lane] = L[(int64_t)le * 64 + lane];
    lnew = -mm8(T, B, r, c);            // couples i to i - 2s
  }
  if (ri < nb) {
    A[lane] = U[(int64_t)i * 64 + lane];
    B[lane] = D[(int64_t)ri * 64 + lane];
    int bad = 0;
    const int neg = gj_inverse8(B, lane, &bad);
    if (report) {
      Dinv[(int64_t)gr * 64 + lane] = B[lane];
      if (lane == 0) {
        if (bad) atomicOr(&flags[0], 1);
        negcnt[gr] = neg;
      }
    }
    const double ga = mm8(A, B, r, c);  // gamma = U_i inv(D_right)
    T[lane] = ga;
    B[lane] = L[(int64_t)ri * 64 + lane];
    if (lane < 8) fs[lane] = F[(int64_t)ri * 8 + lane];
    dv -= mm8(T, B, r, c);              // D_i -= gamma L_right
    if (lane < 8) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fma(T[lane * 8 + k], fs[k], acc);
      fv -= acc;
    }
    B[lane] = U[(int64_t)ri * 64 + lane];
    unew = -mm8(T, B, r, c);            // couples i to i + 2s
  }
  Do[(int64_t)io * 64 + lane] = dv;
  Lo[(int64_t)io * 64 + lane] = lnew;
  Uo[(int64_t)io * 64 + lane] = unew;
  if (lane < 8) Fo[(int64_t)io * 8 + lane] = fv;
}

// x_i = inv(D_i) (f_i - L_i x_{i-s} - U_i x_{i+s});  s == 0: the last remaining block
__device__ __forceinline__ void bcr_back_block(const double *__restrict__ Dinv,
                                               const double *__restrict__ L,
                                               const double *__restrict__ U,
                                               const double *__restrict__ F,
                                               double *__restrict__ X, int nb, int s, int i,
                                               int lane, double *sm) {
  double *t = sm, *xl = sm + 8, *xr = sm + 16;
  const int le = i - s, ri = i + s;
  const bool hl = (s > 0) && le >= 0, hr = (s > 0) && ri < nb;
  if (lane < 8) {
    // L1-bypassing loads: in the two-level launch x of a neighbour comes from another
    // wavefront of the same workgroup a moment ago
    xl[lane] = hl ? __hip_atomic_load(X + (int64_t)le * 8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    xr[lane] = hr ? __hip_atomic_load(X + (int64_t)ri * 8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  }
  if (lane < 8) {
    double acc = F[(int64_t)i * 8 + lane];
    if (hl)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fma(-L[(int64_t)i * 64 + lane * 8 + k], xl[k], acc);
    if (hr)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = fma(-U[(int64_t)i * 64 + lane * 8 + k], xr[k], acc);
    t[lane] = acc;
  }
  if (lane < 8) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fma(Dinv[(int64_t)i * 64 + lane * 8 + k], t[k], acc);
    X[(int64_t)i * 8 + lane] = acc;
  }
}

// invert the blocks eliminated at this level: i = s, 3s, 5s, ... (i mod 2s == s)
__global__ __launch_bounds__(64) void k_bcr_invert(const double *__restrict__ D,
                                                   double *__restrict__ Dinv, int nb, int s,
                                                   int first, int stride, int *__restrict__ flags,
                                                   int *__restrict__ negcnt) {
  __shared__ double sm[BCR_SCRATCH];
  const int i = first + blockIdx.x * stride;
  if (i >= nb) return;
  bcr_invert_block(D, Dinv, i, threadIdx.x, sm, flags, negcnt, i);
  (void)s;
}

// reduce the kept blocks of this level: i = 0, 2s, 4s, ...
__global__ __launch_bounds__(64) void k_bcr_reduce(double *__restrict__ D, double *__restrict__ L,
                                                   double *__restrict__ U, double *__restrict__ F,
                                                   const double *__restrict__ Dinv, int nb, int s) {
  __shared__ double sm[BCR_SCRATCH];
  const int i = blockIdx.x * 2 * s;
  if (i >= nb) return;
  bcr_reduce_block(D, L, U, F, Dinv, nb, s, i, threadIdx.x, sm);
}

// invert + reduce of one level (blocks 0, 2s, 4s, ... are kept)
__global__ __launch_bounds__(64) void k_bcr_level(double *__restrict__ D, double *__restrict__ L,
                                                  double *__restrict__ U, double *__restrict__ F,
                                                  double *__restrict__ Dinv, int nb, int s,
                                                  int *__restrict__ flags,
                                                  int *__restrict__ negcnt) {
  __shared__ double sm[BCR_SCRATCH];
  const int i = blockIdx.x * 2 * s;
  if (i >= nb) return;
  bcr_level_block(D, L, U, F, Dinv, nb, s, i, threadIdx.x, sm, flags, negcnt, D, L, U, F, i, true,
                  i + s);
}

// TWO levels (strides s and 2 s) in one launch: the workgroup of block k = 4 s j keeps k through
// both.  Its three wavefronts reduce k - 2 s, k and k + 2 s by one level side by side (from the
// input set into LDS slots; the outer two are also reduced by the neighbouring workgroups: a
// level of this size is bound by its launch, not by its arithmetic), then wavefront 0 reduces k
// by the second level from the slots.  Nothing is updated in place -- the neighbours still
// read the input set -- so the results go to the OTHER block set: k's new blocks and, for the
// back-substitution, the level-one blocks of k + 2 s (eliminated at the second level; k - 2 s
// is the left neighbour's k + 2 s).  Inverses and pivot counts of the eliminated blocks
// k + s, k + 3 s (first level) and k + 2 s (second level) are this workgroup's to report.
__global__ __launch_bounds__(192) void k_bcr_level2(const double *__restrict__ Di,
                                                    const double *__restrict__ Li,
                                                    const double *__restrict__ Ui,
                                                    const double *__restrict__ Fi,
                                                    double *__restrict__ Do, double *__restrict__ Lo,
                                                    double *__restrict__ Uo, double *__restrict__ Fo,
                                                    double *__restrict__ Dinv, int nb, int s,
                                                    int *__restrict__ flags,
                                                    int *__restrict__ negcnt) {
  __shared__ double sm[3][BCR_SCRATCH];
  __shared__ double Ds[3 * 64], Ls[3 * 64], Us[3 * 64], Fs[3 * 8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = (int)blockIdx.x * 4 * s;
  if (k >= nb) return;
  const int i = k + (w - 1) * 2 * s;  // k - 2 s, k, k + 2 s
  if (i >= 0 && i < nb)
    bcr_level_block(Di, Li, Ui, Fi, Dinv, nb, s, i, lane, sm[w], flags, negcnt, Ds, Ls, Us, Fs, w,
                    /*report=*/w >= 1, i + s);
  __syncthreads();
  if (w == 0) {
    // second level on the slots: slot 1 = k, its neighbours slots 0 and 2 where they exist
    const int lo = (k - 2 * s >= 0) ? 0 : 1;
    const int cnt = ((k + 2 * s < nb) ? 3 : 2) - lo;
    bcr_level_block(Ds + lo * 64, Ls + lo * 64, Us + lo * 64, Fs + lo * 8, Dinv, cnt, 1, 1 - lo, lane,
                    sm[0], flags, negcnt, Do, Lo, Uo, Fo, k, /*report=*/true, k + 2 * s);
  } else if (w == 2 && k + 2 * s < nb) {
    // what the back-substitution of k + 2 s needs (its inverse comes from wavefront 0)
    const int64_t g = (int64_t)(k + 2 * s);
    Lo[g * 64 + lane] = Ls[2 * 64 + lane];
    Uo[g * 64 + lane] = Us[2 * 64 + lane];
    if (lane < 8) Fo[g * 8 + lane] = Fs[2 * 8 + lane];
  }
}

// back-substitution of the blocks eliminated at this level (i mod 2s == s)
__global__ __launch_bounds__(64) void k_bcr_back(const double *__restrict__ Dinv,
                                                 const double *__restrict__ L,
                                                 const double *__restrict__ U,
                                                 const double *__restrict__ F,
                                                 double *__restrict__ X, int nb, int s, int first,
                                                 int stride) {
  __shared__ double sm[BCR_SCRATCH];
  const int i = first + blockIdx.x * stride;
  if (i >= nb) return;
  bcr_back_block(Dinv, L, U, F, X, nb, s, i, threadIdx.x, sm);
}

// Back-substitution of TWO levels in one launch (the mirror of k_bcr_level2): the workgroup of
// block j = 2 s (mod 4 s) solves j at stride 2 s (blocks of set 2: level-one values), then its
// two wavefronts solve j - s and j + s at stride s (set 1: the input set of that launch), which
// only need x_j and the already known x_{j -+ 2 s}.  A workgroup whose j lies beyond the last
// block still owns j - s.
__global__ __launch_bounds__(128) void k_bcr_back2(const double *__restrict__ Dinv,
                                                   const double *__restrict__ L1,
                                                   const double *__restrict__ U1,
                                                   const double *__restrict__ F1,
                                                   const double *__restrict__ L2,
                                                   const double *__restrict__ U2,
                                                   const double *__restrict__ F2,
                                                   double *__restrict__ X, int nb, int s) {
  __shared__ double sm[2][BCR_SCRATCH];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = 2 * s + (int)blockIdx.x * 4 * s;
  if (j - s >= nb) return;
  if (w == 0 && j < nb) {
    bcr_back_block(Dinv, L2, U2, F2, X, nb, 2 * s, j, lane, sm[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // x_j has reached L2 before the barrier
  }
  __syncthreads();
  const int i = j + (w == 0 ? -s : s);
  if (i < nb) bcr_back_block(Dinv, L1, U1, F1, X, nb, s, i, lane, sm[w]);
}

// All levels from stride s0 upwards, the last block, and the matching back-substitution
// levels in ONE workgroup: at most BCR_TAIL_BLOCKS blocks are still in play there, every level
// was two dependent launches of a few microseconds each (launch-bound), and the whole
// remaining system fits in LDS.  The in-play blocks (global index j * s0) are copied to LDS
// in compact order, reduced with strides 1, 2, 4, ... of the compact index by 16 wavefronts
// taking blocks round-robin (a workgroup barrier between the invert / reduce / back phases),
// and only their solution X goes back to HBM for the lower back-substitution levels.
// (A first version ran the tail on the global arrays with up to 512 blocks: every block
// operation is a chain of dependent HBM round trips, one CU working through them was 3x
// slower than the launches it replaced.)
#define BCR_TAIL_BLOCKS 32
#define BCR_PAIR_MAX 4096  // blocks in play up to which two levels share a launch
__global__ __launch_bounds__(1024) void k_bcr_tail(const double *__restrict__ D,
                                                   const double *__restrict__ L,
                                                   const double *__restrict__ U,
                                                   const double *__restrict__ F,
                                                   double *__restrict__ X, int nb, int s0,
                                                   int *__restrict__ flags,
                                                   int *__restrict__ negcnt) {
  __shared__ double Dl[BCR_TAIL_BLOCKS * 64], Ll[BCR_TAIL_BLOCKS * 64], Ul[BCR_TAIL_BLOCKS * 64];
  __shared__ double Il[BCR_TAIL_BLOCKS * 64], Fl[BCR_TAIL_BLOCKS * 8], Xl[BCR_TAIL_BLOCKS * 8];
  __shared__ double smem[16 * BCR_SCRATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *sm = smem + wave * BCR_SCRATCH;
  const int nc = (nb + s0 - 1) / s0;  // compact block count (<= BCR_TAIL_BLOCKS)
  for (int p = tid; p < nc * 64; p += 1024) {
    const int64_t g = (int64_t)(p >> 6) * s0 * 64 + (p & 63);
    Dl[p] = D[g];
    Ll[p] = L[g];
    Ul[p] = U[g];
  }
  for (int p = tid; p < nc * 8; p += 1024) Fl[p] = F[(int64_t)(p >> 3) * s0 * 8 + (p & 7)];
  __syncthreads();
  int top = 0;
  for (int st = 1; st < nc; st *= 2) {
    const int ne = (nc - st + 2 * st - 1) / (2 * st);
    for (int e = wave; e < ne; e += 16) {
      const int j = st + e * 2 * st;
      bcr_invert_block(Dl, Il, j, lane, sm, flags, negcnt, j * s0);
    }
    __syncthreads();
    const int nk = (nc + 2 * st - 1) / (2 * st);
    for (int k = wave; k < nk; k += 16)
      bcr_reduce_block(Dl, Ll, Ul, Fl, Il, nc, st, k * 2 * st, lane, sm);
    __syncthreads();
    top = st;
  }
  if (wave == 0) {
    bcr_invert_block(Dl, Il, 0, lane, sm, flags, negcnt, 0);
    bcr_back_block(Il, Ll, Ul, Fl, Xl, nc, 0, 0, lane, sm);
  }
  __syncthreads();
  for (int st = top; st >= 1; st /= 2) {
    const int ne = (nc - st + 2 * st - 1) / (2 * st);
    for (int e = wave; e < ne; e += 16)
      bcr_back_block(Il, Ll, Ul, Fl, Xl, nc, st, st + e * 2 * st, lane, sm);
    __syncthreads();
  }
  for (int p = tid; p < nc * 8; p += 1024) X[(int64_t)(p >> 3) * s0 * 8 + (p & 7)] = Xl[p];
  // inertia: every block has been inverted exactly once by now (earlier launches or above)
  __shared__ int part[16];
  int cnt = 0;
  for (int i = tid; i < nb; i += 1024) cnt += negcnt[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if (lane == 0) part[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += part[w];
    flags[1] = tot;
  }
}

__global__ void k_bcr_scatter(const double *__restrict__ X, double *__restrict__ out, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = X[i];
}

// Accuracy guard of the banded path: r = rhs0 - K x with K read from the assembled band (cyclic
// reduction leaves it intact; lower band, row i holds K[i][i - d] at [i][d]), max |r| and
// max (|K| |x| + |rhs0|) per workgroup.
__global__ __launch_bounds__(256) void k_band_residual(const double *__restrict__ band, int ldb, int bw,
                                                       int N, const double *__restrict__ x,
                                                       const double *__restrict__ rhs0,
                                                       double *__restrict__ r,
                                                       double *__restrict__ rsmax,
                                                       const int *__restrict__ flags, int nred) {
  // the pivot flags of the solve ride along behind the pairs and the step update's partial sums
  // (one device-to-host copy for all three, pgf_api.hip)
  if (blockIdx.x == 0 && threadIdx.x < 4) rsmax[3 * nred + threadIdx.x] = (double)flags[threadIdx.x];
  // rows [i0, i0 + 256 + bw) of the band (contiguous in memory: coalesced) and x[i0 - bw,
  // i0 + 256 + bw) staged in LDS: row i needs its own band row and, for the upper triangle, the
  // rows i + 1 .. i + bw of its neighbours (bw <= 10, ldb <= 12)
  __shared__ double bs[(256 + 10) * 12];
  __shared__ double xs[256 + 20];
  __shared__ double pr[4], pb[4];
  const int i0 = blockIdx.x * 256, tid = threadIdx.x;
  const int nrow = min(256 + bw, N - i0);
  for (int p = tid; p < nrow * ldb; p += 256) bs[p] = band[(int64_t)i0 * ldb + p];
  for (int p = tid; p < 256 + 2 * bw; p += 256) {
    const int g = i0 - bw + p;
    xs[p] = (g >= 0 && g < N) ? x[g] : 0.0;
  }
  __syncthreads();
  const int i = i0 + tid;
  // ab: the row's share of the normwise backward error's denominator, (|K| |x| + |rhs|)_i <=
  // ||K|| max |x| + max |rhs| -- a backward-stable solve only guarantees |r| <~ eps |K| |x|, which
  // against max |rhs| alone is eps cond(K) (pgf_api.hip, residual_rel)
  double ar = 0.0, ab = 0.0;
  if (i < N) {
    double acc = rhs0[i];
    ab = fabs(acc);
    for (int d = 0; d <= bw && d <= i; ++d) {
      acc = fma(-bs[tid * ldb + d], xs[bw + tid - d], acc);
      ab += fabs(bs[tid * ldb + d] * xs[bw + tid - d]);
    }
    for (int d = 1; d <= bw && i + d < N; ++d) {
      acc = fma(-bs[(tid + d) * ldb + d], xs[bw + tid + d], acc);
      ab += fabs(bs[(tid + d) * ldb + d] * xs[bw + tid + d]);
    }
    r[i] = acc;
    ar = (acc == acc) ? fabs(acc) : __builtin_huge_val();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ar = fmax(ar, __shfl_down(ar, off));
    ab = fmax(ab, __shfl_down(ab, off));
  }
  if ((threadIdx.x & 63) == 0) {
    pr[threadIdx.x >> 6] = ar;
    pb[threadIdx.x >> 6] = ab;
  }
  __syncthreads();
  // one pair per workgroup, reduced on the host after the step's synchronisation (586 atomics on
  // two words cost 14 us: same-address atomics serialise in L2)
  if (threadIdx.x == 0) {
    rsmax[2 * blockIdx.x] = fmax(fmax(pr[0], pr[1]), fmax(pr[2], pr[3]));
    rsmax[2 * blockIdx.x + 1] = fmax(fmax(pb[0], pb[1]), fmax(pb[2], pb[3]));
  }
}

void sp_launch_band_residual(hipStream_t s, const SparseDev &sp, int N, const int *flags) {
  if (N == 0) return;
  hipLaunchKernelGGL(k_band_residual, g1(N), dim3(256), 0, s, sp.band, sp.ldb, sp.bw, N, sp.brhs, sp.brhs0,
                     sp.bres, sp.bred, flags, sp.nred);
}

// x = saved + correction (refinement step of the guard)
__global__ void k_band_axpy(int N, const double *__restrict__ a, double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) x[i] += a[i];
}
void sp_launch_band_axpy(hipStream_t s, int N, const double *a, double *x) {
  if (N) hipLaunchKernelGGL(k_band_axpy, g1(N), dim3(256), 0, s, N, a, x);
}

// Solve the banded system in sp.band / sp.brhs by block cyclic reduction; the solution
// replaces sp.brhs.  flags[0] zero pivot, flags[1] negative pivots.
// guard: keep the right-hand side and finish with the residual of the solution (bres, bred);
// a correction solve of the refinement runs without.
void sp_launch_bcr_solve(hipStream_t s, const SparseDev &sp, int N, int *flags, bool guard) {
  if (N == 0) {
    (void)hipMemsetAsync(flags, 0, 4 * sizeof(int), s);
    return;
  }
  const int nb = (N + 7) / 8;
  hipLaunchKernelGGL(k_bcr_extract, dim3(nb), dim3(64), 0, s, sp.band, sp.ldb, sp.bw, sp.brhs, N, nb,
                     sp.bD, sp.bL, sp.bU, sp.bF, guard ? sp.brhs0 : nullptr, flags);
  // levels with many blocks: one workgroup per block; from the first level with at most
  // BCR_TAIL_BLOCKS blocks left: everything in one workgroup, in LDS
  // PGF_BCR_FUSED=0: separate invert / reduce launches per level
  static const bool fused_levels = !(getenv("PGF_BCR_FUSED") && atoi(getenv("PGF_BCR_FUSED")) == 0);
  // PGF_BCR_PAIRS=0: one level per launch throughout (two per launch where two more levels
  // are due before the tail and the level is launch-bound: at most BCR_PAIR_MAX blocks in play)
  static const bool pairs = !(getenv("PGF_BCR_PAIRS") && atoi(getenv("PGF_BCR_PAIRS")) == 0);
  struct Lev {
    int s, set;     // stride; block set holding the eliminated blocks' L, U, F
    int pair_set2;  // >= 0: first level of a fused pair, second level's blocks are in this set
  };
  Lev lev[40];
  int nlev = 0, cur = 0;  // cur: set holding the blocks in play
  auto Dp = [&](int set) { return sp.bD + (size_t)set * sp.bstride * 64; };
  auto Lp = [&](int set) { return sp.bL + (size_t)set * sp.bstride * 64; };
  auto Up = [&](int set) { return sp.bU + (size_t)set * sp.bstride * 64; };
  auto Fp = [&](int set) { return sp.bF + (size_t)set * sp.bstride * 8; };
  int st = 1;
  while (st < nb) {
    const int left = (nb + st - 1) / st;  // blocks still in play before this level
    if (left <= BCR_TAIL_BLOCKS) break;
    const int left2 = (nb + 2 * st - 1) / (2 * st);
    static const int pair_max = getenv("PGF_BCR_PAIR_MAX") ? atoi(getenv("PGF_BCR_PAIR_MAX")) : BCR_PAIR_MAX;
    if (fused_levels && pairs && left <= pair_max && left2 > BCR_TAIL_BLOCKS && 2 * st < nb) {
      const int nk2 = (nb + 4 * st - 1) / (4 * st);  // kept through both: 0, 4st, ...
      hipLaunchKernelGGL(k_bcr_level2, dim3(nk2), dim3(192), 0, s, Dp(cur), Lp(cur), Up(cur), Fp(cur),
                         Dp(cur ^ 1), Lp(cur ^ 1), Up(cur ^ 1), Fp(cur ^ 1), sp.bDinv, nb, st, flags,
                         sp.bneg);
      lev[nlev++] = {st, cur, cur ^ 1};
      cur ^= 1;
      st *= 4;
      continue;
    }
    const int nk = (nb + 2 * st - 1) / (2 * st);       // kept: 0, 2st, ...
    if (fused_levels) {
      hipLaunchKernelGGL(k_bcr_level, dim3(nk), dim3(64), 0, s, Dp(cur), Lp(cur), Up(cur), Fp(cur),
                         sp.bDinv, nb, st, flags, sp.bneg);
    } else {
      const int ne = (nb - st + 2 * st - 1) / (2 * st);  // eliminated: st, 3st, ...
      if (ne > 0)
        hipLaunchKernelGGL(k_bcr_invert, dim3(ne), dim3(64), 0, s, Dp(cur), sp.bDinv, nb, st, st,
                           2 * st, flags, sp.bneg);
      hipLaunchKernelGGL(k_bcr_reduce, dim3(nk), dim3(64), 0, s, Dp(cur), Lp(cur), Up(cur), Fp(cur),
                         sp.bDinv, nb, st);
    }
    lev[nlev++] = {st, cur, -1};
    st *= 2;
  }
  // st: first level NOT done above (st >= nb: only the last block is left)
  hipLaunchKernelGGL(k_bcr_tail, dim3(1), dim3(1024), 0, s, Dp(cur), Lp(cur), Up(cur), Fp(cur), sp.bX, nb,
                     st, flags, sp.bneg);
  for (int q = nlev - 1; q >= 0; --q) {
    const int bs = lev[q].s;
    if (lev[q].pair_set2 >= 0) {
      const int s2 = lev[q].pair_set2, s1 = lev[q].set;
      const int nj = (nb + bs - 2 * bs + 4 * bs - 1) / (4 * bs);  // j = 2 bs + 4 bs q, j - bs < nb
      if (nj > 0)
        hipLaunchKernelGGL(k_bcr_back2, dim3(nj), dim3(128), 0, s, sp.bDinv, Lp(s1), Up(s1), Fp(s1),
                           Lp(s2), Up(s2), Fp(s2), sp.bX, nb, bs);
      continue;
    }
    const int ne = (nb - bs + 2 * bs - 1) / (2 * bs);
    if (ne > 0)
      hipLaunchKernelGGL(k_bcr_back, dim3(ne), dim3(64), 0, s, sp.bDinv, Lp(lev[q].set), Up(lev[q].set),
                         Fp(lev[q].set), sp.bX, nb, bs, bs, 2 * bs);
  }
  // (sp.bX IS sp.brhs: the back-substitution writes the solution where the caller reads it)
  if (guard) sp_launch_band_residual(s, sp, N, flags);
}
