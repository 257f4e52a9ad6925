// Launch wrappers of pgf_kernels.hip (all asynchronous on the given stream).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

void launch_scale_bounds(hipStream_t s, int n, double lamb, const double *lb, const double *ub,
                         double *slb, double *sub);
void launch_active_set(hipStream_t s, int n, int use_tau, double lamb, double f_x, double f_x0,
                       double f_d, const double *xhat, const double *x, const double *g,
                       const double *slb, const double *sub, uint8_t *mask);
void launch_compact(hipStream_t s, int n, const uint8_t *mask, int *idxI, int *idxA, int *pos,
                    int *counts);
void launch_residual(hipStream_t s, int n, int m, double lamb, double dt, const double *xhat,
                     const double *yhat, const double *x, const double *y, const double *g,
                     const double *c, const double *slb, const double *sub, const uint8_t *mask,
                     double *F, double *b0full);
void launch_reduced_rhs(hipStream_t s, int n, int m, int nI, int nA, double fact, const double *F,
                        const int *idxI, const int *idxA, const double *H, int64_t ldh, const double *J,
                        int64_t ldj, const double *b0full, double *partial, int nparts, double *rhs);
void launch_assemble_kkt(hipStream_t s, double *K, int64_t ldk, const double *H, int64_t ldh,
                         const double *J, int64_t ldj, const int *idxI, int nI, int m,
                         double lamb, double delta);
void launch_copy(hipStream_t s, double *dst, const double *src, int n);
void launch_copy_u8(hipStream_t s, uint8_t *dst, const uint8_t *src, int n);
void launch_mask_diff(hipStream_t s, int n, const uint8_t *a, const uint8_t *b, int *out);
int step_update_blocks(int n, int m);
void launch_step_update(hipStream_t s, int n, int m, int nI, double fact, double rho,
                        const double *x, const double *y, const double *lb, const double *ub,
                        const uint8_t *mask, const int *pos, const double *b0full,
                        const double *F, const double *sol, double *dx, double *dy, double *xn,
                        double *yn, double *red, double *diff_out);
void launch_gemv_rows(hipStream_t s, int rows, int cols, const double *M, int64_t ld,
                      const double *v, const double *add, double sgn, double *out);
void launch_gemvT(hipStream_t s, int rows, int cols, const double *M, int64_t ld,
                  const double *w, const double *base, double *partial, int nparts, double *out);
void launch_mult_vec(hipStream_t s, int m, double rho, const double *c, const double *y,
                     double *w);
void launch_unscaled_res_norm(hipStream_t s, int n, int m, double dt, const double *xhat,
                              const double *yhat, const double *x, const double *y,
                              const double *g, const double *c, const double *lb,
                              const double *ub, double *red, double *out);
void launch_final_reduce(hipStream_t s, const double *red, int cnt, double *out, int take_sqrt);
void launch_measures(hipStream_t s, int n, int m, double active_tol, const double *x,
                     const double *y, const double *r, const double *c, const double *lb,
                     const double *ub, double *red, double *out);
void launch_csr_to_dense(hipStream_t s, int rows, const int *ptr, const int *idx, const double *val,
                         double *dst, int64_t ld);
// r = rhs - K s of the reduced KKT system, K applied from H, J and the mask (never assembled);
// red3 <- max |r|, max |rhs|, max |s|.  v, lv, u: n-vectors of scratch, wy: m, r: nI + m.
void launch_kkt_residual(hipStream_t s, int n, int m, int nI, double lamb, double delta,
                         const double *H, int64_t ldh, const double *J, int64_t ldj,
                         const int *idxI, const int *pos, const uint8_t *mask, const double *rhs,
                         const double *sol, double *v, double *lv, double *u, double *wy,
                         double *partial, int nparts, double *r, double *red3);
void launch_axpy1(hipStream_t s, int N, const double *d, double *x);
void launch_symmetrize(hipStream_t s, double *A, int64_t ld, int N);
// ||H||_inf, ||J||_inf, ||J||_1 -> norms3 (device), for the normwise backward error of the guard
void launch_matrix_norms(hipStream_t s, int n, int m, const double *H, int64_t ldh, const double *J,
                         int64_t ldj, double *norms3);
// condensed KKT system (constraint block eliminated first; pgf_kernels.hip, pgf_api.hip):
// V <- J[:, I]^T zero-padded to mp columns, row nI <- rhs_y (or 0), vd <- -1 / delta
void launch_cond_panel(hipStream_t s, double *V, int64_t ldv, int mp, double *vd, const double *J,
                       int64_t ldj, const int *idxI, int nI, int m, double delta, const double *rhs_y);
// out <- rhs_x + V rhs_y / delta
void launch_cond_rhs(hipStream_t s, int nI, int m, const double *V, int64_t ldv, const double *rhs,
                     double delta, double *out);
// sol_y <- (V^T sol_x - rhs_y) / delta
void launch_cond_y(hipStream_t s, int nI, int m, const double *V, int64_t ldv, const double *solx,
                   const double *rhs_y, double delta, double *partial, size_t partial_cap, double *sol_y);
// launch_kkt_residual of the solve just made AND the evaluation of g, c at the new point (xn, yn)
// in one pass over H and two over J (the separate kernels: two and four); partial: 2 * nparts * n
void launch_residual_and_eval(hipStream_t s, int n, int m, int nI, double lamb, double delta, const double *H,
                              int64_t ldh, const double *J, int64_t ldj, const int *idxI, const int *pos,
                              const uint8_t *mask, const double *rhs, const double *sol, double *v, double *lv,
                              double *u, double *wy, double *partial, int nparts, double *r, double *red3,
                              const double *xn, const double *yn, const double *b, const double *q, double rho,
                              double *c, double *w, double *tmpn, double *g);
