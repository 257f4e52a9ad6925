// Sparse (banded) mode of the step solver: device data and launch wrappers (pgf_sparse.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct SparseDev {
  bool active = false;
  int bw = 0, ldb = 0;
  int nnzH = 0, nnzJ = 0;
  int *pos = nullptr;                       // permuted position of variable i / constraint n + r
  int *Hptr = nullptr, *Hrow = nullptr, *Hcol = nullptr, *Hslot = nullptr;
  int *Jptr = nullptr, *Jcol = nullptr, *Jslot = nullptr;
  int *JTptr = nullptr, *JTrow = nullptr, *JTmap = nullptr;
  double *Hval = nullptr, *Jval = nullptr;
  double *band = nullptr;                   // (N + 1) x ldb
  double *brhs = nullptr;                   // permuted right-hand side / solution
  // accuracy guard of the cyclic-reduction solves: the right-hand side as it was, the residual
  // of the solution, a saved solution (refinement), (max |r|, max |rhs|) per 256 rows
  double *brhs0 = nullptr, *bres = nullptr, *bsol = nullptr;
  // bred: [0, 2 nred) the pairs, [2 nred, 3 nred) partial sums of the step update (guarded
  // steps), [3 nred, 3 nred + 4) the solve's pivot flags
  double *bred = nullptr;
  int nred = 0;
  double *Hb0 = nullptr, *Jb0 = nullptr;
  // block cyclic reduction work arrays: (N/8) blocks of 8 x 8 (D, L, U, inv D), rhs, solution
  double *bD = nullptr, *bL = nullptr, *bU = nullptr, *bDinv = nullptr, *bF = nullptr, *bX = nullptr;
  int *bneg = nullptr;                      // negative pivots met while inverting block i
  // bD, bL, bU, bF hold TWO sets of blocks, the second bstride blocks behind the first: a launch
  // that does two cyclic-reduction levels at once reads one set and writes the other
  int64_t bstride = 0;
  bool values_set = false;
};

void sp_launch_spmv(hipStream_t s, int rows, const int *ptr, const int *col, const double *val,
                    const double *x, const double *add, double sgn, double *y);
void sp_launch_spmvT(hipStream_t s, int cols, const int *tptr, const int *trow, const int *tmap,
                     const double *val, const double *w, const double *base, double *out);
// c = J x - b ; w = rho c + y ; g = H x + (q + J' w)
void sp_launch_eval(hipStream_t s, const SparseDev &sp, int n, int m, const double *x, const double *y,
                    const double *b, const double *q, double rho, double *c, double *w, double *g);
void sp_launch_assemble(hipStream_t s, const SparseDev &sp, int n, int m, const uint8_t *mask,
                        double lamb, double delta);
void sp_launch_rhs(hipStream_t s, const SparseDev &sp, int n, int m, const uint8_t *mask,
                   const double *F, const double *b0full, double fact, double *Hb0, double *Jb0);
// out[pos[i]] = in[i] (gather == 0) or out[i] = in[pos[i]] (gather != 0)
void sp_launch_permute(hipStream_t s, const SparseDev &sp, int N, const double *in, double *out,
                       int gather);
void sp_launch_factor(hipStream_t s, const SparseDev &sp, int N, int *flags);
void sp_launch_bcr_solve(hipStream_t s, const SparseDev &sp, int N, int *flags, bool guard = true);
void sp_launch_band_residual(hipStream_t s, const SparseDev &sp, int N, const int *flags);
void sp_launch_band_axpy(hipStream_t s, int N, const double *a, double *x);
void sp_launch_fwdsolve(hipStream_t s, const SparseDev &sp, int N);
void sp_launch_backsolve(hipStream_t s, const SparseDev &sp, int N);
void sp_launch_step_update(hipStream_t s, const SparseDev &sp, int n, int m, double fact,
                           double rho, const double *x, const double *y, const double *lb,
                           const double *ub, const double *F, double *dx, double *dy, double *xn,
                           double *yn, double *red);
