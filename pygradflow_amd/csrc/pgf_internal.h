// Internal declarations shared by the HIP translation units of libpgf_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#define PGF_NB 64  // diagonal-block / triangular-solve block size (one wavefront of rows)

struct PgfProfile {
  bool enabled = false;
  // 1: the factorisation's kernels as separate launches, one span each (per-kernel figures);
  // 2: the PRODUCTION launches (diagonal chain beside the trailing update, T(k) with the next
  //    diagonal block's update), one span per launch
  int mode = 1;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> fused_spans, trsmud_spans;
  std::vector<double> fused_flops, fused_bytes;
  double acc_fused_ms = 0, acc_fused_flops = 0, acc_fused_bytes = 0, acc_trsmud_ms = 0;
  int64_t acc_fused_launches = 0;
  std::vector<hipEvent_t> pool;  // recycled event pairs
  std::vector<std::pair<hipEvent_t, hipEvent_t>> update_spans;
  std::vector<double> update_flops;
  std::vector<double> update_bytes;  // algorithmic bytes: C tile read + write, both panels once
  std::vector<std::pair<hipEvent_t, hipEvent_t>> factor_spans;
  // look-ahead schedule, instrumented (unfused) pass: the diagonal chain, the TRSM below it
  // and the diagonal-block update, one span per launch
  std::vector<std::pair<hipEvent_t, hipEvent_t>> chain_spans, trsm_spans, udiag_spans;
  double acc_update_ms = 0, acc_update_flops = 0, acc_update_bytes = 0, acc_factor_ms = 0;
  double acc_chain_ms = 0, acc_trsm_ms = 0, acc_udiag_ms = 0;
  int64_t acc_update_launches = 0, acc_chain_launches = 0;
};

// Dense symmetric factor  K = L D L^T  (unit lower L, diagonal D), lower triangle,
// row-major in HBM with row stride ldk.  Row N (when nrows == N + 1) carries a
// right-hand side through the elimination (forward substitution for free).
// update launches of one factorisation that can have their own persistent-tile counter (behind
// flags[0..3]; reused modulo this number: a launch is long finished 256 launches later)
#define LDLT_UPD_COUNTERS 256

struct DenseLdlt {
  int Nmax = 0;
  int64_t ldk = 0;
  double *K = nullptr;      // (Nmax + 1) x ldk
  double *W = nullptr;      // (Nmax + 1) x panel-width workspace: W = L * D of the panel
  double *dvec = nullptr;   // D
  double *dinv = nullptr;   // 1 / D
  double *zwork = nullptr;  // solve work vector (Nmax)
  double *Linv = nullptr;   // inverse of every 64 x 64 diagonal block of L, [block][row][64]
  double *LinvT = nullptr;  // the transposes
  int *flags = nullptr;     // [0] zero-pivot flag, [1] negative pivots, [2] chain helpers failed
  // chained solves: [2S] XCC slot, [2S+1] bad (S = chain_stride = 64-row blocks of capacity);
  // xpub: two halves of S * 64 published solution entries (sentinel = all bits set)
  int *chain = nullptr;
  double *xpub = nullptr;
  int chain_stride = 0;
  int *hctl = nullptr;      // stamps between the diagonal chain and its helper workgroups
  int *h_flags = nullptr;   // pinned host mirror ([3]: status word of the chained solves)
  hipStream_t stream = nullptr;
  int OB = 256;             // outer block width (K-depth of the bulk trailing update)
  size_t wstride = 0;       // doubles per W buffer (two buffers)
  int N = 0;
  bool factored = false;
  int n_neg = 0;
  PgfProfile *prof = nullptr;
  // A pre-eliminated diagonal block (the condensed KKT system, pgf_api.hip): when vdepth > 0 the
  // matrix to factorise is K - V diag(vd) V^T, V = (N + 1) x vdepth row-major (row N rides along
  // like row N of K); the look-ahead schedule applies it as `virtual' column blocks that are
  // already factorised, lazily, like any other pending block.  vneg = its negative pivots.
  double *V = nullptr;
  int64_t ldv = 0;
  double *vd = nullptr;
  int vdepth = 0;  // multiple of 32 (zero-padded columns)
  int vneg = 0;
  size_t vcap = 0, vdcap = 0;  // allocated doubles
  int inject_chain_failure = 0;  // test hook: the next chained solve reports a failure
  int inject_helper_failure = 0;  // test hook: the next factorisation reports failed helpers
};

hipError_t ldlt_alloc(DenseLdlt &f, int Nmax, hipStream_t stream);
void ldlt_free(DenseLdlt &f);
// enqueue the factorisation of the leading N x N lower triangle (+ rows up to nrows)
hipError_t ldlt_factor_async(DenseLdlt &f, int N, int nrows);
// wait and read flags: returns 0 ok / 1 singular / 2 the diagonal chain's helper workgroups
// failed their checks (they are switched off, factorise again); sets f.n_neg.  (A chained solve that failed
// its own checks is reported by ldlt_chain_check after any host synchronisation.)
int ldlt_finish(DenseLdlt &f, hipError_t *err);
// sol <- K^{-1} rhs on device vectors of length N (rhs preserved if rhs != sol)
hipError_t ldlt_solve_async(DenseLdlt &f, const double *rhs, double *sol);
// backward half only: sol <- L^{-T} w, w already equals D^{-1} L^{-1} rhs (row N trick)
hipError_t ldlt_backsolve_async(DenseLdlt &f, const double *w, double *sol);
// after a host sync: nonzero if a chained solve reported a timeout / placement problem
int ldlt_chain_check(DenseLdlt &f);
void ldlt_chain_set_enabled(bool on);  // test hook: undo the switch-off of a failed check
// look-ahead schedule (pgf_factor2.hip): the default; PGF_FACTOR=1 selects the round-1 one
bool ldlt_use_lookahead();
void ldlt_chain_helpers_off();
void ldlt_inject_helper_failure(hipStream_t s, int *flags);  // test hook: flags[2] |= 1
bool ldlt_chain_helpers_enabled();     // helpers requested (PGF_CHAIN_HELP) and not switched off
void ldlt_chain_helpers_set(bool on);  // test hook: undo / force the switch-off
hipError_t ldlt_factor2_async(DenseLdlt &f, int N, int nrows);
void ldlt_chain_timing_dump();  // PGF_CHAIN_TIMING diagnostic
// shared launch helpers (pgf_ldlt.hip)
hipEvent_t prof_event(PgfProfile *p);
void launch_update(DenseLdlt &f, hipStream_t s, const double *Wp, int64_t ldw, int N, int nrows,
                   int row0, int col0, int colEnd, int kc0, int KB, PgfProfile *p, int any_order);

// ---- dense LU with partial pivoting (pgf_lu.hip): P A = L U, row-major, in place ----------
struct DenseLu {
  int N = 0;
  int64_t ld = 0;
  double *A = nullptr;     // N x ld: unit-lower L below the diagonal, U on and above it
  int *piv = nullptr;      // row interchanged with row i at step i (LAPACK convention)
  int *perm = nullptr;     // the interchanges as one permutation: (P b)[i] = b[perm[i]]
  double *work = nullptr;  // N
  double *PT = nullptr;    // the current panel, column-major (32 x ldp): pivot search and rank-1
  int64_t ldp = 0;         // updates run along rows, coalesced
  int *flags = nullptr;    // [0] zero / non-finite pivot
  hipStream_t stream = nullptr;
  bool factored = false;
};
hipError_t lu_alloc(DenseLu &f, int N, hipStream_t stream);
void lu_free(DenseLu &f);
int lu_factor(DenseLu &f, hipError_t *err);  // 0 ok / 1 singular / -1 HIP error
hipError_t lu_solve_async(DenseLu &f, const double *rhs, double *sol, int trans);

// ---- batched mode ----------------------------------------------------------
// One entry per instance of a batch (all instances share n, m): the device addresses of an
// ordinary solver handle.  Batched kernels pick their instance with blockIdx.z and read the
// instance's own reduced size N = counts[0] + m on the device, so a whole batched Newton
// step is enqueued without a host round trip.  ctl: [0] factor this step, [1] factor
// valid, [2] mask valid, [3] frozen (the instance sits this Newton step out).
// ps: the instance's own outer-step scalars [dt, lambda, rho, fact, delta] -- every instance
// has its own step-size controller.
struct BInst {
  const double *H, *J;
  int64_t ldh, ldj;
  const double *lb, *ub, *q, *b;
  double *slb, *sub, *xhat, *yhat, *x, *y, *xn, *yn, *g, *c, *F, *b0full, *rhs, *sol, *dx, *dy;
  double *w, *tmpn, *partial, *red;
  uint8_t *mask, *mask_new;
  int *idxI, *idxA, *pos, *counts, *ctl;
  const double *ps;
  double *K;
  int64_t ldk;
  double *W;
  int64_t wstride;
  double *dvec, *dinv, *zwork, *Linv, *LinvT;
  int *flags;
  int *hctl;  // stamps between the instance's chain and its helper workgroups (small batches)
  // chained triangular solves: both publication halves (capblk * 64 entries each) and the
  // control words [0] XCC slot, [1] status, [2] solve counter of the instance's handle
  double *xpub;
  int *cctl;
  int capblk;
  // condensed order (pgf_api.hip, batch_condensed_wanted): the instance's panel V = J_I^T
  // ((n + 1) x ldv, row nI = the constraint part of the right-hand side) and its scaling -1 / delta
  double *V;
  int64_t ldv;
  double *vd;
};

struct BatchScalars {
  int n, m;
};
#define BPS_DT 0
#define BPS_LAMB 1
#define BPS_RHO 2
#define BPS_FACT 3
#define BPS_DELTA 4
#define BPS_STRIDE 8

// device-resident step controller: per-instance state (cs) and constants (cp)
#define DCS_LAMB 0      // lambda of the next outer iteration
#define DCS_INTEGRAL 1  // integral of the PI law (log scale)
#define DCS_FIRST 2     // step length of the first Newton step
#define DCS_ACCEPTED 3  // 1: the last outer step was accepted
#define DCS_DONE 4      // 1: decided after the first Newton step
#define DCS_NEXT 5      // lambda decided so far
#define DCS_USED 6      // lambda the current iteration ran with
#define DCS_STRIDE 8
#define DCP_RHO 0
#define DCP_NEWTON_TOL 1
#define DCP_LAMB_RED 2
#define DCP_LAMB_MIN 3
#define DCP_LAMB_INC 4
#define DCP_THETA_MAX 5
#define DCP_K_P 6
#define DCP_K_I 7
#define DCP_LOG_THETA_REF 8
#define DCP_COUNT 16
void batch_launch_dctl_begin(hipStream_t s, int B, const double *cs, const double *cp, double *ps,
                             uint8_t *accept);
void batch_launch_dctl_mid(hipStream_t s, const BInst *tab, int B, double *cs, const double *cp,
                           const double *diff, const int *flags, const double *norm);
void batch_launch_dctl_end(hipStream_t s, int B, double *cs, const double *cp, const double *diff,
                           const int *flags, double *log3);

// pgf_kernels.hip
// accept[i] != 0: (x^, y^) <- (x, y); == 0: (x, y) <- (x^, y^) (a rejected step goes back)
void batch_launch_advance(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                          const uint8_t *accept);
void batch_launch_set_frozen(hipStream_t s, const BInst *tab, int B, const uint8_t *frozen);
void batch_launch_eval(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc, int nparts);
// mode 1: adopt mask_new where it differs (or no mask yet); 2: always adopt; 0: keep
void batch_launch_mask(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc, int mode,
                       double tau);
// cond_mp > 0: the condensed order -- only A = H[I,I] + lamb I is assembled (+ b_x in row nI), the
// constraint block goes into the instances' panels V (cond_mp columns, zero-padded)
void batch_launch_rhs_assemble(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                               int cond_mp = 0);
// condensed order: zwork <- b_x + V b_y / delta before a forward solve (instances that reuse their
// factor), sol_y <- (V^T sol_x - b_y) / delta after the backward solve (all instances)
void batch_launch_cond_prep_fwd(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc);
void batch_launch_cond_y(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc);
// flags_out: [3 i] zero-pivot flag, [3 i + 1] negative pivots, [3 i + 2] |I| of instance i
void batch_launch_step_update(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                              double *diff_out, int *flags_out);
void batch_launch_res_norm(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                           double *norm_out);
void batch_launch_measures(hipStream_t s, const BInst *tab, int B, const BatchScalars &sc,
                           int nparts, double active_tol, double *red4, double *out);
// pgf_ldlt.hip
// batched wrappers of the look-ahead schedule's chain and T(k) kernels (pgf_factor2.hip)
// helpers: 3 B workgroups of 1024 threads must be resident together (small batches only)
void ldlt_batch_launch_chain(hipStream_t s, const BInst *tab, int B, int m, int c0, bool helpers);
void ldlt_batch_launch_update_diag(hipStream_t s, const BInst *tab, int B, int m, int wbuf, int c1);
void ldlt_batch_launch_chain_update(hipStream_t s, const BInst *tab, int B, int Nmax, int m, int wbuf,
                                    int c1, bool helpers, int vdepth = 0);
void ldlt_batch_launch_virtual_diag(hipStream_t s, const BInst *tab, int B, int vdepth);
// the batched factorisation runs the fused look-ahead schedule for this batch size (the only one
// that knows the condensed order)
bool ldlt_batch_fused_schedule(int B, int OB, bool profiling);
bool ldlt_batch_condensed_schedule(int OB);
void ldlt_batch_launch_trsm(hipStream_t s, const BInst *tab, int B, int per, int m, int wbuf, int c0);
void ldlt_batch_factor_async(hipStream_t s, const BInst *tab, int B, int Nmax, int m, int OB,
                             PgfProfile *p, int vdepth = 0);
bool ldlt_chain_enabled();  // chained solves requested (PGF_TRSV_CHAIN) and not switched off
void ldlt_batch_solve_async(hipStream_t s, const BInst *tab, int B, int Nmax, int m,
                            bool any_unfactored_solve, bool cond_prep = false);

// ---- elementwise / assembly launches (pgf_kernels.hip) --------------------
struct StepDev;  // opaque here

// micro-benchmark of the trailing update (pgf_ldlt.hip)
hipError_t ldlt_bench_update(int N, int KB, int variant, int reps, double *ms_out,
                             double *flops_out);
