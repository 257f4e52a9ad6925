/*
 * pgf_hip.h -- C ABI of the MI355X-native semi-smooth Newton / KKT step.
 *
 * Drop-in boundary for ONE hot path of chrhansk/pygradflow (v0.5.24): everything
 * executed inside one NewtonMethod.step(iterate) with the default
 * StepSolverType.Symmetric.  Plain C: opaque handles,
 * raw pointers + sizes, int status returns, no callbacks, no exceptions.
 *
 * Each entry point cites the reference interface (file:line, relative to the
 * pygradflow checkout) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub a pygradflow maintainer would add.
 *
 * Conventions
 *   - all floating point is IEEE binary64 (Params.dtype, params.py:275-277);
 *   - masks are one byte per variable, 0 / 1 (numpy bool_);
 *   - `loc` says where an input pointer lives: PGF_HOST or PGF_DEVICE;
 *     outputs are host pointers unless the name says `_dev`;
 *   - one caller thread per handle; one HIP stream per handle;
 *   - caller owns every host buffer, the library owns every device buffer.
 *
 * Status codes (mapped by the Python shim):
 *   0 ok | 1 singular / zero pivot -> LinearSolverError
 *   2 inertia mismatch -> LinearSolverError | 3 invalid argument -> ValueError
 *   4 called out of order -> RuntimeError | >=100 HIP runtime error -> RuntimeError
 */
#ifndef PGF_HIP_H
#define PGF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGF_OK 0
#define PGF_SINGULAR 1
#define PGF_INERTIA 2
#define PGF_INVALID 3
#define PGF_NOT_READY 4
#define PGF_HIP_ERROR 100

#define PGF_HOST 0
#define PGF_DEVICE 1

/* pgf_create flags */
#define PGF_CREATE_SPARSE 1u /* banded mode: CSR derivatives, no dense N x N storage */

/* pgf_qp_step policy bits (NewtonMethod policies, newton.py:35-60, 63-89, 181-215) */
#define PGF_STEP_RECOMPUTE_MASK 1u  /* Full, ActiveSet: mask from the current point  */
#define PGF_STEP_REFACTOR 2u        /* Full: assemble + factor every step             */
#define PGF_STEP_REFACTOR_ON_CHANGE 4u /* ActiveSet: refactor iff the mask changed     */

typedef struct pgf_solver *pgf_handle;
typedef struct pgf_linsolver *pgf_ls_handle;

/* ---- library ---------------------------------------------------------- */
int pgf_version(void);
int pgf_device_count(int *count);
/* static description of the last error on this handle (never NULL) */
const char *pgf_last_error(pgf_handle h);

/* ---- step-solver handle: one per (n, m); reusable across outer steps --- */
/* replaces SymmetricStepSolver.__init__ (step/solver/symmetric_step_solver.py:14-25,
 * scaled_step_solver.py:16-33) -- allocation only */
int pgf_create(int n, int m, int device, unsigned flags, pgf_handle *out);
int pgf_destroy(pgf_handle h);

/* Problem.var_lb / var_ub (problem.py:57-62); +-inf allowed */
int pgf_set_bounds(pgf_handle h, const double *lb, const double *ub);

/* one outer step: x^, y^, dt, rho.  Pre-scales the bounds lamb*lb, lamb*ub
 * (ScaledImplicitFunc.__init__, implicit_func.py:211-216); drops mask, K, factor */
int pgf_set_outer(pgf_handle h, const double *xhat, const double *yhat, double dt, double rho);

/* ScaledStepSolver.update_derivs (scaled_step_solver.py:76-79,
 * symmetric_step_solver.py:41-43): H = lag_hess(x, y) (no rho J'J), J = cons_jac.
 * Dense row-major; only the lower triangle of H is read.  PGF_HOST pointers are
 * copied to HBM; PGF_DEVICE pointers are adopted (caller keeps them alive).
 * Invalidates K and its factor. */
int pgf_set_derivs_dense(pgf_handle h, const double *H, int64_t ldh, const double *J,
                         int64_t ldj, int loc);
/* The same matrices as scipy CSR (what Iterate.aug_lag_deriv_xx / _xy return,
 * iterate.py:99-110): host arrays, int32 indices, duplicates are summed.  Only the
 * non-zeros cross PCIe; the dense H, J the factorisation reads are rebuilt on the device.
 * For problems whose derivatives are sparse but not banded (the banded path has its own
 * entry points below). */
int pgf_set_derivs_csr(pgf_handle h, const int *Hptr, const int *Hidx, const double *Hval,
                       const int *Jptr, const int *Jidx, const double *Jval);

/* StepFunc.compute_active_set (implicit_func.py:72-74) =
 * projection_initial (:233-246, tau = NaN means None) + compute_active_set_box (:21-44).
 * x, g: current point and g = aug_lag_deriv_x(rho) (iterate.py:91-94). mask_out[n]. */
int pgf_active_set(pgf_handle h, const double *x, const double *g, double tau,
                   uint8_t *mask_out);

/* ScaledStepSolver.update_active_set (scaled_step_solver.py:81-83) */
int pgf_set_active_set(pgf_handle h, const uint8_t *mask);

/* SymmetricStepSolver._compute_deriv (:49-77) + linear_solver()/LUSolver.__init__
 * (:129-133, linear_solver/lu_solver.py:9-17): gather-assemble the reduced KKT matrix
 * for the current mask in HBM and factor it (LDL^T, K symmetric quasi-definite).
 * n_neg = number of negative pivots = LinearSolver.num_neg_eigvals().
 * Returns PGF_SINGULAR on a zero / non-finite pivot. */
int pgf_factor(pgf_handle h, int *n_neg);

/* ScaledStepSolver.solve (scaled_step_solver.py:85-107) for the current mask/factor:
 * residual (implicit_func.py:219-231), rhs split (:38-60), reduced rhs
 * (symmetric_step_solver.py:79-94), solve (:135-158), scatter (:115-121),
 * dy (:104), StepResult clipping and diff (step_solver.py:16-63).
 * c may be NULL when m == 0.  Factors first if needed (status as pgf_factor).
 * inertia_check != 0 and n_neg != m  ->  PGF_INERTIA (:146-153). */
int pgf_newton_solve(pgf_handle h, const double *x, const double *y, const double *g,
                     const double *c, int inertia_check, double *dx, double *dy,
                     double *xn, double *yn, double *diff);

/* ScaledImplicitFunc.value_at (implicit_func.py:219-231); mask NULL = recompute at p */
int pgf_residual(pgf_handle h, const double *x, const double *y, const double *g,
                 const double *c, const uint8_t *mask, double *F_out);

/* LinearSolver.solve(rhs, trans) (linear_solver/linear_solver.py:23-25) against the
 * current reduced KKT factor; rhs/sol have |I| + m entries */
int pgf_linear_solve(pgf_handle h, const double *rhs, int trans, double *sol);

/* size of the reduced system of the current mask: |I|, |I| + m */
int pgf_reduced_dims(pgf_handle h, int *n_inactive, int *n_reduced);
/* copy the assembled (pre-factor) lower triangle of K to host, row-major N x N
 * (debug / parity; re-assembles, does not disturb the factor) */
int pgf_get_kkt(pgf_handle h, double *K_out, int64_t ldk);

/* ---- sparse (banded) mode: handle created with PGF_CREATE_SPARSE --------------------- */
/* Fixed sparsity pattern of H (n x n CSR, both triangles) and J (m x n CSR) and the band
 * plan computed on the host (pygradflow_amd/sparse.py): pos[i] = row of variable i /
 * constraint n + r in the permuted banded KKT matrix, bw = half bandwidth, H/J slot =
 * linear index into the band array for each stored entry (H: -1 for the mirrored upper
 * duplicates), JT* = column-ordered copy of J's pattern (JTmap indexes Jval).  Replaces the
 * scipy fancy-slicing / bmat assembly of symmetric_step_solver.py:27-39, 49-77 for sparse
 * problems; active variables become identity rows instead of being sliced out. */
int pgf_sparse_set_pattern(pgf_handle h, int bw, const int *pos, int nnzH, const int *Hptr,
                           const int *Hrow, const int *Hcol, const int *Hslot, int nnzJ,
                           const int *Jptr, const int *Jcol, const int *Jslot, const int *JTptr,
                           const int *JTrow, const int *JTmap);
/* values of H = lag_hess(x, y) and J = cons_jac(x) in pattern order (update_derivs) */
int pgf_sparse_set_values(pgf_handle h, const double *Hval, const double *Jval);
/* q and b of a linear-quadratic problem whose Q, A were given through the sparse pattern */
int pgf_qp_set_vectors(pgf_handle h, const double *q, const double *b);

/* ---- device-resident linear-quadratic mode (bench, batched mode) -------- */
/* f = 1/2 x'Qx + q'x, c = Ax - b: H = Q and J = A stay in HBM; g and c are
 * evaluated on device (SURVEY.md 8d "Problem data device-resident"). */
int pgf_qp_set_problem(pgf_handle h, const double *Q, int64_t ldq, const double *q,
                       const double *A, int64_t lda, const double *b, int loc);
int pgf_qp_set_point(pgf_handle h, const double *x, const double *y);
int pgf_qp_get_point(pgf_handle h, double *x, double *y);
int pgf_qp_get_mask(pgf_handle h, uint8_t *mask);
/* mask <- compute_active_set at the device point (SimplifiedNewtonMethod.__init__,
 * newton.py:52-56: call once at (x^, y^)); *changed = 1 if the stored mask was replaced */
int pgf_qp_update_active_set(pgf_handle h, double tau, int *changed);
/* new outer step at the current device point: (x^, y^) <- (x, y) on device, then as
 * pgf_set_outer (what Solver.solve does between accepted steps, solver.py:357-378) */
int pgf_qp_advance_outer(pgf_handle h, double dt, double rho);
/* One NewtonMethod.step on the device-resident point (policy bits above; tau NaN = None).
 * (x, y) <- (xn, yn).  n_neg / diff may be NULL.  One host sync at the end
 * (plus one to read |I| when the mask is recomputed). */
int pgf_qp_step(pgf_handle h, unsigned policy, double tau, int inertia_check, int *n_neg,
                double *diff);
/* enqueue-only variant for timing loops: no host sync, status checked by pgf_qp_sync */
int pgf_qp_step_async(pgf_handle h, unsigned policy, double tau);
int pgf_qp_sync(pgf_handle h, int *n_neg, double *diff);
/* ||F(z)||_2 of the UNSCALED residual (ImplicitFunc.value_at, implicit_func.py:150-161)
 * at the device point; host result, and optionally a device slot (RCCL all-gather input) */
int pgf_qp_residual_norm(pgf_handle h, double *norm_out, double *norm_out_dev);
/* Termination measures of the device point: out[0] = stat_res
 * (iterate.py:174-177, with bounds_dual :140-152 and the ActiveSet of active_set.py:4-29 at
 * tolerance active_tol), out[1] = cons_violation (:166-171), out[2] = bound_violation
 * (:155-163), out[3] = ||y||_inf (DualNormUpdate, penalty.py:60-74). */
int pgf_qp_measures(pgf_handle h, double active_tol, double *out);
/* HIP stream of the handle (void* = hipStream_t) for event timing by the caller */
int pgf_stream(pgf_handle h, void **stream_out);
/* device time (ms) spent in the factor's trailing-update launches since the last call,
 * their count and algorithmic flops -- measured with HIP events on the handle's stream */
int pgf_profile_enable(pgf_handle h, int on);
int pgf_profile_read(pgf_handle h, double *update_ms, int64_t *update_launches,
                     double *update_flops, double *factor_ms);
/* the same and more, as one array (count >= PGF_PROF_COUNT).  While profiling is enabled the
 * dense factorisation runs its kernels as separate launches (the production schedule fuses
 * the diagonal chain with the trailing update in one launch) so that each can be timed:
 * update (k_ldlt_update) ms / launches / algorithmic flops / algorithmic bytes, whole
 * factorisation ms, diagonal chain (k_diag_chain) ms / launches, TRSM below the diagonal
 * block (k_trsm_block) ms, diagonal-block update (k_update_diag) ms */
#define PGF_PROF_UPDATE_MS 0
#define PGF_PROF_UPDATE_LAUNCHES 1
#define PGF_PROF_UPDATE_FLOPS 2
#define PGF_PROF_UPDATE_BYTES 3
#define PGF_PROF_FACTOR_MS 4
#define PGF_PROF_CHAIN_MS 5
#define PGF_PROF_CHAIN_LAUNCHES 6
#define PGF_PROF_TRSM_MS 7
#define PGF_PROF_UDIAG_MS 8
#define PGF_PROF_COUNT 9
/* pgf_profile_enable(h, 2): the PRODUCTION launches are timed instead, one HIP-event span per
 * launch: k_chain_update (the diagonal chain beside the trailing-update tiles; flops / bytes =
 * the algorithmic work of its update jobs) and k_trsm_ud (T(k) with the next diagonal block's
 * update); read with count >= PGF_PROF_COUNT2 */
#define PGF_PROF_FUSED_MS 9
#define PGF_PROF_FUSED_LAUNCHES 10
#define PGF_PROF_FUSED_FLOPS 11
#define PGF_PROF_FUSED_BYTES 12
#define PGF_PROF_TRSMUD_MS 13
#define PGF_PROF_COUNT2 14
int pgf_profile_read_ex(pgf_handle h, double *out, int count);

/* ---- batched mode: many device-resident instances advanced by ONE launch sequence ---- */
/* The reference's only parallelism is a process pool over independent instances
 * (runners/runner.py:107-153); BASELINE configs[3] is 256 instances of (n=1024, m=256).
 * A batch groups existing dense linear-quadratic handles of identical (n, m) on one
 * device (each set up through pgf_set_bounds / pgf_qp_set_problem / pgf_qp_set_point);
 * every kernel of the Newton step then runs over all instances at once (instance =
 * blockIdx.z, reduced sizes read on the device: no host round trip inside a step).
 * While a handle belongs to a batch, drive it only through the batch. */
typedef struct pgf_batch_s *pgf_batch;
int pgf_batch_create(const pgf_handle *handles, int count, pgf_batch *out);
int pgf_batch_destroy(pgf_batch b);
const char *pgf_batch_last_error(pgf_batch b);
/* pgf_qp_advance_outer for every instance: (x^, y^) <- (x, y), new dt and rho */
int pgf_batch_advance_outer(pgf_batch b, double dt, double rho);
/* the same with per-instance dt[i], rho[i] -- every instance runs its own step-size
 * controller -- and accept[i]: nonzero (or accept == NULL) = the instance's last steps were
 * accepted; zero = rejected, the instance goes back to its outer point (x, y) <- (x^, y^) and
 * retries with the new dt[i] (StepController.compute_step, step_control.py:80-107) */
int pgf_batch_advance_outer_each(pgf_batch b, const double *dt, const double *rho,
                                 const uint8_t *accept);
/* frozen[i] != 0: instance i sits out the following Newton steps until the next
 * pgf_batch_advance_outer* (the controller's early exits: converged after the first step,
 * distance_ratio_control.py:36-45, or a failed factorisation); NULL clears all */
int pgf_batch_set_frozen(pgf_batch b, const uint8_t *frozen);
/* pgf_qp_update_active_set for every instance (SimplifiedNewtonMethod.__init__) */
int pgf_batch_update_active_set(pgf_batch b, double tau);
/* one NewtonMethod.step per instance (policy bits as pgf_qp_step); enqueue only */
int pgf_batch_step_async(pgf_batch b, unsigned policy, double tau);
/* wait; per instance: status (PGF_OK / PGF_SINGULAR), negative pivots of the factor in
 * use, ||(dx, dy)||.  Any output may be NULL.  Returns PGF_OK even when single instances
 * failed -- their status says so (the reference would reject only those steps). */
int pgf_batch_sync(pgf_batch b, int *status, int *n_neg, double *diff);
/* unscaled residual norms of all instances: host array and / or device array (the
 * rank-local block of the RCCL all-gather) */
int pgf_batch_residual_norms(pgf_batch b, double *norms_out, double *norms_out_dev);
/* ---- device-resident step controller -------------------------------
 * The reference's default DistanceRatioController (step/distance_ratio_control.py:12-78: two
 * Newton steps per outer iteration, theta = |step 2| / |step 1|, accept iff theta <= theta_max,
 * PI law on log(theta) for lambda, controller.py:54-77; early exits :34-45; a failed
 * factorisation = rejected step with 2 lambda, step_control.py:103-107) for every instance of
 * the batch, decided ON THE DEVICE: pgf_batch_ctl_iterate enqueues `iterations` outer
 * iterations back to back with no host synchronisation in between (per-instance lambda, PI
 * integral, accept flag and frozen flags live in HBM).  params: newton_tol, lamb_red,
 * lamb_min, lamb_inc, theta_max, K_P, K_I, theta_ref (Params fields of the same names).
 * pgf_batch_ctl_read waits and returns the next lambda and the last accept flag of every
 * instance and, per outer iteration and instance, (lambda used, lambda next, accepted). */
int pgf_batch_ctl_init(pgf_batch b, double lamb_init, double rho, const double *params,
                       int max_iterations);
int pgf_batch_ctl_iterate(pgf_batch b, unsigned policy, double tau, int iterations);
int pgf_batch_ctl_read(pgf_batch b, double *lamb, uint8_t *accepted, double *log3, int log_rows);

/* all points / masks, instance-major: x[count][n], y[count][m], mask[count][n] */
int pgf_batch_get_points(pgf_batch b, double *x, double *y);
int pgf_batch_get_masks(pgf_batch b, uint8_t *mask);
/* pgf_qp_measures for every instance: out[count][4] */
int pgf_batch_measures(pgf_batch b, double active_tol, double *out);
int pgf_batch_stream(pgf_batch b, void **stream_out);
/* Instances whose step the accuracy guard REPAIRED inside pgf_batch_sync since the batch was
 * created: the sampled residual of the batched solve failed, and the single-instance guard on the
 * instance's handle (full residual, refinement, pivoted LU) produced the step. */
int pgf_batch_refinement_stats(pgf_batch b, int *repaired);
/* ---- the batched mode's one collective, without Python ----------------
 * One process per GPU; every rank advances its shard of the instances as a device batch and the
 * residual norms of ALL instances are all-gathered once per step (the reference gathers per-
 * instance results from its process pool, runners/runner.py:107-153).  pygradflow_amd/batched.py
 * does this through torch.distributed (backend "nccl" = RCCL); these entry points do the same for
 * a host without PyTorch: librccl.so is opened at run time (dlopen; the library has no link-time
 * dependency on it), rank 0 creates the 128-byte id and hands it to the other ranks by whatever
 * means the host has, every rank creates its communicator on its own device.
 * pgf_batch_allgather_norms: ||F|| of the batch's current points (as pgf_batch_residual_norms)
 * into slot `rank' of all_dev -- a DEVICE array of nranks * count doubles -- and one
 * ncclAllGather on the batch's stream; returns after the stream has drained.  Every rank must
 * hold the same number of instances.  PGF_NOT_READY if librccl.so cannot be loaded. */
typedef struct pgf_comm_s *pgf_comm;
#define PGF_COMM_ID_BYTES 128
int pgf_comm_unique_id(void *id_out);
int pgf_comm_create(int nranks, int rank, const void *id, int device, pgf_comm *out);
int pgf_comm_destroy(pgf_comm c);
int pgf_batch_allgather_norms(pgf_batch b, pgf_comm c, double *all_dev);
/* as pgf_profile_enable / pgf_profile_read, for the batch's trailing-update launches */
int pgf_batch_profile_enable(pgf_batch b, int on);
int pgf_batch_profile_read(pgf_batch b, double *update_ms, int64_t *update_launches,
                           double *update_flops);

/* ---- stand-alone dense linear solver (LinearSolver ABC) ------------------ */
/* LinearSolver.__init__ factorises in the constructor
 * (linear_solver/linear_solver.py:18-21, lu_solver.py:9-17).  A: dense row-major N x N.
 * symmetric != 0: LDL^T on the lower triangle; symmetric == 0: LU with partial pivoting of the
 * full matrix (what the reference's LUSolver does for every matrix; needed by its Standard /
 * Extended / Asymmetric step-solver formulations, step/solver/).  PGF_SINGULAR on failure.
 * pgf_ls_solve(trans != 0) solves with A^T (cond_estimate.py:82).  pgf_ls_num_neg: LDL^T only
 * (PGF_NOT_READY for an LU: LUSolver.num_neg_eigvals() is None).  pgf_ls_get_factor copies
 * L below / D on the diagonal, or for the LU: unit-lower L below, U on and above. */
int pgf_ls_create_dense(int N, const double *A, int64_t lda, int symmetric, int device,
                        pgf_ls_handle *out);
int pgf_ls_solve(pgf_ls_handle ls, const double *rhs, int trans, double *sol);
int pgf_ls_num_neg(pgf_ls_handle ls, int *out);
/* debug / parity: copy the factor (unit-lower L below the diagonal, D on it) to host */
int pgf_ls_get_factor(pgf_ls_handle ls, double *LD_out, int64_t ld);
int pgf_ls_destroy(pgf_ls_handle ls);

/* ---- kernel micro-benchmark hook (tools/, bench diagnostics) ------------------ */
/* Times `reps` launches of the factorisation's trailing-update kernel on a synthetic
 * N x N lower-triangular region with K-depth KB (variant selects the tile kernel);
 * returns the mean launch time in ms and the algorithmic flops of one launch. */
int pgf_bench_update(int N, int KB, int variant, int reps, int device, double *ms_out,
                     double *flops_out);

/* out <- K v: the reduced KKT matrix of the current active set applied to a host vector of
 * length N = |I| + m on the device, from H, J and the mask (symmetric: K' v is the same).
 * These are the `mat @ x` / `mat.T @ x` products of the reference's ConditionEstimator
 * (step/cond_estimate.py:60-82); dense mode only. */
int pgf_kkt_apply(pgf_handle h, const double *v, double *out);

/* ---- accuracy guard of the dense path ---------------------------------------------- */
/* The reference factorises the reduced KKT matrix with a PIVOTED sparse LU (SuperLU through
 * scipy.sparse.linalg.splu, linear_solver/lu_solver.py:14) and therefore stays accurate when
 * H[I,I] + lambda I is indefinite (symmetric_step_solver.py:146-153 only checks the inertia on
 * request).  The unpivoted LDL^T used here is checked instead: after every solve
 * max |rhs - K s| is compared with tol * max |rhs| (K applied from H, J and the mask); beyond
 * that, up to two steps of iterative refinement, then a dense LU with partial pivoting of the
 * same matrix; PGF_SINGULAR only if that fails as well.  mode 0 switches the check off;
 * tol / fail_tol <= 0 keep the defaults (1e-11, 1e-7). */
int pgf_set_refinement(pgf_handle h, int mode, double tol, double fail_tol);
/* counters since pgf_create: refinement steps taken, LU factorisations, and the relative
 * residual of the last checked solve */
int pgf_refinement_stats(pgf_handle h, int *refined, int *lu_fallbacks, double *last_rel_residual);

/* ---- test hooks ------------------------------------------------------------------ */
/* The triangular solves of the dense path run as ONE launch whose workgroups hand the
 * solution over block by block; every such solve checks itself (placement of its workers,
 * bounded waits).  A failed check never surfaces: the call that notices it repeats the solve
 * with the per-block kernels before it returns and the chained kernels stay off for the
 * rest of the process.  pgf_debug_fail_next_chain makes the next chained solve of the
 * handle look failed (status word set, solution overwritten with NaN) so that the recovery
 * can be tested; pgf_debug_chain_enable(1) switches the chained kernels back on. */
int pgf_debug_fail_next_chain(pgf_handle h);
int pgf_debug_chain_enable(int on);
/* The dense factorisation's diagonal chain hands work to two helper workgroups of the same
 * launch; a failed placement check or a timed-out hand-over switches them off
 * for the rest of the process and the factorisation (and the step built on it) is repeated
 * inside the call that notices.  pgf_debug_fail_next_helper makes the next factorisation of
 * this handle report such a failure; pgf_debug_chain_helpers(1 / 0) switches the helpers on /
 * off, (-1) only queries; returns the previous state (1 = on). */
int pgf_debug_fail_next_helper(pgf_handle h);
/* batched mode (chains with helpers up to 32 instances): instance 0's next factorisation reports
 * failed helpers -- its step comes back as failed (status PGF_SINGULAR), which the controllers
 * reject and repeat, and the helpers are switched off */
int pgf_batch_debug_fail_next_helper(pgf_batch b);
int pgf_debug_chain_helpers(int on);
/* Which factorisation the handle's current dense factor is (tests, tools/check_condensed.py):
 * 0 none / stale, 1 LDL^T of the reduced KKT matrix in its natural order, 2 LDL^T of the
 * condensed system (constraint block eliminated first, pgf_api.hip condensed_wanted), 3 the
 * pivoted LU that took over after a failed residual check. */
int pgf_debug_factor_kind(pgf_handle h);

#ifdef __cplusplus
}
#endif
#endif /* PGF_HIP_H */
