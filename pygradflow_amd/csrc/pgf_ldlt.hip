// Dense LDL^T factorisation and triangular solves for the reduced KKT matrix
// (replaces SuperLU gstrf/gstrs reached through scipy.sparse.linalg.splu at
// reference pygradflow/linear_solver/lu_solver.py:14,21).
//
// K is symmetric quasi-definite ([[H_II + lamb I, J_I'],[J_I, -delta I]]), so an
// unpivoted LDL^T exists for the natural order; the number of negative pivots is
// the inertia the reference asks its linear solver for (num_neg_eigvals).
//
// Layout: lower triangle, row-major, row stride ldk (multiple of 16 doubles, so
// every row starts on a 128-byte line).  Two-level right-looking blocked algorithm
// (outer block 256 columns, inner panels of 64):
//   k_ldlt_panel        fused 64-column panel: 16-blocked LDL^T of the diagonal block +
//                       TRSM of the workgroup's own 64 rows, all in LDS
//   k_ldlt_update       trailing update, 64 x 64 tiles, v_mfma_f64_16x16x4_f64
//                       (the FP64-MFMA-bound kernel the roofline is quoted on)
//   k_inv_diag_blocks   inverse of every 64 x 64 diagonal block of L (once per factor)
//   k_trsv_{fwd,bwd}_super  triangular solves, 256-row super-blocks per launch
#include <hip/hip_ext.h>

#include "pgf_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "pgf_ldlt_dev.h"

// ------------------------------------------------------------------ fused panel kernel
// One launch per 64-column panel.  Every workgroup (4 wavefronts) holds in LDS the
// 64 x 64 diagonal block (rows 0..63 of M) stacked on its OWN 64 panel rows (rows
// 64..127) and runs a 16-blocked right-looking LDL^T on the 128 x 64 stack:
//   (a) 16 x 16 diagonal tile: unblocked, rows in VGPRs of 16 lanes, v_readlane broadcasts
//   (b) rows below the tile: X L_bb^T = A by substitution, one lane per row (16 VGPRs),
//       L_bb broadcast from LDS; emits W = X (to LDS for (c), to the W workspace for the
//       own rows) and L = X D^-1
//   (c) remaining tiles to the right: M_tile -= W_ti L_tj^T with v_mfma_f64_16x16x4_f64
// The diagonal block is factored redundantly by every workgroup (it is on the critical
// path anyway and this removes one launch boundary per panel); workgroup 0 writes it
// back together with D, 1/D, the zero-pivot flag and the negative-pivot count.
// All loops over tiles are rolled (LDS offsets computed at run time): the code stays a
// few KB, unlike a fully unrolled in-register 64 x 64 elimination, which is
// instruction-fetch bound when launched cold.
#define PNL_LD 66   // LDS row stride of M: conflict-free MFMA fragment reads, 16 B rows
#define PNL_WLD 18

// M[128][PNL_LD]; Wt[192][PNL_WLD]: rows 0..63 diagonal block, 64..127 / 128..191 the own
// rows' W of even / odd sub-block steps (double buffered); D, 1/D; flag
#define PNL_SMEM (128 * PNL_LD * 8 + 192 * PNL_WLD * 8 + 2 * 64 * 8 + 16)

template <int NB, bool PRE = false>
__device__ __forceinline__ void panel_body(unsigned char *smem, const int wg,
                                           double *__restrict__ K, int64_t ldk,
                                           double *__restrict__ W, int64_t ldw, int wofs, int N,
                                           int nrows, int c0, double *__restrict__ dvec,
                                           double *__restrict__ dinv, int *__restrict__ flags,
                                           int skip) {
  static_assert(NB == 64, "panel kernel is written for 64-column panels");
  double(*M)[PNL_LD] = reinterpret_cast<double(*)[PNL_LD]>(smem);
  double(*Wt)[PNL_WLD] = reinterpret_cast<double(*)[PNL_WLD]>(smem + 128 * PNL_LD * 8);
  double *dD = reinterpret_cast<double *>(smem + 128 * PNL_LD * 8 + 192 * PNL_WLD * 8);
  double *dI = dD + 64;
  int &s_bad = *reinterpret_cast<int *>(dI + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nb = min(NB, N - c0);
  const int rbase = c0 + nb + wg * 64;  // first own row (global)
  const int coh = 0;
  (void)skip;
  if (tid == 0) s_bad = 0;

  // ---- load: diag block (identity outside the valid lower triangle) + own rows.
  // All 16 global loads of a lane are issued before the first LDS store (one memory
  // latency instead of sixteen).  PRE: the caller has filled M already.
  if (!PRE) {
    double2_t v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int p = q * 256 + tid;
      const int row = p >> 5, c2 = (p & 31) * 2;
      double2_t t = (double2_t){0.0, 0.0};
      if (row < 64) {
        if (row < nb) {
          const double *src = K + (int64_t)(c0 + row) * ldk + c0 + c2;
          if (c2 + 1 <= row) t = ld_f64x2(src, coh);
          else if (c2 <= row) t.x = ld_f64(src, coh);
        } else {  // identity padding keeps the elimination well defined
          if (c2 == row) t.x = 1.0;
          if (c2 + 1 == row) t.y = 1.0;
        }
      } else {
        const int r = rbase + row - 64;
        if (r < nrows) {
          const double *src = K + (int64_t)r * ldk + c0 + c2;
          if (c2 + 1 < nb) t = ld_f64x2(src, coh);
          else if (c2 < nb) t.x = ld_f64(src, coh);
        }
      }
      v[q] = t;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int p = q * 256 + tid;
      *reinterpret_cast<double2_t *>(&M[p >> 5][(p & 31) * 2]) = v[q];
    }
  }
  __syncthreads();

  // Schedule of one 16-column sub-block `sb` (cb = 16 sb).  Critical path = wavefront 0:
  //   (a+) wavefront 0, lane <-> row of the 64 x 64 DIAGONAL block: right-looking elimination
  //        of the 16 columns for the tile rows AND, in the same instruction stream, for all
  //        diagonal-block rows below the tile (their W = L D and L come for free);
  //   (c-diag) rank-16 update of the diagonal block's remaining tiles (MFMA);
  // off the critical path, one step behind, for the workgroup's OWN 64 rows (stack rows
  // 64..127): (b-own) substitution by wavefront 1 while wavefront 0 runs the next (a+);
  // (c-own) their tile updates alongside (c-diag).
  // The earlier version substituted all 112 rows below the tile between (a) and (c): 1.4 us
  // per sub-block on the critical path.
  auto update_tile16 = [&](int psb, int ti, int tj) {  // M[ti][tj] -= W_psb[ti] L_psb[tj]^T
    const int pcb = psb * 16;
    const int wrow = ti * 16 + ((ti >= 4) ? 64 * (psb & 1) : 0);  // own rows: buffer of step psb
    double4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = M[ti * 16 + l4 + 4 * r][tj * 16 + l15];
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const double av = -Wt[wrow + l15][ks + l4];
      const double bv = M[tj * 16 + l15][pcb + ks + l4];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) M[ti * 16 + l4 + 4 * r][tj * 16 + l15] = acc[r];
  };
  auto a_plus = [&](int sb) {  // wavefront 0
    const int cb = sb * 16;
    double a[16], w[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const double2_t v = *reinterpret_cast<const double2_t *>(&M[lane][cb + k]);
      a[k] = v.x;
      a[k + 1] = v.y;
    }
    // Per column the wavefront issues ~35 dependent instructions, so every one counts: a
    // zero / inf / NaN pivot is detected with one v_cmp_class and only recorded (the factor
    // is rejected as a whole afterwards, whatever it then contains); each tile lane picks
    // its own pivot out of w[] once, after the loop.
    bool bad_any = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double d = lane_bcast(a[j], cb + j);
      // classes: sNaN, qNaN, -inf, -0, +0, +inf
      const bool bad = __builtin_amdgcn_class(d, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200);
      bad_any |= bad && (cb + j) < nb;
      const double di = fast_recip(d);
      w[j] = a[j];
      const double l = a[j] * di;
#pragma unroll
      for (int k = j + 1; k < 16; ++k) a[k] = fma(-l, lane_bcast(a[j], cb + k), a[k]);
      a[j] = l;
    }
    const int tr = lane - cb;  // row inside the tile (tile lanes: 0..15)
    if (tr >= 0 && tr < 16) {
      double d_mine = 1.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < tr) M[lane][cb + k] = a[k];
        if (k == tr) d_mine = w[k];
      }
      const bool ok = !__builtin_amdgcn_class(d_mine, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200);
      M[lane][lane] = d_mine;
      dD[lane] = d_mine;
      dI[lane] = ok ? fast_recip(d_mine) : 0.0;
      if (tr == 0 && bad_any) s_bad = 1;
    } else if (tr >= 16) {  // diagonal-block rows below the tile
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        double2_t wv, lv;
        wv.x = w[k];
        wv.y = w[k + 1];
        lv.x = a[k];
        lv.y = a[k + 1];
        *reinterpret_cast<double2_t *>(&Wt[lane][k]) = wv;
        *reinterpret_cast<double2_t *>(&M[lane][cb + k]) = lv;
      }
    }
  };
  auto b_own = [&](int sb) {  // wavefront 1: substitution for the own rows, one lane per row
    const int cb = sb * 16;
    const int row = 64 + lane;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const double2_t v = *reinterpret_cast<const double2_t *>(&M[row][cb + k]);
      x[k] = v.x;
      x[k + 1] = v.y;
    }
#pragma unroll
    for (int t = 0; t < 15; ++t) {
      const double xt = x[t];
#pragma unroll
      for (int j = t + 1; j < 16; ++j) x[j] = fma(-xt, M[cb + j][cb + t], x[j]);
    }
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      double2_t wv, lv;
      wv.x = x[k];
      wv.y = x[k + 1];
      lv.x = x[k] * dI[cb + k];
      lv.y = x[k + 1] * dI[cb + k + 1];
      *reinterpret_cast<double2_t *>(&Wt[row + 64 * (sb & 1)][k]) = wv;
      *reinterpret_cast<double2_t *>(&M[row][cb + k]) = lv;
    }
    const int r = rbase + lane;
    if (r < nrows) {
      double *wp = W + (int64_t)r * ldw + wofs + cb;
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        double2_t wv;
        wv.x = x[k];
        wv.y = x[k + 1];
        st_f64x2(wp + k, wv, coh);
      }
    }
  };
  // phase 2 of sub-block sb: (c-diag) of step sb: tj in (sb, 3], ti in [tj, 3]; then the part
  // of (c-own) of step sb - 1 that the next (b-own) needs: column block tj = sb, ti in [4, 7]
  auto phase2 = [&](int sb) {
    const int nd = (3 - sb) * (4 - sb) / 2;
    const int no = sb > 0 ? 4 : 0;
    for (int e0 = wave; e0 < nd + no; e0 += 4) {
      if (e0 < nd) {
        int e = e0, tj = sb + 1;
        while (e >= 4 - tj) {
          e -= 4 - tj;
          ++tj;
        }
        update_tile16(sb, tj + e, tj);
      } else {
        update_tile16(sb - 1, 4 + (e0 - nd), sb);
      }
    }
  };
  // the rest of (c-own) of step sb - 2 (column blocks tj in [sb, 3]): wavefronts 2, 3 during
  // phase 1 of sub-block sb; its W sits in the other own-row buffer than the one (b-own) of
  // step sb - 1 is writing
  auto own_deferred = [&](int sb) {
    const int cnt = 4 * (4 - sb);
    for (int e = wave - 2; e < cnt; e += 2) update_tile16(sb - 2, 4 + (e & 3), sb + (e >> 2));
  };

  for (int sb = 0; sb < 4; ++sb) {
    // phase 1: (a+) of this step | (b-own) of the previous one | deferred (c-own) tiles
    if (wave == 0) a_plus(sb);
    if (wave == 1 && sb > 0) b_own(sb - 1);
    if (wave >= 2 && sb >= 2) own_deferred(sb);
    __syncthreads();
    // phase 2: (c-diag) of this step, urgent (c-own) column of the previous one
    phase2(sb);
    __syncthreads();
  }
  if (wave == 1) b_own(3);  // no tiles are left to update after the last step
  __syncthreads();

  // ---- write back: own rows (L), and by workgroup 0 the factored diagonal block
  for (int p = tid; p < 64 * 32; p += 256) {
    const int row = p >> 5, c2 = (p & 31) * 2;
    const int r = rbase + row;
    if (r < nrows) {
      const double2_t v = *reinterpret_cast<const double2_t *>(&M[64 + row][c2]);
      double *dst = K + (int64_t)r * ldk + c0 + c2;
      if (c2 + 1 < nb) st_f64x2(dst, v, coh);
      else if (c2 < nb) st_f64(dst, v.x, coh);
    }
  }
  if (wg == 0) {
    for (int p = tid; p < 64 * 64; p += 256) {
      const int row = p >> 6, c = p & 63;
      if (row < nb && c <= row) st_f64(K + (int64_t)(c0 + row) * ldk + c0 + c, M[row][c], coh);
    }
    if (tid < nb) {
      st_f64(dvec + c0 + tid, dD[tid], coh);
      st_f64(dinv + c0 + tid, dI[tid], coh);
    }
    if (wave == 0) {
      const unsigned long long negs = __ballot(lane < nb && dD[lane] < 0.0);
      if (lane == 0) {
        if (s_bad) atomicOr(&flags[0], 1);
        const int neg = __popcll(negs);
        if (neg) atomicAdd(&flags[1], neg);
      }
    }
  }
}

// bench only (EXP & 32): shader-clock and 100 MHz wall-clock stamps around one tile, to read
// the clock the chip actually sustains under this kernel's load
__device__ long long g_upd_clk[4];

template <int BM, int BN, int BK, int WR = 2, int WC = 2, int DB = 0, int EXP = 0>
__global__ __launch_bounds__(64 * WR * WC) void k_ldlt_update(
    double *__restrict__ K, int64_t ldk, const double *__restrict__ W, int64_t ldw, int N,
    int nrows, int row0, int col0, int colEnd, int kc0, int KB) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(DB ? 2 : 1) * (BM + BN) * (BK + 2) * 8];
  // (an XCD-contiguous tile order was tried and loses badly here: on the triangular domain
  // it starves the XCDs that get the short upper tile rows, 42 -> 24 TFLOP/s)
  const int bx = blockIdx.x, by = blockIdx.y;
  const int i0 = row0 + by * BM;
  const int j0 = col0 + bx * BN;
  if (j0 > i0 + BM - 1) return;  // tile entirely above the diagonal
  const bool stamp = (EXP & 32) && threadIdx.x == 0 && bx == 0 && by == (int)gridDim.y / 2;
  if (stamp) {
    g_upd_clk[0] = clock64();
    g_upd_clk[1] = wall_clock64();
  }
  update_tile<BM, BN, BK, WR, WC, DB, false, (EXP & 31)>(smem, threadIdx.x, i0, j0, K, ldk, W, ldw, N,
                                                         nrows, colEnd, kc0, KB);
  if (stamp) {
    g_upd_clk[2] = clock64();
    g_upd_clk[3] = wall_clock64();
  }
}

template <int NB>
__global__ __launch_bounds__(256) void k_ldlt_panel(double *__restrict__ K, int64_t ldk,
                                                    double *__restrict__ W, int64_t ldw, int wofs,
                                                    int N, int nrows, int c0,
                                                    double *__restrict__ dvec,
                                                    double *__restrict__ dinv,
                                                    int *__restrict__ flags, int skip) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PNL_SMEM];
  panel_body<NB>(smem, blockIdx.x, K, ldk, W, ldw, wofs, N, nrows, c0, dvec, dinv, flags, skip);
}

// ------------------------------------------------------------------ triangular solves
// Inverse of every 64 x 64 unit-lower diagonal block of L (one workgroup = one wavefront per
// block, all blocks in parallel, once per factorisation).  Lane c computes column c of
// inv(L_bb) by right-looking substitution on e_c (the recurrence of a row of X L^T = I);
// L_bb^T is broadcast from LDS.  Stored twice, inv(L_bb) and its transpose, both as
// [block][row][64], so that forward and backward solves read coalesced rows.
// The triangular solves then need no serial 63-step chain per block, only mat-vecs.
__device__ __forceinline__ void inv_diag_body(const double *__restrict__ K, int64_t ldk, int N,
                                              double *__restrict__ Linv,
                                              double *__restrict__ LinvT) {
  __shared__ __attribute__((aligned(16))) double Lt[64][64];  // Lt[t][j] = L_bb[j][t], j > t
  const int lane = threadIdx.x;
  const int b0 = blockIdx.x * 64;
  const int nb = min(64, N - b0);
  for (int idx = lane; idx < 64 * 64; idx += 64) {
    const int j = idx >> 6, t = idx & 63;
    Lt[t][j] = (j < nb && t < j) ? K[(int64_t)(b0 + j) * ldk + b0 + t] : 0.0;
  }
  __syncthreads();
  double y[64];  // y[j] = inv(L_bb)[j][lane]
#pragma unroll
  for (int j = 0; j < 64; ++j) y[j] = (j == lane) ? 1.0 : 0.0;
#pragma unroll
  for (int t = 0; t < 63; ++t) {
    const double yt = y[t];
#pragma unroll
    for (int j = t + 1; j < 64; ++j) y[j] = fma(-yt, Lt[t][j], y[j]);
  }
  double *o = Linv + (size_t)blockIdx.x * 4096;
  double *ot = LinvT + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int j = 0; j < 64; ++j) o[j * 64 + lane] = y[j];  // row j of inv(L_bb), coalesced
  // transpose through LDS (Lt is free now): row `lane` of the transpose = column written
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 64; ++j) Lt[lane][j] = y[j];         // Lt[c][j] = inv[j][c]
  __syncthreads();
  for (int idx = lane; idx < 64 * 64; idx += 64) ot[idx] = Lt[idx >> 6][idx & 63];
}

// Backward solve L^T s = w, one launch per SUPER-row super-block (4 x 64 sub-blocks),
// 4 wavefronts per workgroup.  Sub-blocks are walked from last to first.  Per sub-block:
//   * every wavefront first issues the 64 loads of the block-row segment it will fold
//     (wavefront q < sb: the earlier sub-block q of this super-block, redundantly in every
//     workgroup; wavefront 3: this workgroup's own 64 entries left of the super-block);
//   * wavefront 0 applies the transposed inverse of the diagonal block (a 64 x 64 mat-vec,
//     operands prefetched one sub-block ahead) and publishes the solved sub-block in LDS;
//   * after a barrier each wavefront folds the solved values into its target entries.
template <int SUPER>
__device__ __forceinline__ void trsv_bwd_body(const double *__restrict__ K, int64_t ldk,
                                              const double *__restrict__ Linv,
                                              double *__restrict__ z, double *__restrict__ x,
                                              int N, int c0) {
  static_assert(SUPER == 256, "four sub-blocks, four wavefronts");
  __shared__ double zs[SUPER];  // work entries of the super-block
  __shared__ double xs[64];     // solved sub-block
  __shared__ double rs[64];     // right-hand side of the sub-block being solved
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int width = min(SUPER, N - c0);
  const int nsub = (width + 63) / 64;
  zs[tid] = (tid < width) ? z[c0 + tid] : 0.0;
  // wavefront 3: this workgroup's own entry to the left of the super-block
  const int t = blockIdx.x * 64 + lane;
  const bool ext_live = (wave == 3) && (c0 > 0) && t < c0;
  double zext = ext_live ? z[t] : 0.0;
  __syncthreads();
  for (int sb = nsub - 1; sb >= 0; --sb) {
    const int b0 = c0 + sb * 64;
    const int nb = min(64, N - b0);
    // fold operand of this wavefront: column segment L[b0 .. b0+63][target]
    const bool in_fold = wave < sb;               // earlier sub-block `wave`
    const bool do_fold = in_fold || ext_live;
    const int tcol = in_fold ? (c0 + wave * 64 + lane) : t;
    const double *cp = K + (int64_t)b0 * ldk + (do_fold ? tcol : 0);
    double lv[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) lv[j] = (do_fold && j < nb) ? cp[(int64_t)j * ldk] : 0.0;
    if (wave == 0) {
      // x_i = sum_j inv(L_bb)[j][i] r_j : column i of the inverse, rows coalesced over lanes
      const double *ip = Linv + (size_t)(b0 / 64) * 4096 + lane;
      double iv[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) iv[j] = ip[j * 64];
      rs[lane] = zs[sb * 64 + lane];
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int j = 0; j < 64; j += 4) {
        a0 = fma(iv[j], rs[j], a0);
        a1 = fma(iv[j + 1], rs[j + 1], a1);
        a2 = fma(iv[j + 2], rs[j + 2], a2);
        a3 = fma(iv[j + 3], rs[j + 3], a3);
      }
      const double xv = (lane < nb) ? (a0 + a1) + (a2 + a3) : 0.0;
      xs[lane] = xv;
      if (blockIdx.x == 0 && lane < nb) x[b0 + lane] = xv;
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
      s0 = fma(lv[j], xs[j], s0);
      s1 = fma(lv[j + 1], xs[j + 1], s1);
      s2 = fma(lv[j + 2], xs[j + 2], s2);
      s3 = fma(lv[j + 3], xs[j + 3], s3);
    }
    const double sum = (s0 + s1) + (s2 + s3);
    if (in_fold) zs[wave * 64 + lane] -= sum;
    if (ext_live) zext -= sum;
    __syncthreads();
  }
  if (ext_live) z[t] = zext;
}

// Forward solve L y = b, mirror image of k_trsv_bwd_super: sub-blocks first to last;
// wavefront 0 applies inv(L_bb) (rows of the stored transpose are coalesced over lanes) and
// then folds into this workgroup's own 64 rows below the super-block; wavefront q > sb
// folds into the later sub-block q of the same super-block (redundantly per workgroup).
template <int SUPER>
__device__ __forceinline__ void trsv_fwd_body(const double *__restrict__ K, int64_t ldk,
                                              const double *__restrict__ LinvT,
                                              double *__restrict__ z, double *__restrict__ x,
                                              int N, int c0) {
  static_assert(SUPER == 256, "four sub-blocks, four wavefronts");
  __shared__ double zs[SUPER];
  __shared__ double xs[64];
  __shared__ double rs[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int width = min(SUPER, N - c0);
  const int nsub = (width + 63) / 64;
  zs[tid] = (tid < width) ? z[c0 + tid] : 0.0;
  // wavefront 0: this workgroup's own row below the super-block
  const int r = c0 + SUPER + blockIdx.x * 64 + lane;
  const bool ext_live = (wave == 0) && r < N;
  double zext = ext_live ? z[r] : 0.0;
  __syncthreads();
  for (int sb = 0; sb < nsub; ++sb) {
    const int b0 = c0 + sb * 64;
    const int nb = min(64, N - b0);
    const bool in_fold = (wave > sb) && (wave < nsub);  // later sub-block `wave`
    const bool do_fold = in_fold || ext_live;
    const int trow = in_fold ? (c0 + wave * 64 + lane) : r;
    const bool row_ok = do_fold && trow < N;
    const double *rp = K + (int64_t)(row_ok ? trow : 0) * ldk + b0;
    double lv[64];
#pragma unroll
    for (int j = 0; j < 64; j += 2) {
      double2_t v = (double2_t){0.0, 0.0};
      if (row_ok) v = *reinterpret_cast<const double2_t *>(rp + j);
      lv[j] = (j < nb) ? v.x : 0.0;
      lv[j + 1] = (j + 1 < nb) ? v.y : 0.0;
    }
    if (wave == 0) {
      const double *ip = LinvT + (size_t)(b0 / 64) * 4096 + lane;  // LinvT[j][i] = inv[i][j]
      double iv[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) iv[j] = ip[j * 64];
      rs[lane] = zs[sb * 64 + lane];
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int j = 0; j < 64; j += 4) {
        a0 = fma(iv[j], rs[j], a0);
        a1 = fma(iv[j + 1], rs[j + 1], a1);
        a2 = fma(iv[j + 2], rs[j + 2], a2);
        a3 = fma(iv[j + 3], rs[j + 3], a3);
      }
      const double xv = (lane < nb) ? (a0 + a1) + (a2 + a3) : 0.0;
      xs[lane] = xv;
      if (blockIdx.x == 0 && lane < nb) x[b0 + lane] = xv;
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
      s0 = fma(lv[j], xs[j], s0);
      s1 = fma(lv[j + 1], xs[j + 1], s1);
      s2 = fma(lv[j + 2], xs[j + 2], s2);
      s3 = fma(lv[j + 3], xs[j + 3], s3);
    }
    const double sum = (s0 + s1) + (s2 + s3);
    if (in_fold) zs[wave * 64 + lane] -= sum;
    if (ext_live) zext -= sum;
    __syncthreads();
  }
  if (ext_live) z[r] = zext;
}

__global__ __launch_bounds__(64) void k_inv_diag_blocks(const double *__restrict__ K, int64_t ldk,
                                                        int N, double *__restrict__ Linv,
                                                        double *__restrict__ LinvT) {
  inv_diag_body(K, ldk, N, Linv, LinvT);
}

template <int SUPER>
__global__ __launch_bounds__(256) void k_trsv_bwd_super(const double *__restrict__ K, int64_t ldk,
                                                         const double *__restrict__ Linv,
                                                         double *__restrict__ z,
                                                         double *__restrict__ x, int N, int c0) {
  trsv_bwd_body<SUPER>(K, ldk, Linv, z, x, N, c0);
}

template <int SUPER>
__global__ __launch_bounds__(256) void k_trsv_fwd_super(const double *__restrict__ K, int64_t ldk,
                                                         const double *__restrict__ LinvT,
                                                         double *__restrict__ z,
                                                         double *__restrict__ x, int N, int c0) {
  trsv_fwd_body<SUPER>(K, ldk, LinvT, z, x, N, c0);
}

// ------------------------------------------------------------------ chained triangular solves
// One launch for a whole triangular solve instead of one per 256-row super-block (20 dependent
// launches of ~15 us at N = 5120).  One workgroup per 64-row block; block b accumulates the
// products with the already solved blocks AS THEY BECOME AVAILABLE and finishes with the
// mat-vec by its inverted diagonal block.  The off-diagonal blocks are final data, so they are
// fetched before the entries they wait for.
//  * Hand-over: the DATA is the signal.  Every solve publishes into one half of `xpub`, which
//    holds the sentinel CHAIN_EMPTY (all bits set: a NaN no arithmetic produces) in every slot
//    when the solve starts; a producer stores its 64 solution entries there, a consumer polls
//    the very words it needs until none of them is the sentinel.  An aligned 8-byte store is
//    indivisible, so a word is either the sentinel or final -- no stamp, no store-to-store
//    ordering between data and stamp to get right (round 1's hand-over published x and then a
//    stamp: two stores of one wavefront to different L2 channels are not ordered without a
//    wait in between), and one L2 round trip per hand-over less.  The solve with epoch e uses
//    half e & 1 and its workers put the sentinel back into the OTHER half (kernel boundaries
//    order that against the next solve of the handle's stream).
//  * Placement: the workers are the workgroups with id % 8 == 0, which the round-robin dispatch
//    puts on ONE XCD; published entries are plain stores (the L1 is write-through: they land in
//    that XCD's L2 and stay there) read with L1-bypassing (agent-scope relaxed atomic) loads,
//    and therefore meet in that XCD's L2 -- no L2 write-back / invalidate per hand-over (with
//    agent-scope release / acquire fences a hand-over costs 11 us).  The placement is CHECKED,
//    not assumed: every worker reads its XCC id (s_getreg HW_REG_XCC_ID) and compares it with
//    the other workers' through one atomicMax; a mismatch or a timed-out wait sets ctl[1],
//    which the host reads at the next synchronisation: the solve is repeated with the
//    per-super-block kernels inside the same call and the chained solves are switched off.
//  * Termination: polls are bounded, every workgroup walks a finite loop, and a workgroup only
//    waits for workers with a SMALLER index, which are dispatched before it.
#define CHAIN_SPIN_LIMIT (1 << 18)  // ~0.15 s per wait; a real hand-over takes about a microsecond
#define CHAIN_EMPTY (-1LL)

// all workers of a launch must sit on one XCD: ctl[0] = max over workers of (epoch << 4 | xcc)
__device__ __forceinline__ void chain_check_xcc(int epoch, int *ctl) {
  if (threadIdx.x == 0) {
    // 27 bits of epoch: after 2^27 solves the slot stops changing and the check goes quiet
    // (no false alarms)
    const int ep = epoch & 0x7ffffff;
    const int mine = (ep << 4) | (int)(__builtin_amdgcn_s_getreg(6164) & 15);  // XCC_ID[3:0]
    const int old = atomicMax(&ctl[0], mine);
    if ((old >> 4) == ep && old != mine) atomicOr(&ctl[1], 2);
  }
}

__device__ __forceinline__ long long chain_peek(const double *p) {
  return __hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// lanes [0, cnt) of the wavefront hold one published entry each (`bits`, possibly still the
// sentinel: the caller issued that load earlier); returns once none is
__device__ __forceinline__ double chain_take(const double *p, bool mine, long long bits, int *ctl) {
  for (int it = 0; it < CHAIN_SPIN_LIMIT; ++it) {
    if (!__builtin_amdgcn_ballot_w64(mine && bits == CHAIN_EMPTY)) return __longlong_as_double(bits);
    __builtin_amdgcn_s_sleep(1);
    if (mine) bits = chain_peek(p);
  }
  atomicOr(&ctl[1], 1);
  return 0.0;
}

__device__ __forceinline__ double chain_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// workers put the sentinel back into the half the NEXT solve publishes into
__device__ __forceinline__ void chain_reset_other(double *other, int capblk, int w, int nw) {
  if (threadIdx.x < 64)
    for (int j = w; j < capblk; j += nw)
      reinterpret_cast<long long *>(other)[(size_t)j * 64 + threadIdx.x] = CHAIN_EMPTY;
}

// backward: L^T x = z.  x_b = inv(L_bb)^T (z_b - sum_{a > b} L_ab^T x_a).  In place (x == z) ok.
// w: the worker's place in the chain (worker 0 solves the last block); xpub2: both publication
// halves (capblk * 64 entries each); ctl: [0] XCC slot, [1] status, [2] the handle's solve
// counter -- kept on the DEVICE, so that single-instance and batched solves of a pooled handle
// agree on which half is clean: every worker reads it when it starts, the worker that ends the
// chain (and therefore runs after every other worker has read it) advances it.
__device__ __forceinline__ void chain_bwd_body(const double *__restrict__ K, int64_t ldk,
                                               const double *__restrict__ Linv, const double *z,
                                               double *x, int N, double *xpub2, int capblk,
                                               int *__restrict__ ctl, int w) {
  const int epoch = ctl[2] + 1;
  double *xpub = xpub2 + (size_t)(epoch & 1) * capblk * 64;
  double *xother = xpub2 + (size_t)((epoch + 1) & 1) * capblk * 64;
  chain_check_xcc(epoch, ctl);
  __shared__ double part[4][64];
  __shared__ double rs[64];
  const int nblk = (N + 63) / 64;
  const int b = nblk - 1 - w;  // last block first
  const int b0 = b * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  chain_reset_other(xother, capblk, w, nblk);
  // inverse of the diagonal block, rows [16 wave, 16 wave + 16): inv[j][lane] -- parked in LDS
  // until the block's own turn.  All workers of a solve must be RESIDENT on the one XCD (32
  // CUs): a worker that starts late walks the whole chain behind the others.  In registers the
  // inverse took the kernel to 146 VGPRs = 3 workgroups per CU = 96 workers (N <= 6144).
  __shared__ double ivs[64][64];
  {
    const double *ip = Linv + (size_t)b * 4096 + (size_t)(16 * wave) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) ivs[16 * wave + t][lane] = ip[t * 64];
  }
  const double zb = (wave == 0 && b0 + lane < N) ? z[b0 + lane] : 0.0;
  // column `lane` of L_ab, rows [16 wave, 16 wave + 16), for a = nblk - 1 ... b + 1
  auto fetch = [&](int a, double (&lv)[16]) {
    const int r0 = a * 64 + 16 * wave;
    const double *cp = K + (int64_t)r0 * ldk + b0 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) lv[t] = (r0 + t < N) ? cp[(int64_t)t * ldk] : 0.0;
  };
  // this wavefront's 16 entries of block a: lane t < 16 holds entry 16 wave + t (rows beyond N
  // are never published and count as zero)
  auto slot = [&](int a) { return xpub + a * 64 + 16 * wave + lane; };
  auto wanted = [&](int a) { return lane < 16 && a * 64 + 16 * wave + lane < N; };
  double acc4[4] = {0.0, 0.0, 0.0, 0.0};
  double cur[16], nxt[16];
  int a = nblk - 1;
  long long pend = 0;
  if (a > b) {
    fetch(a, cur);
    if (wanted(a)) pend = chain_peek(slot(a));
  }
  for (; a > b; --a) {
    if (a - 1 > b) fetch(a - 1, nxt);
    const double xa = chain_take(slot(a), wanted(a), pend, ctl);
    // the next block's entries are asked for now: behind the front they are there already
    pend = 0;
    if (a - 1 > b && wanted(a - 1)) pend = chain_peek(slot(a - 1));
    // four independent sums: a dependent fp64 FMA of a lone wavefront costs ~35 cycles
#pragma unroll
    for (int t = 0; t < 16; ++t) acc4[t & 3] = fma(cur[t], chain_bcast(xa, t), acc4[t & 3]);
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[t] = nxt[t];
  }
  part[wave][lane] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  __syncthreads();
  if (wave == 0) rs[lane] = zb - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
  __syncthreads();
  // x_i = sum_j inv[j][i] r_j  (j >= i; the stored inverse is zero above the diagonal)
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 16; ++t) s4[t & 3] = fma(ivs[16 * wave + t][lane], rs[16 * wave + t], s4[t & 3]);
  part[wave][lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  __syncthreads();
  if (wave == 0) {
    const double xv = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (b0 + lane < N) {
      // plain stores: they stay in the XCD's L2, where the consumers' L1-bypassing loads find
      // them (sc1 stores are write-through AND drop the line from L2)
      xpub[b0 + lane] = xv;
      x[b0 + lane] = xv;
    }
    if (b == 0 && lane == 0) ctl[2] = epoch;  // end of the chain: every worker has read it
  }
}

__global__ __launch_bounds__(256) void k_trsv_bwd_chain(const double *__restrict__ K, int64_t ldk,
                                                        const double *__restrict__ Linv,
                                                        const double *z, double *x, int N,
                                                        double *xpub2, int capblk,
                                                        int *__restrict__ ctl) {
  if (blockIdx.x & 7) return;
  chain_bwd_body(K, ldk, Linv, z, x, N, xpub2, capblk, ctl, (int)(blockIdx.x >> 3));
}

// forward: L y = z.  y_b = inv(L_bb) (z_b - sum_{a < b} L_ba y_a).  In place ok.
__device__ __forceinline__ void chain_fwd_body(const double *__restrict__ K, int64_t ldk,
                                               const double *__restrict__ LinvT, const double *z,
                                               double *x, int N, double *xpub2, int capblk,
                                               int *__restrict__ ctl, int b) {
  const int epoch = ctl[2] + 1;
  double *xpub = xpub2 + (size_t)(epoch & 1) * capblk * 64;
  double *xother = xpub2 + (size_t)((epoch + 1) & 1) * capblk * 64;
  chain_check_xcc(epoch, ctl);
  __shared__ double part[4][64];
  __shared__ double rs[64];
  const int nblk = (N + 63) / 64;  // b: first block first
  const int b0 = b * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  chain_reset_other(xother, capblk, b, nblk);
  // LinvT[j][i] = inv[i][j]: rows j in [16 wave, 16 wave + 16), column i = lane (parked in LDS,
  // see k_trsv_bwd_chain)
  __shared__ double ivs[64][64];
  {
    const double *ip = LinvT + (size_t)b * 4096 + (size_t)(16 * wave) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) ivs[16 * wave + t][lane] = ip[t * 64];
  }
  const double zb = (wave == 0 && b0 + lane < N) ? z[b0 + lane] : 0.0;
  // Wavefront w takes columns [16 w, 16 w + 16) of every block a.  Two layouts:
  //  * blocks a < b - 1 (the bulk: 3160 of the 3240 blocks at N = 5120) COALESCED: lane
  //    (r4, c) = (lane >> 4, lane & 15) holds column 16 w + c of the rows 4 i + r4, i < 16 --
  //    four full cache lines per load instruction -- its multiplier is the published entry
  //    16 w + c (all four lane groups poll the same 16 words, no broadcast), and the 16 row sums
  //    over c are formed ONCE, by a reduce-scatter over the 16-lane groups while the wavefront
  //    waits for the last block's entries.
  //  * block b - 1 (the hand-over path) with lane <-> ROW: 128 contiguous bytes per lane, the
  //    multipliers wavefront-uniform (v_readlane), no cross-lane reduction behind the wait.
  // With lane <-> row for every block the 64 cache lines per load instruction made the ONE
  // XCD's texture addressers the bound: 2.2 us per hand-over against 1.4 us backward.
  const int r4 = lane >> 4, c = lane & 15;
  auto fetch = [&](int a, double (&lv)[16]) {
    const double *rp = K + (int64_t)(b0 + r4) * ldk + a * 64 + 16 * wave + c;
#pragma unroll
    for (int i = 0; i < 16; ++i) lv[i] = (b0 + 4 * i + r4 < N) ? rp[(int64_t)(4 * i) * ldk] : 0.0;
  };
  const bool row_ok = b0 + lane < N;
  auto fetch_rows = [&](int a, double (&lv)[16]) {
    const double *rp = K + (int64_t)(b0 + lane) * ldk + a * 64 + 16 * wave;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      double2_t v = (double2_t){0.0, 0.0};
      if (row_ok) v = *reinterpret_cast<const double2_t *>(rp + t);
      lv[t] = v.x;
      lv[t + 1] = v.y;
    }
  };
  // blocks a < b are full (only the last block of a solve can be ragged)
  __shared__ double bulk[4][64];
  const int nbulk = b - 1;  // blocks 0 .. b - 2
  // (the last block's rows travel in the prefetch registers of the bulk loop: behind the last
  // bulk block there is nothing else to fetch)
  double cur[16], nxt[16];
  if (b == 1) fetch_rows(0, cur);
  {
    auto slot = [&](int a) { return xpub + a * 64 + 16 * wave + c; };
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    long long pend = 0;
    if (nbulk > 0) {
      fetch(0, cur);
      pend = chain_peek(slot(0));
    }
    for (int a = 0; a < nbulk; ++a) {
      if (a + 1 < nbulk) fetch(a + 1, nxt);
      else fetch_rows(b - 1, nxt);
      const double xa = chain_take(slot(a), true, pend, ctl);
      pend = 0;
      if (a + 1 < nbulk) pend = chain_peek(slot(a + 1));
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fma(cur[i], xa, acc[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
    }
    // reduce-scatter over c: lane (r4, c) ends with the sum of row 4 c + r4
    double v8[8], v4[4], v2[2];
    const bool h3 = c & 8, h2 = c & 4, h1 = c & 2, h0 = c & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v8[j] = (h3 ? acc[j + 8] : acc[j]) + __shfl_xor(h3 ? acc[j] : acc[j + 8], 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) v4[j] = (h2 ? v8[j + 4] : v8[j]) + __shfl_xor(h2 ? v8[j] : v8[j + 4], 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) v2[j] = (h1 ? v4[j + 2] : v4[j]) + __shfl_xor(h1 ? v4[j] : v4[j + 2], 2);
    bulk[wave][4 * c + r4] = (h0 ? v2[1] : v2[0]) + __shfl_xor(h0 ? v2[0] : v2[1], 1);
  }
  double accr = 0.0;
  if (b > 0) {
    const bool mine = lane < 16;
    const double *sl = xpub + (b - 1) * 64 + 16 * wave + lane;
    const double xa = chain_take(sl, mine, mine ? chain_peek(sl) : 0, ctl);
    // four independent sums: a dependent fp64 FMA of a lone wavefront costs ~35 cycles
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 16; ++t) acc4[t & 3] = fma(cur[t], chain_bcast(xa, t), acc4[t & 3]);
    accr = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  }
  part[wave][lane] = accr + bulk[wave][lane];  // (own wavefront's LDS words: in program order)
  __syncthreads();
  if (wave == 0) rs[lane] = zb - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
  __syncthreads();
  // y_i = sum_j inv[i][j] r_j
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 16; ++t) s4[t & 3] = fma(ivs[16 * wave + t][lane], rs[16 * wave + t], s4[t & 3]);
  part[wave][lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  __syncthreads();
  if (wave == 0) {
    const double xv = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (b0 + lane < N) {
      xpub[b0 + lane] = xv;
      x[b0 + lane] = xv;
    }
    if (b == nblk - 1 && lane == 0) ctl[2] = epoch;  // end of the chain
  }
}

__global__ __launch_bounds__(256, 3) void k_trsv_fwd_chain(const double *__restrict__ K, int64_t ldk,
                                                           const double *__restrict__ LinvT,
                                                           const double *z, double *x, int N,
                                                           double *xpub2, int capblk,
                                                           int *__restrict__ ctl) {
  if (blockIdx.x & 7) return;
  chain_fwd_body(K, ldk, LinvT, z, x, N, xpub2, capblk, ctl, (int)(blockIdx.x >> 3));
}

// ------------------------------------------------------------------ batched variants
// One instance per blockIdx.z (BInst, pgf_internal.h).  Every instance has its own reduced
// size N = counts[0] + m, known only on the device: the host walks the panel / update
// schedule of the largest possible system (n + m) and workgroups beyond an instance's own
// N return at once.  ctl[0] == 0 (factor still valid) skips the factorisation kernels.
// Workgroup -> (instance, tile): consecutive workgroup ids go round-robin over the 8 XCDs, each
// with its own L2.  Instance i is pinned to XCD i % 8 (id % 8 selects the residue class, the
// rest of the id walks that class instance by instance, tile by tile), so the tiles of one
// instance that run together share one L2 instead of every L2 seeing every instance:
// measured on the K = 256 trailing update of 256 instances: 19.7 -> 29.9 TFLOP/s.
// (batch_decode / batch_grid: pgf_ldlt_dev.h)

template <int NB>
__global__ __launch_bounds__(256) void kb_ldlt_panel(const BInst *__restrict__ tab, int B, int per,
                                                     int m, int64_t ldw, int wbuf, int ob0,
                                                     int c0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PNL_SMEM];
  int inst, wg;
  if (!batch_decode(B, per, inst, wg)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m, nrows = N + 1;
  if (c0 >= N) return;
  const int below = nrows - min(c0 + NB, N);
  const int npw = max(1, (below + 63) / 64);
  if (wg >= npw) return;
  panel_body<NB>(smem, wg, I.K, I.ldk, I.W + (int64_t)wbuf * I.wstride, ldw, c0 - ob0, N, nrows, c0,
                 I.dvec, I.dinv, I.flags, 0);
}

__global__ __launch_bounds__(256) void kb_ldlt_update(const BInst *__restrict__ tab, int B, int tc,
                                                      int tr, int m, int64_t ldw, int wbuf,
                                                      int wcol, int row0, int col0, int colEndArg,
                                                      int kc0, int KB) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(64 + 64) * (16 + 2) * 8];
  int inst, t;
  if (!batch_decode(B, tc * tr, inst, t)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m, nrows = N + 1;
  const int colEnd = min(colEndArg, N);
  const int by = t / tc, bx = t - by * tc;
  const int i0 = row0 + by * 64;
  const int j0 = col0 + bx * 64;
  if (i0 >= nrows || j0 >= colEnd || j0 > i0 + 63) return;
  update_tile<64, 64, 16>(smem, threadIdx.x, i0, j0, I.K, I.ldk,
                          I.W + (int64_t)wbuf * I.wstride + wcol, ldw, N, nrows, colEnd, kc0, KB);
}
// the same tiles for a pre-eliminated block (condensed order): C -= (V vd) V^T over the whole
// lower triangle (+ row N), depth KB, operands from the instance's panel
__global__ __launch_bounds__(256) void kb_ldlt_update_virtual(const BInst *__restrict__ tab, int B, int tc,
                                                              int tr, int KB) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(64 + 64) * (16 + 2) * 8];
  int inst, t;
  if (!batch_decode(B, tc * tr, inst, t)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0], nrows = N + 1;
  const int by = t / tc, bx = t - by * tc;
  const int i0 = by * 64, j0 = bx * 64;
  if (i0 >= nrows || j0 >= N || j0 > i0 + 63) return;
  update_tile<64, 64, 16, 2, 2, 0, true>(smem, threadIdx.x, i0, j0, I.K, I.ldk, I.V, I.ldv, N, nrows, N, 0, KB,
                                         I.vd);
}

// ---- left-looking panel step of the batched schedule --------------------------------------
// In batched mode throughput counts, not the latency of one panel, so the panel step is split:
//   kb_diag_ll : ONE workgroup per instance brings the 64 x 64 diagonal tile up to date with
//                the earlier panels of its outer block (left-looking, MFMA), factorises it
//                (panel_body on a preloaded tile) and inverts the unit-lower factor;
//   kb_trsm_ll : one workgroup per 64 rows below: left-looking update of its tile, then
//                X = T inv(L_bb)^T as a 64^3 MFMA product; stores W = X and L = X D^-1.
// No workgroup repeats the diagonal factorisation and there are no K = 64 trailing updates
// (their C traffic and launches are gone); the inverses are the ones the triangular solves
// need anyway.
#define LL_LD 66
#define TRSM_SMEM (2 * 64 * LL_LD * 8 + 64 * 8)

// acc (2 x 2 MFMA tiles per wavefront, quadrant (wr, wc) of a 64 x 64 tile) -=
//   sum_k A[i0 + i][k] * B[j0 + j][kb0 + k], k < kp, through the LDS staging area `stg`
__device__ __forceinline__ void ll_accumulate(double4_t (&acc)[2][2], unsigned char *stg,
                                              const double *__restrict__ A, int64_t lda, int i0,
                                              int ilim, const double *__restrict__ Bm, int64_t ldb,
                                              int j0, int jlim, int kb0, int kp) {
  constexpr int LD = 18;
  double(*As)[LD] = reinterpret_cast<double(*)[LD]>(stg);
  double(*Bs)[LD] = reinterpret_cast<double(*)[LD]>(stg + 64 * LD * 8);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  // 64 rows x 8 double2 pieces = 512 pieces per operand: two per lane
  double2_t pa[2], pb[2];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = q * 256 + tid;
      const int row = p >> 3, kofs = (p & 7) * 2;
      double2_t va = (double2_t){0.0, 0.0}, vb = (double2_t){0.0, 0.0};
      if (i0 + row < ilim)
        va = *reinterpret_cast<const double2_t *>(A + (int64_t)(i0 + row) * lda + kk + kofs);
      if (j0 + row < jlim)
        vb = *reinterpret_cast<const double2_t *>(Bm + (int64_t)(j0 + row) * ldb + kb0 + kk + kofs);
      pa[q] = va;
      pb[q] = vb;
    }
  };
  if (kp > 0) fetch(0);
  for (int kk = 0; kk < kp; kk += 16) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = q * 256 + tid;
      *reinterpret_cast<double2_t *>(&As[p >> 3][(p & 7) * 2]) = -pa[q];
      *reinterpret_cast<double2_t *>(&Bs[p >> 3][(p & 7) * 2]) = pb[q];
    }
    __syncthreads();
    if (kk + 16 < kp) fetch(kk + 16);
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      double a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[wr * 32 + t * 16 + l15][ks + l4];
        b[t] = Bs[wc * 32 + t * 16 + l15][ks + l4];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
  }
  __syncthreads();  // staging area free again
}

// Two tiles that share their B operand (the L rows of the diagonal block): accD uses A rows
// iD0.., accO uses A rows iO0..; one staging pass and one barrier pair per 16-deep chunk for
// both.  Staging: 3 x 64 x 18 doubles.
__device__ __forceinline__ void ll_accumulate2(double4_t (&accD)[2][2], double4_t (&accO)[2][2],
                                               unsigned char *stg,
                                               const double *__restrict__ A, int64_t lda, int iD0,
                                               int iDlim, int iO0, int iOlim,
                                               const double *__restrict__ Bm, int64_t ldb, int j0,
                                               int jlim, int kb0, int kp) {
  constexpr int LD = 18;
  double(*Ad)[LD] = reinterpret_cast<double(*)[LD]>(stg);
  double(*Ao)[LD] = reinterpret_cast<double(*)[LD]>(stg + 64 * LD * 8);
  double(*Bs)[LD] = reinterpret_cast<double(*)[LD]>(stg + 2 * 64 * LD * 8);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  double2_t pd[2], po[2], pb[2];
  auto fetch = [&](int kk) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = q * 256 + tid;
      const int row = p >> 3, kofs = (p & 7) * 2;
      double2_t vd = (double2_t){0.0, 0.0}, vo = vd, vb = vd;
      if (iD0 + row < iDlim)
        vd = *reinterpret_cast<const double2_t *>(A + (int64_t)(iD0 + row) * lda + kk + kofs);
      if (iO0 + row < iOlim)
        vo = *reinterpret_cast<const double2_t *>(A + (int64_t)(iO0 + row) * lda + kk + kofs);
      if (j0 + row < jlim)
        vb = *reinterpret_cast<const double2_t *>(Bm + (int64_t)(j0 + row) * ldb + kb0 + kk + kofs);
      pd[q] = vd;
      po[q] = vo;
      pb[q] = vb;
    }
  };
  if (kp > 0) fetch(0);
  for (int kk = 0; kk < kp; kk += 16) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = q * 256 + tid;
      *reinterpret_cast<double2_t *>(&Ad[p >> 3][(p & 7) * 2]) = -pd[q];
      *reinterpret_cast<double2_t *>(&Ao[p >> 3][(p & 7) * 2]) = -po[q];
      *reinterpret_cast<double2_t *>(&Bs[p >> 3][(p & 7) * 2]) = pb[q];
    }
    __syncthreads();
    if (kk + 16 < kp) fetch(kk + 16);
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      double ad[2], ao[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ad[t] = Ad[wr * 32 + t * 16 + l15][ks + l4];
        ao[t] = Ao[wr * 32 + t * 16 + l15][ks + l4];
        b[t] = Bs[wc * 32 + t * 16 + l15][ks + l4];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          accD[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[mi], b[nj], accD[mi][nj], 0, 0, 0);
          accO[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[mi], b[nj], accO[mi][nj], 0, 0, 0);
        }
    }
  }
  __syncthreads();
}

// Single-instance left-looking panel (default; PGF_PANEL_LL=0 disables): the fused panel kernel with a prologue
// that brings the diagonal tile and the workgroup's own tile up to date with the earlier
// panels of the outer block (MFMA, operands from the L2-resident W and L), so that NO K = 64
// trailing-update launches are needed between the panels of an outer block.
template <int NB>
__global__ __launch_bounds__(256) void k_ldlt_panel_ll(double *__restrict__ K, int64_t ldk,
                                                       double *__restrict__ W, int64_t ldw,
                                                       int ob0, int N, int nrows, int c0,
                                                       double *__restrict__ dvec,
                                                       double *__restrict__ dinv,
                                                       int *__restrict__ flags) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PNL_SMEM];
  double(*M)[PNL_LD] = reinterpret_cast<double(*)[PNL_LD]>(smem);
  unsigned char *stg = smem + 128 * PNL_LD * 8;  // the W-tile area is free until (a+)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int nb = min(NB, N - c0);
  const int rbase = c0 + nb + (int)blockIdx.x * 64;
  const int kp = c0 - ob0;
  double4_t accD[2][2], accO[2][2];
  // diagonal tile (lower triangle) and own tile (rows rbase .., columns c0 .. c0 + nb)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int j = wc * 32 + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = wr * 32 + mi * 16 + l4 + 4 * r;
        accD[mi][nj][r] = (i < nb && j <= i) ? K[(int64_t)(c0 + i) * ldk + c0 + j] : 0.0;
        accO[mi][nj][r] = (rbase + i < nrows && j < nb) ? K[(int64_t)(rbase + i) * ldk + c0 + j] : 0.0;
      }
    }
  ll_accumulate2(accD, accO, stg, W, ldw, c0, c0 + nb, rbase, nrows, K, ldk, c0, c0 + nb, ob0, kp);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int j = wc * 32 + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = wr * 32 + mi * 16 + l4 + 4 * r;
        double v = accD[mi][nj][r];
        if (i >= nb) v = (i == j) ? 1.0 : 0.0;  // identity outside the valid part
        else if (j > i) v = 0.0;
        M[i][j] = v;
        M[64 + i][j] = accO[mi][nj][r];
      }
    }
  __syncthreads();
  panel_body<NB, true>(smem, blockIdx.x, K, ldk, W, ldw, kp, N, nrows, c0, dvec, dinv, flags, 0);
}

__global__ __launch_bounds__(256) void kb_diag_ll(const BInst *__restrict__ tab, int m, int64_t ldw,
                                                  int wbuf, int ob0, int c0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PNL_SMEM];
  const BInst &I = tab[blockIdx.x];  // workgroup i -> XCD i % 8, as batch_decode pins it
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m;
  if (c0 >= N) return;
  const int nb = min(64, N - c0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  double(*M)[PNL_LD] = reinterpret_cast<double(*)[PNL_LD]>(smem);
  const double *Wb = I.W + (int64_t)wbuf * I.wstride;
  // lower triangle of the diagonal tile, updated by the panels [ob0, c0) of this outer block
  double4_t acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int j = wc * 32 + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = wr * 32 + mi * 16 + l4 + 4 * r;
        acc[mi][nj][r] = (i < nb && j <= i) ? I.K[(int64_t)(c0 + i) * I.ldk + c0 + j] : 0.0;
      }
    }
  ll_accumulate(acc, smem + 64 * PNL_LD * 8, Wb, ldw, c0, c0 + nb, I.K, I.ldk, c0, c0 + nb, ob0,
                c0 - ob0);
  // M rows 0..63 <- tile (identity outside the valid lower triangle), rows 64..127 <- 0
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int j = wc * 32 + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = wr * 32 + mi * 16 + l4 + 4 * r;
        double v = acc[mi][nj][r];
        if (i >= nb) v = (i == j) ? 1.0 : 0.0;
        else if (j > i) v = 0.0;
        M[i][j] = v;
      }
    }
  for (int p = tid; p < 64 * 32; p += 256)
    *reinterpret_cast<double2_t *>(&M[64 + (p >> 5)][(p & 31) * 2]) = (double2_t){0.0, 0.0};
  __syncthreads();
  // factorise in place (no rows below: nrows = c0 + nb); writes the tile, D, 1/D, flags
  panel_body<PGF_NB, true>(smem, 0, I.K, I.ldk, I.W + (int64_t)wbuf * I.wstride, ldw, c0 - ob0, N,
                           c0 + nb, c0, I.dvec, I.dinv, I.flags, 0);
  __syncthreads();
  // inverse of the unit-lower factor: lane c of wavefront 0 owns column c (substitution on
  // e_c, L broadcast from LDS); rows >= nb of M are identity rows, so is their inverse
  double(*Tt)[PNL_LD] = reinterpret_cast<double(*)[PNL_LD]>(smem + 64 * PNL_LD * 8);
  const size_t blk = (size_t)(c0 / 64) * 4096;
  if (wave == 0) {
    double y[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) y[j] = (j == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < 63; ++t) {
      const double yt = y[t];
#pragma unroll
      for (int j = t + 1; j < 64; ++j) y[j] = fma(-yt, M[j][t], y[j]);
    }
    double *o = I.Linv + blk;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      o[j * 64 + lane] = y[j];  // row j of the inverse, coalesced
      Tt[lane][j] = y[j];       // Tt[c][j] = inv[j][c]
    }
  }
  __syncthreads();
  double *ot = I.LinvT + blk;
  for (int p = tid; p < 64 * 64; p += 256) ot[p] = Tt[p >> 6][p & 63];
}

__global__ __launch_bounds__(256, 2) void kb_trsm_ll(const BInst *__restrict__ tab, int B, int per,
                                                     int m, int64_t ldw, int wbuf, int ob0,
                                                     int c0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TRSM_SMEM];
  int inst, wg;
  if (!batch_decode(B, per, inst, wg)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m, nrows = N + 1;
  if (c0 >= N) return;
  const int nb = min(64, N - c0);
  const int r0 = c0 + nb + wg * 64;
  if (r0 >= nrows) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  double(*Ts)[LL_LD] = reinterpret_cast<double(*)[LL_LD]>(smem);
  double(*Is)[LL_LD] = reinterpret_cast<double(*)[LL_LD]>(smem + 64 * LL_LD * 8);
  double *ds = reinterpret_cast<double *>(smem + 2 * 64 * LL_LD * 8);
  double *Wb = I.W + (int64_t)wbuf * I.wstride;
  // inverse of the diagonal factor and 1/D: issued first, consumed after the update loop
  const double *ip = I.Linv + (size_t)(c0 / 64) * 4096;
  double2_t iv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    iv[q] = *reinterpret_cast<const double2_t *>(ip + (size_t)(q * 256 + tid) * 2);
  const double dv = (tid < nb) ? I.dinv[c0 + tid] : 0.0;
  // own tile, columns < nb
  double4_t acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int j = wc * 32 + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = r0 + wr * 32 + mi * 16 + l4 + 4 * r;
        acc[mi][nj][r] = (i < nrows && j < nb) ? I.K[(int64_t)i * I.ldk + c0 + j] : 0.0;
      }
    }
  ll_accumulate(acc, smem, Wb, ldw, r0, nrows, I.K, I.ldk, c0, c0 + nb, ob0, c0 - ob0);
  // T and inv(L) to LDS
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Ts[wr * 32 + mi * 16 + l4 + 4 * r][wc * 32 + nj * 16 + l15] = acc[mi][nj][r];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int p = q * 256 + tid;
    *reinterpret_cast<double2_t *>(&Is[p >> 5][(p & 31) * 2]) = iv[q];
  }
  if (tid < 64) ds[tid] = dv;
  __syncthreads();
  // X[i][j] = sum_k T[i][k] inv[j][k]
  double4_t x[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) x[mi][nj] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int ks = 0; ks < 64; ks += 4) {
    double a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t] = Ts[wr * 32 + t * 16 + l15][ks + l4];
      b[t] = Is[wc * 32 + t * 16 + l15][ks + l4];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
        x[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], x[mi][nj], 0, 0, 0);
  }
  __syncthreads();  // all reads of Ts done: reuse it for X
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Ts[wr * 32 + mi * 16 + l4 + 4 * r][wc * 32 + nj * 16 + l15] = x[mi][nj][r];
  __syncthreads();
  // coalesced stores: W = X (the L D the updates multiply with), L = X D^-1
  const int wofs = c0 - ob0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int p = q * 256 + tid;
    const int row = p >> 5, c2 = (p & 31) * 2;
    const int r = r0 + row;
    if (r >= nrows || c2 >= nb) continue;
    const double2_t w = *reinterpret_cast<const double2_t *>(&Ts[row][c2]);
    double2_t l;
    l.x = w.x * ds[c2];
    l.y = w.y * ds[c2 + 1];
    double *wp = Wb + (int64_t)r * ldw + wofs + c2;
    double *kp = I.K + (int64_t)r * I.ldk + c0 + c2;
    if (c2 + 1 < nb) {
      *reinterpret_cast<double2_t *>(wp) = w;
      *reinterpret_cast<double2_t *>(kp) = l;
    } else {
      wp[0] = w.x;
      kp[0] = l.x;
    }
  }
}

__global__ __launch_bounds__(64) void kb_inv_diag_blocks(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m;
  if ((int)blockIdx.x * 64 >= N) return;
  inv_diag_body(I.K, I.ldk, N, I.Linv, I.LinvT);
}

// forward half for instances whose factor is reused: zwork <- rhs
__global__ void kb_solve_prep_fwd(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] != 0 || I.ctl[3]) return;
  const int N = I.counts[0] + m;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) I.zwork[i] = I.rhs[i];
}

__global__ __launch_bounds__(256) void kb_trsv_fwd_super(const BInst *__restrict__ tab, int m,
                                                         int c0) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] != 0 || I.ctl[3]) return;
  const int N = I.counts[0] + m;
  if (c0 >= N) return;
  const int below = N - (c0 + 256);
  const int g = below > 0 ? (below + 63) / 64 : 1;
  if ((int)blockIdx.x >= g) return;
  trsv_fwd_body<256>(I.K, I.ldk, I.LinvT, I.zwork, I.sol, N, c0);
}

// zwork <- D^-1 L^-1 rhs: row N of K after a factorisation, D^-1 * (forward result) otherwise
__global__ void kb_solve_prep_bwd(const BInst *__restrict__ tab, int m) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  const int N = I.counts[0] + m;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) I.flags[3] = 0;  // the sampled residual check of this step (kb_sample_residual)
  if (i >= N) return;
  I.zwork[i] = I.ctl[0] ? I.K[(int64_t)N * I.ldk + i] : I.sol[i] * I.dinv[i];
}

__global__ __launch_bounds__(256) void kb_trsv_bwd_super(const BInst *__restrict__ tab, int m,
                                                         int c0) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[3]) return;
  const int N = I.counts[0] + m;
  if (c0 >= N) return;
  const int g = c0 > 0 ? (c0 + 63) / 64 : 1;
  if ((int)blockIdx.x >= g) return;
  trsv_bwd_body<256>(I.K, I.ldk, I.Linv, I.zwork, I.sol, N, c0);
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
  if ((e = hipMalloc((void **)&f.K, rows * f.ldk * sizeof(double))) != hipSuccess) return e;
  f.OB = 256;
  if (const char *ob = getenv("PGF_OB")) {  // legacy schedule only (PGF_FACTOR=1)
    const int v = atoi(ob);
    if (v == 64 || v == 128 || v == 192 || v == 256) f.OB = v;
  }
  f.wstride = rows * (size_t)256;
  if ((e = hipMalloc((void **)&f.W, 2 * f.wstride * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.dvec, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.dinv, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.zwork, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.Linv, (rows / 64 + 1) * 4096 * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.LinvT, (rows / 64 + 1) * 4096 * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.flags, (4 + LDLT_UPD_COUNTERS) * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMemset(f.flags, 0, (4 + LDLT_UPD_COUNTERS) * sizeof(int))) != hipSuccess) return e;
  f.chain_stride = (int)(rows / 64 + 2);
  if ((e = hipMalloc(&f.chain, (2 * f.chain_stride + 4) * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMemset(f.chain, 0, (2 * f.chain_stride + 4) * sizeof(int))) != hipSuccess) return e;
  // publication halves of the chained solves, every slot = the sentinel (all bits set)
  if ((e = hipMalloc(&f.xpub, 2 * (size_t)f.chain_stride * 64 * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMemset(f.xpub, 0xff, 2 * (size_t)f.chain_stride * 64 * sizeof(double))) != hipSuccess) return e;
  // [0, 16): chain <-> helpers, [32, 96): row groups of T(k) -> tiles of the next diagonal block
  if ((e = hipMalloc(&f.hctl, 128 * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMemset(f.hctl, 0, 128 * sizeof(int))) != hipSuccess) return e;
  if ((e = hipHostMalloc(&f.h_flags, 4 * sizeof(int))) != hipSuccess) return e;
  for (int i = 0; i < 4; ++i) f.h_flags[i] = 0;
  return hipSuccess;
}

void ldlt_free(DenseLdlt &f) {
  if (f.K) (void)hipFree(f.K);
  if (f.W) (void)hipFree(f.W);
  if (f.dvec) (void)hipFree(f.dvec);
  if (f.dinv) (void)hipFree(f.dinv);
  if (f.zwork) (void)hipFree(f.zwork);
  if (f.Linv) (void)hipFree(f.Linv);
  if (f.LinvT) (void)hipFree(f.LinvT);
  if (f.flags) (void)hipFree(f.flags);
  if (f.chain) (void)hipFree(f.chain);
  if (f.xpub) (void)hipFree(f.xpub);
  if (f.hctl) (void)hipFree(f.hctl);
  if (f.V) (void)hipFree(f.V);
  if (f.vd) (void)hipFree(f.vd);
  if (f.h_flags) (void)hipHostFree(f.h_flags);
  f = DenseLdlt();
}

hipEvent_t prof_event(PgfProfile *p) {
  if (!p->pool.empty()) {
    hipEvent_t e = p->pool.back();
    p->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

// Launch one trailing-update region (see k_ldlt_update) on stream `s`, optionally
// bracketed by profiling events.
// any_order: launched with hipExtAnyOrderLaunch, i.e. without the barrier that makes a launch
// wait for the packets queued before it (it then runs beside the previous kernel of the
// stream; the NEXT ordinary launch still waits for both).  Ignored while profiling.
void launch_update(DenseLdlt &f, hipStream_t s, const double *Wp, int64_t ldw, int N,
                   int nrows, int row0, int col0, int colEnd, int kc0, int KB,
                   PgfProfile *p, int any_order) {
  if (row0 >= nrows || col0 >= colEnd) return;
  // 64 x 64 tiles (96 VGPRs, 5 wavefronts per SIMD) beat 128 x 128 tiles (249 VGPRs, 2 per
  // SIMD) at every region size measured (tools/bench_update.py): the kernel lives on
  // occupancy to hide its LDS / global latencies.
  const int bt = 64;
  const int tr = (nrows - row0 + bt - 1) / bt;
  const int tc = (colEnd - col0 + bt - 1) / bt;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (p) {
    e0 = prof_event(p);
    e1 = prof_event(p);
    (void)hipEventRecord(e0, s);
  }
  if (any_order && !p)
    hipExtLaunchKernelGGL((k_ldlt_update<64, 64, 16>), dim3(tc, tr), dim3(256), 0, s, nullptr,
                          nullptr, hipExtAnyOrderLaunch, f.K, f.ldk, Wp, ldw, N, nrows, row0, col0,
                          colEnd, kc0, KB);
  else
    hipLaunchKernelGGL((k_ldlt_update<64, 64, 16>), dim3(tc, tr), dim3(256), 0, s, f.K, f.ldk, Wp,
                       ldw, N, nrows, row0, col0, colEnd, kc0, KB);
  if (p) {
    (void)hipEventRecord(e1, s);
    p->update_spans.emplace_back(e0, e1);
    // algorithmic flops: entries (i, j) of the region with j <= i, 2*KB flops each
    double cnt = 0.0;
    const double lo = col0, hi = colEnd;  // columns [lo, hi)
    // rows i in [row0, nrows): columns j in [lo, min(hi, i + 1))
    for (int i = row0; i < nrows; i += 1) {
      const double top = (i + 1 < hi) ? (double)(i + 1) : hi;
      if (top > lo) cnt += top - lo;
    }
    p->update_flops.push_back(2.0 * cnt * KB);
    // algorithmic HBM bytes: every C entry read and written once, the W rows and the L rows of
    // the region once each
    p->update_bytes.push_back(16.0 * cnt + 8.0 * KB * ((double)(nrows - row0) + (double)(colEnd - col0)));
  }
}

// Two-level factorisation, the schedule of one call (one queue):
//   per outer block of f.OB = 256 columns:
//     4 x k_ldlt_panel_ll   64-column panels, left-looking inside the block: a panel's
//                           prologue applies the earlier panels of the block to the two tiles
//                           it needs, so nothing is launched between them
//     1 x k_ldlt_update     bulk right-looking update of everything to the right, K = 256
//   k_inv_diag_blocks       inverses of the 64 x 64 diagonal blocks for the solves
// This is the round-1 schedule, kept as the reference schedule (PGF_FACTOR=1) for the
// look-ahead schedule of pgf_factor2.hip, which is the default.
hipError_t ldlt_factor_async(DenseLdlt &f, int N, int nrows) {
  if (ldlt_use_lookahead()) return ldlt_factor2_async(f, N, nrows);
  f.N = N;
  f.factored = false;
  hipStream_t sA = f.stream;
  hipError_t e = hipMemsetAsync(f.flags, 0, 4 * sizeof(int), sA);
  if (e != hipSuccess) return e;
  PgfProfile *p = (f.prof && f.prof->enabled) ? f.prof : nullptr;
  if (p) {
    p->factor_spans.emplace_back(prof_event(p), prof_event(p));
    (void)hipEventRecord(p->factor_spans.back().first, sA);
  }
  const int OB = f.OB;
  int buf = 0;
  for (int ob0 = 0; ob0 < N; ob0 += OB, buf ^= 1) {
    const int obEnd = std::min(ob0 + OB, N);
    double *Wb = f.W + (size_t)buf * f.wstride;
    for (int c0 = ob0; c0 < obEnd; c0 += PGF_NB) {
      const int below = nrows - std::min(c0 + PGF_NB, N);
      const int npw = std::max(1, (below + 63) / 64);
      hipLaunchKernelGGL(k_ldlt_panel_ll<PGF_NB>, dim3(npw), dim3(256), 0, sA, f.K, f.ldk, Wb,
                         (int64_t)OB, ob0, N, nrows, c0, f.dvec, f.dinv, f.flags);
    }
    if (obEnd < N)  // bulk update of the whole trailing matrix, K-depth = the block width
      launch_update(f, sA, Wb, OB, N, nrows, obEnd, obEnd, N, ob0, obEnd - ob0, p, 0);
  }
  if (N > 0)
    hipLaunchKernelGGL(k_inv_diag_blocks, dim3((N + 63) / 64), dim3(64), 0, sA, f.K, f.ldk, N,
                       f.Linv, f.LinvT);
  if (p) (void)hipEventRecord(p->factor_spans.back().second, sA);
  e = hipMemcpyAsync(f.h_flags, f.flags, 4 * sizeof(int), hipMemcpyDeviceToHost, sA);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

int ldlt_finish(DenseLdlt &f, hipError_t *err) {
  hipError_t e = hipStreamSynchronize(f.stream);
  if (err) *err = e;
  if (e != hipSuccess) return -1;
  ldlt_chain_timing_dump();  // no-op unless PGF_CHAIN_TIMING is set
  if (f.h_flags[2]) {  // the diagonal chain's helpers failed a check: nothing of this factor is
    f.h_flags[2] = 0;  // to be trusted; they are off from now on and the caller factorises again
    ldlt_chain_helpers_off();
    f.factored = false;
    return 2;
  }
  f.n_neg = f.h_flags[1] + (f.vdepth > 0 ? f.vneg : 0);
  f.factored = (f.h_flags[0] == 0);
  return f.h_flags[0] ? 1 : 0;
}

// PGF_TRSV_CHAIN=0: one launch per 256-row super-block (the earlier scheme) instead of the
// chained single-launch solves; also switched off for the rest of the process when a chained
// solve reports a placement / timeout problem (ldlt_chain_check)
static bool g_chain_off = false;
static bool use_chain() {
  static const bool on = !(getenv("PGF_TRSV_CHAIN") && atoi(getenv("PGF_TRSV_CHAIN")) == 0);
  return on && !g_chain_off;
}

void ldlt_chain_set_enabled(bool on) { g_chain_off = !on; }
bool ldlt_chain_enabled() { return use_chain(); }

// after a host synchronisation of f.stream: did a chained solve since the last check fail its
// own checks?  (bit 0: a wait timed out, bit 1: workers on different XCDs)
int ldlt_chain_check(DenseLdlt &f) {
  if (!f.chain || !f.h_flags) return 0;  // banded handles have no dense factor
  const int bad = f.h_flags[3];
  if (!bad) return 0;
  f.h_flags[3] = 0;
  g_chain_off = true;
  (void)hipMemsetAsync(f.chain + 2 * f.chain_stride + 1, 0, sizeof(int), f.stream);
  // a solve that timed out may have left its half partly published
  (void)hipMemsetAsync(f.xpub, 0xff, 2 * (size_t)f.chain_stride * 64 * sizeof(double), f.stream);
  return bad;
}

// test hook (pgf_debug_fail_next_chain): make the chained solve just enqueued look like one that
// failed its checks -- status word set, solution overwritten with NaN
__global__ void k_chain_inject(double *__restrict__ sol, int N, int *__restrict__ ctl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) sol[i] = __builtin_nan("");
  if (i == 0) atomicOr(&ctl[1], 1);
}

static hipError_t chain_report(DenseLdlt &f, double *sol) {
  if (f.inject_chain_failure) {
    f.inject_chain_failure = 0;
    hipLaunchKernelGGL(k_chain_inject, dim3((f.N + 255) / 256), dim3(256), 0, f.stream, sol, f.N,
                       f.chain + 2 * f.chain_stride);
  }
  return hipMemcpyAsync(f.h_flags + 3, f.chain + 2 * f.chain_stride + 1, sizeof(int),
                        hipMemcpyDeviceToHost, f.stream);
}

hipError_t ldlt_backsolve_async(DenseLdlt &f, const double *w, double *sol) {
  const int N = f.N;
  hipStream_t s = f.stream;
  if (N == 0) return hipSuccess;
  if (use_chain()) {
    const int nblk = (N + 63) / 64;
    hipLaunchKernelGGL(k_trsv_bwd_chain, dim3(8 * nblk), dim3(256), 0, s, f.K, f.ldk, f.Linv, w, sol,
                       N, f.xpub, f.chain_stride, f.chain + 2 * f.chain_stride);
    return chain_report(f, sol);
  }
  hipLaunchKernelGGL(k_vec_copy_strided, dim3((N + 255) / 256), dim3(256), 0, s, f.zwork, w, N);
  constexpr int SUPER = 256;
  const int last = ((N - 1) / SUPER) * SUPER;
  for (int c0 = last; c0 >= 0; c0 -= SUPER) {
    const int g = c0 > 0 ? (c0 + 63) / 64 : 1;
    hipLaunchKernelGGL(k_trsv_bwd_super<SUPER>, dim3(g), dim3(256), 0, s, f.K, f.ldk, f.Linv,
                       f.zwork, sol, N, c0);
  }
  return hipGetLastError();
}

hipError_t ldlt_solve_async(DenseLdlt &f, const double *rhs, double *sol) {
  const int N = f.N;
  hipStream_t s = f.stream;
  if (N == 0) return hipSuccess;
  // forward: L y = rhs  (y lands in sol)
  if (use_chain()) {
    const int nblk = (N + 63) / 64;
    hipLaunchKernelGGL(k_trsv_fwd_chain, dim3(8 * nblk), dim3(256), 0, s, f.K, f.ldk, f.LinvT, rhs,
                       sol, N, f.xpub, f.chain_stride, f.chain + 2 * f.chain_stride);
  } else {
    hipLaunchKernelGGL(k_vec_copy_strided, dim3((N + 255) / 256), dim3(256), 0, s, f.zwork, rhs, N);
    for (int c0 = 0; c0 < N; c0 += 256) {
      const int below = N - (c0 + 256);
      const int g = below > 0 ? (below + 63) / 64 : 1;
      hipLaunchKernelGGL(k_trsv_fwd_super<256>, dim3(g), dim3(256), 0, s, f.K, f.ldk, f.LinvT,
                         f.zwork, sol, N, c0);
    }
  }
  // diagonal: y <- D^-1 y
  hipLaunchKernelGGL(k_vec_scale, dim3((N + 255) / 256), dim3(256), 0, s, sol, f.dinv, N);
  // backward: L^T s = y
  return ldlt_backsolve_async(f, sol, sol);
}

// ------------------------------------------------------------------ batched host schedule
static bool batch_chain_sched() {
  static const bool on = !(getenv("PGF_BATCH_CHAIN") && atoi(getenv("PGF_BATCH_CHAIN")) == 0);
  return on;
}
static int batch_fused_max() {
  static const int v = getenv("PGF_BATCH_FUSED_MAX") ? atoi(getenv("PGF_BATCH_FUSED_MAX")) : 64;
  return v;
}
bool ldlt_batch_fused_schedule(int B, int OB, bool profiling) {
  return OB == 256 && batch_chain_sched() && B <= batch_fused_max() && !profiling;
}
// the schedules that know the condensed order (vdepth > 0): the two chain schedules
bool ldlt_batch_condensed_schedule(int OB) { return OB == 256 && batch_chain_sched(); }

// vdepth > 0 (fused look-ahead schedule only): every instance's K is preceded by a pre-eliminated
// block with panel BInst::V (the condensed order, pgf_api.hip); m is then 0 for the factor kernels
void ldlt_batch_factor_async(hipStream_t s, const BInst *tab, int B, int Nmax, int m, int OB,
                             PgfProfile *p, int vdepth) {
  if (Nmax <= 0 || B <= 0) return;
  // PGF_BATCH_LL=0: the single-instance schedule with a batch dimension (fused panel
  // kernel + K = 64 inner updates); default: the left-looking split panel step
  const bool ll = !(getenv("PGF_BATCH_LL") && atoi(getenv("PGF_BATCH_LL")) == 0);
  // Default (PGF_BATCH_CHAIN=0 restores the split panel steps below): per outer block ONE chain
  // launch (the 256 x 256 diagonal block of every instance by one workgroup each,
  // pgf_factor2.hip) and ONE T(k) launch instead of four (diagonal tile, rows below) pairs.  A
  // small batch is bound by the number of dependent launches, not by throughput -- a rank of
  // an 8-GPU run of BASELINE config 4 holds 32 instances: 1.98 -> 1.71 ms per batched step;
  // 256 instances: 8.48 -> 8.17 ms.
  const bool chain_sched = batch_chain_sched();
  // small batches (PGF_BATCH_FUSED_MAX, default 64 instances): the single-instance look-ahead
  // too -- the next block's chains run beside the previous block's bulk update in one launch
  // (k_update_diag first), most CUs being idle during a chain launch of a few workgroups
  if (ldlt_batch_fused_schedule(B, OB, p != nullptr)) {
    // chain helpers (three workgroups per instance, resident together) up to 32 instances:
    // 8: 0.96 -> 0.83 ms, 16: 1.09 -> 0.98, 32: 1.56 -> 1.52; at 64 they cost more CUs than
    // they save chain time (2.52 -> 2.57)
    const bool helpers = B <= 32;
    int buf = 0;
    if (vdepth > 0) {
      // condensed order: the rank-m term's update of the first diagonal block, then the first
      // chains BESIDE its update of everything else (they used to run alone, the chip idle)
      ldlt_batch_launch_virtual_diag(s, tab, B, vdepth);
      ldlt_batch_launch_chain_update(s, tab, B, Nmax, m, 0, 0, helpers, vdepth);
    } else {
      ldlt_batch_launch_chain(s, tab, B, m, 0, helpers);
    }
    if (Nmax + 1 - std::min(OB, Nmax) > 0)
      ldlt_batch_launch_trsm(s, tab, B, (Nmax + 1 - std::min(OB, Nmax) + 15) / 16, m, 0, 0);
    for (int c0 = 0; c0 + OB < Nmax; c0 += OB, buf ^= 1) {
      const int c1 = c0 + OB, obEnd = std::min(c1 + OB, Nmax);
      ldlt_batch_launch_update_diag(s, tab, B, m, buf, c1);
      ldlt_batch_launch_chain_update(s, tab, B, Nmax, m, buf, c1, helpers);
      const int below = Nmax + 1 - obEnd;
      if (below > 0) ldlt_batch_launch_trsm(s, tab, B, (below + 15) / 16, m, buf ^ 1, c1);
    }
    return;
  }
  if (OB == 256 && chain_sched) {
    int buf = 0;
    if (vdepth > 0) {
      // condensed order: the rank-m term on the whole lower triangle first (the chains of a large
      // batch fill the chip: nothing to run it beside)
      const int tr = (Nmax + 1 + 63) / 64, tc = (Nmax + 63) / 64;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (p) {
        e0 = prof_event(p);
        e1 = prof_event(p);
        (void)hipEventRecord(e0, s);
      }
      hipLaunchKernelGGL(kb_ldlt_update_virtual, dim3(batch_grid(B, tc * tr)), dim3(256), 0, s, tab, B, tc, tr,
                         vdepth);
      if (p) {
        (void)hipEventRecord(e1, s);
        p->update_spans.emplace_back(e0, e1);
        p->update_flops.push_back(-(double)vdepth);  // (marks the virtual launch: pgf_batch_profile_read)
      }
    }
    for (int ob0 = 0; ob0 < Nmax; ob0 += OB, buf ^= 1) {
      const int obEnd = std::min(ob0 + OB, Nmax);
      ldlt_batch_launch_chain(s, tab, B, m, ob0, false);
      const int below = Nmax + 1 - obEnd;
      if (below > 0) ldlt_batch_launch_trsm(s, tab, B, (below + 15) / 16, m, buf, ob0);
      if (obEnd < Nmax) {
        const int tr = (Nmax + 1 - obEnd + 63) / 64, tc = (Nmax - obEnd + 63) / 64;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (p) {
          e0 = prof_event(p);
          e1 = prof_event(p);
          (void)hipEventRecord(e0, s);
        }
        hipLaunchKernelGGL(kb_ldlt_update, dim3(batch_grid(B, tc * tr)), dim3(256), 0, s, tab, B, tc,
                           tr, m, (int64_t)OB, buf, 0, obEnd, obEnd, 0x7fffffff, ob0, OB);
        if (p) {
          (void)hipEventRecord(e1, s);
          p->update_spans.emplace_back(e0, e1);
          p->update_flops.push_back((double)obEnd);
        }
      }
    }
    return;
  }
  int buf = 0;
  for (int ob0 = 0; ob0 < Nmax; ob0 += OB, buf ^= 1) {
    const int obEnd = std::min(ob0 + OB, Nmax);
    for (int c0 = ob0; c0 < obEnd; c0 += PGF_NB) {
      const int below = Nmax + 1 - std::min(c0 + PGF_NB, Nmax);
      const int npw = std::max(1, (below + 63) / 64);
      if (ll) {
        hipLaunchKernelGGL(kb_diag_ll, dim3(B), dim3(256), 0, s, tab, m, (int64_t)OB, buf, ob0, c0);
        hipLaunchKernelGGL(kb_trsm_ll, dim3(batch_grid(B, npw)), dim3(256), 0, s, tab, B, npw, m,
                           (int64_t)OB, buf, ob0, c0);
        continue;
      }
      hipLaunchKernelGGL(kb_ldlt_panel<PGF_NB>, dim3(batch_grid(B, npw)), dim3(256), 0, s, tab, B,
                         npw, m, (int64_t)OB, buf, ob0, c0);
      const int c1 = c0 + PGF_NB;
      if (c1 < obEnd) {
        const int tr = (Nmax + 1 - c1 + 63) / 64, tc = (obEnd - c1 + 63) / 64;
        hipLaunchKernelGGL(kb_ldlt_update, dim3(batch_grid(B, tc * tr)), dim3(256), 0, s, tab, B,
                           tc, tr, m, (int64_t)OB, buf, c0 - ob0, c1, c1, ob0 + OB, c0, PGF_NB);
      }
    }
    if (obEnd < Nmax) {
      const int tr = (Nmax + 1 - obEnd + 63) / 64, tc = (Nmax - obEnd + 63) / 64;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (p) {
        e0 = prof_event(p);
        e1 = prof_event(p);
        (void)hipEventRecord(e0, s);
      }
      hipLaunchKernelGGL(kb_ldlt_update, dim3(batch_grid(B, tc * tr)), dim3(256), 0, s, tab, B, tc,
                         tr, m, (int64_t)OB, buf, 0, obEnd, obEnd, 0x7fffffff, ob0, OB);
      if (p) {
        (void)hipEventRecord(e1, s);
        p->update_spans.emplace_back(e0, e1);
        p->update_flops.push_back((double)obEnd);  // region start; flops need the N_i (sync)
      }
    }
  }
  if (!ll)
    hipLaunchKernelGGL(kb_inv_diag_blocks, dim3((Nmax + 63) / 64, 1, B), dim3(64), 0, s, tab, m);
}

// Chained solves of a batch: the single-instance chain bodies, instance i on XCD i % 8 (ids
// 8 q + x: q walks the instances of residue class x one after the other, nwmax workers each, so
// that a worker only ever waits for workers with smaller ids).  Every block of L is read once
// -- the per-super-block kernels re-solve the 256-row super-block in every workgroup -- and
// the instances' chains overlap: 256 instances of N = 1280: 0.9 -> ~0.2 ms per batched step.
__global__ __launch_bounds__(256) void kb_trsv_bwd_chain(const BInst *__restrict__ tab, int B, int m,
                                                         int nwmax) {
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int inst = xcd + 8 * (q / nwmax), w = q % nwmax;
  if (inst >= B) return;
  const BInst &I = tab[inst];
  if (I.ctl[3]) return;
  const int N = I.counts[0] + m;
  if (w >= (N + 63) / 64) return;
  chain_bwd_body(I.K, I.ldk, I.Linv, I.zwork, I.sol, N, I.xpub, I.capblk, I.cctl, w);
}

__global__ __launch_bounds__(256, 3) void kb_trsv_fwd_chain(const BInst *__restrict__ tab, int B,
                                                            int m, int nwmax) {
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int inst = xcd + 8 * (q / nwmax), w = q % nwmax;
  if (inst >= B) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] != 0 || I.ctl[3]) return;
  const int N = I.counts[0] + m;
  if (w >= (N + 63) / 64) return;
  chain_fwd_body(I.K, I.ldk, I.LinvT, I.zwork, I.sol, N, I.xpub, I.capblk, I.cctl, w);
}

// cond_prep: the caller has filled zwork for the forward solves itself (condensed order)
void ldlt_batch_solve_async(hipStream_t s, const BInst *tab, int B, int Nmax, int m,
                            bool any_unfactored_solve, bool cond_prep) {
  if (Nmax <= 0 || B <= 0) return;
  const dim3 gv((Nmax + 255) / 256, 1, B);
  if (use_chain()) {
    const int nwmax = (Nmax + 63) / 64;
    const dim3 gc(8 * ((B + 7) / 8) * nwmax);
    if (any_unfactored_solve) {
      if (!cond_prep) hipLaunchKernelGGL(kb_solve_prep_fwd, gv, dim3(256), 0, s, tab, m);
      hipLaunchKernelGGL(kb_trsv_fwd_chain, gc, dim3(256), 0, s, tab, B, m, nwmax);
    }
    hipLaunchKernelGGL(kb_solve_prep_bwd, gv, dim3(256), 0, s, tab, m);
    hipLaunchKernelGGL(kb_trsv_bwd_chain, gc, dim3(256), 0, s, tab, B, m, nwmax);
    return;
  }
  if (any_unfactored_solve) {
    if (!cond_prep) hipLaunchKernelGGL(kb_solve_prep_fwd, gv, dim3(256), 0, s, tab, m);
    for (int c0 = 0; c0 < Nmax; c0 += 256) {
      const int below = Nmax - (c0 + 256);
      const int g = below > 0 ? (below + 63) / 64 : 1;
      hipLaunchKernelGGL(kb_trsv_fwd_super, dim3(g, 1, B), dim3(256), 0, s, tab, m, c0);
    }
  }
  hipLaunchKernelGGL(kb_solve_prep_bwd, gv, dim3(256), 0, s, tab, m);
  const int last = ((Nmax - 1) / 256) * 256;
  for (int c0 = last; c0 >= 0; c0 -= 256) {
    const int g = c0 > 0 ? (c0 + 63) / 64 : 1;
    hipLaunchKernelGGL(kb_trsv_bwd_super, dim3(g, 1, B), dim3(256), 0, s, tab, m, c0);
  }
}

// ------------------------------------------------------------------ micro-benchmark
__global__ void k_fill_pattern(double *p, size_t n, double scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = scale * (double)((i * 2654435761ull) % 1000003ull) / 1000003.0 - 0.5 * scale;
}

hipError_t ldlt_bench_update(int N, int KB, int variant, int reps, double *ms_out,
                             double *flops_out) {
  DenseLdlt f;
  hipStream_t s;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) return e;
  f.OB = KB;
  f.ldk = pick_ldk(N + KB);
  f.stream = s;
  const size_t rows = (size_t)N + KB + 1;
  if ((e = hipMalloc((void **)&f.K, rows * f.ldk * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.W, rows * KB * sizeof(double))) != hipSuccess) return e;
  hipLaunchKernelGGL(k_fill_pattern, dim3((rows * f.ldk + 255) / 256), dim3(256), 0, s, f.K,
                     rows * f.ldk, 1.0);
  hipLaunchKernelGGL(k_fill_pattern, dim3((rows * KB + 255) / 256), dim3(256), 0, s, f.W,
                     rows * (size_t)KB, 1e-3);
  // region: rows/cols [KB, KB + N), L panel in columns [0, KB)
  const int Nt = N + KB;
  auto launch = [&]() {
#define PGF_LAUNCH_VARIANT(BM_, BN_, BK_, WR_, WC_, DB_)                                     \
  hipLaunchKernelGGL((k_ldlt_update<BM_, BN_, BK_, WR_, WC_, DB_>),                             \
                     dim3((N + BN_ - 1) / BN_, (N + BM_ - 1) / BM_), dim3(64 * WR_ * WC_), 0, s, \
                     f.K, f.ldk, f.W, (int64_t)KB, Nt, Nt, KB, KB, Nt, 0, KBx)
    const int KBx = KB;
    switch (variant) {
      case 0: PGF_LAUNCH_VARIANT(64, 64, 16, 2, 2, 0); break;
      case 1: PGF_LAUNCH_VARIANT(64, 64, 16, 2, 2, 1); break;
      case 3: PGF_LAUNCH_VARIANT(128, 128, 16, 4, 4, 1); break;
      case 9: PGF_LAUNCH_VARIANT(64, 64, 64, 2, 2, 0); break;
      case 10: PGF_LAUNCH_VARIANT(64, 64, 32, 2, 2, 0); break;
      case 11: PGF_LAUNCH_VARIANT(128, 128, 32, 4, 4, 1); break;  // the fused update role's tile
      // (256 x 128 and 128 x 256 tiles per 16-wavefront workgroup measured the same as 128 x 128
      // at N >= 4864 and half of it at N = 2560)
#define PGF_LAUNCH_EXP(X_)                                                                     \
  hipLaunchKernelGGL((k_ldlt_update<128, 128, 32, 4, 4, 1, X_>), dim3((N + 127) / 128, (N + 127) / 128), \
                     dim3(1024), 0, s, f.K, f.ldk, f.W, (int64_t)KB, Nt, Nt, KB, KB, Nt, 0, KBx)
      case 21: PGF_LAUNCH_EXP(1); break;
      case 22: PGF_LAUNCH_EXP(2); break;
      case 23: PGF_LAUNCH_EXP(3); break;
      case 27: PGF_LAUNCH_EXP(7); break;
      case 28: PGF_LAUNCH_EXP(8); break;
      case 36: PGF_LAUNCH_EXP(16); break;
      case 43: PGF_LAUNCH_EXP(23); break;
      case 32: PGF_LAUNCH_EXP(32); break;
      case 75: PGF_LAUNCH_EXP(32 + 23); break;
#undef PGF_LAUNCH_EXP
      default: PGF_LAUNCH_VARIANT(64, 64, 16, 2, 2, 0); break;
    }
#undef PGF_LAUNCH_VARIANT
  };
  launch();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, s);
  e = hipStreamSynchronize(s);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / reps;
  *flops_out = (double)N * ((double)N + 1.0) * KB;
  if (variant == 32 || variant == 75) {
    long long c[4] = {0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_upd_clk), sizeof(c));
    const double us = (double)(c[3] - c[1]) * 0.01;
    fprintf(stderr, "one tile (middle row): %.1f us, %.0f shader cycles -> %.0f MHz\n", us,
            (double)(c[2] - c[0]), us > 0 ? (double)(c[2] - c[0]) / us : 0.0);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(f.K);
  (void)hipFree(f.W);
  (void)hipStreamDestroy(s);
  return e;
}
