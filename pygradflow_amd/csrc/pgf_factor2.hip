// Look-ahead schedule of the dense LDL^T factorisation of the reduced KKT matrix (the default;
// replaces SuperLU gstrf reached through scipy.sparse.linalg.splu at reference
// pygradflow/linear_solver/lu_solver.py:14).  Same arithmetic as pgf_ldlt.hip (unpivoted
// right-looking LDL^T in the natural order, inertia = number of negative pivots), different
// division of labour.  Per outer block k of 256 columns:
//
//   k_update_diag  (k-1 -> k)   the 256 x 256 DIAGONAL block of block k receives block k-1's
//                               update first (36 tiles of 32 x 32, one workgroup each)
//   k_chain_update              ONE launch with three kinds of workgroups:
//     workgroup 0     D(k)      the chain: 16 wavefronts factorise that diagonal block -- four
//                               64-column sub-panels; per sub-panel the 64 x 64 tile is
//                               eliminated by wavefront 0 (lane <-> row, v_readlane broadcasts),
//                               the block's rows below ride one 16-column step behind on
//                               wavefronts 1-3, MFMA updates on all 16.  This is the
//                               factorisation's critical chain of N sequential pivots -- and
//                               nothing else is.
//     workgroups 8, 16          its helpers on the same XCD (stamps in global memory): the part
//                               of the in-block update the chain does not need for its next
//                               sub-panel, and the inverses of the unit-lower diagonal tiles
//                               (the solves and k_trsm_block multiply with them)
//     all others                one 128 x 128 tile each of trailing-update work from the lazy
//                               plan (plan_updates): column blocks are kept complete only when
//                               the chain is about to need them, the rest on a budget that
//                               hides behind the chain; W = L D formed from L and D on the fly
//   k_trsm_block   T(k)         all rows below the block: X = T inv(L_kk)^T blocked by 64
//                               (MFMA, B operands straight from L2 into registers); writes
//                               W = X = L D (operand of k_update_diag) and L = X D^-1.
//
// The chain and the update tiles touch disjoint cache lines and hand nothing to each other, so
// the launch is correct whatever order its workgroups run in: look-ahead inside one queue,
// without a second stream (hipExtAnyOrderLaunch starts kernels early but does not run them
// side by side on this stack).  The round-1 schedule ran 80 panel launches of 22-33 us one
// after the other with the bulk updates between them (2.3 ms of 4.0 ms per step at N = 5120);
// here the serial part is D(k), ~86 us per 256 columns, and the update hides behind it.
// Batched mode (kb_*): the same chain / T(k) / update-diag device code with a batch dimension.
#include <hip/hip_ext.h>

#include "pgf_internal.h"
#include "pgf_ldlt_dev.h"
#include "pgf_chain3.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#define C_LD 66    // LDS row stride of 64-column tiles: conflict-free MFMA fragment reads
#define C_WLD 18
#define CH_ROWS 256
// M[256][66] | Wt[2][64][18] | D[64] | 1/D[64] | flag | 1/D of the previous sub-panel [64] |
// Lt[2][16][16]: the factored 16 x 16 tile of a step, transposed (chain_b_own's multipliers)
#define CH_LT_OFF (CH_ROWS * C_LD * 8 + 2 * 64 * C_WLD * 8 + 3 * 64 * 8 + 16)
#define CH_SMEM (CH_LT_OFF + 2 * 16 * 16 * 8)

// ------------------------------------------------------------------ TS x TS tile product
// One 16 x 16 MFMA tile per wavefront of a (TS / 16)^2-wavefront workgroup:
//   acc -= sum_{k < kd} A[i0 + i][k] * B[j0 + j][k]
// A and B are row-major in global memory (k contiguous, first column already applied to the
// pointers), staged through LDS in 64-deep chunks with the next chunk's loads in flight behind
// the current chunk's MFMAs.  v_mfma_f64_16x16x4_f64 operand layout: A: lane l holds
// A[l & 15][l >> 4], B: B[l >> 4][l & 15], C/D: row (l >> 4) + 4 reg, col l & 15.
template <int TS>
__device__ __forceinline__ void tile_msub(double4_t &acc, double (*As)[C_LD], double (*Bs)[C_LD],
                                          const double *A, int64_t lda, int i0, int ilim,
                                          const double *B, int64_t ldb, int j0, int jlim, int kd) {
  constexpr int WPR = TS / 16, NT = 64 * WPR * WPR, NQ = TS * 32 / NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WPR, wc = wave % WPR, l15 = lane & 15, l4 = lane >> 4;
  double2_t va[NQ], vb[NQ];
  auto fetch = [&](int kk) {  // chunk [kk, kk + 64) -> registers
    const int kc = min(64, kd - kk);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = q * NT + tid;
      const int row = p >> 5, c2 = (p & 31) * 2;
      double2_t a = (double2_t){0.0, 0.0}, b = (double2_t){0.0, 0.0};
      if (i0 + row < ilim) {
        const double *src = A + (int64_t)(i0 + row) * lda + kk + c2;
        if (c2 + 1 < kc) a = *reinterpret_cast<const double2_t *>(src);
        else if (c2 < kc) a.x = *src;
      }
      if (j0 + row < jlim) {
        const double *src = B + (int64_t)(j0 + row) * ldb + kk + c2;
        if (c2 + 1 < kc) b = *reinterpret_cast<const double2_t *>(src);
        else if (c2 < kc) b.x = *src;
      }
      va[q] = a;
      vb[q] = b;
    }
  };
  if (kd > 0) fetch(0);
  for (int kk = 0; kk < kd; kk += 64) {
    const int kc = min(64, kd - kk);
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = q * NT + tid;
      const int row = p >> 5, c2 = (p & 31) * 2;
      *reinterpret_cast<double2_t *>(&As[row][c2]) = -va[q];
      *reinterpret_cast<double2_t *>(&Bs[row][c2]) = vb[q];
    }
    __syncthreads();
    if (kk + 64 < kd) fetch(kk + 64);  // in flight during this chunk's MFMAs
    const int kr = (kc + 3) & ~3;
#pragma unroll 4
    for (int ks = 0; ks < kr; ks += 4) {
      const double a = As[16 * wr + l15][ks + l4];
      const double b = Bs[16 * wc + l15][ks + l4];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------ U_diag
// Diagonal block of the NEXT outer block (rows / columns [c1, c1 + nb1)) -= W L^T of the
// current one (K-depth kd, L in columns [kc0, kc0 + kd) of K, W block-relative).  One
// workgroup per TS x TS tile of the lower triangle; TS = 32: 36 workgroups of 4 wavefronts
// (a tile is 2 TS^2 kd flops on ONE CU's matrix pipes: 6.8 us at TS = 64, 1.7 us at 32).
template <int TS>
__device__ __forceinline__ void update_diag_tile(unsigned char *smem, int t, double *K, int64_t ldk,
                                                 const double *W, int64_t ldw, int kc0, int kd,
                                                 int c1, int nb1) {
  constexpr int WPR = TS / 16;
  double(*As)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem);
  double(*Bs)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem + TS * C_LD * 8);
  int I = 0;
  while (t > I) {
    t -= I + 1;
    ++I;
  }
  const int J = t;  // J <= I
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WPR, wc = wave % WPR, l15 = lane & 15, l4 = lane >> 4;
  const int lim = c1 + nb1;
  const int i0 = c1 + TS * I, j0 = c1 + TS * J;
  double4_t acc;
  const int j = j0 + 16 * wc + l15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 16 * wr + l4 + 4 * r;
    acc[r] = (i < lim && j < lim && j <= i) ? K[(int64_t)i * ldk + j] : 0.0;
  }
  tile_msub<TS>(acc, As, Bs, W, ldw, i0, lim, K + kc0, ldk, j0, lim, kd);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 16 * wr + l4 + 4 * r;
    if (i < lim && j < lim && j <= i) K[(int64_t)i * ldk + j] = acc[r];
  }
}

template <int TS>
__global__ __launch_bounds__(4 * TS * TS / 16) void k_update_diag(double *K, int64_t ldk,
                                                                  const double *W, int64_t ldw,
                                                                  int kc0, int kd, int c1, int nb1) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TS * C_LD * 8];
  update_diag_tile<TS>(smem, (int)blockIdx.x, K, ldk, W, ldw, kc0, kd, c1, nb1);
}

// batched: the 36 tiles of every instance's next diagonal block (block [c1 - 256, c1) applied)
__global__ __launch_bounds__(256) void kb_update_diag(const BInst *__restrict__ tab, int B, int m,
                                                      int wbuf, int c1) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 32 * C_LD * 8];
  int inst, t;
  if (!batch_decode(B, 36, inst, t)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m;
  if (c1 >= N) return;
  update_diag_tile<32>(smem, t, I.K, I.ldk, I.W + (int64_t)wbuf * I.wstride, 256, c1 - 256, 256, c1,
                       min(256, N - c1));
}

// ------------------------------------------------------------------ T(k)
// Rows below outer block [c0, c0 + nb): workgroup g owns the 16 rows r0 = c0 + nb + 16 g ...
// (row N, a carried right-hand side, included), wavefront wc the columns 16 wc ... of every
// 64-column sub-panel.  Per sub-panel s, left-looking:
//   T_s = K[rows, sub-panel s] - sum_{t < s} X_t L_st^T ;  X_s = T_s inv(L_ss)^T
//   W[rows, 64 s ..] = X_s (= L D) ;  K[rows, sub-panel s] = X_s D^-1 (= L)
// The workgroup's 16 x 256 strip starts in registers (one 16 x 16 accumulator per wavefront and
// sub-panel); the X_t it produces stay in LDS as the A operands of the later sub-panels.
// The B operands (L_10, inv_1, L_20, L_21, inv_2, ...: ten 64 x 64 tiles of the L2-resident
// diagonal block) are NOT staged through LDS: with 16 rows per workgroup every B element
// feeds exactly one wavefront, so each wavefront loads its own 16-row slice straight into MFMA
// operand registers, three tiles ahead of their use, and the six update steps need no barrier
// at all -- only the four solves exchange T_s / X_s through LDS.  (Staged through LDS with two
// barriers per tile the kernel took 20 us, 1.0-1.6 us per step against 0.5 us of MFMA work: one
// wavefront per SIMD has nothing to hide an LDS round trip behind.  Now 17-19 us: an update
// step takes 0.5-0.6 us when its B slice has arrived, but 300 workgroups pulling the same
// 320 KB each out of L2 -- ~100 MB per launch -- make the loads, not the MFMAs, the limit;
// in-kernel stamps: 3.5 us of cold strip loads, ~11 us of steps.)
// The k index of an MFMA step is free as long as A and B agree: step q of lane group l4 takes
// k = 8 (q / 2) + 2 l4 + (q & 1), so that both operands come as aligned 16-byte pairs.
// 16 rows per workgroup because the MFMA work of a workgroup runs on ONE CU's matrix pipes
// (0.3 TFLOP/s): 64 rows took 34 us.
// POST (k_trsm_ud): after the stores of sub-panel s have reached L2 the workgroup posts
// post[s] = postval -- the update of the next diagonal block runs in the same launch and takes
// the rows sub-panel by sub-panel.
// RT: 16-row tiles per workgroup.  1 where the launch is bound by the latency of one row group
// (single instance: ~250 row groups, one per CU); the batched launches -- thousands of row
// groups, three resident per CU, each pulling its own copy of the B slices through the CU's
// texture path -- take 2: every B register feeds two MFMAs (kb_trsm_block, 32 instances:
// 100 -> ... us).
template <bool POST = false, int RT = 1>
__device__ __forceinline__ void trsm_block_body(double (*Xs)[16 * RT][C_LD], double *ds, int wg, double *K,
                                                int64_t ldk, double *W, int64_t ldw, int nrows,
                                                int c0, int nb, const double *__restrict__ dinv,
                                                const double *__restrict__ Linv, int *post = nullptr,
                                                int postval = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bend = c0 + nb;
  const int r0 = bend + 16 * RT * wg;
  const int ns = (nb + 63) / 64;
  constexpr int DEPTH = 3;
  double2_t bq[DEPTH][8];
  auto fetch_b = [&](double2_t (&dst)[8], int idx) {
    // idx -> tile (s_, t_), s_ (s_ + 1) / 2 + t_ = idx; t_ < s_: L_st, t_ == s_: inv(L_ss)
    const int s_ = (idx >= 6) ? 3 : (idx >= 3) ? 2 : (idx >= 1) ? 1 : 0;
    const int t_ = idx - s_ * (s_ + 1) / 2;
    if (idx >= 10 || s_ >= ns) return;
    const int cb = c0 + 64 * s_;
    const int row = 16 * wc + l15;
    const bool ok = (t_ == s_) || row < min(64, bend - cb);
    const double *src = (t_ == s_) ? Linv + (size_t)(cb / 64) * 4096 + row * 64
                                   : K + (int64_t)(cb + row) * ldk + c0 + 64 * t_;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dst[j] = ok ? *reinterpret_cast<const double2_t *>(src + 8 * j + 2 * l4) : (double2_t){0.0, 0.0};
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch_b(bq[d], d);
  double4_t acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = c0 + 64 * s + 16 * wc + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = r0 + 16 * rt + l4 + 4 * r;
        acc[rt][s][r] = (i < nrows && j < bend) ? K[(int64_t)i * ldk + j] : 0.0;
      }
    }
  ds[tid] = (tid < nb) ? dinv[c0 + tid] : 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < ns) {
      const int cb = c0 + 64 * s;
      const int ncol = min(64, bend - cb);
#pragma unroll
      for (int t = 0; t < s; ++t) {
        const int idx = s * (s + 1) / 2 + t;  // compile-time after unrolling
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          double4_t a4 = acc[rt][s];
          double2_t av[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            av[j] = *reinterpret_cast<const double2_t *>(&Xs[t][16 * rt + l15][8 * j + 2 * l4]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[j].x, bq[idx % DEPTH][j].x, a4, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[j].y, bq[idx % DEPTH][j].y, a4, 0, 0, 0);
          }
          acc[rt][s] = a4;
        }
        fetch_b(bq[idx % DEPTH], idx + DEPTH);  // the slot just emptied
      }
      {
        // X[i][j] = sum_k T[i][k] inv[j][k]: T_s goes through LDS (every wavefront needs the
        // whole 64-column row), X_s replaces it there
        const int idx = s * (s + 1) / 2 + s;
        double(*St)[C_LD] = Xs[s];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) St[16 * rt + l4 + 4 * r][16 * wc + l15] = acc[rt][s][r];
        __syncthreads();
        double4_t x[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          x[rt] = (double4_t){0.0, 0.0, 0.0, 0.0};
          double2_t av[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            av[j] = *reinterpret_cast<const double2_t *>(&St[16 * rt + l15][8 * j + 2 * l4]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            x[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j].x, bq[idx % DEPTH][j].x, x[rt], 0, 0, 0);
            x[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j].y, bq[idx % DEPTH][j].y, x[rt], 0, 0, 0);
          }
        }
        fetch_b(bq[idx % DEPTH], idx + DEPTH);
        __syncthreads();  // all reads of T done: X replaces it
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) St[16 * rt + l4 + 4 * r][16 * wc + l15] = x[rt][r];
        __syncthreads();
        for (int p = tid; p < 16 * RT * 32; p += 256) {
          const int row = p >> 5, c2 = (p & 31) * 2;
          const int r = r0 + row;
          if (r >= nrows || c2 >= ncol) continue;
          const double2_t w = *reinterpret_cast<const double2_t *>(&St[row][c2]);
          double2_t l;
          l.x = w.x * ds[64 * s + c2];
          l.y = w.y * ds[64 * s + c2 + 1];
          double *wp = W + (int64_t)r * ldw + 64 * s + c2;
          double *kp = K + (int64_t)r * ldk + cb + c2;
          if (c2 + 1 < ncol) {
            *reinterpret_cast<double2_t *>(wp) = w;
            *reinterpret_cast<double2_t *>(kp) = l;
          } else {
            wp[0] = w.x;
            kp[0] = l.x;
          }
        }
        if (POST) {
          // every thread's stores are in L2 (the L1 is write-through) before the stamp is
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) post[s] = postval;
        }
      }
    } else if (POST) {
      if (tid == 0) post[s] = postval;  // no such sub-panel (ragged last block): nothing to wait for
    }
  }
}

__global__ __launch_bounds__(256) void k_trsm_block(double *K, int64_t ldk, double *W, int64_t ldw,
                                                    int nrows, int c0, int nb,
                                                    const double *__restrict__ dinv,
                                                    const double *__restrict__ Linv) {
  __shared__ __attribute__((aligned(16))) double Xs[4][16][C_LD];
  __shared__ double ds[256];
  trsm_block_body(Xs, ds, (int)blockIdx.x, K, ldk, W, ldw, nrows, c0, nb, dinv, Linv);
}

// batched: instance = tab[..] (its own N, known on the device), workgroup wg of `per` (32 rows each)
#define KB_TRSM_RT 2
__global__ __launch_bounds__(256) void kb_trsm_block(const BInst *__restrict__ tab, int B, int per,
                                                     int m, int wbuf, int c0) {
  __shared__ __attribute__((aligned(16))) double Xs[4][16 * KB_TRSM_RT][C_LD];
  __shared__ double ds[256];
  int inst, wg;
  if (!batch_decode(B, per, inst, wg)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m, nrows = N + 1;
  if (c0 >= N) return;
  const int nb = min(256, N - c0);
  if (c0 + nb + 16 * KB_TRSM_RT * wg >= nrows) return;
  trsm_block_body<false, KB_TRSM_RT>(Xs, ds, wg, I.K, I.ldk, I.W + (int64_t)wbuf * I.wstride, 256, nrows, c0,
                                     nb, I.dinv, I.Linv);
}

// ------------------------------------------------------------------ D(k)
#ifndef PGF_CHAIN_DPP
#define PGF_CHAIN_DPP 1  // 0: the v_readlane / LDS-broadcast elimination of round 2 (below)
#endif
#if PGF_CHAIN_DPP
// Round 3: the elimination with DPP row broadcasts.  gfx950 has 64-bit DPP operands for
// row_newbcast ("DP ALU DPP"): v_fmac_f64_dpp acc, src row_newbcast:k, mult adds (lane k's src of
// the own row of 16 lanes) * mult -- ONE instruction per entry of a rank-1 update instead of two
// v_readlane + FMA, issued every 4 cycles (tools/chain_dpp_test.hip: a 16 x 16 tile in 0.83 us
// against 1.45, with 64 rows riding along 1.06 against 1.65).  Every row of 16 lanes therefore
// holds the whole pivot tile (lane r <-> pivot row r, all four rows of lanes the same), and each
// lane one more row that rides along: its row of the 64 x 64 diagonal tile (chain_a_plus) or of
// the rows below (chain_b_own).
template <int L>
__device__ __forceinline__ double dpp_bcast(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xf, 0xf, false);
}
// acc += (lane L's src) * mult.  FRESH: two wait states first -- a DPP operand written by the
// instruction right before is read stale otherwise (inline assembly is invisible to the
// compiler's hazard recogniser).
template <int L, bool FRESH = false>
__device__ __forceinline__ void dpp_fmac(double &acc, double src, double mult) {
  if (FRESH)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mult), "n"(L));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mult), "n"(L));
}

// column C of chain_a_plus: a[] = the lane's pivot row (entries <= row valid), b[] = its riding
// row; r = 1 / a[C] of the own lane (meaningful in lane C), npc = -a[C] * a[C + 1][C].  The
// riding row's L entry and W entry of column C go to LDS at once (mrow / wrow: byte addresses of
// M[row][cb] and Wt[row][0]): the stores drain beside the arithmetic.
template <int C>
__device__ __forceinline__ void chain_a_col(double (&a)[16], double (&b)[16], double &r, double &npc,
                                            double &dmine, int prow, double *mrow, double *wrow) {
  constexpr int C1 = (C + 1) & 15, C2 = (C + 2) & 15;
  double rn = 0.0, npn = 0.0;
  if (C + 1 < 16) {
    // the pivot chain: next pivot's column first, its reciprocal in flight behind the rest
    dpp_fmac<C, true>(a[C1], r, npc);
    rn = __builtin_amdgcn_rcp(a[C1]);
  }
  const double rb = dpp_bcast<C>(r);
  const double l = -a[C] * rb;   // -L[pivot row][C] (junk on and above the diagonal: never used)
  const double mr = -b[C] * rb;  // -L[riding row][C]
  if (C + 2 < 16) npn = -a[C1] * dpp_bcast<C2>(a[C1]);
  mrow[C] = -mr;
  wrow[C] = b[C];
#define CH_UA(K) \
  if (K > C + 1) dpp_fmac<K>(a[K], a[C], l);
  CH_UA(2) CH_UA(3) CH_UA(4) CH_UA(5) CH_UA(6) CH_UA(7) CH_UA(8) CH_UA(9) CH_UA(10) CH_UA(11)
  CH_UA(12) CH_UA(13) CH_UA(14) CH_UA(15)
#undef CH_UA
#define CH_UB(K) \
  if (K > C) dpp_fmac<K>(b[K], a[C], mr);
  CH_UB(1) CH_UB(2) CH_UB(3) CH_UB(4) CH_UB(5) CH_UB(6) CH_UB(7) CH_UB(8) CH_UB(9) CH_UB(10)
  CH_UB(11) CH_UB(12) CH_UB(13) CH_UB(14) CH_UB(15)
#undef CH_UB
  dmine = (prow == C) ? a[C] : dmine;
  if (C + 1 < 16) {
    rn = fma(rn, fma(-a[C1], rn, 1.0), rn);
    r = rn;
    npc = npn;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// wavefront 0: right-looking elimination of the 16 columns of sub-block sb of the 64 x 64
// diagonal tile; emits L into M (rows of the tile at and below the sub-block), W = L D of the
// rows below the 16 x 16 pivot tile into Wt, D and 1 / D.  (Rows ABOVE the sub-block ride along
// with zeros, and the pivot rows ride along as copies of themselves: what they store above the
// diagonal of the tile is never read.)
__device__ __forceinline__ void chain_a_plus(double (*M)[C_LD], double (*Wt)[C_WLD], double *dD,
                                             double *dI, int &s_bad, int lane, int sb, int ncol,
                                             double (*Lt)[16]) {
  (void)Lt;
  const int cb = sb * 16, prow = lane & 15;
  double a[16], b[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t *>(&M[cb + prow][cb + k]);
    const double2_t u = *reinterpret_cast<const double2_t *>(&M[lane][cb + k]);
    a[k] = v.x;
    a[k + 1] = v.y;
    b[k] = u.x;
    b[k + 1] = u.y;
  }
  double r = __builtin_amdgcn_rcp(a[0]);
  r = fma(r, fma(-a[0], r, 1.0), r);
  double npc = -a[0] * dpp_bcast<1>(a[0]);
  double dmine = 1.0;
  double *mrow = &M[lane][cb], *wrow = &Wt[lane][0];
  chain_a_col<0>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<1>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<2>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<3>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<4>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<5>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<6>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<7>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<8>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<9>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<10>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<11>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<12>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<13>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<14>(a, b, r, npc, dmine, prow, mrow, wrow);
  chain_a_col<15>(a, b, r, npc, dmine, prow, mrow, wrow);
  const int tr = lane - cb;  // row inside the 16 x 16 pivot tile
  const bool piv = tr >= 0 && tr < 16;
  // classes flagged bad: sNaN, qNaN, -inf, -0, +0, +inf
  const bool isbad = __builtin_amdgcn_class(dmine, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200);
  const bool bad_any = __ballot(piv && isbad && lane < ncol) != 0ull;
  if (piv) {
    M[lane][lane] = dmine;  // (behind the riding copy's store of a 1 at this place)
    dD[lane] = dmine;
    dI[lane] = isbad ? 0.0 : fast_recip(dmine);
    if (tr == 0 && bad_any) s_bad = 1;
  }
}

// one lane per stack row below the diagonal tile: x L_bb^T = a_row for sub-block sbp; X (= L D)
// replaces the row's entries in place.  The lane holds pivot row (lane & 15) of W = L D of the
// factored tile (from M and D: chain_a_plus left L below the tile's diagonal), the multipliers
// come as DPP broadcasts: x[K] -= (x[C] / d_C) * W[K][C].
template <int C>
__device__ __forceinline__ void chain_b_col(double (&x)[16], const double (&w)[16], double di) {
  const double m = -x[C] * dpp_bcast<C>(di);
#define CH_UX(K) \
  if (K > C) dpp_fmac<K>(x[K], w[C], m);
  CH_UX(1) CH_UX(2) CH_UX(3) CH_UX(4) CH_UX(5) CH_UX(6) CH_UX(7) CH_UX(8) CH_UX(9) CH_UX(10)
  CH_UX(11) CH_UX(12) CH_UX(13) CH_UX(14) CH_UX(15)
#undef CH_UX
  __builtin_amdgcn_sched_barrier(0);  // eager (right-looking) order, see chain_a_plus
}
__device__ __forceinline__ void chain_b_own(double (*M)[C_LD], int row, int sbp, int lane,
                                            const double (*Lt)[16], const double *dD, const double *dI) {
  (void)Lt;
  const int cb = sbp * 16, prow = lane & 15;
  double x[16], w[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t *>(&M[row][cb + k]);
    const double2_t l = *reinterpret_cast<const double2_t *>(&M[cb + prow][cb + k]);
    const double2_t d = *reinterpret_cast<const double2_t *>(&dD[cb + k]);
    x[k] = v.x;
    x[k + 1] = v.y;
    w[k] = l.x * d.x;  // (entries on and above the diagonal: D itself or junk, never broadcast)
    w[k + 1] = l.y * d.y;
  }
  const double di = dI[cb + prow];
  chain_b_col<0>(x, w, di);
  chain_b_col<1>(x, w, di);
  chain_b_col<2>(x, w, di);
  chain_b_col<3>(x, w, di);
  chain_b_col<4>(x, w, di);
  chain_b_col<5>(x, w, di);
  chain_b_col<6>(x, w, di);
  chain_b_col<7>(x, w, di);
  chain_b_col<8>(x, w, di);
  chain_b_col<9>(x, w, di);
  chain_b_col<10>(x, w, di);
  chain_b_col<11>(x, w, di);
  chain_b_col<12>(x, w, di);
  chain_b_col<13>(x, w, di);
  chain_b_col<14>(x, w, di);
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    double2_t wv;
    wv.x = x[k];
    wv.y = x[k + 1];
    *reinterpret_cast<double2_t *>(&M[row][cb + k]) = wv;
  }
}
#else
// wavefront 0: right-looking elimination of the 16 columns of sub-block sb of the 64 x 64
// diagonal tile, lane <-> row (the scheme of panel_body's (a+), pgf_ldlt.hip): the tile's rows
// and, in the same instruction stream, the tile rows below it; emits L into M, W = L D of the
// rows below the 16 x 16 tile into Wt, D and 1/D.
__device__ __forceinline__ void chain_a_plus(double (*M)[C_LD], double (*Wt)[C_WLD], double *dD,
                                             double *dI, int &s_bad, int lane, int sb, int ncol,
                                             double (*Lt)[16]) {
  const int cb = sb * 16;
  double a[16], w[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t *>(&M[lane][cb + k]);
    a[k] = v.x;
    a[k + 1] = v.y;
  }
  // The serial chain of the whole factorisation runs through this loop: pivot -> reciprocal ->
  // multiplier column -> the ONE entry the next pivot needs -> next pivot.  Written software-
  // pipelined, with a scheduling barrier per column: left alone, the compiler's list scheduler
  // turns the right-looking updates into a lazy (left-looking) order in which column j waits
  // for a chain of j dependent FMAs right before its pivot -- 3.5 us per 16 columns instead
  // of about one.  Per column: the next pivot's entry is updated first and its reciprocal
  // chain started, the other 14 - j updates (independent FMAs, two v_readlane each) fill in.
  // classes flagged bad: sNaN, qNaN, -inf, -0, +0, +inf
  // Dependent fp64 operations cost ~30 cycles each on a lone wavefront, so the chain carries
  // as few as possible: reciprocal seed + ONE Newton step (v_rcp_f64 delivers > 26 bits; the
  // pivots only enter through products and the 1e-10 bar leaves five digits), and the product
  // of the next pivot's two factors is formed while the reciprocal is still in flight:
  //   a[j+1] -= (w[j] * c) * (1 / d)   instead of   a[j+1] -= (w[j] / d) * c.
  double d = lane_bcast(a[0], cb);
  bool bad_any = __builtin_amdgcn_class(d, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200) && cb < ncol;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    w[j] = a[j];
    double r = __builtin_amdgcn_rcp(d);
    double pc = 0.0;
    if (j + 1 < 16) pc = a[j] * lane_bcast(a[j], cb + j + 1);  // beside the reciprocal
    r = fma(r, fma(-d, r, 1.0), r);
    if (j + 1 < 16) {
      a[j + 1] = fma(-pc, r, a[j + 1]);
      d = lane_bcast(a[j + 1], cb + j + 1);
      bad_any |= __builtin_amdgcn_class(d, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200) &&
                 (cb + j + 1) < ncol;
    }
    const double l = a[j] * r;
#pragma unroll
    for (int k = j + 2; k < 16; ++k) a[k] = fma(-l, lane_bcast(w[j], cb + k), a[k]);
    a[j] = l;
    __builtin_amdgcn_sched_barrier(0);
  }
  const int tr = lane - cb;  // row inside the 16 x 16 tile
  if (tr >= 0 && tr < 16) {
    double d_mine = 1.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < tr) {
        M[lane][cb + k] = a[k];
        Lt[k][tr] = a[k];  // transposed copy: row t = the multipliers of column t (chain_b_own)
      }
      if (k == tr) d_mine = w[k];
    }
    const bool ok = !__builtin_amdgcn_class(d_mine, 0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200);
    M[lane][lane] = d_mine;
    dD[lane] = d_mine;
    dI[lane] = ok ? fast_recip(d_mine) : 0.0;
    if (tr == 0 && bad_any) s_bad = 1;
  } else if (tr >= 16) {
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
}

// one lane per stack row below the diagonal tile: x L_bb^T = a_row for sub-block sbp; X (= L D)
// replaces the row's entries in place.  The multipliers L_bb[j][t], j > t, come as 16-byte
// broadcast reads from the TRANSPOSED copy of the factored tile that chain_a_plus leaves in LDS
// (row t of Lt = column t of L_bb, contiguous in j): 64 reads + 120 FMAs.  With v_readlane
// broadcasts out of a register copy of the tile (two per FMA, 360 instructions) this function,
// not the pivot recurrence, bounded an elimination step: 2.05 us against 1.65 us for
// chain_a_plus, each with a SIMD to itself (tools/chain_step_test.hip).
// The reads are inline assembly with immediate offsets off ONE base register: as C++ loads the
// compiler hoists the reads of all 15 columns to the top of the function and spills them -- a
// lane has 128 registers in the 16-wavefront workgroup and the chain kernel uses 126 of them.
template <int T, int P>
__device__ __forceinline__ void chain_b_read(double2_t &m, unsigned base) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m) : "v"(base), "n"((T * 16 + 2 * P) * 8));
}
template <int T>
__device__ __forceinline__ void chain_b_col(double (&x)[16], unsigned base) {
  // pairs (2 p, 2 p + 1) with an entry beyond column T: p >= (T + 1) / 2
  constexpr int P0 = (T + 1) >> 1;
  double2_t m[8];
  if (0 >= P0) chain_b_read<T, 0>(m[0], base);
  if (1 >= P0) chain_b_read<T, 1>(m[1], base);
  if (2 >= P0) chain_b_read<T, 2>(m[2], base);
  if (3 >= P0) chain_b_read<T, 3>(m[3], base);
  if (4 >= P0) chain_b_read<T, 4>(m[4], base);
  if (5 >= P0) chain_b_read<T, 5>(m[5], base);
  if (6 >= P0) chain_b_read<T, 6>(m[6], base);
  chain_b_read<T, 7>(m[7], base);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const double xt = x[T];
#pragma unroll
  for (int p = P0; p < 8; ++p) {
    if (2 * p > T) x[2 * p] = fma(-xt, m[p].x, x[2 * p]);
    x[2 * p + 1] = fma(-xt, m[p].y, x[2 * p + 1]);
  }
  __builtin_amdgcn_sched_barrier(0);  // eager (right-looking) order, see chain_a_plus
}
__device__ __forceinline__ void chain_b_own(double (*M)[C_LD], int row, int sbp, int lane,
                                            const double (*Lt)[16], const double *, const double *) {
  const int cb = sbp * 16;
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const double2_t v = *reinterpret_cast<const double2_t *>(&M[row][cb + k]);
    x[k] = v.x;
    x[k + 1] = v.y;
  }
  const unsigned base = (unsigned)(uintptr_t)&Lt[0][0];  // LDS byte address
  chain_b_col<0>(x, base);
  chain_b_col<1>(x, base);
  chain_b_col<2>(x, base);
  chain_b_col<3>(x, base);
  chain_b_col<4>(x, base);
  chain_b_col<5>(x, base);
  chain_b_col<6>(x, base);
  chain_b_col<7>(x, base);
  chain_b_col<8>(x, base);
  chain_b_col<9>(x, base);
  chain_b_col<10>(x, base);
  chain_b_col<11>(x, base);
  chain_b_col<12>(x, base);
  chain_b_col<13>(x, base);
  chain_b_col<14>(x, base);
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    double2_t wv;
    wv.x = x[k];
    wv.y = x[k + 1];
    *reinterpret_cast<double2_t *>(&M[row][cb + k]) = wv;
  }
}
#endif  // PGF_CHAIN_DPP

// C (16 x 16 at M[ci][cj]) -= A B^T over 16 k: A rows at (ar, ak) of Am (stride lda doubles),
// B rows at M[br][bk]
template <int LDA>
__device__ __forceinline__ void chain_tile16(double (*M)[C_LD], int ci, int cj,
                                             const double (*Am)[LDA], int ar, int ak, int br,
                                             int bk, int l15, int l4) {
  double4_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = M[ci + l4 + 4 * r][cj + l15];
#pragma unroll
  for (int ks = 0; ks < 16; ks += 4) {
    const double av = -Am[ar + l15][ak + ks + l4];
    const double bv = M[br + l15][bk + ks + l4];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) M[ci + l4 + 4 * r][cj + l15] = acc[r];
}

// block column Q of the inverse of a unit-lower 64 x 64 tile whose diagonal 16 x 16 sub-tiles
// already hold their inverses (see the end of k_diag_chain); one wavefront.  Xo[p - Q - 1] =
// block (p, Q), p > Q, in MFMA C layout.
template <int Q>
__device__ __forceinline__ void inv_block_column(const double (*Lg)[C_LD], double4_t (&Xo)[3],
                                                 int l15, int l4) {
  double4_t X[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) X[Q][rr] = Lg[16 * Q + 4 * rr + l4][16 * Q + l15];
#pragma unroll
  for (int p = Q + 1; p < 4; ++p) {
    double4_t S = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = Q; r < p; ++r)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        S = __builtin_amdgcn_mfma_f64_16x16x4f64(Lg[16 * p + l15][16 * r + 4 * rr + l4], X[r][rr], S,
                                                 0, 0, 0);
    double4_t Xp = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      Xp = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lg[16 * p + l15][16 * p + 4 * rr + l4], S[rr], Xp, 0,
                                                0, 0);
    X[p] = Xp;
    Xo[p - Q - 1] = Xp;
  }
}

// ---- helper workgroups of the chain (same launch, same XCD: workgroup ids 0, 8 and 16)
// The chain workgroup hands each finished 64-column sub-panel to two helpers through stamps in
// global memory: helper T applies the sub-panel to the part of the diagonal block the chain
// does not need for its NEXT sub-panel, helper I inverts the sub-panel's unit-lower tile.  Both
// used to sit at the end of the chain's own critical path.  Protocol as for the chained solves
// (pgf_ldlt.hip): producer drains its stores (s_waitcnt vmcnt(0)) behind a barrier, then ONE
// lane stores the epoch stamp with an L1-bypassing access; the consumer polls it (bounded) and
// reads the data through its own, freshly invalidated L1 or with L1-bypassing loads; producer
// and consumer share an L2 because ids that are multiples of 8 land on one XCD -- checked at run
// time through HW_REG_XCC_ID.  A failed check or a timed-out wait sets flags[2]; the host then
// repeats the factorisation without helpers.
#define HC_STAMP 0  // [0, 4): sub-panel s written back (chain -> helpers)
#define HC_DONE 4   // [4, 8): deferred tiles of sub-panel s updated (helper T -> chain)
#define HC_XCC 8    // max over the three roles of (epoch << 4 | xcc)
#define HC_WORDS 16
#define HELP_SPIN_LIMIT (1 << 18)

__device__ __forceinline__ void help_wait(const int *stamp, int epoch, int *flags) {
  for (int it = 0; it < HELP_SPIN_LIMIT; ++it) {
    if (__hip_atomic_load(stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return;
    __builtin_amdgcn_s_sleep(1);
  }
  atomicOr(&flags[2], 2);
}
__device__ __forceinline__ void help_post(int *stamp, int epoch) {
  __hip_atomic_store(stamp, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void help_check_xcc(int *hc, int epoch, int *flags) {
  const int ep = epoch & 0x7ffffff;
  const int mine = (ep << 4) | (int)(__builtin_amdgcn_s_getreg(6164) & 15);  // XCC_ID[3:0]
  const int old = atomicMax(&hc[HC_XCC], mine);
  if ((old >> 4) == ep && old != mine) atomicOr(&flags[2], 1);
}
__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ T(k) + U_diag in ONE launch
// k_update_diag (k -> k + 1) only needs the rows of the NEXT diagonal block from T(k): the (at
// most) 16 row groups right below block k.  As its own launch it cost 7.9 us per outer block on
// the factorisation's critical path.  Here those row groups and the 36 diagonal tiles are
// workgroups of one launch on ONE XCD (ids that are multiples of 8, checked through
// HW_REG_XCC_ID like the chain's helpers): a row group posts a stamp per 64-column sub-panel once
// the sub-panel's W / L entries are in that XCD's L2, a diagonal tile takes its K-depth-256
// product sub-panel by sub-panel behind the stamps of the four row groups it reads (operands
// with L1-bypassing loads) and is done about 2 us after the last one -- inside the time the
// other ~290 row groups of T(k) take anyway.  Workgroups only wait for workgroups with smaller
// ids; polls are bounded; a failed placement check or a timed-out wait sets flags[2] and the
// host repeats the factorisation with separate launches (as for the chain's helpers).
#define TUD_BASE 32  // hctl words [32, 96): stamp of (row group g < 16, sub-panel s < 4) at 4 g + s

// wait until the four row groups (g0, g0 + 1, g1, g1 + 1) have posted sub-panel s
__device__ __forceinline__ void tud_wait(const int *st, int g0, int g1, int s, int val, int ngroups,
                                         int *flags) {
  const int lane = threadIdx.x & 63;
  int g = (lane < 2) ? g0 + lane : g1 + (lane - 2);
  const bool mine = lane < 4 && g < ngroups;
  for (int it = 0; it < HELP_SPIN_LIMIT; ++it) {
    int v = val;
    if (mine) v = __hip_atomic_load(st + 4 * g + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool same_epoch = (v >> 4) == (val >> 4);
    if (!__builtin_amdgcn_ballot_w64(!same_epoch)) {
      if (__builtin_amdgcn_ballot_w64(v != val)) atomicOr(&flags[2], 1);  // another XCD
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  atomicOr(&flags[2], 2);
}

// update_diag_tile<32> behind the stamps: chunk kk = sub-panel kk / 64
__device__ __forceinline__ void update_diag_tile_live(unsigned char *smem, int t, double *K, int64_t ldk,
                                                      const double *W, int64_t ldw, int kc0, int kd,
                                                      int c1, int nb1, const int *st, int val,
                                                      int ngroups, int *flags) {
  constexpr int TS = 32, WPR = 2, NT = 256, NQ = 4;
  double(*As)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem);
  double(*Bs)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem + TS * C_LD * 8);
  int I = 0;
  while (t > I) {
    t -= I + 1;
    ++I;
  }
  const int J = t;  // J <= I
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WPR, wc = wave % WPR, l15 = lane & 15, l4 = lane >> 4;
  const int lim = c1 + nb1;
  const int i0 = c1 + TS * I, j0 = c1 + TS * J;
  double4_t acc;
  const int j = j0 + 16 * wc + l15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 16 * wr + l4 + 4 * r;
    acc[r] = (i < lim && j < lim && j <= i) ? K[(int64_t)i * ldk + j] : 0.0;
  }
  const double *B = K + kc0;
  for (int kk = 0; kk < kd; kk += 64) {
    const int kc = min(64, kd - kk);
    tud_wait(st, 2 * I, 2 * J, kk >> 6, val, ngroups, flags);
    double2_t va[NQ], vb[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = q * NT + tid;
      const int row = p >> 5, c2 = (p & 31) * 2;
      double2_t a = (double2_t){0.0, 0.0}, b = (double2_t){0.0, 0.0};
      if (i0 + row < lim) {
        const double *src = W + (int64_t)(i0 + row) * ldw + kk + c2;
        if (c2 < kc) a.x = ld_agent(src);
        if (c2 + 1 < kc) a.y = ld_agent(src + 1);
      }
      if (j0 + row < lim) {
        const double *src = B + (int64_t)(j0 + row) * ldk + kk + c2;
        if (c2 < kc) b.x = ld_agent(src);
        if (c2 + 1 < kc) b.y = ld_agent(src + 1);
      }
      va[q] = a;
      vb[q] = b;
    }
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = q * NT + tid;
      const int row = p >> 5, c2 = (p & 31) * 2;
      *reinterpret_cast<double2_t *>(&As[row][c2]) = -va[q];
      *reinterpret_cast<double2_t *>(&Bs[row][c2]) = vb[q];
    }
    __syncthreads();
    const int kr = (kc + 3) & ~3;
#pragma unroll 4
    for (int ks = 0; ks < kr; ks += 4) {
      const double a = As[16 * wr + l15][ks + l4];
      const double b = Bs[16 * wc + l15][ks + l4];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 16 * wr + l4 + 4 * r;
    if (i < lim && j < lim && j <= i) K[(int64_t)i * ldk + j] = acc[r];
  }
}

// grid: tud_grid() workgroups.  nA = row groups of the next diagonal block, nU = its tiles,
// S = nA + nU special workgroups at the ids 8 q; every other id takes one of the other row groups.
__global__ __launch_bounds__(256) void k_trsm_ud(double *K, int64_t ldk, double *W, int64_t ldw,
                                                 int nrows, int c0, int nb,
                                                 const double *__restrict__ dinv,
                                                 const double *__restrict__ Linv, int N, int *hctl,
                                                 int epoch, int *flags) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 16 * C_LD * 8 + 256 * 8];
  double(*Xs)[16][C_LD] = reinterpret_cast<double(*)[16][C_LD]>(smem);
  double *ds = reinterpret_cast<double *>(smem + 4 * 16 * C_LD * 8);
  const int c1 = c0 + nb, nb1 = min(256, N - c1);
  const int nT = (nrows - c1 + 15) / 16;
  const int nA = min(nT, (nb1 + 15) / 16);
  const int nt = (nb1 + 31) / 32, nU = nt * (nt + 1) / 2;
  const int S = nA + nU;
  const int id = (int)blockIdx.x;
  int *st = hctl + TUD_BASE;
  const int val = ((epoch & 0x7ffffff) << 4) | (int)(__builtin_amdgcn_s_getreg(6164) & 15);
  if (id < 8 * S && (id & 7) == 0) {
    const int q = id >> 3;
    if (q < nA) {
      trsm_block_body<true>(Xs, ds, q, K, ldk, W, ldw, nrows, c0, nb, dinv, Linv, st + 4 * q, val);
    } else {
      update_diag_tile_live(smem, q - nA, K, ldk, W, ldw, c0, nb, c1, nb1, st, val, nA, flags);
    }
    return;
  }
  const int r = id < 8 * S ? id - (id >> 3) - 1 : id - S;
  if (nA + r >= nT) return;
  trsm_block_body<false>(Xs, ds, nA + r, K, ldk, W, ldw, nrows, c0, nb, dinv, Linv);
}

// inverse of the unit-lower 64 x 64 diagonal tile g of the block (rows / columns from b0), by
// the four wavefronts [4 slot, 4 slot + 4) of the workgroup; every thread of the workgroup
// calls this (barriers), `live` says whether its wavefront group has a tile.  Blocked by 16:
// wavefront v inverts the 16 x 16 diagonal sub-tile v by substitution (lane c <-> column c),
// then wavefront q < 3 builds block column q of the inverse top down,
//   X_pq = -D_p sum_{r = q}^{p-1} L_pr X_rq   (D_p = inv(L_pp), X_qq = D_q),
// with MFMA: a 16 x 16 accumulator (row (l >> 4) + 4 reg, column l & 15) IS the B operand
// of the next four k-steps, so the X_rq stay in registers.  Stored as inv and as its
// transpose, [tile][row][64]: forward and backward solves both read coalesced rows.
__device__ __forceinline__ void invert_tile(unsigned char *smem, const double *K, int64_t ldk,
                                            int b0, int nbw, bool live, double *__restrict__ Linv,
                                            double *__restrict__ LinvT) {
  // wave: uniform per wavefront -> scalar register, role tests and tile numbers on the SALU
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4, v = wave & 3;
  double(*Lg)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem + (size_t)(wave >> 2) * 64 * C_LD * 8);
  if (live) {
    for (int idx = tid & 255; idx < 64 * 32; idx += 256) {
      const int row = idx >> 5, c2 = (idx & 31) * 2;
      double2_t t = (double2_t){0.0, 0.0};
      if (row < nbw) {
        const double *src = K + (int64_t)(b0 + row) * ldk + b0 + c2;
        if (c2 + 1 < row) t = *reinterpret_cast<const double2_t *>(src);
        else if (c2 < row) t.x = *src;
      }
      if (c2 == row) t.x = 1.0;
      if (c2 + 1 == row) t.y = 1.0;
      *reinterpret_cast<double2_t *>(&Lg[row][c2]) = t;
    }
  }
  __syncthreads();
  if (live && lane < 16) {
    double y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) y[j] = (j == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < 15; ++t) {
      const double yt = y[t];
#pragma unroll
      for (int j = t + 1; j < 16; ++j) y[j] = fma(-yt, Lg[16 * v + j][16 * v + t], y[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) Lg[16 * v + j][16 * v + lane] = y[j];
  }
  __syncthreads();
  double4_t Xo[3];
  if (live) {
    if (v == 0) inv_block_column<0>(Lg, Xo, l15, l4);
    else if (v == 1) inv_block_column<1>(Lg, Xo, l15, l4);
    else if (v == 2) inv_block_column<2>(Lg, Xo, l15, l4);
  }
  __syncthreads();  // every wavefront is done reading the L blocks: the X blocks go in place
  if (live && v < 3) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int pb = v + 1 + t;
      if (pb < 4) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Lg[16 * pb + l4 + 4 * rr][16 * v + l15] = Xo[t][rr];
      }
    }
  }
  __syncthreads();
  if (live) {
    double *o = Linv + (size_t)(b0 / 64) * 4096;
    double *ot = LinvT + (size_t)(b0 / 64) * 4096;
    for (int idx = tid & 255; idx < 64 * 64; idx += 256) {
      const int row = idx >> 6, col = idx & 63;
      o[idx] = Lg[row][col];
      ot[idx] = Lg[col][row];
    }
  }
  __syncthreads();  // Lg is refilled by the next pass
}

// 16 x 16 sub-tiles on or below the diagonal of a lower-triangular region of nt 64-row tiles,
// tile (I, J), J <= I, holding 10 (I == J) or 16 of them; (mi, mj) = offsets inside the region
__device__ __forceinline__ void decode_subtile(int e, int &mi, int &mj) {
  int I = 0, J = 0;
  while (true) {
    const int cnt = (I == J) ? 10 : 16;
    if (e < cnt) break;
    e -= cnt;
    if (++J > I) {
      J = 0;
      ++I;
    }
  }
  int ti, tj;
  if (I == J) {
    ti = (e >= 6) ? 3 : (e >= 3) ? 2 : (e >= 1) ? 1 : 0;
    tj = e - ti * (ti + 1) / 2;
  } else {
    ti = e >> 2;
    tj = e & 3;
  }
  mi = 64 * I + 16 * ti;
  mj = 64 * J + 16 * tj;
}

// helper T (workgroup 8): for every sub-panel s with rows beyond the NEXT sub-panel, apply it to
// the lower triangle of those rows / columns [cb + 128, bend): C -= (L D) L^T with L read back
// from global memory (the chain wrote L = X D^-1; X itself stays in its LDS)
template <int NW>
__device__ __forceinline__ void helper_tiles(unsigned char *smem, double *K, int64_t ldk, int c0,
                                             int nb, const double *dvec, int *hc, int epoch,
                                             int *flags) {
  constexpr int NT = 64 * NW;
  double(*Mh)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem);
  double *dDs = reinterpret_cast<double *>(smem + 128 * C_LD * 8);
  // wave: uniform per wavefront -> scalar register, role tests and tile numbers on the SALU
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bend = c0 + nb, ns = (nb + 63) / 64;
  if (tid == 0) help_check_xcc(hc, epoch, flags);
  for (int s = 0; s + 2 < ns; ++s) {
    const int cb = c0 + 64 * s, r0 = cb + 128;
    const int rows = bend - r0, rowsp = (rows + 63) & ~63, nt = rowsp / 64;
    if (tid == 0) help_wait(hc + HC_STAMP + s, epoch, flags);
    __syncthreads();
    for (int p = tid; p < rowsp * 32; p += NT) {
      const int row = p >> 5, c2 = (p & 31) * 2;
      double2_t t = (double2_t){0.0, 0.0};
      if (row < rows) t = *reinterpret_cast<const double2_t *>(K + (int64_t)(r0 + row) * ldk + cb + c2);
      *reinterpret_cast<double2_t *>(&Mh[row][c2]) = t;
    }
    if (tid < 64) dDs[tid] = dvec[cb + tid];
    __syncthreads();
    const int total = nt * (nt + 1) / 2 * 16 - nt * 6;
    for (int e = wave; e < total; e += NW) {
      int mi, mj;
      decode_subtile(e, mi, mj);
      const int gi = r0 + mi, gj = r0 + mj, j = gj + l15;
      double4_t c = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = gi + l4 + 4 * r;
        if (i < bend && j < bend && j <= i) c[r] = ld_agent(K + (int64_t)i * ldk + j);
      }
#pragma unroll 4
      for (int ks = 0; ks < 64; ks += 4) {
        const double av = -Mh[mi + l15][ks + l4] * dDs[ks + l4];
        const double bv = Mh[mj + l15][ks + l4];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = gi + l4 + 4 * r;
        if (i < bend && j < bend && j <= i) K[(int64_t)i * ldk + j] = c[r];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) help_post(hc + HC_DONE + s, epoch);
  }
}

// helper I (workgroup 16): the inverse of every sub-panel's diagonal tile as soon as it is final
template <int NW>
__device__ __forceinline__ void helper_inverses(unsigned char *smem, const double *K, int64_t ldk,
                                                int c0, int nb, int *hc, int epoch, int *flags,
                                                double *__restrict__ Linv,
                                                double *__restrict__ LinvT) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const int bend = c0 + nb, ns = (nb + 63) / 64;
  if (tid == 0) help_check_xcc(hc, epoch, flags);
  for (int s = 0; s < ns; ++s) {
    const int b0 = c0 + 64 * s;
    if (tid == 0) help_wait(hc + HC_STAMP + s, epoch, flags);
    __syncthreads();
    invert_tile(smem, K, ldk, b0, min(64, bend - b0), wave < 4, Linv, LinvT);
  }
}

template <int NW, bool HELP>
__device__ __forceinline__ void chain_body(unsigned char *smem, double *K, int64_t ldk, int c0,
                                           int nb, double *__restrict__ dvec,
                                           double *__restrict__ dinv, int *__restrict__ flags,
                                           double *__restrict__ Linv, double *__restrict__ LinvT,
                                           long long *__restrict__ dbg, int *hc, int epoch) {
  double(*M)[C_LD] = reinterpret_cast<double(*)[C_LD]>(smem);
  double(*Wt)[C_WLD] = reinterpret_cast<double(*)[C_WLD]>(smem + CH_ROWS * C_LD * 8);
  double *dD = reinterpret_cast<double *>(smem + CH_ROWS * C_LD * 8 + 2 * 64 * C_WLD * 8);
  double *dI = dD + 64;
  int &s_bad = *reinterpret_cast<int *>(dI + 64);
  double(*Lt)[16][16] = reinterpret_cast<double(*)[16][16]>(smem + CH_LT_OFF);
  constexpr int NT = 64 * NW;  // NW = 16 or 8 wavefronts (8: 256 registers per lane)
  // wave: uniform per wavefront -> scalar register, role tests and tile numbers on the SALU
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bend = c0 + nb;
  const int ns = (nb + 63) / 64;
  // PGF_CHAIN_TIMING: phase stamps of the first sub-panel (100 MHz wall clock), thread 0
  int dbi = 0;
#define CH_STAMP()                                               \
  do {                                                           \
    if (dbg && tid == 0 && dbi < 32) dbg[dbi++] = wall_clock64(); \
  } while (0)
  CH_STAMP();
  if (HELP && tid == 0) help_check_xcc(hc, epoch, flags);

  bool preloaded = false, early0 = false;
  double *dIo = dI + 64 + 2;  // 1/D of the sub-panel just factored (behind the flag word)
  for (int s = 0; s < ns; ++s) {
    // Thread and wavefront indices are laundered once per sub-panel: otherwise every address
    // and role predicate of the loop body is hoisted to the kernel entry and kept alive across
    // the eliminations, where a lane has no register to spare (128 in a 16-wavefront workgroup):
    // 60 spilled registers, and a first scratch access costs microseconds.
    int tl_ = threadIdx.x, wv_ = wave;
    asm volatile("" : "+v"(tl_), "+s"(wv_));
    const int tid = tl_, lane = tid & 63, wave = wv_;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int cb = c0 + 64 * s;
    const int ncol = min(64, bend - cb);
    const int own = max(0, bend - cb - 64);  // block rows below the tile (ncol == 64 if any)
    const int ownp = (own + 63) & ~63;       // padded to whole wavefronts of rows
    if (tid == 0 && !early0) s_bad = 0;
    // ---- load the stack: diagonal tile (identity outside the valid lower triangle) + the
    // block's rows below, all loads of a lane in flight before its first LDS store.  Not for
    // a stack the previous sub-panel's in-block update has left in M already (see there).
    if (!preloaded) {
      constexpr int NQ = 8192 / NT;
      double2_t v[NQ];
      const int np = (64 + ownp) * 32;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int p = q * NT + tid;
        const int row = p >> 5, c2 = (p & 31) * 2;
        double2_t t = (double2_t){0.0, 0.0};
        if (p < np) {
          if (row < 64) {
            if (row < ncol) {
              const double *src = K + (int64_t)(cb + row) * ldk + cb + c2;
              if (c2 + 1 <= row) t = *reinterpret_cast<const double2_t *>(src);
              else if (c2 <= row) t.x = *src;
            } else {
              if (c2 == row) t.x = 1.0;
              if (c2 + 1 == row) t.y = 1.0;
            }
          } else if (row < 64 + own) {
            t = *reinterpret_cast<const double2_t *>(K + (int64_t)(cb + row) * ldk + cb + c2);
          }
        }
        v[q] = t;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int p = q * NT + tid;
        if (p < np) *reinterpret_cast<double2_t *>(&M[p >> 5][(p & 31) * 2]) = v[q];
      }
    }
    __syncthreads();
    CH_STAMP();  // stack loaded

    // ---- 64-column panel, four 16-column steps.  Critical path = wavefront 0 (a+); the rows
    // below the tile follow one step behind on wavefronts 1..3 (64 rows each).  The rank-16
    // MFMA updates are split by urgency: what the NEXT phase 1 reads -- column block sb + 1 of
    // the diagonal tile (step sb) and column block sb of the rows below (step sb - 1) -- is
    // updated between the two barriers of the step; every other tile waits for the idle
    // wavefronts 4.. of the following phase 1, in the shadow of (a+).  W = L D of the diagonal
    // rows is double buffered (Wt[sb & 1]) because of that.
    //   update of diagonal tile (ti, tj) by step t: urgent if tj == t + 1 (phase 2 of step t),
    //                                               else phase 1 of step t + 1;
    //   update of lower    tile (ti, tj) by step t: urgent if tj == t + 1 (phase 2 of step t + 1),
    //                                               else phase 1 of step t + 2.
    const int ot = ownp / 16;  // 16-row tiles below the diagonal tile
    auto diag_tile = [&](int t, int ti, int tj) {  // step t applied to diagonal tile (ti, tj)
      chain_tile16<C_WLD>(M, ti * 16, tj * 16, Wt + (t & 1) * 64, ti * 16, 0, tj * 16, t * 16, l15, l4);
    };
    auto own_tile = [&](int t, int ti, int tj) {  // step t applied to lower tile (ti, tj), ti >= 4
      chain_tile16<C_LD>(M, ti * 16, tj * 16, M, ti * 16, t * 16, tj * 16, t * 16, l15, l4);
    };
    for (int sb = 0; sb < 4; ++sb) {
      if (wave == 0) {
        // (step 0 of a stack built in place has been eliminated beside the tail of the
        // previous sub-panel's in-block update already)
        if (!(early0 && sb == 0)) chain_a_plus(M, Wt + (sb & 1) * 64, dD, dI, s_bad, lane, sb, ncol, Lt[sb & 1]);
      } else if (wave <= 3) {
        if (sb > 0 && 64 * (wave - 1) < own) chain_b_own(M, 64 * wave + lane, sb - 1, lane, Lt[(sb - 1) & 1], dD, dI);
      } else {
        // deferred tiles: diagonal (ti, tj), tj in [sb + 1, 3], of step sb - 1; lower (ti, tj),
        // tj in [sb, 3], of step sb - 2
        const int ndd = sb >= 1 ? (3 - sb) * (4 - sb) / 2 : 0;
        const int ndo = sb >= 2 ? ot * (4 - sb) : 0;
        // (wavefronts 4, 8, 12 share wavefront 0's SIMD: they stay out of its way)
        const int w4 = wave - 4;
        const int rk = (NW == 16) ? ((w4 & 3) ? w4 - (w4 >> 2) - 1 : -1) : w4;
        const int nwk = (NW == 16) ? 9 : NW - 4;
        for (int e0 = rk; rk >= 0 && e0 < ndd + ndo; e0 += nwk) {
          if (e0 < ndd) {
            int e = e0, tj = sb + 1;
            while (e >= 4 - tj) {
              e -= 4 - tj;
              ++tj;
            }
            diag_tile(sb - 1, tj + e, tj);
          } else {
            const int e = e0 - ndd;
            own_tile(sb - 2, 4 + e % ot, sb + e / ot);
          }
        }
        // helper T has had three steps to finish what it was handed one sub-panel ago:
        // everything this sub-panel's in-block update fetches after step 3
        if (HELP && sb == 3 && tid == NT - 64 && s >= 1 && s + 1 < ns)
          help_wait(hc + HC_DONE + s - 1, epoch, flags);
      }
      __syncthreads();
      if (s == 0) CH_STAMP();  // phase 1 of step sb
      // urgent tiles: diagonal (ti, sb + 1), ti in [sb + 1, 3], of step sb; lower (ti, sb) of
      // step sb - 1
      const int nud = 3 - sb;
      const int nuo = sb >= 1 ? ot : 0;
      for (int e0 = wave; e0 < nud + nuo; e0 += NW) {
        if (e0 < nud) diag_tile(sb, sb + 1 + e0, sb + 1);
        else own_tile(sb - 1, 4 + (e0 - nud), sb);
      }
      __syncthreads();
    }
    if (s == 0) CH_STAMP();  // four steps done
    // ---- trailing update inside the block (rows / columns below the tile, K-depth 64, A = -X
    // from M, B = X D^-1): the C tiles live in global memory; ALL of a wavefront's tiles are
    // fetched in one burst here, so that their latency (they were last written by another
    // kernel: HBM / Infinity Cache, ~2 us) is paid once and hides behind the write-back.
    // Only 16 x 16 sub-tiles on or below the diagonal are enumerated, dealt round-robin: the
    // phase is bound by the CU's matrix pipes, every wavefront should carry the same number.
    // With helpers only the tiles of the NEXT sub-panel's columns are this workgroup's: tile
    // column 0 of the region, 10 + 16 (nt - 1) sub-tiles; the rest is helper T's.
    constexpr int MT = ((HELP ? 42 : 78) + NW - 1) / NW;  // sub-tiles per wavefront, at most
    const int nt = ownp / 64;
    const int total = HELP ? (nt ? 10 + 16 * (nt - 1) : 0) : nt * (nt + 1) / 2 * 16 - nt * 6;
    auto decode = [&](int e, int &gi, int &gj, int &mi, int &mj) {
      if (HELP) {
        if (e < 10) {
          const int ti = (e >= 6) ? 3 : (e >= 3) ? 2 : (e >= 1) ? 1 : 0;
          mi = 16 * ti;
          mj = 16 * (e - ti * (ti + 1) / 2);
        } else {
          mi = 64 + 16 * ((e - 10) >> 2);  // tile row 1 + (e - 10) / 16, sub-row ((e - 10) / 4) % 4
          mj = 16 * ((e - 10) & 3);
        }
      } else {
        decode_subtile(e, mi, mj);
      }
      mi += 64;
      mj += 64;
      gi = cb + mi;
      gj = cb + mj;
    };
    // whole 64-row tiles below (no ragged edge): the next stack is built in place
    const bool direct = own > 0 && (own & 63) == 0;
    double4_t ct[MT];
    early0 = false;
    if (HELP && NW == 16 && direct) {
      // ---- With helpers and whole tiles the sub-panel boundary is pipelined:
      //  A  wavefronts 1-3 finish the lagging rows (step 3); the others fetch their C sub-tiles
      //     and write the factored tile, D, 1/D and the flags back meanwhile
      //  B  L rows written back; first round of MFMA sub-tiles = the NEXT diagonal tile (plus
      //     six others), which goes straight into M[0..63]; stamp to the helpers
      //  C  wavefront 0 eliminates step 0 of the next sub-panel while the others finish the
      //     remaining sub-tiles (in registers: M's rows 64.. are still their operands)
      //  D  those go into M as the rest of the next stack
      auto eidx = [&](int q) {  // sub-tile of round q: round 0 one per wavefront, then 1..15 only
        if (q == 0) return wave;
        return wave == 0 ? total : NW + (wave - 1) + (NW - 1) * (q - 1);
      };
      // (the C loads of wavefronts 1-3 are in flight while they finish the lagging rows)
#pragma unroll
      for (int q = 0; q < MT; ++q) {
        const int e = eidx(q);
        ct[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (e < total) {
          int gi, gj, mi, mj;
          decode(e, gi, gj, mi, mj);
          const int j = gj + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) ct[q][r] = ld_agent(K + (int64_t)(gi + l4 + 4 * r) * ldk + j);
        }
      }
      if (s == 0) CH_STAMP();  // C loads issued
      if (wave >= 1 && wave <= 3 && 64 * (wave - 1) < own) chain_b_own(M, 64 * wave + lane, 3, lane, Lt[1], dD, dI);
      for (int p = tid; p < 64 * 64; p += NT) {
        const int row = p >> 6, c = p & 63;
        if (c <= row) K[(int64_t)(cb + row) * ldk + cb + c] = M[row][c];
      }
      if (s == 0) CH_STAMP();  // tile written back
      if (tid < 64) {
        dvec[cb + tid] = dD[tid];
        dinv[cb + tid] = dI[tid];
        dIo[tid] = dI[tid];
      }
      if (wave == 0) {
        const unsigned long long negs = __ballot(dD[lane] < 0.0);
        if (lane == 0) {
          if (s_bad) atomicOr(&flags[0], 1);
          s_bad = 0;  // for the early step 0 below
          const int neg = __popcll(negs);
          if (neg) atomicAdd(&flags[1], neg);
        }
      }
      if (s == 0) CH_STAMP();  // (wavefront 0) before the barrier
      __syncthreads();  // A -> B
      CH_STAMP();       // panel factored
      for (int p = tid; p < own * 32; p += NT) {
        const int row = 64 + (p >> 5), c2 = (p & 31) * 2;
        double2_t v = *reinterpret_cast<const double2_t *>(&M[row][c2]);
        v.x *= dIo[c2];
        v.y *= dIo[c2 + 1];
        *reinterpret_cast<double2_t *>(K + (int64_t)(cb + row) * ldk + cb + c2) = v;
      }
      auto mfma_sub = [&](int q) {
        const int e = eidx(q);
        if (e < total) {
          int gi, gj, mi, mj;
          decode(e, gi, gj, mi, mj);
          double4_t c = ct[q];
#pragma unroll 4
          for (int ks = 0; ks < 64; ks += 4) {
            const double av = -M[mi + l15][ks + l4];
            const double bv = M[mj + l15][ks + l4] * dIo[ks + l4];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
          }
          ct[q] = c;
        }
      };
      auto to_stack = [&](int q) {  // sub-tile of round q -> its place in the next stack
        const int e = eidx(q);
        if (e < total) {
          int gi, gj, mi, mj;
          decode(e, gi, gj, mi, mj);
          const int cj = mj - 64 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ri = mi - 64 + l4 + 4 * r;
            M[ri][cj] = (cj <= ri) ? ct[q][r] : 0.0;
          }
        }
      };
      if (s == 0) CH_STAMP();  // L rows written back
      mfma_sub(0);
      if (s == 0) CH_STAMP();  // first MFMA round
      // rows 0..63 of M (the old tile) were last read by the write-back before A -> B
      if (wave < 10) to_stack(0);
      for (int p = tid; p < 6 * 256; p += NT) {  // the six sub-tiles above the new diagonal
        const int t6 = p >> 8, rr = (p >> 4) & 15, cc = p & 15;
        const int ti = (t6 >= 5) ? 2 : (t6 >= 3) ? 1 : 0;  // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
        const int tj = (ti == 0) ? 1 + t6 : (ti == 1) ? t6 - 1 : 3;
        M[16 * ti + rr][16 * tj + cc] = 0.0;
      }
      if (s == 0) CH_STAMP();  // next diagonal tile stored
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-backs have left
      if (s == 0) CH_STAMP();  // stores drained
      __syncthreads();  // B -> C
      if (tid == 0) help_post(hc + HC_STAMP + s, epoch);
      CH_STAMP();  // next diagonal tile in place
      // (one barrier, reached on two paths: the branch is uniform per wavefront, and this way
      // no accumulator of the other path is live across the elimination, which has no
      // registers to spare)
      if (wave == 0) {
        chain_a_plus(M, Wt, dD, dI, s_bad, lane, 0, 64, Lt[0]);
        __syncthreads();  // C -> D
      } else {
#pragma unroll
        for (int q = 1; q < MT; ++q) mfma_sub(q);
        __syncthreads();  // C -> D: nobody reads the old rows 64.. of M any more
        if (wave >= 10) to_stack(0);
#pragma unroll
        for (int q = 1; q < MT; ++q) to_stack(q);
      }
      preloaded = true;
      early0 = true;
    } else {
      if (wave >= 1 && wave <= 3 && 64 * (wave - 1) < own) chain_b_own(M, 64 * wave + lane, 3, lane, Lt[1], dD, dI);
      __syncthreads();
      CH_STAMP();  // panel factored
#pragma unroll
      for (int q = 0; q < MT; ++q) {
        const int e = wave + q * NW;
        ct[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (e < total) {
          int gi, gj, mi, mj;
          decode(e, gi, gj, mi, mj);
          const int j = gj + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = gi + l4 + 4 * r;
            if (i < bend && j < bend && j <= i)
              ct[q][r] = HELP ? ld_agent(K + (int64_t)i * ldk + j) : K[(int64_t)i * ldk + j];
          }
        }
      }
      // ---- write back: factored tile, D, 1/D, flags; rows below: L = X D^-1
      for (int p = tid; p < 64 * 64; p += NT) {
        const int row = p >> 6, c = p & 63;
        if (row < ncol && c <= row) K[(int64_t)(cb + row) * ldk + cb + c] = M[row][c];
      }
      if (tid < ncol) {
        dvec[cb + tid] = dD[tid];
        dinv[cb + tid] = dI[tid];
      }
      if (wave == 0) {
        const unsigned long long negs = __ballot(lane < ncol && dD[lane] < 0.0);
        if (lane == 0) {
          if (s_bad) atomicOr(&flags[0], 1);
          const int neg = __popcll(negs);
          if (neg) atomicAdd(&flags[1], neg);
        }
      }
      for (int p = tid; p < own * 32; p += NT) {
        const int row = 64 + (p >> 5), c2 = (p & 31) * 2;
        double2_t v = *reinterpret_cast<const double2_t *>(&M[row][c2]);
        v.x *= dI[c2];
        v.y *= dI[c2 + 1];
        *reinterpret_cast<double2_t *>(K + (int64_t)(cb + row) * ldk + cb + c2) = v;
      }
      if (s == 0) CH_STAMP();  // written back
#pragma unroll
      for (int q = 0; q < MT; ++q) {
        const int e = wave + q * NW;
        if (e < total) {
          int gi, gj, mi, mj;
          decode(e, gi, gj, mi, mj);
          {
            double4_t c = ct[q];
#pragma unroll 4
            for (int ks = 0; ks < 64; ks += 4) {
              const double av = -M[mi + l15][ks + l4];
              const double bv = M[mj + l15][ks + l4] * dI[ks + l4];
              c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
            }
            if (direct && mj < 128) {
              ct[q] = c;  // next sub-panel's stack: stays in registers until M is free
            } else {
              const int j = gj + l15;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = gi + l4 + 4 * r;
                if (i < bend && j < bend && j <= i) K[(int64_t)i * ldk + j] = c[r];
              }
            }
          }
        }
      }
      if (HELP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-backs have left
      __syncthreads();  // M is refilled next; the global tiles written above are read back
      if (HELP && tid == 0) help_post(hc + HC_STAMP + s, epoch);
      preloaded = direct;
      if (direct) {
        // the next sub-panel's stack (columns 64..127 of this one's rows 64..) goes from the
        // accumulators straight into M: no round trip through global memory
#pragma unroll
        for (int q = 0; q < MT; ++q) {
          const int e = wave + q * NW;
          if (e < total) {
            int gi, gj, mi, mj;
            decode(e, gi, gj, mi, mj);
            if (mj < 128) {
              const int cj = mj - 64 + l15;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int ri = mi - 64 + l4 + 4 * r;
                M[ri][cj] = (cj <= ri) ? ct[q][r] : 0.0;
              }
            }
          }
        }
        // the six 16 x 16 sub-tiles above the diagonal of the new diagonal tile
        for (int p = tid; p < 6 * 256; p += NT) {
          const int t6 = p >> 8, rr = (p >> 4) & 15, cc = p & 15;
          const int ti = (t6 >= 5) ? 2 : (t6 >= 3) ? 1 : 0;  // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
          const int tj = (ti == 0) ? 1 + t6 : (ti == 1) ? t6 - 1 : 3;
          M[16 * ti + rr][16 * tj + cc] = 0.0;
        }
      }
    }
    CH_STAMP();  // in-block update done
  }

  // ---- inverses of the block's unit-lower diagonal tiles (invert_tile), four wavefronts per
  // tile; with helpers this is helper I's work
  if (!HELP) {
    for (int g0 = 0; g0 < ns; g0 += NW / 4) {
      const int g = g0 + (wave >> 2);
      const int b0 = c0 + 64 * g;
      invert_tile(smem, K, ldk, b0, min(64, bend - b0), g < ns, Linv, LinvT);
    }
  }
  CH_STAMP();  // inverses done
#undef CH_STAMP
}

// workgroup 0: the chain; with HELP (grid of 17) workgroups 8 and 16 are its helpers, the
// others leave at once
template <int NW, bool HELP>
__global__ __launch_bounds__(64 * NW) void k_diag_chain(double *K, int64_t ldk, int c0, int nb,
                                                     double *__restrict__ dvec,
                                                     double *__restrict__ dinv,
                                                     int *__restrict__ flags,
                                                     double *__restrict__ Linv,
                                                     double *__restrict__ LinvT,
                                                     long long *__restrict__ dbg, int *hc,
                                                     int epoch) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SMEM];
  if (blockIdx.x == 0) {
    chain_body<NW, HELP>(smem, K, ldk, c0, nb, dvec, dinv, flags, Linv, LinvT, dbg, hc, epoch);
  } else if (HELP && blockIdx.x == 8) {
    helper_tiles<NW>(smem, K, ldk, c0, nb, dvec, hc, epoch, flags);
  } else if (HELP && blockIdx.x == 16) {
    helper_inverses<NW>(smem, K, ldk, c0, nb, hc, epoch, flags, Linv, LinvT);
  }
}

// ------------------------------------------------------------------ D(k + 1) beside update tiles
// ONE launch: workgroup 0 is the chain D(k + 1) of the next outer block (workgroups 8 and 16
// its helpers); every other workgroup applies pending blocks to one 128 x 128 tile of the
// trailing matrix below the next diagonal block, as the launch's job table says (update_tile of
// pgf_ldlt_dev.h with the A operand scaled by D while it is staged).  The two roles touch
// disjoint cache lines and hand nothing to each other, so the launch is correct whatever order
// the workgroups run in; dispatched first, the chain has its CU from the start.
// Jobs of one lazy update launch: job q brings column block [col0, col0 + 256) -- rows from
// rowstart (or the diagonal, whichever is lower) -- from "blocks < kc0 / 256 applied" to
// "blocks < (kc0 + KB) / 256 applied"; its 128 x 128 tiles are numbered tile_begin[q] ...
#define UPD_MAXJOBS 96
#define UPD_TM 64  // tile rows (x 128 columns): with the persistent tile loop below the finer unit packs
                   // a launch's work into fewer idle CU-rounds (128: ~260 units of 37 us over 253 CUs = two
                   // rounds, half of the second one idle)
struct UpdJobs {
  int njobs;
  int tile_begin[UPD_MAXJOBS + 1];
  int col0[UPD_MAXJOBS], rowstart[UPD_MAXJOBS], kc0[UPD_MAXJOBS], KB[UPD_MAXJOBS];
  int ntc[UPD_MAXJOBS];  // 128-wide tile columns of the job (2 per column block; adjacent column
                         // blocks with the same K-range share a job: large N, eager plan)
  // a leading segment of the K-range in the pre-eliminated block's panel (UpdVirt): columns
  // [kc0v, kc0v + KBv) of V first, then columns [kc0, kc0 + KB) of K (either may be empty) -- ONE
  // job, one pass over the tiles: two jobs on the same tiles of a launch would race
  int kc0v[UPD_MAXJOBS], KBv[UPD_MAXJOBS];
};
// The pre-eliminated block (DenseLdlt::V): panel rows V[i][.], D-scaling vd.
struct UpdVirt {
  const double *V;
  int64_t ldv;
  const double *vd;
};

// tile t of a launch's job table.  ONE 128 x 128 tile per workgroup, 16 wavefronts as 4 x 4 with
// 2 x 2 MFMA tiles each and two LDS stages of K-depth 32: the chain's LDS footprint allows one
// workgroup per CU, so the 16 wavefronts share one staged panel pair.
__device__ __forceinline__ void update_job_tile(unsigned char *smem, int t, double *K, int64_t ldk,
                                                const double *__restrict__ dvec, int N, int nrows,
                                                const UpdJobs &jobs, const UpdVirt &uv) {
  if (t >= jobs.tile_begin[jobs.njobs]) return;
  int q = 0;
  while (t >= jobs.tile_begin[q + 1]) ++q;
  t -= jobs.tile_begin[q];
  const int col0 = jobs.col0[q], rs = jobs.rowstart[q];
  // tile columns in turn; rows of each from max(rowstart, column start)
  int j0 = col0, i0 = max(rs, j0);
  for (int c = 0; c < jobs.ntc[q]; ++c) {
    j0 = col0 + 128 * c;
    i0 = max(rs, j0);
    const int nc = (j0 < N && i0 < nrows) ? (nrows - i0 + UPD_TM - 1) / UPD_TM : 0;
    if (t < nc) break;
    t -= nc;
  }
  i0 += UPD_TM * t;
  const int kc0 = jobs.kc0[q], KBr = jobs.KB[q], kc0v = jobs.kc0v[q], KBv = jobs.KBv[q];
  if (KBv > 0)
    update_tile<UPD_TM, 128, 32, 4, 4, 1, true>(smem, threadIdx.x, i0, j0, K, ldk, uv.V + kc0v, uv.ldv, N,
                                                nrows, N, kc0, KBv + KBr, uv.vd + kc0v, KBv, K + kc0, ldk,
                                                dvec + kc0);
  else
    update_tile<UPD_TM, 128, 32, 4, 4, 1, true>(smem, threadIdx.x, i0, j0, K, ldk, K + kc0, ldk, N, nrows, N,
                                                kc0, KBr, dvec + kc0);
}

// The update role, persistent: a workgroup takes tile after tile of the launch's job table from an
// atomic counter (zeroed with the factorisation's flags) until the table is exhausted -- deepest
// jobs first, so the long tiles start early and the short ones fill the gaps (round 2: one tile
// per workgroup; ~260 tiles over 253 CUs ran as two rounds, the second nearly empty).
__device__ __forceinline__ void update_worker(unsigned char *smem, double *K, int64_t ldk,
                                              const double *__restrict__ dvec, int N, int nrows,
                                              const UpdJobs &jobs, const UpdVirt &uv, int *ctr) {
  __shared__ int s_next;
  const int total = jobs.tile_begin[jobs.njobs];
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(ctr, 1);
    __syncthreads();
    const int t = s_next;
    __syncthreads();  // (s_next is rewritten next round; the previous tile's LDS reads are done)
    if (t >= total) return;
    update_job_tile(smem, t, K, ldk, dvec, N, nrows, jobs, uv);
  }
}

template <bool HELP>
__global__ __launch_bounds__(1024) void k_chain_update(double *K, int64_t ldk, int c0, int nb,
                                                       double *__restrict__ dvec,
                                                       double *__restrict__ dinv,
                                                       int *__restrict__ flags,
                                                       double *__restrict__ Linv,
                                                       double *__restrict__ LinvT, int *hc,
                                                       int epoch, int N, int nrows,
                                                       const UpdJobs jobs, const UpdVirt uv, int *ctr) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SMEM];
  const int b = (int)blockIdx.x;
  if (b == 0) {
    chain_body<16, HELP>(smem, K, ldk, c0, nb, dvec, dinv, flags, Linv, LinvT, nullptr, hc, epoch);
    return;
  }
  if (HELP && b == 8) {
    helper_tiles<16>(smem, K, ldk, c0, nb, dvec, hc, epoch, flags);
    return;
  }
  if (HELP && b == 16) {
    helper_inverses<16>(smem, K, ldk, c0, nb, hc, epoch, flags, Linv, LinvT);
    return;
  }
  update_worker(smem, K, ldk, dvec, N, nrows, jobs, uv, ctr);
}

// the update role alone (per-kernel profiling, PGF_FUSED=0): same tiles, same job table
__global__ __launch_bounds__(1024) void k_update_jobs(double *K, int64_t ldk,
                                                      const double *__restrict__ dvec, int N,
                                                      int nrows, const UpdJobs jobs, const UpdVirt uv,
                                                      int *ctr) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 256 * 34 * 8];
  update_worker(smem, K, ldk, dvec, N, nrows, jobs, uv, ctr);
}

// The pre-eliminated block's update of ONE diagonal block, C -= V diag(vd) V^T on rows and columns
// [c0, c0 + nb): the only piece of it the first chain waits for.  Latency, not throughput (a
// 128 x 128 job tile of depth 1024 would take ~85 us): one 16 x 16 tile of the lower triangle per
// workgroup, its four wavefronts taking the 16-column chunks of the depth in turn (c = w, w + 4,
// ...; two chunks in flight per wavefront: the operands come straight from L2, ~1 us per dependent
// round trip) and summed through LDS in a fixed order.  Lane (l15, l4) reads four consecutive
// doubles of its row per chunk and feeds component t to MFMA t: the sum over k does not care
// which lane group carries which k as long as A and B agree.
__device__ __forceinline__ void virtual_diag_body(double (*part)[4][64], int t, double *K, int64_t ldk, int c0,
                                                  int nb, const double *__restrict__ V, int64_t ldv,
                                                  const double *__restrict__ vd, int depth, int vrows) {
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = 16 * ti, j0 = 16 * tj;
  const double *pa = V + (int64_t)min(c0 + i0 + l15, vrows - 1) * ldv + 4 * l4;
  const double *pb = V + (int64_t)min(c0 + j0 + l15, vrows - 1) * ldv + 4 * l4;
  const double *pd = vd + 4 * l4;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int nc = depth / 16;
  for (int c = wave; c < nc; c += 8) {
    const int k0 = 16 * c;
    const bool two = c + 4 < nc;
    const int k1 = two ? k0 + 64 : k0;
    const double2_t a0 = *reinterpret_cast<const double2_t *>(pa + k0), a1 = *reinterpret_cast<const double2_t *>(pa + k0 + 2);
    const double2_t b0 = *reinterpret_cast<const double2_t *>(pb + k0), b1 = *reinterpret_cast<const double2_t *>(pb + k0 + 2);
    const double2_t d0 = *reinterpret_cast<const double2_t *>(pd + k0), d1 = *reinterpret_cast<const double2_t *>(pd + k0 + 2);
    const double2_t e0 = *reinterpret_cast<const double2_t *>(pa + k1), e1 = *reinterpret_cast<const double2_t *>(pa + k1 + 2);
    const double2_t f0 = *reinterpret_cast<const double2_t *>(pb + k1), f1 = *reinterpret_cast<const double2_t *>(pb + k1 + 2);
    const double2_t g0 = *reinterpret_cast<const double2_t *>(pd + k1), g1 = *reinterpret_cast<const double2_t *>(pd + k1 + 2);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[0] * d0[0], b0[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[1] * d0[1], b0[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[0] * d1[0], b1[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[1] * d1[1], b1[1], acc, 0, 0, 0);
    if (two) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-e0[0] * g0[0], f0[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-e0[1] * g0[1], f0[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-e1[0] * g1[0], f1[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-e1[1] * g1[1], f1[1], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave != 0) return;
  const int j = j0 + l15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + l4 + 4 * r;
    if (i < nb && j <= i) {
      double *p = K + (int64_t)(c0 + i) * ldk + c0 + j;
      *p = *p + (((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane]);
    }
  }
}
__global__ __launch_bounds__(256) void k_virtual_diag(double *K, int64_t ldk, int c0, int nb,
                                                      const double *__restrict__ V, int64_t ldv,
                                                      const double *__restrict__ vd, int depth, int vrows) {
  __shared__ double part[4][4][64];
  virtual_diag_body(part, (int)blockIdx.x, K, ldk, c0, nb, V, ldv, vd, depth, vrows);
}
// batched: the first diagonal block of every instance that factorises (blockIdx.z = instance).
// Throughput, not the latency of one tile: a wavefront takes one 16 x 16 tile over the whole
// depth, four chunks (64 columns) in flight -- 34 workgroups per instance, all of a small batch
// resident at once (the split-depth version above: 4352 workgroups for 32 instances, 42 us).
__global__ __launch_bounds__(256) void kb_virtual_diag(const BInst *__restrict__ tab, int depth) {
  const BInst &I = tab[blockIdx.z];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0];
  if (N <= 0) return;
  const int nb = min(256, N), nt = (nb + 15) / 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int t = 4 * (int)blockIdx.x + wave;
  if (t >= nt * (nt + 1) / 2) return;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int i0 = 16 * ti, j0 = 16 * tj;
  const double *pa = I.V + (int64_t)min(i0 + l15, N) * I.ldv + 4 * l4;
  const double *pb = I.V + (int64_t)min(j0 + l15, N) * I.ldv + 4 * l4;
  const double *pd = I.vd + 4 * l4;
  double4_t acc;
  const int j = j0 + l15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + l4 + 4 * r;
    acc[r] = (i < nb && j <= i) ? I.K[(int64_t)i * I.ldk + j] : 0.0;
  }
  for (int k0 = 0; k0 < depth; k0 += 64) {
    double2_t a[4][2], b[4][2], d[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = min(k0 + 16 * c, depth - 16);  // (a clamped chunk is skipped below)
      a[c][0] = *reinterpret_cast<const double2_t *>(pa + k);
      a[c][1] = *reinterpret_cast<const double2_t *>(pa + k + 2);
      b[c][0] = *reinterpret_cast<const double2_t *>(pb + k);
      b[c][1] = *reinterpret_cast<const double2_t *>(pb + k + 2);
      d[c][0] = *reinterpret_cast<const double2_t *>(pd + k);
      d[c][1] = *reinterpret_cast<const double2_t *>(pd + k + 2);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (k0 + 16 * c >= depth) break;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[c][0][0] * d[c][0][0], b[c][0][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[c][0][1] * d[c][0][1], b[c][0][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[c][1][0] * d[c][1][0], b[c][1][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[c][1][1] * d[c][1][1], b[c][1][1], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + l4 + 4 * r;
    if (i < nb && j <= i) I.K[(int64_t)i * I.ldk + j] = acc[r];
  }
}

// ------------------------------------------------------------------ batched wrappers
// The same chain and T(k) kernels with a batch dimension: instance = workgroup (chain) or
// batch_decode (T); every instance has its own N on the device, workgroups beyond it return.
// The chain runs WITHOUT helper workgroups here: with hundreds of instances there is one
// chain per CU and a spinning helper could wait for a CU its own chain occupies.
// ---- the register-resident chain (pgf_chain3.h; PGF_CHAIN=3): no helper workgroups
static_assert(C3_SMEM <= CH_SMEM, "chain3 LDS footprint");
// (ONE kernel for the chain alone -- an empty job table -- and beside update tiles: with two
// kernels inlining chain3_body in one translation unit the compiler spills 58 registers in each,
// with one 5)
__global__ __launch_bounds__(1024) void k_chain3_update(double *K, int64_t ldk, int c0, int nb,
                                                        double *__restrict__ dvec, double *__restrict__ dinv,
                                                        int *__restrict__ flags, double *__restrict__ Linv,
                                                        double *__restrict__ LinvT, int N, int nrows,
                                                        const UpdJobs jobs, const UpdVirt uv, int *ctr) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SMEM];
  const int b = (int)blockIdx.x;
  if (b == 0) {
    chain3_body(smem, K, ldk, c0, nb, dvec, dinv, flags, Linv, LinvT, nullptr);
    return;
  }
  update_worker(smem, K, ldk, dvec, N, nrows, jobs, uv, ctr);
}

template <bool HELP>
__global__ __launch_bounds__(1024) void kb_diag_chain(const BInst *__restrict__ tab, int B, int Bp, int m,
                                                      int c0, int epoch) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SMEM];
  // roles: [0, Bp) chains, [Bp, 2 Bp) helpers T, [2 Bp, 3 Bp) helpers I; Bp = B rounded up to 8,
  // so that the three workgroups of an instance land on one XCD (ids equal modulo 8)
  const int id = (int)blockIdx.x, role = id / Bp, inst = id - role * Bp;
  if (inst >= B) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m;
  if (c0 >= N) return;
  const int nb = min(256, N - c0);
  if (role == 0)
    chain_body<16, HELP>(smem, I.K, I.ldk, c0, nb, I.dvec, I.dinv, I.flags, I.Linv, I.LinvT, nullptr,
                         I.hctl, epoch);
  else if (HELP && role == 1)
    helper_tiles<16>(smem, I.K, I.ldk, c0, nb, I.dvec, I.hctl, epoch, I.flags);
  else if (HELP && role == 2)
    helper_inverses<16>(smem, I.K, I.ldk, c0, nb, I.hctl, epoch, I.flags, I.Linv, I.LinvT);
}

// chains of the outer block at c1 beside the previous block's trailing update (everything below
// the diagonal block at c1), all instances in ONE launch: workgroups [0, Bp) are the chains
// (Bp = B rounded up to 8, so that the tiles behind them keep the instance -> XCD pinning),
// the rest one 128 x 128 update tile each.  Small batches only: with a chain per CU the tiles
// would queue behind them.
template <bool HELP>
__global__ __launch_bounds__(1024) void kb_chain_update(const BInst *__restrict__ tab, int B, int Bp,
                                                        int per, int m, int wbuf, int c1, int epoch,
                                                        int vdepth) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CH_SMEM];
  const int id = (int)blockIdx.x;
  constexpr int NR = HELP ? 3 : 1;  // roles of the chain: chain, helper T, helper I
  if (id < NR * Bp) {
    const int role = id / Bp, ci = id - role * Bp;
    if (ci >= B) return;
    const BInst &I = tab[ci];
    if (I.ctl[0] == 0) return;
    const int N = I.counts[0] + m;
    if (c1 >= N) return;
    const int nb = min(256, N - c1);
    if (role == 0)
      chain_body<16, HELP>(smem, I.K, I.ldk, c1, nb, I.dvec, I.dinv, I.flags, I.Linv, I.LinvT, nullptr,
                           I.hctl, epoch);
    else if (role == 1)
      helper_tiles<16>(smem, I.K, I.ldk, c1, nb, I.dvec, I.hctl, epoch, I.flags);
    else
      helper_inverses<16>(smem, I.K, I.ldk, c1, nb, I.hctl, epoch, I.flags, I.Linv, I.LinvT);
    return;
  }
  int inst, t;
  if (!batch_decode_id(id - NR * Bp, B, per, inst, t)) return;
  const BInst &I = tab[inst];
  if (I.ctl[0] == 0) return;
  const int N = I.counts[0] + m, nrows = N + 1;
  if (c1 >= N) return;
  const int row0 = c1 + min(256, N - c1);
  if (row0 >= nrows) return;
  const int tr = (nrows - row0 + 127) / 128, tc = (N - c1 + 127) / 128;
  int by = 0;
  while (by < tr) {
    const int nc = min(tc, (row0 + 128 * by + 127 - c1) / 128 + 1);
    if (t < nc) break;
    t -= nc;
    ++by;
  }
  if (by >= tr) return;
  if (vdepth > 0)
    // c1 = 0, condensed order: the "previous block" is the pre-eliminated constraint block -- its
    // panel V with the scaling -1 / delta, everything but the first diagonal block (kb_virtual_diag)
    update_tile<128, 128, 32, 4, 4, 1, true>(smem, threadIdx.x, row0 + 128 * by, c1 + 128 * t, I.K, I.ldk, I.V,
                                             I.ldv, N, nrows, N, 0, vdepth, I.vd);
  else
    update_tile<128, 128, 32, 4, 4, 1>(smem, threadIdx.x, row0 + 128 * by, c1 + 128 * t, I.K, I.ldk,
                                       I.W + (int64_t)wbuf * I.wstride, 256, N, nrows, N, c1 - 256, 256);
}

void ldlt_batch_launch_update_diag(hipStream_t s, const BInst *tab, int B, int m, int wbuf, int c1) {
  hipLaunchKernelGGL(kb_update_diag, dim3(batch_grid(B, 36)), dim3(256), 0, s, tab, B, m, wbuf, c1);
}
// per: tiles of the largest possible instance (Nmax) in this launch
// Epochs of the chain <-> helper stamps: ONE counter for the process, single-instance and
// batched launches alike.  A handle's stamp words outlive the launches that wrote them (handles
// are pooled and move in and out of batches): with a counter per handle or per mode an old
// stamp could equal a new launch's epoch and release a helper before its chain had produced
// anything (seen once in ~10 runs of the test suite as a wrong inertia in a batched test that
// followed single-instance tests on the same pooled handles).
static std::atomic<int> g_help_epoch{0};
static int next_help_epoch() {
  int e = ++g_help_epoch;
  if (e <= 0 || e == 0x7fffffff) {  // wrapped: start over (2^31 launches)
    g_help_epoch = 1;
    e = 1;
  }
  return e;
}
static bool chain_helpers();  // (below: PGF_CHAIN_HELP and the process-wide switch-off)
void ldlt_batch_launch_chain_update(hipStream_t s, const BInst *tab, int B, int Nmax, int m, int wbuf,
                                    int c1, bool helpers, int vdepth) {
  const int nrows = Nmax + 1, row0 = std::min(c1 + 256, Nmax);
  int per = 0;
  if (row0 < nrows) {
    const int tr = (nrows - row0 + 127) / 128, tc = (Nmax - c1 + 127) / 128;
    for (int by = 0; by < tr; ++by) per += std::min(tc, (row0 + 128 * by + 127 - c1) / 128 + 1);
  }
  const int Bp = 8 * ((B + 7) / 8);
  const int tiles = per ? batch_grid(B, per) : 0;
  // The helper workgroups (two per instance, a CU each for the length of the chain) shorten the
  // chain from ~86 to ~70 us -- which only pays while the launch is bound by its chains: with more
  // than ~1.5 tiles per CU (a 128 x 128 x 256 tile: ~40 us) the update tiles are the longer role
  // and want those CUs (32 instances, N = 1280: the first two of four launches)
  static const int ncu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
      n = pr.multiProcessorCount;
    return n;
  }();
  if (helpers && 2 * B * per > 3 * std::max(1, ncu - B)) helpers = false;
  if (helpers && chain_helpers())
    hipLaunchKernelGGL(kb_chain_update<true>, dim3(3 * Bp + tiles), dim3(1024), 0, s, tab, B, Bp,
                       std::max(per, 1), m, wbuf, c1, next_help_epoch(), vdepth);
  else
    hipLaunchKernelGGL(kb_chain_update<false>, dim3(Bp + tiles), dim3(1024), 0, s, tab, B, Bp,
                       std::max(per, 1), m, wbuf, c1, 0, vdepth);
}
void ldlt_batch_launch_virtual_diag(hipStream_t s, const BInst *tab, int B, int vdepth) {
  hipLaunchKernelGGL(kb_virtual_diag, dim3(34, 1, B), dim3(256), 0, s, tab, vdepth);
}
void ldlt_batch_launch_chain(hipStream_t s, const BInst *tab, int B, int m, int c0, bool helpers) {
  const int Bp = 8 * ((B + 7) / 8);
  if (helpers && chain_helpers())
    hipLaunchKernelGGL(kb_diag_chain<true>, dim3(3 * Bp), dim3(1024), 0, s, tab, B, Bp, m, c0,
                       next_help_epoch());
  else
    hipLaunchKernelGGL(kb_diag_chain<false>, dim3(Bp), dim3(1024), 0, s, tab, B, Bp, m, c0, 0);
}
// per: 16-row groups of the largest possible instance below the block
void ldlt_batch_launch_trsm(hipStream_t s, const BInst *tab, int B, int per, int m, int wbuf, int c0) {
  const int wgs = (per + KB_TRSM_RT - 1) / KB_TRSM_RT;
  hipLaunchKernelGGL(kb_trsm_block, dim3(batch_grid(B, wgs)), dim3(256), 0, s, tab, B, wgs, m, wbuf, c0);
}

// ------------------------------------------------------------------ host schedule
bool ldlt_use_lookahead() {
  static const bool on = !(getenv("PGF_FACTOR") && atoi(getenv("PGF_FACTOR")) == 1);
  return on;
}

// PGF_CHAIN=3: the register-resident chain of pgf_chain3.h instead of the LDS-panel chain
static bool chain3_on() {
  static const bool on = getenv("PGF_CHAIN") && atoi(getenv("PGF_CHAIN")) == 3;
  return on;
}

// wavefronts of the chain workgroup: 16 (128 registers per lane, a few spilled) or 8
static int chain_waves() {
  static const int nw = (getenv("PGF_CHAIN_WAVES") && atoi(getenv("PGF_CHAIN_WAVES")) == 8) ? 8 : 16;
  return nw;
}


// helper workgroups of the diagonal chain (PGF_CHAIN_HELP=0: the chain does everything itself);
// switched off for the process after a failed placement check or a timed-out hand-over
static bool g_help_off = false;
static bool g_fused_ud_off = false;
static bool chain_helpers() {
  static const bool on = !(getenv("PGF_CHAIN_HELP") && atoi(getenv("PGF_CHAIN_HELP")) == 0);
  return on && !g_help_off;
}
void ldlt_chain_helpers_off() {
  // (flags[2] does not say which of the in-launch hand-overs failed: both kinds go)
  g_help_off = true;
  g_fused_ud_off = true;
}
bool ldlt_chain_helpers_enabled() { return chain_helpers(); }
void ldlt_chain_helpers_set(bool on) {
  g_help_off = !on;
  g_fused_ud_off = !on;
}

// test hook (pgf_debug_fail_next_helper): make the factorisation just enqueued look like one
// whose helpers failed their checks
__global__ void k_helper_inject(int *__restrict__ flags) { atomicOr(&flags[2], 1); }
void ldlt_inject_helper_failure(hipStream_t s, int *flags) {
  hipLaunchKernelGGL(k_helper_inject, dim3(1), dim3(1), 0, s, flags);
}

// budget of one lazy update launch in tile-blocks (128 x 128 tile x K-depth 256; 255 CUs take
// one each per ~47 us) and the number of pending blocks an optional job may take at once;
// PGF_LAZY_BUDGET=0: no limit = the eager schedule (every launch applies block k everywhere)
static int lazy_budget() {
  static const int b = getenv("PGF_LAZY_BUDGET") ? atoi(getenv("PGF_LAZY_BUDGET")) : 420;
  return b > 0 ? b : (1 << 30);
}
// pending blocks an optional job takes at once: 2 in the natural order; 4 with a pre-eliminated
// block (its virtual blocks are all pending from the start: deeper passes over the same C tiles
// re-read them less often; measured 2.03 -> 2.015 ms at config 2, 3 and 5+ are slower)
static int lazy_cap(int vdepth = 0) {
  static const int c = getenv("PGF_LAZY_CAP") ? std::max(1, atoi(getenv("PGF_LAZY_CAP"))) : 0;
  return c ? c : (vdepth > 0 ? 4 : 2);
}

// T(k) and the next diagonal block's update in one launch (k_trsm_ud); PGF_FUSED_UD=0 or a failed
// placement check: two launches
static bool fused_ud() {
  static const bool on = !(getenv("PGF_FUSED_UD") && atoi(getenv("PGF_FUSED_UD")) == 0);
  return on && !g_fused_ud_off;
}

static bool fused() {
  static const bool on = !(getenv("PGF_FUSED") && atoi(getenv("PGF_FUSED")) == 0);
  return on;
}

// PGF_CHAIN_TIMING=1 (diagnostic): the chain kernel of the FIRST block of every factorisation
// stamps its phases into this buffer; ldlt_chain_timing_dump prints the last set
static long long *g_chain_dbg = nullptr;
static long long *chain_dbg_buffer() {
  static const bool on = getenv("PGF_CHAIN_TIMING") != nullptr;
  if (!on) return nullptr;
  if (!g_chain_dbg) {
    if (hipMalloc((void **)&g_chain_dbg, 32 * sizeof(long long)) != hipSuccess) return nullptr;
    (void)hipMemset(g_chain_dbg, 0, 32 * sizeof(long long));
  }
  return g_chain_dbg;
}
void ldlt_chain_timing_dump() {
  if (!g_chain_dbg) return;
  long long h[32];
  if (hipMemcpy(h, g_chain_dbg, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
  fprintf(stderr, "k_diag_chain phase stamps (us since kernel start):");
  for (int i = 1; i < 32 && h[i]; ++i) fprintf(stderr, " %.2f", (double)(h[i] - h[0]) * 0.01);
  fprintf(stderr, "\n");
}

// ------------------------------------------------------------------ lazy update plan
// The trailing update beside the chain.  Launch L (chain D(L), panels of blocks < L available)
// must leave column block L complete below its diagonal block (T(L) reads it) and column block
// L + 1 complete through block L - 1 (k_update_diag / D(L + 1)); every other (column block,
// block) pair may wait.  Work is counted in tile-blocks (one 128 x 128 tile x K-depth 256 =
// one workgroup for ~45 us); what hides behind a chain of ~86 us is a little less than two
// rounds over 255 CUs.  Earliest deadline first with a per-launch budget: nearest column block
// first, an optional job takes at most `cap` pending blocks at once (its tiles run for cap x
// 45 us).  Too small a budget pushes work against the deadlines, where it comes back as a few
// very deep tiles on a few CUs; too large a one front-loads the launches as the eager schedule
// does.  The budget is therefore chosen per factorisation: the candidate with the smallest
// estimated total time (plan_cost) -- the reduced size changes from step to step.
struct UpdPlan {
  UpdJobs first;                // beside the chain of column block 0 (virtual blocks only)
  std::vector<UpdJobs> launch;  // [L - 1]: beside the chain of column block k + 1
  double cost = 0.0;            // estimated sum of launch times in units of one tile-block
};

// A pre-eliminated block of depth vdepth (DenseLdlt::V) counts as nv = ceil(vdepth / OB) column
// blocks that are factorised before the first one: block indices below are unified, virtual blocks
// [0, nv) first, real block k at nv + k.  Updates commute, so the only deadlines are the usual
// ones -- a column block's diagonal tile complete before its chain, its rows below before its T --
// and the virtual blocks are pending work like any other: only the first diagonal block is due
// before the first chain (k_virtual_diag: small tiles, a few microseconds, the only exposed part);
// the rows below it and column block 1 follow beside that chain (stage `first'), the rest lazily.
static void plan_updates(UpdPlan &pl, int N, int nrows, int OB, int budget, int cap, double chain_units,
                         int vdepth = 0) {
  const int nblk = (N + OB - 1) / OB;
  const int nv = (vdepth + OB - 1) / OB;
  std::vector<int> done(nblk + 2, 0);
  pl.launch.assign(std::max(0, nblk - 1), UpdJobs());
  pl.first.njobs = 0;
  pl.first.tile_begin[0] = 0;
  pl.cost = 0.0;
  auto tiles = [&](int col0, int rowstart) {
    int n = 0;
    for (int c = 0; c < 2; ++c) {
      const int j0 = col0 + 128 * c;
      if (j0 >= N) continue;
      const int i0 = std::max(rowstart, j0);
      if (i0 < nrows) n += (nrows - i0 + UPD_TM - 1) / UPD_TM;
    }
    return n;
  };
  // stage -1: first, k >= 0: the launch beside the chain of column block k + 1
  for (int st = (nv > 0 ? -1 : 0); st < nblk - 1; ++st) {
    const int k = st;
    const int avail = st < 0 ? nv - 1 : nv + k;  // newest block whose panel exists
    UpdJobs jb;
    jb.njobs = 0;
    int units = 0, maxdepth = 0, cnt[UPD_MAXJOBS];
    // unified blocks [p0, p1]: the virtual part and the real part are the two segments of one job
    auto add = [&](int J, int rowstart, int p0, int p1) {
      if (p1 < p0) return;
      int kc0v = 0, KBv = 0, kc0 = 0, KB = 0;
      if (p0 < nv) {
        const int v1 = std::min(p1, nv - 1);
        kc0v = p0 * OB;
        KBv = std::min((v1 + 1) * OB, vdepth) - p0 * OB;
      }
      if (p1 >= nv) {
        const int r0 = std::max(p0, nv) - nv, r1 = p1 - nv;
        kc0 = r0 * OB;
        KB = (r1 - r0 + 1) * OB;
      }
      const int depth = p1 - p0 + 1;
      const int n = tiles(J * OB, rowstart);
      if (!n || KB + KBv <= 0) return;
      units += n * depth;
      maxdepth = std::max(maxdepth, depth);
      // a whole column block right behind the previous job's, same K-range: one job
      if (jb.njobs > 0 && rowstart == J * OB) {
        const int q = jb.njobs - 1;
        if (jb.rowstart[q] == jb.col0[q] && jb.col0[q] + 128 * jb.ntc[q] == J * OB && jb.kc0[q] == kc0 &&
            jb.KB[q] == KB && jb.kc0v[q] == kc0v && jb.KBv[q] == KBv) {
          jb.ntc[q] += OB / 128;
          cnt[q] += n;
          return;
        }
      }
      const int q = jb.njobs++;
      jb.col0[q] = J * OB;
      jb.rowstart[q] = rowstart;
      jb.kc0[q] = kc0;
      jb.KB[q] = KB;
      jb.kc0v[q] = kc0v;
      jb.KBv[q] = KBv;
      jb.ntc[q] = OB / 128;
      cnt[q] = n;
    };
    int Jopt;  // first column block whose pending work is optional at this stage
    const int lim = budget;
    if (st == -1) {
      add(0, std::min(OB, N), done[0], avail);  // (its diagonal block: k_virtual_diag)
      done[0] = avail + 1;
      if (nblk > 1) {
        add(1, OB, done[1], avail);
        done[1] = avail + 1;
      }
      Jopt = 2;
    } else {
      const int c1 = (k + 1) * OB, nb1 = std::min(OB, N - c1), row0 = c1 + nb1;
      if (done[k + 1] <= avail) add(k + 1, row0, done[k + 1], avail);
      done[k + 1] = avail + 1;
      if (k + 2 < nblk && done[k + 2] <= avail) {
        add(k + 2, (k + 2) * OB, done[k + 2], avail);
        done[k + 2] = avail + 1;
      }
      Jopt = k + 3;
    }
    for (int J = Jopt; J < nblk; ++J) {
      const int pend = avail + 1 - done[J];
      if (pend <= 0) continue;
      // (room is kept for the jobs that must run; what is skipped here stays pending)
      if (units >= lim || jb.njobs >= UPD_MAXJOBS - 2) break;
      const int take = std::min(pend, cap);
      add(J, J * OB, done[J], done[J] + take - 1);
      done[J] += take;
    }
    // deepest jobs first: their tiles take longest
    int order[UPD_MAXJOBS];
    for (int q = 0; q < jb.njobs; ++q) order[q] = q;
    std::stable_sort(order, order + jb.njobs,
                     [&](int a, int b) { return jb.KB[a] + jb.KBv[a] > jb.KB[b] + jb.KBv[b]; });
    UpdJobs &js = st == -1 ? pl.first : pl.launch[k];
    js.njobs = jb.njobs;
    js.tile_begin[0] = 0;
    for (int q = 0; q < jb.njobs; ++q) {
      const int o = order[q];
      js.col0[q] = jb.col0[o];
      js.rowstart[q] = jb.rowstart[o];
      js.kc0[q] = jb.kc0[o];
      js.KB[q] = jb.KB[o];
      js.ntc[q] = jb.ntc[o];
      js.kc0v[q] = jb.kc0v[o];
      js.KBv[q] = jb.KBv[o];
      js.tile_begin[q + 1] = js.tile_begin[q] + cnt[o];
    }
    // list-scheduling estimate of the launch: work / 253 CUs, at least the deepest tile, in
    // whole tile times; and never less than the chain
    const double t = std::max((double)maxdepth, std::ceil(units / 253.0));
    pl.cost += std::max(chain_units, t);
  }
}

hipError_t ldlt_factor2_async(DenseLdlt &f, int N, int nrows) {
  f.N = N;
  f.factored = false;
  hipStream_t s = f.stream;
  // flags [0, 4) and the update launches' tile counters behind them
  hipError_t e = hipMemsetAsync(f.flags, 0, (4 + LDLT_UPD_COUNTERS) * sizeof(int), s);
  if (e != hipSuccess) return e;
  static const int ncu = []() {
    int dev = 0, n = 256;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
      n = pr.multiProcessorCount;
    return n;
  }();
  PgfProfile *p = (f.prof && f.prof->enabled) ? f.prof : nullptr;
  if (p) {
    p->factor_spans.emplace_back(prof_event(p), prof_event(p));
    (void)hipEventRecord(p->factor_spans.back().first, s);
  }
  constexpr int OB = 256;
  const int64_t ldw = OB;
  auto span_begin = [&](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
    if (!p) return;
    v.emplace_back(prof_event(p), prof_event(p));
    (void)hipEventRecord(v.back().first, s);
  };
  auto span_end = [&](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
    if (p) (void)hipEventRecord(v.back().second, s);
  };
  PgfProfile dummy;
  PgfProfile &pr = p ? *p : dummy;
  const bool help = chain_helpers();
  const int vdepth = f.vdepth;
  const UpdVirt uv{f.V, f.ldv, f.vd};
  auto launch_d = [&](int c0) {
    span_begin(pr.chain_spans);
    long long *dbg = (c0 == 0) ? chain_dbg_buffer() : nullptr;
    const int nb = std::min(OB, N - c0);
    const int ep = next_help_epoch();
    if (chain3_on()) {
      UpdJobs none;
      none.njobs = 0;
      none.tile_begin[0] = 0;
      hipLaunchKernelGGL(k_chain3_update, dim3(1), dim3(1024), 0, s, f.K, f.ldk, c0, nb, f.dvec, f.dinv,
                         f.flags, f.Linv, f.LinvT, N, nrows, none, uv, f.flags + 4);
    } else if (chain_waves() == 16) {
      if (help)
        hipLaunchKernelGGL((k_diag_chain<16, true>), dim3(17), dim3(1024), 0, s, f.K, f.ldk, c0, nb,
                           f.dvec, f.dinv, f.flags, f.Linv, f.LinvT, dbg, f.hctl, ep);
      else
        hipLaunchKernelGGL((k_diag_chain<16, false>), dim3(1), dim3(1024), 0, s, f.K, f.ldk, c0, nb,
                           f.dvec, f.dinv, f.flags, f.Linv, f.LinvT, dbg, f.hctl, ep);
    } else {
      if (help)
        hipLaunchKernelGGL((k_diag_chain<8, true>), dim3(17), dim3(512), 0, s, f.K, f.ldk, c0, nb,
                           f.dvec, f.dinv, f.flags, f.Linv, f.LinvT, dbg, f.hctl, ep);
      else
        hipLaunchKernelGGL((k_diag_chain<8, false>), dim3(1), dim3(512), 0, s, f.K, f.ldk, c0, nb,
                           f.dvec, f.dinv, f.flags, f.Linv, f.LinvT, dbg, f.hctl, ep);
    }
    span_end(pr.chain_spans);
  };
  // T(c0) -- with the update of the next diagonal block in the same launch (k_trsm_ud) unless
  // per-kernel events are wanted or there is no next block
  const bool prod = p && p->mode == 2;  // production launches, one span each
  const bool fud = fused_ud() && (!p || prod);
  auto launch_t = [&](int c0, double *Wb) {
    const int nb = std::min(OB, N - c0);
    const int below = nrows - (c0 + nb);
    if (below <= 0) return;
    if (fud && c0 + nb < N) {
      const int c1 = c0 + nb, nb1 = std::min(OB, N - c1);
      const int nT = (below + 15) / 16, nA = std::min(nT, (nb1 + 15) / 16);
      const int nt = (nb1 + 31) / 32, S = nA + nt * (nt + 1) / 2;
      const int rest = nT - nA;  // row groups at the ordinary ids
      const int last = rest > 0 ? (rest - 1 < 7 * S ? (rest - 1) + (rest - 1) / 7 + 1 : rest - 1 + S) : 0;
      const int grid = std::max(8 * (S - 1) + 1, last + 1);
      if (prod) span_begin(pr.trsmud_spans);
      hipLaunchKernelGGL(k_trsm_ud, dim3(grid), dim3(256), 0, s, f.K, f.ldk, Wb, ldw, nrows, c0, nb,
                         f.dinv, f.Linv, N, f.hctl, next_help_epoch(), f.flags);
      if (prod) span_end(pr.trsmud_spans);
      return;
    }
    span_begin(pr.trsm_spans);
    hipLaunchKernelGGL(k_trsm_block, dim3((below + 15) / 16), dim3(256), 0, s, f.K, f.ldk, Wb, ldw,
                       nrows, c0, nb, f.dinv, f.Linv);
    span_end(pr.trsm_spans);
  };
  // Lazy trailing update (production; plan_updates above).  PGF_LAZY_BUDGET fixes the budget
  // (0 = no limit = the eager schedule: every launch applies its block everywhere).
  const bool lazy = fused() && (!p || prod);  // one launch for chain + update; else two, same jobs
  const int nblk = (N + OB - 1) / OB;
  UpdPlan plan;
  if (nblk > 1 || vdepth > 0) {
    // cached per (N, nrows, depth of the pre-eliminated block): a Newton iteration refactorises
    // the same size many times
    static thread_local int cN = -1, cR = -1, cV = -1;
    static thread_local UpdPlan cplan;
    if (cN != N || cR != nrows || cV != vdepth) {
      // ~86 us chain / time of one tile-block (64 x 128 x 256: ~21 us, 128 x 128: ~40 us)
      // ~66 us chain (DPP elimination) / time of one tile-block (64 x 128 x 256: ~21 us, 128 x 128: ~40 us)
      const double chain_units = 66.0 / (UPD_TM == 64 ? 21.0 : 40.0);
      if (getenv("PGF_LAZY_BUDGET")) {
        plan_updates(cplan, N, nrows, OB, lazy_budget(), lazy_cap(vdepth), chain_units, vdepth);
      } else {
        // the search (60 candidate plans) once per 128-row size class: the winning budget
        // depends on the tile counts, and the reduced size moves by a few rows from step to
        // step when the active set churns (config 5b: a new N every step)
        static thread_local std::unordered_map<int, int> budget_of;
        const int key = (nrows + 127) / 128 + 4096 * ((vdepth + 127) / 128);
        auto it = budget_of.find(key);
        if (it == budget_of.end()) {
          UpdPlan best;
          int best_b = 1 << 30;
          plan_updates(best, N, nrows, OB, best_b, lazy_cap(vdepth), chain_units, vdepth);  // eager
          for (int b = 200 * (128 / UPD_TM); b <= 1400 * (128 / UPD_TM); b += 20 * (128 / UPD_TM)) {
            UpdPlan cand;
            plan_updates(cand, N, nrows, OB, b, lazy_cap(vdepth), chain_units, vdepth);
            if (cand.cost < best.cost - 1e-9) {
              best = std::move(cand);
              best_b = b;
            }
          }
          budget_of.emplace(key, best_b);
          cplan = std::move(best);
        } else {
          plan_updates(cplan, N, nrows, OB, it->second, lazy_cap(vdepth), chain_units, vdepth);
        }
      }
      cN = N;
      cR = nrows;
      cV = vdepth;
    }
    plan = cplan;
  }
  // algorithmic work of a launch's update jobs: entries (i, j), j <= i, of every job's region, 2 KB
  // flops each; bytes: every such entry read and written once, the L rows of the region's rows
  // and of its columns once per job
  auto job_work = [&](const UpdJobs &js, double &fl, double &by) {
    fl = by = 0.0;
    for (int q = 0; q < js.njobs; ++q) {
      const int col0 = js.col0[q], colEnd = std::min(N, col0 + 128 * js.ntc[q]);
      const int rs = std::max(js.rowstart[q], col0);
      double cnt = 0.0;
      for (int i = rs; i < nrows; ++i) cnt += std::min(colEnd, i + 1) - col0;
      const double kd = js.KB[q] + js.KBv[q];
      fl += 2.0 * cnt * kd;
      by += 16.0 * cnt + 8.0 * kd * ((double)(nrows - rs) + (double)(colEnd - col0));
    }
  };
  // chain of column block [c1, c1 + nb1) beside the update jobs js, one launch
  auto launch_fused = [&](int c1, int nb1, const UpdJobs &js, int *ctr) {
    const int ntiles = js.tile_begin[js.njobs];
    const int ep = next_help_epoch();
    if (prod) {
      span_begin(pr.fused_spans);
      double fl, by;
      job_work(js, fl, by);
      p->fused_flops.push_back(fl);
      p->fused_bytes.push_back(by);
    }
    const int workers = std::min(ntiles, ncu - 3);  // one workgroup per CU: persistent tile loops
    if (chain3_on())
      hipLaunchKernelGGL(k_chain3_update, dim3(1 + workers), dim3(1024), 0, s, f.K, f.ldk, c1, nb1, f.dvec,
                         f.dinv, f.flags, f.Linv, f.LinvT, N, nrows, js, uv, ctr);
    else if (help)
      hipLaunchKernelGGL(k_chain_update<true>, dim3(std::max(17, workers + 3)), dim3(1024), 0, s, f.K, f.ldk,
                         c1, nb1, f.dvec, f.dinv, f.flags, f.Linv, f.LinvT, f.hctl, ep, N, nrows, js, uv, ctr);
    else
      hipLaunchKernelGGL(k_chain_update<false>, dim3(1 + workers), dim3(1024), 0, s, f.K, f.ldk, c1, nb1,
                         f.dvec, f.dinv, f.flags, f.Linv, f.LinvT, f.hctl, ep, N, nrows, js, uv, ctr);
    if (prod) span_end(pr.fused_spans);
  };
  // update jobs alone
  auto launch_jobs = [&](const UpdJobs &js, int *ctr) {
    const int ntiles = js.tile_begin[js.njobs];
    if (ntiles <= 0) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (p && !prod) {
      e0 = prof_event(p);
      e1 = prof_event(p);
      (void)hipEventRecord(e0, s);
    }
    hipLaunchKernelGGL(k_update_jobs, dim3(std::min(ntiles, ncu)), dim3(1024), 0, s, f.K, f.ldk, f.dvec, N,
                       nrows, js, uv, ctr);
    if (p && !prod) {
      (void)hipEventRecord(e1, s);
      p->update_spans.emplace_back(e0, e1);
      double fl, by;
      job_work(js, fl, by);
      p->update_flops.push_back(fl);
      p->update_bytes.push_back(by);
    }
  };
  int buf = 0;
  if (N > 0) {
    // a pre-eliminated block: its update of the first diagonal block first (nothing hides it), the
    // rest of column block 0, column block 1 -- and whatever the plan adds -- beside the first chain
    if (vdepth > 0) {
      const int nb0 = std::min(OB, N), nt = (nb0 + 15) / 16;
      hipLaunchKernelGGL(k_virtual_diag, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, f.K, f.ldk, 0, nb0, f.V,
                         f.ldv, f.vd, vdepth, nrows);
    }
    if (vdepth > 0 && lazy && plan.first.njobs > 0) {
      launch_fused(0, std::min(OB, N), plan.first, f.flags + 4);
    } else {
      launch_d(0);
      if (vdepth > 0) launch_jobs(plan.first, f.flags + 4);
    }
    launch_t(0, f.W);
  }
  for (int c0 = 0; c0 + OB < N; c0 += OB, buf ^= 1) {
    const int c1 = c0 + OB, nb1 = std::min(OB, N - c1);
    const double *Wb = f.W + (size_t)buf * f.wstride;
    const int nt = (nb1 + 31) / 32;
    if (!fud) {
      span_begin(pr.udiag_spans);
      hipLaunchKernelGGL(k_update_diag<32>, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, f.K, f.ldk, Wb,
                         ldw, c0, OB, c1, nb1);
      span_end(pr.udiag_spans);
    }
    // D(k + 1) beside trailing-update work, in one launch; while profiling (per-kernel events)
    // and on request (PGF_FUSED=0) D(k + 1) and the whole of U(k) as two launches
    int *ctr = f.flags + 4 + ((c0 / OB + 1) % (LDLT_UPD_COUNTERS - 1));
    if (lazy) {
      launch_fused(c1, nb1, plan.launch[c0 / OB], ctr);
    } else {
      launch_d(c1);
      launch_jobs(plan.launch[c0 / OB], ctr);
    }
    launch_t(c1, f.W + (size_t)(buf ^ 1) * f.wstride);
  }
  if (p) (void)hipEventRecord(p->factor_spans.back().second, s);
  if (f.inject_helper_failure) {
    f.inject_helper_failure = 0;
    hipLaunchKernelGGL(k_helper_inject, dim3(1), dim3(1), 0, s, f.flags);
  }
  e = hipMemcpyAsync(f.h_flags, f.flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
