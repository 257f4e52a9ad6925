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

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ helpers
// broadcast lane `src` (wave-uniform, compile-time after unrolling) of a double through
// two v_readlane_b32 (scalar result; no LDS round trip as with __shfl / ds_bpermute)
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Loads of data another kernel may have written while a second queue was resident (the
// look-ahead schedule): system-scope relaxed atomic loads (global_load ... sc0 sc1) bypass
// the per-XCD L2, which is not coherent with the other XCDs' L2s.
__device__ __forceinline__ double ld_f64(const double *p, int coh) {
  if (coh) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return *p;
}
__device__ __forceinline__ double2_t ld_f64x2(const double *p, int coh) {
  double2_t v;
  if (coh) {
    v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    v = *reinterpret_cast<const double2_t *>(p);
  }
  return v;
}

__device__ __forceinline__ void st_f64(double *p, double v, int coh) {
  if (coh)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // write-through
  else
    *p = v;
}
__device__ __forceinline__ void st_f64x2(double *p, double2_t v, int coh) {
  if (coh) {
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    *reinterpret_cast<double2_t *>(p) = v;
  }
}

// 1 / d to within an ulp or two: v_rcp_f64 seed + two Newton steps (5 dependent ops
// instead of the ~12 of an IEEE-correct division; the pivots only enter through
// products, which the 1e-10 iterate tolerance covers with 5 digits to spare)
__device__ __forceinline__ double fast_recip(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
#ifndef PGF_RECIP_ONE_STEP
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
#endif
  return r;
}

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
  const int coh = (skip >> 3) & 1;      // bit 3 of `skip`: system-scope loads (look-ahead)
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
    if (wave == 0 && !(skip & 1)) a_plus(sb);
    if (wave == 1 && sb > 0 && !(skip & 2)) b_own(sb - 1);
    if (wave >= 2 && sb >= 2 && !(skip & 4)) own_deferred(sb);
    __syncthreads();
    // phase 2: (c-diag) of this step, urgent (c-own) column of the previous one
    if (!(skip & 4)) phase2(sb);
    __syncthreads();
  }
  if (wave == 1 && !(skip & 2)) b_own(3);  // no tiles are left to update after the last step
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

// ------------------------------------------------------------------ trailing update
// C[i][j] -= sum_k W[i][k] * L[j][k]   for row0 <= i < nrows, col0 <= j < colEnd, j <= i
// (rows >= N are carried right-hand sides: every column < N is "below" them).
// 128 x 128 tile per workgroup, 4 wavefronts as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles
// of v_mfma_f64_16x16x4_f64 (A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15],
// C/D: row = (l>>4) + 4*reg, col = l&15).  The accumulators START as the C tile (all 64
// loads in flight at once, hidden behind the first operand fetch) and -W is staged, so
// the epilogue is store-only.  K-chunks of 16 go through LDS with the next chunk
// prefetched into registers; LDS rows padded to 18 doubles (conflict-free ds_read_b64
// for the fragment pattern, 16-byte aligned ds_write_b128).
#define UPD_BM 128

// BM x BN = tile (rows x columns); BK = K-chunk staged per barrier pair; 4 wavefronts as
// 2 x 2, each (BM/2) x (BN/2) = TM x TN MFMA tiles.  LDS rows are padded to BK + 2 doubles.
template <int BM, int BN, int BK, int WR = 2, int WC = 2, int DB = 0>
__device__ __forceinline__ void update_tile(unsigned char *smem, const int i0, const int j0,
                                            double *__restrict__ K, int64_t ldk,
                                            const double *__restrict__ W, int64_t ldw, int N,
                                            int nrows, int colEnd, int kc0, int KBc) {
  const int KB = KBc & 0xFFFFF;        // K-depth
  const int coh = (KBc >> 20) & 1;     // bit 20: system-scope loads (look-ahead schedule)
  constexpr int NT = 64 * WR * WC;           // threads per workgroup
  constexpr int WM = BM / WR, WN = BN / WC;  // rows / columns per wavefront
  constexpr int TM = WM / 16, TN = WN / 16;  // MFMA tiles per wavefront
  constexpr int LD = BK + 2;
  constexpr int PPR = BK / 2;                // 16-byte pieces per row
  constexpr int PA = BM * PPR / NT, PB = BN * PPR / NT;
  static_assert(PA >= 1 && PB >= 1, "tile too small for the workgroup");
  constexpr int STAGE = (BM + BN) * LD * 8;  // bytes of one LDS stage (A then B)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int l15 = lane & 15, l4 = lane >> 4;

  // accumulators <- C tile (entries above the diagonal / outside the region are never
  // stored back; whatever they hold stays confined to its own accumulator element)
  double4_t acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
    for (int nj = 0; nj < TN; ++nj) {
      const int j = j0 + wc * WN + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * WM + mi * 16 + l4 + 4 * r;
        double v = 0.0;
        if (i < nrows && j < colEnd && j <= i) v = ld_f64(K + (int64_t)i * ldk + j, coh);
        acc[mi][nj][r] = v;
      }
    }
  }

  // staging map: piece p = q*256 + tid -> row p / PPR, two doubles at column (p % PPR)*2
  double2_t pa[PA], pb[PB];
  auto fetch = [&](int kk, int = 0) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int p = q * NT + tid;
      const int row = p / PPR, kofs = (p % PPR) * 2;
      const int gi = i0 + row;
      double2_t va = (double2_t){0.0, 0.0};
      if (gi < nrows) va = ld_f64x2(W + (int64_t)gi * ldw + kk + kofs, coh);
      pa[q] = va;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int p = q * NT + tid;
      const int row = p / PPR, kofs = (p % PPR) * 2;
      const int gj = j0 + row;
      double2_t vb = (double2_t){0.0, 0.0};
      if (gj < colEnd)
        vb = ld_f64x2(K + (int64_t)gj * ldk + kc0 + kk + kofs, coh);
      pb[q] = vb;
    }
  };
  auto stage = [&](int buf, int = 0) {
    double(*As)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE);
    double(*Bs)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE + BM * LD * 8);
    // negate here, not at the fetch: touching the loaded value there would make the
    // wavefront wait for the prefetch before it starts the current chunk's MFMAs
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int p = q * NT + tid;
      *reinterpret_cast<double2_t *>(&As[p / PPR][(p % PPR) * 2]) = -pa[q];
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int p = q * NT + tid;
      *reinterpret_cast<double2_t *>(&Bs[p / PPR][(p % PPR) * 2]) = pb[q];
    }
  };
  auto compute = [&](int buf) {
    double(*As)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE);
    double(*Bs)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE + BM * LD * 8);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      double a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = As[wr * WM + t * 16 + l15][ks + l4];
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = Bs[wc * WN + t * 16 + l15][ks + l4];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int nj = 0; nj < TN; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
  };

  fetch(0);
  if (DB == 2) {
    // two LDS stages and TWO chunks in flight, in two STATICALLY named register sets (a
    // runtime-indexed set goes to scratch): the loads of chunk c+2 are issued before chunk c
    // is computed, so every fetch has two compute phases to land.
    double2_t qa[PA], qb[PB];  // second register set (the first is pa2[0] / pb2[0])
    auto fetch1 = [&](int kk) {
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int p = q * NT + tid;
        const int row = p / PPR, kofs = (p % PPR) * 2;
        const int gi = i0 + row;
        double2_t va = (double2_t){0.0, 0.0};
        if (gi < nrows) va = ld_f64x2(W + (int64_t)gi * ldw + kk + kofs, coh);
        qa[q] = va;
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int p = q * NT + tid;
        const int row = p / PPR, kofs = (p % PPR) * 2;
        const int gj = j0 + row;
        double2_t vb = (double2_t){0.0, 0.0};
        if (gj < colEnd) vb = ld_f64x2(K + (int64_t)gj * ldk + kc0 + kk + kofs, coh);
        qb[q] = vb;
      }
    };
    auto stage1 = [&](int buf) {
      double(*As)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE);
      double(*Bs)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE + BM * LD * 8);
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int p = q * NT + tid;
        *reinterpret_cast<double2_t *>(&As[p / PPR][(p % PPR) * 2]) = -qa[q];
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int p = q * NT + tid;
        *reinterpret_cast<double2_t *>(&Bs[p / PPR][(p % PPR) * 2]) = qb[q];
      }
    };
    const int nc = KB / BK;
    if (nc > 1) fetch1(BK);
    stage(0, 0);
    __syncthreads();
    for (int c = 0; c < nc; c += 2) {
      // even chunk c: LDS stage 0; set 1 holds chunk c+1 (in flight); set 0 is free
      if (c + 2 < nc) fetch((c + 2) * BK, 0);
      compute(0);
      if (c + 1 < nc) stage1(1);
      __syncthreads();
      if (c + 1 >= nc) break;
      // odd chunk c+1: LDS stage 1; set 0 holds chunk c+2 (in flight); set 1 is free
      if (c + 3 < nc) fetch1((c + 3) * BK);
      compute(1);
      if (c + 2 < nc) stage(0, 0);
      __syncthreads();
    }
  } else if (DB) {
    // two LDS stages, ONE barrier per chunk: chunk c is computed from stage c&1 while the
    // prefetched chunk c+1 is written to the other stage (last read one iteration ago)
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int kk = 0; kk < KB; kk += BK) {
      const bool more = kk + BK < KB;
      if (more) fetch(kk + BK);
      compute(cur);
      if (more) stage(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    for (int kk = 0; kk < KB; kk += BK) {
      __syncthreads();  // previous chunk's fragment reads are done
      stage(0);
      __syncthreads();
      if (kk + BK < KB) fetch(kk + BK);
      compute(0);
    }
  }

  // epilogue: store-only, lower triangle of the region
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
    for (int nj = 0; nj < TN; ++nj) {
      const int j = j0 + wc * WN + nj * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * WM + mi * 16 + l4 + 4 * r;
        if (i < nrows && j < colEnd && j <= i) st_f64(K + (int64_t)i * ldk + j, acc[mi][nj][r], coh);
      }
    }
  }
}

template <int BM, int BN, int BK, int WR = 2, int WC = 2, int DB = 0>
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
  update_tile<BM, BN, BK, WR, WC, DB>(smem, i0, j0, K, ldk, W, ldw, N, nrows, colEnd, kc0, KB);
}

// ------------------------------------------------------------------ 64-bit DPP helpers
// gfx950 has 64-bit DPP with the row_newbcast control: the source operand is read from lane
// L of the reader's own 16-lane row.  One v_fmac_f64_dpp therefore does
//   acc += (value held by lane L of my row) * mult
// which is exactly the rank-1 update of a 16 x 16 tile whose rows live on the 16 lanes of a
// row -- no v_readlane pair, no LDS round trip.  hipcc adds no hazard padding inside asm
// statements, so the two wait states a DPP read needs after a VALU write are in the string.
template <int L>
__device__ __forceinline__ void fmac_bcast(double &acc, double from_lane, double mult) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(from_lane), "v"(mult), "n"(L));
}
template <int L>
__device__ __forceinline__ double bcast16(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "=v"(r)
               : "v"(v), "n"(L));
  return r;
}

// rank-1 update of columns K..15 with pivot column J:  a[K] += A[K][J] * (-l_i)
template <int J, int K>
__device__ __forceinline__ void tile_elim(double (&a)[16], double negl) {
  if constexpr (K < 16) {
    fmac_bcast<K>(a[K], a[J], negl);
    tile_elim<J, K + 1>(a, negl);
  }
}
// unblocked LDL^T of the 16 x 16 tile whose row (lane & 15) is in a[0..15]
template <int J>
__device__ __forceinline__ void tile_factor(double (&a)[16], int l15, int cb, int nb, double &d_mine,
                                            double &di_mine, int &bad_any) {
  if constexpr (J < 16) {
    const double d = bcast16<J>(a[J]);
    const bool bad = (d == 0.0) || !(fabs(d) <= 1.79e308);
    const double di = bad ? 0.0 : fast_recip(d);
    bad_any |= (bad && (cb + J) < nb) ? 1 : 0;
    if (l15 == J) {
      d_mine = d;
      di_mine = di;
    }
    const double l = a[J] * di;
    tile_elim<J, J + 1>(a, -l);
    a[J] = l;
    tile_factor<J + 1>(a, l15, cb, nb, d_mine, di_mine, bad_any);
  }
}
// substitution x L_bb^T = a_row for one row per lane; tl[c] of lane j holds L_bb[j][c]
template <int T, int J>
__device__ __forceinline__ void subst_inner(double (&x)[16], const double (&tl)[16], double negxt) {
  if constexpr (J < 16) {
    fmac_bcast<J>(x[J], tl[T], negxt);  // x[J] -= x[T] * L_bb[J][T]
    subst_inner<T, J + 1>(x, tl, negxt);
  }
}
template <int T>
__device__ __forceinline__ void subst_rows(double (&x)[16], const double (&tl)[16]) {
  if constexpr (T < 15) {
    subst_inner<T, T + 1>(x, tl, -x[T]);
    subst_rows<T + 1>(x, tl);
  }
}

// ------------------------------------------------------------------ wide panel kernel
// Same algorithm as panel_body, generalised to a PW-column panel (PW = 128: half as many
// panel launches and two of three inner updates per outer block disappear) with OWN rows
// per workgroup.  The PW x PW diagonal block is kept in LDS as a packed lower trapezoid:
// row-tile ti (16 rows) stores 16 (ti + 1) + 2 doubles per row, so the 128 x 128 block
// takes 76 KB instead of 133 KB and leaves room for the own rows and W; the +2 padding
// keeps MFMA fragment reads conflict-free (row strides of 4 or 36 dwords mod 64).
template <int PW, int OWN>
struct PanelLayout {
  static constexpr int CT = PW / 16;
  static constexpr int R = PW + OWN;
  static constexpr int RT = R / 16;
  static constexpr int DIAG = 128 * CT * (CT + 1) + 32 * CT;
  static constexpr int OWN_LD = PW + 2;
  static constexpr int OWND = OWN * OWN_LD;
  static constexpr int WTD = R * 18;
  static constexpr int SMEM = (DIAG + OWND + WTD + 2 * PW) * 8 + 16;
  __device__ __forceinline__ static int off(int row, int col) {
    if (row < PW) {
      const int ti = row >> 4;
      return 128 * ti * (ti + 1) + 32 * ti + (row & 15) * (16 * (ti + 1) + 2) + col;
    }
    return DIAG + (row - PW) * OWN_LD + col;
  }
};

template <int PW, int OWN>
__device__ __forceinline__ void panel_body2(unsigned char *smem, const int wg,
                                            double *__restrict__ K, int64_t ldk,
                                            double *__restrict__ W, int64_t ldw, int wofs, int N,
                                            int nrows, int c0, double *__restrict__ dvec,
                                            double *__restrict__ dinv, int *__restrict__ flags,
                                            int skip) {
  using LY = PanelLayout<PW, OWN>;
  constexpr int CT = LY::CT, R = LY::R, RT = LY::RT;
  double *M = reinterpret_cast<double *>(smem);
  double(*Wt)[18] = reinterpret_cast<double(*)[18]>(M + LY::DIAG + LY::OWND);
  double *dD = M + LY::DIAG + LY::OWND + LY::WTD;
  double *dI = dD + PW;
  int &s_bad = *reinterpret_cast<int *>(dI + PW);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nb = min(PW, N - c0);
  const int rbase = c0 + nb + wg * OWN;  // first own row (global)
  const int coh = (skip >> 3) & 1;
  if (tid == 0) s_bad = 0;

  // ---- load the diagonal block (lower trapezoid; identity padding beyond nb) ...
  {
    constexpr int HP = PW / 2;             // 16-byte pieces per full row
    constexpr int DP = PW * HP / 256;      // per lane
    double2_t v[DP];
#pragma unroll
    for (int q = 0; q < DP; ++q) {
      const int p = q * 256 + tid;
      const int row = p / HP, c2 = (p % HP) * 2;
      double2_t t = (double2_t){0.0, 0.0};
      if (c2 < 16 * ((row >> 4) + 1)) {
        if (row < nb) {
          const double *src = K + (int64_t)(c0 + row) * ldk + c0 + c2;
          if (c2 + 1 <= row) t = ld_f64x2(src, coh);
          else if (c2 <= row) t.x = ld_f64(src, coh);
        } else {
          if (c2 == row) t.x = 1.0;
          if (c2 + 1 == row) t.y = 1.0;
        }
      }
      v[q] = t;
    }
#pragma unroll
    for (int q = 0; q < DP; ++q) {
      const int p = q * 256 + tid;
      const int row = p / HP, c2 = (p % HP) * 2;
      if (c2 < 16 * ((row >> 4) + 1)) *reinterpret_cast<double2_t *>(&M[LY::off(row, c2)]) = v[q];
    }
    // ... and the own rows
    constexpr int OP = OWN * HP / 256;
    double2_t u[OP];
#pragma unroll
    for (int q = 0; q < OP; ++q) {
      const int p = q * 256 + tid;
      const int row = p / HP, c2 = (p % HP) * 2;
      const int r = rbase + row;
      double2_t t = (double2_t){0.0, 0.0};
      if (r < nrows) {
        const double *src = K + (int64_t)r * ldk + c0 + c2;
        if (c2 + 1 < nb) t = ld_f64x2(src, coh);
        else if (c2 < nb) t.x = ld_f64(src, coh);
      }
      u[q] = t;
    }
#pragma unroll
    for (int q = 0; q < OP; ++q) {
      const int p = q * 256 + tid;
      *reinterpret_cast<double2_t *>(&M[LY::off(PW + p / HP, (p % HP) * 2)]) = u[q];
    }
  }
  __syncthreads();

  for (int sb = 0; sb < CT; ++sb) {
    const int cb = sb * 16;
    // ---- (a) 16 x 16 diagonal tile: wavefront 0, lane (l & 15) <-> row (every 16-lane row
    // of the wavefront holds the same tile, so row_newbcast works in all four)
    double tl[16];  // the factored tile's rows, kept in registers for (b)
    if (wave == 0 && !(skip & 1)) {
#pragma unroll
      for (int k = 0; k < 16; ++k) tl[k] = M[LY::off(cb + l15, cb + k)];
      double d_mine = 1.0, di_mine = 1.0;
      int bad_any = 0;
      tile_factor<0>(tl, l15, cb, nb, d_mine, di_mine, bad_any);
      if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < lane) M[LY::off(cb + lane, cb + k)] = tl[k];
        M[LY::off(cb + lane, cb + lane)] = d_mine;
        dD[cb + lane] = d_mine;
        dI[cb + lane] = di_mine;
        if (lane == 0 && bad_any) s_bad = 1;
      }
    }
    __syncthreads();
    // ---- (b) rows below the tile: substitution, one lane per row; L_bb comes from the
    // tile registers of the lane's own 16-lane row through row_newbcast (wavefront 0 still
    // has them from (a); the others reload the factored tile from LDS)
    {
      const int row = cb + 16 + wave * 64 + lane;
      const bool wave_has_rows = (cb + 16 + wave * 64) < R;  // wave-uniform
      if (wave_has_rows && !(skip & 2)) {
        if (wave != 0) {
#pragma unroll
          for (int k = 0; k < 16; ++k) tl[k] = M[LY::off(cb + l15, cb + k)];
        }
        const int rowc = min(row, R - 1);  // lanes past the stack compute on a valid row
        double x[16];
        const double *xr = &M[LY::off(rowc, cb)];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          const double2_t v = *reinterpret_cast<const double2_t *>(xr + k);
          x[k] = v.x;
          x[k + 1] = v.y;
        }
        subst_rows<0>(x, tl);
        if (row < R) {
        double *xw = &M[LY::off(row, cb)];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          double2_t w, l;
          w.x = x[k];
          w.y = x[k + 1];
          l.x = x[k] * dI[cb + k];
          l.y = x[k + 1] * dI[cb + k + 1];
          *reinterpret_cast<double2_t *>(&Wt[row][k]) = w;
          *reinterpret_cast<double2_t *>(xw + k) = l;
        }
        if (row >= PW) {
          const int r = rbase + row - PW;
          if (r < nrows) {
            double *wp = W + (int64_t)r * ldw + wofs + cb;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
              double2_t w;
              w.x = x[k];
              w.y = x[k + 1];
              st_f64x2(wp + k, w, coh);
            }
          }
        }
        }
      }
    }
    __syncthreads();
    // ---- (c) tiles to the right: M[ti][tj] -= W[ti] L[tj]^T, tj in (sb, CT), ti in [tj, RT)
    if (sb + 1 < CT && !(skip & 4)) {
      int total = 0;
      for (int tj = sb + 1; tj < CT; ++tj) total += RT - tj;
      for (int e0 = wave; e0 < total; e0 += 4) {
        int e = e0, tj = sb + 1;
        while (e >= RT - tj) {
          e -= RT - tj;
          ++tj;
        }
        const int ti = tj + e;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = M[LY::off(ti * 16 + l4 + 4 * r, tj * 16 + l15)];
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
          const double av = -Wt[ti * 16 + l15][ks + l4];
          const double bv = M[LY::off(tj * 16 + l15, cb + ks + l4)];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) M[LY::off(ti * 16 + l4 + 4 * r, tj * 16 + l15)] = acc[r];
      }
    }
    __syncthreads();
  }

  // ---- write back: own rows (L), and by workgroup 0 the factored diagonal block
  {
    constexpr int HP = PW / 2;
    for (int p = tid; p < OWN * HP; p += 256) {
      const int row = p / HP, c2 = (p % HP) * 2;
      const int r = rbase + row;
      if (r < nrows) {
        const double2_t v = *reinterpret_cast<const double2_t *>(&M[LY::off(PW + row, c2)]);
        double *dst = K + (int64_t)r * ldk + c0 + c2;
        if (c2 + 1 < nb) st_f64x2(dst, v, coh);
        else if (c2 < nb) st_f64(dst, v.x, coh);
      }
    }
  }
  if (wg == 0) {
    for (int p = tid; p < PW * PW; p += 256) {
      const int row = p / PW, c = p % PW;
      if (row < nb && c <= row) st_f64(K + (int64_t)(c0 + row) * ldk + c0 + c, M[LY::off(row, c)], coh);
    }
    for (int i = tid; i < nb; i += 256) {
      st_f64(dvec + c0 + i, dD[i], coh);
      st_f64(dinv + c0 + i, dI[i], coh);
    }
    if (wave == 0) {
      int neg = 0;
      for (int i = lane; i < nb; i += 64) neg += (dD[i] < 0.0) ? 1 : 0;
      for (int o = 32; o > 0; o >>= 1) neg += __shfl_down(neg, o);
      if (lane == 0) {
        if (s_bad) atomicOr(&flags[0], 1);
        if (neg) atomicAdd(&flags[1], neg);
      }
    }
  }
}

template <int PW, int OWN>
__global__ __launch_bounds__(256) void k_ldlt_panel2(double *__restrict__ K, int64_t ldk,
                                                     double *__restrict__ W, int64_t ldw, int wofs,
                                                     int N, int nrows, int c0,
                                                     double *__restrict__ dvec,
                                                     double *__restrict__ dinv,
                                                     int *__restrict__ flags, int skip) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PanelLayout<PW, OWN>::SMEM];
  panel_body2<PW, OWN>(smem, blockIdx.x, K, ldk, W, ldw, wofs, N, nrows, c0, dvec, dinv, flags,
                       skip);
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

// Panel launch that also carries "filler" workgroups: 128 x 128 tiles of the PREVIOUS outer
// block's bulk trailing update (columns beyond the current outer block), which touch
// nothing the panel reads or writes.  This is the look-ahead of the blocked
// factorisation, obtained inside one queue: the CUs the panel cannot use (it has ~N/64
// workgroups and is latency-bound) run MFMA tiles instead of idling.  No data is handed
// between workgroups of the launch, so ordinary kernel-boundary visibility suffices.
// Tiles are numbered over the lower triangle of the region: t -> (ti, tj), tj <= ti.
template <int NB>
__global__ __launch_bounds__(256, 2) void k_ldlt_panel_fused(
    double *__restrict__ K, int64_t ldk, double *__restrict__ W, int64_t ldw, int wofs, int N,
    int nrows, int c0, double *__restrict__ dvec, double *__restrict__ dinv,
    int *__restrict__ flags, int skip, int n_panel_wg, const double *__restrict__ Wprev,
    int64_t ldwp, int ureg0, int ukc0, int uKB, int tile_start) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PNL_SMEM];
  if ((int)blockIdx.x < n_panel_wg) {
    panel_body<NB>(smem, blockIdx.x, K, ldk, W, ldw, wofs, N, nrows, c0, dvec, dinv, flags, skip);
  } else {
    const int t = tile_start + (int)blockIdx.x - n_panel_wg;
    int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    update_tile<UPD_BM, UPD_BM, 16>(smem, ureg0 + ti * UPD_BM, ureg0 + tj * UPD_BM, K, ldk, Wprev,
                                    ldwp, N, nrows, N, ukc0, uKB);
  }
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
// products with the already solved blocks AS THEY BECOME AVAILABLE -- every block publishes
// its 64 solution entries and then an epoch stamp, consumers poll the stamp -- and finishes
// with the mat-vec by its inverted diagonal block.  The off-diagonal blocks are final data, so
// they are fetched before the stamp they wait for.
//  * Hand-over protocol: the workers are the workgroups with id % 8 == 0, which the
//    round-robin dispatch puts on ONE XCD; solution entries and stamps are written and read
//    with L1-bypassing (agent-scope relaxed atomic) accesses and therefore meet in that XCD's
//    L2 -- no L2 write-back / invalidate per hand-over (with agent-scope release / acquire
//    fences a hand-over costs 11 us instead of 2.5 us and the chain is slower than the
//    launches it replaces).  The placement is CHECKED, not assumed: every worker reads its
//    XCC id (s_getreg HW_REG_XCC_ID) and compares it with the other workers' through one
//    atomicMax; a mismatch or a timed-out wait sets ctl[1], which the host reads at the next
//    synchronisation: the call fails loudly and the chained solves are switched off.
//  * Termination: polls are bounded, every workgroup walks a finite loop, and a workgroup only
//    waits for workers with a SMALLER index, which are dispatched before it.
#define CHAIN_SPIN_LIMIT (1 << 18)  // ~0.15 s per wait; a real hand-over takes microseconds

__device__ __forceinline__ void chain_wait(const int *stamp, int epoch, int *ctl) {
  for (int it = 0; it < CHAIN_SPIN_LIMIT; ++it) {
    if (__hip_atomic_load(stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return;
    __builtin_amdgcn_s_sleep(1);
  }
  atomicOr(&ctl[1], 1);
}

// all workers of a launch must sit on one XCD: ctl[0] = max over workers of (epoch << 4 | xcc)
__device__ __forceinline__ void chain_check_xcc(int epoch, int *ctl) {
  if (threadIdx.x == 0) {
    // 27 bits of epoch: after 2^27 solves the slot stops changing and the check goes quiet
    // (no false alarms); the stamps themselves compare full 32-bit epochs
    const int ep = epoch & 0x7ffffff;
    const int mine = (ep << 4) | (int)(__builtin_amdgcn_s_getreg(6164) & 15);  // XCC_ID[3:0]
    const int old = atomicMax(&ctl[0], mine);
    if ((old >> 4) == ep && old != mine) atomicOr(&ctl[1], 2);
  }
}

// backward: L^T x = z.  x_b = inv(L_bb)^T (z_b - sum_{a > b} L_ab^T x_a).  In place (x == z) ok.
__global__ __launch_bounds__(256) void k_trsv_bwd_chain(const double *__restrict__ K, int64_t ldk,
                                                        const double *__restrict__ Linv,
                                                        const double *z, double *x, int N,
                                                        int *__restrict__ stamps, int epoch,
                                                        int *__restrict__ ctl) {
  if (blockIdx.x & 7) return;
  chain_check_xcc(epoch, ctl);
  __shared__ double part[4][64];
  __shared__ double rs[64];
  const int nblk = (N + 63) / 64;
  const int b = nblk - 1 - (int)(blockIdx.x >> 3);  // last block first
  const int b0 = b * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // inverse of the diagonal block, rows [16 wave, 16 wave + 16): inv[j][lane]
  double iv[16];
  const double *ip = Linv + (size_t)b * 4096 + (size_t)(16 * wave) * 64 + lane;
#pragma unroll
  for (int t = 0; t < 16; ++t) iv[t] = ip[t * 64];
  const double zb = (wave == 0 && b0 + lane < N) ? z[b0 + lane] : 0.0;
  // column `lane` of L_ab, rows [16 wave, 16 wave + 16), for a = nblk - 1 ... b + 1
  auto fetch = [&](int a, double (&lv)[16]) {
    const int r0 = a * 64 + 16 * wave;
    const double *cp = K + (int64_t)r0 * ldk + b0 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) lv[t] = (r0 + t < N) ? cp[(int64_t)t * ldk] : 0.0;
  };
  double acc = 0.0;
  double cur[16], nxt[16];
  int a = nblk - 1;
  if (a > b) fetch(a, cur);
  for (; a > b; --a) {
    if (a - 1 > b) fetch(a - 1, nxt);
    chain_wait(stamps + a, epoch, ctl);
    const int xr0 = a * 64 + 16 * wave;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const double xv = (xr0 + t < N) ? __hip_atomic_load(x + xr0 + t, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                      : 0.0;
      acc = fma(cur[t], xv, acc);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[t] = nxt[t];
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) rs[lane] = zb - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
  __syncthreads();
  // x_i = sum_j inv[j][i] r_j  (j >= i; the stored inverse is zero above the diagonal)
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < 16; ++t) s = fma(iv[t], rs[16 * wave + t], s);
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const double xv = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (b0 + lane < N)
      __hip_atomic_store(x + b0 + lane, xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0)
      __hip_atomic_store(stamps + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// forward: L y = z.  y_b = inv(L_bb) (z_b - sum_{a < b} L_ba y_a).  In place ok.
__global__ __launch_bounds__(256) void k_trsv_fwd_chain(const double *__restrict__ K, int64_t ldk,
                                                        const double *__restrict__ LinvT,
                                                        const double *z, double *x, int N,
                                                        int *__restrict__ stamps, int epoch,
                                                        int *__restrict__ ctl) {
  if (blockIdx.x & 7) return;
  chain_check_xcc(epoch, ctl);
  __shared__ double part[4][64];
  __shared__ double rs[64];
  const int b = (int)(blockIdx.x >> 3);  // first block first
  const int b0 = b * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // LinvT[j][i] = inv[i][j]: rows j in [16 wave, 16 wave + 16), column i = lane
  double iv[16];
  const double *ip = LinvT + (size_t)b * 4096 + (size_t)(16 * wave) * 64 + lane;
#pragma unroll
  for (int t = 0; t < 16; ++t) iv[t] = ip[t * 64];
  const double zb = (wave == 0 && b0 + lane < N) ? z[b0 + lane] : 0.0;
  // lane <-> ROW b0 + lane; wavefront w takes columns [16 w, 16 w + 16) of every block a: 128
  // contiguous bytes per lane (one cache line each, not coalesced across lanes), so that the
  // multiplier x_a[t] is wavefront-uniform and no cross-lane reduction sits on the hand-over
  // path (a coalesced layout with 16 row sums over the lanes made the forward chain 2x slower
  // than the backward one)
  const bool row_ok = b0 + lane < N;
  auto fetch = [&](int a, double (&lv)[16]) {
    const double *rp = K + (int64_t)(b0 + lane) * ldk + a * 64 + 16 * wave;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      double2_t v = (double2_t){0.0, 0.0};
      if (row_ok) v = *reinterpret_cast<const double2_t *>(rp + t);
      lv[t] = v.x;
      lv[t + 1] = v.y;
    }
  };
  double acc = 0.0;
  double cur[16], nxt[16];
  if (b > 0) fetch(0, cur);
  for (int a = 0; a < b; ++a) {
    if (a + 1 < b) fetch(a + 1, nxt);
    chain_wait(stamps + a, epoch, ctl);
    const int xr0 = a * 64 + 16 * wave;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const double xv = __hip_atomic_load(x + xr0 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc = fma(cur[t], xv, acc);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[t] = nxt[t];
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) rs[lane] = zb - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
  __syncthreads();
  // y_i = sum_j inv[i][j] r_j
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < 16; ++t) s = fma(iv[t], rs[16 * wave + t], s);
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const double xv = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (b0 + lane < N)
      __hip_atomic_store(x + b0 + lane, xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0)
      __hip_atomic_store(stamps + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
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
__device__ __forceinline__ bool batch_decode(int B, int per, int &inst, int &t) {
  const int id = blockIdx.x;
  const int slot = id >> 3;
  const int il = slot / per;
  t = slot - il * per;
  inst = il * 8 + (id & 7);
  return inst < B;
}
static inline int batch_grid(int B, int per) { return 8 * ((B + 7) / 8) * per; }

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
  update_tile<64, 64, 16>(smem, i0, j0, I.K, I.ldk, I.W + (int64_t)wbuf * I.wstride + wcol, ldw, N,
                          nrows, colEnd, kc0, KB);
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
  // PGF_FINEGRAINED=1 (diagnostic): K and W in fine-grained device memory, i.e. coherent
  // across the XCDs' L2s without kernel-boundary cache maintenance
  const bool fg = getenv("PGF_FINEGRAINED") != nullptr;
  auto dmalloc = [&](double **ptr, size_t bytes) {
    return fg ? hipExtMallocWithFlags((void **)ptr, bytes, hipDeviceMallocFinegrained)
              : hipMalloc((void **)ptr, bytes);
  };
  if ((e = dmalloc(&f.K, rows * f.ldk * sizeof(double))) != hipSuccess) return e;
  f.OB = 256;
  if (const char *ob = getenv("PGF_OB")) {
    const int v = atoi(ob);
    if (v == 64 || v == 128 || v == 192 || v == 256) f.OB = v;
  }
  f.wstride = rows * (size_t)f.OB;
  if ((e = dmalloc(&f.W, 2 * f.wstride * sizeof(double))) != hipSuccess) return e;
  if ((e = hipStreamCreateWithFlags(&f.stream2, hipStreamNonBlocking)) != hipSuccess) return e;
  if ((e = hipEventCreateWithFlags(&f.ev_panel, hipEventDisableTiming)) != hipSuccess) return e;
  if ((e = hipEventCreateWithFlags(&f.ev_update, hipEventDisableTiming)) != hipSuccess) return e;
  if ((e = hipMalloc(&f.dvec, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.dinv, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.zwork, rows * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.Linv, (rows / 64 + 1) * 4096 * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.LinvT, (rows / 64 + 1) * 4096 * sizeof(double))) != hipSuccess) return e;
  if ((e = hipMalloc(&f.flags, 4 * sizeof(int))) != hipSuccess) return e;
  f.chain_stride = (int)(rows / 64 + 2);
  if ((e = hipMalloc(&f.chain, (2 * f.chain_stride + 4) * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMemset(f.chain, 0, (2 * f.chain_stride + 4) * sizeof(int))) != hipSuccess) return e;
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
  if (f.h_flags) (void)hipHostFree(f.h_flags);
  if (f.ev_panel) (void)hipEventDestroy(f.ev_panel);
  if (f.ev_update) (void)hipEventDestroy(f.ev_update);
  for (hipEvent_t ev : f.ev_ring) (void)hipEventDestroy(ev);
  if (f.stream2) (void)hipStreamDestroy(f.stream2);
  f = DenseLdlt();
}

static hipEvent_t prof_event(PgfProfile *p) {
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
static void launch_update(DenseLdlt &f, hipStream_t s, const double *Wp, int64_t ldw, int N,
                          int nrows, int row0, int col0, int colEnd, int kc0, int KB,
                          PgfProfile *p, int coh = 0) {
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
  hipLaunchKernelGGL((k_ldlt_update<64, 64, 16>), dim3(tc, tr), dim3(256), 0, s, f.K, f.ldk, Wp,
                     ldw, N, nrows, row0, col0, colEnd, kc0, KB | (coh << 20));
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
  }
}

// ------------------------------------------------------------------ overlapped schedule
// Look-ahead inside ONE queue.  A kernel launched with hipExtAnyOrderLaunch does not wait for
// the packets before it (tools/launch_gap_test.hip: it really runs unordered on this stack),
// so after every panel launch of outer block i + 1 a PIECE of outer block i's bulk trailing
// update is launched any-order and runs beside that panel:
//     U_next(i)                      columns of block i + 1, normal launch
//     P0(i+1) | piece 0 of B_i       P0 normal, the piece any-order
//     P1(i+1) | piece 1 of B_i       P1's barrier waits for P0 AND piece 0, then both start
//     ...
//     fence kernel                   normal; its end-of-kernel release writes the last
//                                    piece's C tiles back before the next block reads them
// The pieces touch only columns >= nextEnd, the panels of block i + 1 only their own columns
// and the other W buffer, so a pair never shares data.  Everything a piece reads was written
// at least two ordinary kernel boundaries earlier; only the panels, which hand data from one
// to the next across XCDs, keep the ordinary (fenced) launch.
__global__ void k_fence() {}

static void launch_piece(DenseLdlt &f, hipStream_t s, const double *Wp, int64_t ldw, int N,
                         int row_lo, int row_hi, int col0, int kc0, int KB) {
  if (row_lo >= row_hi || col0 >= N) return;
  const int tr = (row_hi - row_lo + 63) / 64;
  const int tc = (std::min(N, row_hi) - col0 + 63) / 64;
  if (tc <= 0) return;
  hipExtLaunchKernelGGL((k_ldlt_update<64, 64, 16>), dim3(tc, tr), dim3(256), 0, s, nullptr,
                        nullptr, hipExtAnyOrderLaunch, f.K, f.ldk, Wp, ldw, N, row_hi, row_lo, col0,
                        N, kc0, KB);
}

static void factor_overlapped(DenseLdlt &f, int N, int nrows) {
  // Per outer block i (bulk update B_i = strip A: the next block's columns, + far region):
  //   A (ordinary) | B1 (any-order)   the strip alone under-fills the chip; B1 = the first
  //                                    ~FILL tiles of the far region runs beside it
  //   P0 (ordinary: waits for A, B1) | B2 (any-order) = the rest of the far region
  //   P1 (ordinary: waits for P0, B2), P2, P3
  // Ordinary launches end with a cache release, and each one here starts only after the
  // any-order piece before it has completed, so every piece's C tiles are written back
  // before anything reads them.
  hipStream_t s = f.stream;
  const int OB = f.OB;
  const int FILL = getenv("PGF_OVERLAP_FILL") ? atoi(getenv("PGF_OVERLAP_FILL")) : 1000;
  struct {
    bool active = false;
    const double *Wp = nullptr;
    int reg0 = 0, kc0 = 0, KB = 0, r1 = 0;
  } pend;
  int buf = 0;
  for (int ob0 = 0; ob0 < N; ob0 += OB, buf ^= 1) {
    const int obEnd = std::min(ob0 + OB, N);
    double *Wb = f.W + (size_t)buf * f.wstride;
    int k = 0;
    for (int c0 = ob0; c0 < obEnd; c0 += PGF_NB, ++k) {
      const int below = nrows - std::min(c0 + PGF_NB, N);
      const int npw = std::max(1, (below + 63) / 64);
      hipLaunchKernelGGL(k_ldlt_panel_ll<PGF_NB>, dim3(npw), dim3(256), 0, s, f.K, f.ldk, Wb,
                         (int64_t)OB, ob0, N, nrows, c0, f.dvec, f.dinv, f.flags);
      if (k == 0 && pend.active) {  // B2 of the previous block beside this block's first panel
        launch_piece(f, s, pend.Wp, OB, N, pend.r1, nrows, pend.reg0, pend.kc0, pend.KB);
        pend.active = false;
      }
    }
    if (obEnd < N) {
      const int KB = obEnd - ob0;
      const int nextEnd = std::min(obEnd + OB, N);
      launch_update(f, s, Wb, OB, N, nrows, obEnd, obEnd, nextEnd, ob0, KB, nullptr, 0);  // A
      if (nextEnd < N) {
        // far region: rows / columns >= nextEnd; B1 = its first tile rows holding ~FILL tiles
        const int TR = (nrows - nextEnd + 63) / 64;
        int t1 = (int)std::ceil((std::sqrt(8.0 * FILL + 1.0) - 1.0) * 0.5);
        t1 = std::max(1, std::min(TR, t1));
        const int r1 = std::min(nrows, nextEnd + t1 * 64);
        launch_piece(f, s, Wb, OB, N, nextEnd, r1, nextEnd, ob0, KB);  // B1
        if (r1 < nrows) {
          pend.active = true;
          pend.Wp = Wb;
          pend.reg0 = nextEnd;
          pend.kc0 = ob0;
          pend.KB = KB;
          pend.r1 = r1;
        }
      }
    }
  }
}

// Two-level factorisation, the schedule of one call (default path, one queue):
//   per outer block of f.OB = 256 columns:
//     4 x k_ldlt_panel_ll   64-column panels, left-looking inside the block: a panel's
//                           prologue applies the earlier panels of the block to the two tiles
//                           it needs, so nothing is launched between them
//     1 x k_ldlt_update     bulk right-looking update of everything to the right, K = 256
//   k_inv_diag_blocks       inverses of the 64 x 64 diagonal blocks for the solves
// W (= L D of the current outer block, OB columns wide) is double buffered: the experimental
// overlapped / look-ahead schedules still read block i's W while block i + 1 writes its own.
// The branches behind PGF_LOOKAHEAD, PGF_FUSE, PGF_OVERLAP, PGF_PANEL_LL=0, PGF_PW and
// PGF_PANEL2 are the experiments DESIGN.md describes; none of them is the default.
hipError_t ldlt_factor_async(DenseLdlt &f, int N, int nrows) {
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
  // PGF_OVERLAP=1: any-order look-ahead inside the one queue (factor_overlapped); the
  // instrumented pass (events around every update launch) keeps the plain schedule
  if (getenv("PGF_OVERLAP") && atoi(getenv("PGF_OVERLAP")) != 0 && !p &&
      !getenv("PGF_LOOKAHEAD") && !getenv("PGF_FUSE")) {
    factor_overlapped(f, N, nrows);
    if (N > 0)
      hipLaunchKernelGGL(k_inv_diag_blocks, dim3((N + 63) / 64), dim3(64), 0, sA, f.K, f.ldk, N,
                         f.Linv, f.LinvT);
    e = hipMemcpyAsync(f.h_flags, f.flags, 2 * sizeof(int), hipMemcpyDeviceToHost, sA);
    if (e != hipSuccess) return e;
    return hipGetLastError();
  }
  const int OB = f.OB;
  // PGF_LOOKAHEAD=1: bulk update of outer block i on a second stream, overlapped with the
  // panels of block i + 1; PGF_COHERENT (default 1 with look-ahead): stream-A kernels and
  // (bit 1) the bulk kernel load with system scope, see ld_f64
  const bool la = getenv("PGF_LOOKAHEAD") != nullptr;
  const int cohm = getenv("PGF_COHERENT") ? atoi(getenv("PGF_COHERENT")) : 0;
  const int cohA = cohm & 1, cohB = (cohm >> 1) & 1;
  // PGF_LA_ONEQ: the look-ahead launch sequence on ONE queue (separates schedule logic from
  // multi-queue effects)
  hipStream_t sB = (la && !getenv("PGF_LA_ONEQ")) ? f.stream2 : f.stream;
  int evi = 0;
  auto next_event = [&]() -> hipEvent_t {
    if ((size_t)evi >= f.ev_ring.size()) {
      hipEvent_t ev;
      (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      f.ev_ring.push_back(ev);
    }
    return f.ev_ring[evi++];
  };
  hipEvent_t ev_b_done = nullptr;
  bool b_pending = false;
  if (la) {  // stream B must not start before everything already queued on A
    hipEvent_t ev0 = next_event();
    (void)hipEventRecord(ev0, sA);
    (void)hipStreamWaitEvent(sB, ev0, 0);
  }
  const int la_dbg = getenv("PGF_LA_DEBUG") ? atoi(getenv("PGF_LA_DEBUG")) : 0;
  // panel width: 128 (wide kernel, 32 own rows per workgroup) or 64
  const int pw_env = getenv("PGF_PW") ? atoi(getenv("PGF_PW")) : 64;
  const bool pnl2 = getenv("PGF_PANEL2") != nullptr;  // 64-wide panel through the new body
  const int PWh = (pw_env == 128 && OB % 128 == 0 && getenv("PGF_FUSE") == nullptr) ? 128 : 64;
  const int OWNh = (PWh == 128) ? 32 : 64;
  // left-looking panels inside the outer block (no K = 64 update launches): the default on the
  // single-queue schedule; PGF_PANEL_LL=0 brings the separate inner updates back
  const bool pnl_ll = !(getenv("PGF_PANEL_LL") && atoi(getenv("PGF_PANEL_LL")) == 0) && !la &&
                      getenv("PGF_FUSE") == nullptr && PWh == 64 && !pnl2;
  // With a second queue active, consecutive kernels of ONE stream were observed to overlap
  // (the last workgroups of an inner update still running when the next panel started:
  // wrong factors, periodic in 8 workgroups).  An explicit record + wait on the same stream
  // between dependent launches restores the in-order semantics; without look-ahead (one
  // queue) it is not needed.
  auto fence_on = [&](hipStream_t st) {
    hipEvent_t ev = next_event();
    (void)hipEventRecord(ev, st);
    (void)hipStreamWaitEvent(st, ev, 0);
  };
  // PGF_LA_DEBUG: 9 = no extra fences at all, 10 = only at the cross-queue edges
  auto self_fence = [&]() {
    if (la && la_dbg != 9 && la_dbg != 10) fence_on(sA);
  };
  auto edge_fence = [&](hipStream_t st) {
    if (la && la_dbg == 10) fence_on(st);
  };
  const int skip = (getenv("PGF_SKIP") ? atoi(getenv("PGF_SKIP")) : 0) | (cohA << 3);
  // filler tiles in the panel launches: correct, but not yet a win (the panel's 87 KB of
  // LDS leaves one filler workgroup per CU, and the tile kernel needs several per SIMD)
  const bool fuse = getenv("PGF_FUSE") != nullptr;
  // pending bulk update (previous outer block): region [reg0, nrows) x [reg0, N)
  struct {
    bool active = false;
    const double *Wp = nullptr;
    int reg0 = 0, kc0 = 0, KB = 0, total = 0, done = 0;
  } pend;
  int buf = 0;
  for (int ob0 = 0; ob0 < N; ob0 += OB, buf ^= 1) {
    const int obEnd = std::min(ob0 + OB, N);
    double *Wb = f.W + (size_t)buf * f.wstride;
    const int npanels = (obEnd - ob0 + PWh - 1) / PWh;
    int k = 0;
    for (int c0 = ob0; c0 < obEnd; c0 += PWh, ++k) {
      const int below = nrows - std::min(c0 + PWh, N);
      const int npw = std::max(1, (below + OWNh - 1) / OWNh);
      const int remaining = pend.active ? pend.total - pend.done : 0;
      if (remaining > 0) {
        const int share = (remaining + (npanels - k) - 1) / (npanels - k);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (p) {
          e0 = prof_event(p);
          e1 = prof_event(p);
          (void)hipEventRecord(e0, sA);
        }
        hipLaunchKernelGGL(k_ldlt_panel_fused<PGF_NB>, dim3(npw + share), dim3(256), 0, sA, f.K,
                           f.ldk, Wb, (int64_t)OB, c0 - ob0, N, nrows, c0, f.dvec, f.dinv, f.flags,
                           skip, npw, pend.Wp, (int64_t)OB, pend.reg0, pend.kc0, pend.KB,
                           pend.done);
        if (p) {
          (void)hipEventRecord(e1, sA);
          p->update_spans.emplace_back(e0, e1);
          // algorithmic flops of the filler tiles of this launch (lower triangle only)
          double cnt = 0.0;
          for (int t = pend.done; t < pend.done + share; ++t) {
            int ti = (int)((std::sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            while (ti * (ti + 1) / 2 > t) --ti;
            const int tj = t - ti * (ti + 1) / 2;
            const int r0 = pend.reg0 + ti * UPD_BM, c0t = pend.reg0 + tj * UPD_BM;
            const int r1 = std::min(r0 + UPD_BM, nrows), c1t = std::min(c0t + UPD_BM, N);
            for (int i = r0; i < r1; ++i) {
              const int top = std::min(c1t, i + 1);
              if (top > c0t) cnt += top - c0t;
            }
          }
          p->update_flops.push_back(2.0 * cnt * pend.KB);
        }
        pend.done += share;
      } else if (PWh == 128) {
        hipLaunchKernelGGL((k_ldlt_panel2<128, 32>), dim3(npw), dim3(256), 0, sA, f.K, f.ldk, Wb,
                           (int64_t)OB, c0 - ob0, N, nrows, c0, f.dvec, f.dinv, f.flags, skip);
      } else if (pnl2) {
        hipLaunchKernelGGL((k_ldlt_panel2<64, 64>), dim3(npw), dim3(256), 0, sA, f.K, f.ldk, Wb,
                           (int64_t)OB, c0 - ob0, N, nrows, c0, f.dvec, f.dinv, f.flags, skip);
      } else if (pnl_ll) {
        hipLaunchKernelGGL(k_ldlt_panel_ll<PGF_NB>, dim3(npw), dim3(256), 0, sA, f.K, f.ldk, Wb,
                           (int64_t)OB, ob0, N, nrows, c0, f.dvec, f.dinv, f.flags);
        continue;  // no inner updates: the next panel's prologue applies this one
      } else {
        hipLaunchKernelGGL(k_ldlt_panel<PGF_NB>, dim3(npw), dim3(256), 0, sA, f.K, f.ldk, Wb,
                           (int64_t)OB, c0 - ob0, N, nrows, c0, f.dvec, f.dinv, f.flags, skip);
      }
      const int c1 = c0 + PWh;
      self_fence();
      if (c1 < obEnd)  // inner update: the rest of this outer block's columns, K = panel width
        launch_update(f, sA, Wb + (c0 - ob0), OB, N, nrows, c1, c1, obEnd, c0, PWh, p, cohA);
      self_fence();
    }
    pend.active = false;
    if (obEnd < N) {
      const int KB = obEnd - ob0;  // == OB here (only the last outer block may be short)
      const int nextEnd = std::min(obEnd + OB, N);
      if (!la && !fuse) {
        // one queue, no look-ahead: nothing is gained by splitting off the next block's
        // columns -- update the whole trailing matrix in one (large, efficient) launch
        launch_update(f, sA, Wb, OB, N, nrows, obEnd, obEnd, N, ob0, KB, p, 0);
        continue;
      }
      if (la && b_pending) {
        (void)hipStreamWaitEvent(sA, ev_b_done, 0);  // RMW order
        edge_fence(sA);
      }
      launch_update(f, sA, Wb, OB, N, nrows, obEnd, obEnd, nextEnd, ob0, KB, p, cohA);
      self_fence();
      if (nextEnd < N) {
        if (la) {
          hipEvent_t ev_a = next_event();
          (void)hipEventRecord(ev_a, sA);
          (void)hipStreamWaitEvent(sB, ev_a, 0);
          edge_fence(sB);
          launch_update(f, sB, Wb, OB, N, nrows, nextEnd, nextEnd, N, ob0, KB, p, cohB);
          ev_b_done = next_event();
          (void)hipEventRecord(ev_b_done, sB);
          b_pending = true;
        } else if (fuse) {
          const int TR = (nrows - nextEnd + UPD_BM - 1) / UPD_BM;
          pend.active = true;
          pend.Wp = Wb;
          pend.reg0 = nextEnd;
          pend.kc0 = ob0;
          pend.KB = KB;
          pend.total = TR * (TR + 1) / 2;
          pend.done = 0;
        } else {
          launch_update(f, sA, Wb, OB, N, nrows, nextEnd, nextEnd, N, ob0, KB, p);
        }
      }
    }
  }
  if (la && b_pending) {
    (void)hipStreamWaitEvent(sA, ev_b_done, 0);
    edge_fence(sA);
  }
  self_fence();
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
  f.n_neg = f.h_flags[1];
  f.factored = (f.h_flags[0] == 0);
  if (f.h_flags[0]) return 1;
  return ldlt_chain_check(f) ? 2 : 0;
}

// PGF_TRSV_CHAIN=0: one launch per 256-row super-block (the earlier scheme) instead of the
// chained single-launch solves; also switched off for the rest of the process when a chained
// solve reports a placement / timeout problem (ldlt_chain_check)
static bool g_chain_off = false;
static bool use_chain() {
  static const bool on = !(getenv("PGF_TRSV_CHAIN") && atoi(getenv("PGF_TRSV_CHAIN")) == 0);
  return on && !g_chain_off;
}

// after a host synchronisation of f.stream: did a chained solve since the last check fail its
// own checks?  (bit 0: a wait timed out, bit 1: workers on different XCDs)
int ldlt_chain_check(DenseLdlt &f) {
  if (!f.chain || !f.h_flags) return 0;  // banded handles have no dense factor
  const int bad = f.h_flags[3];
  if (!bad) return 0;
  f.h_flags[3] = 0;
  g_chain_off = true;
  (void)hipMemsetAsync(f.chain + 2 * f.chain_stride + 1, 0, sizeof(int), f.stream);
  return bad;
}

static hipError_t chain_report(DenseLdlt &f) {
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
                       N, f.chain, ++f.chain_epoch, f.chain + 2 * f.chain_stride);
    return chain_report(f);
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
                       sol, N, f.chain + f.chain_stride, ++f.chain_epoch,
                       f.chain + 2 * f.chain_stride);
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
void ldlt_batch_factor_async(hipStream_t s, const BInst *tab, int B, int Nmax, int m, int OB,
                             PgfProfile *p) {
  if (Nmax <= 0 || B <= 0) return;
  // PGF_BATCH_LL=0: the single-instance schedule with a batch dimension (fused panel
  // kernel + K = 64 inner updates); default: the left-looking split panel step
  const bool ll = !(getenv("PGF_BATCH_LL") && atoi(getenv("PGF_BATCH_LL")) == 0);
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

void ldlt_batch_solve_async(hipStream_t s, const BInst *tab, int B, int Nmax, int m,
                            bool any_unfactored_solve) {
  if (Nmax <= 0 || B <= 0) return;
  const dim3 gv((Nmax + 255) / 256, 1, B);
  if (any_unfactored_solve) {
    hipLaunchKernelGGL(kb_solve_prep_fwd, gv, dim3(256), 0, s, tab, m);
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
  // PGF_FINEGRAINED=1 (diagnostic): K and W in fine-grained device memory, i.e. coherent
  // across the XCDs' L2s without kernel-boundary cache maintenance
  const bool fg = getenv("PGF_FINEGRAINED") != nullptr;
  auto dmalloc = [&](double **ptr, size_t bytes) {
    return fg ? hipExtMallocWithFlags((void **)ptr, bytes, hipDeviceMallocFinegrained)
              : hipMalloc((void **)ptr, bytes);
  };
  if ((e = dmalloc(&f.K, rows * f.ldk * sizeof(double))) != hipSuccess) return e;
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
      case 1: PGF_LAUNCH_VARIANT(64, 64, 16, 2, 2, 2); break;
      case 2: PGF_LAUNCH_VARIANT(128, 128, 16, 2, 2, 2); break;
      case 3: PGF_LAUNCH_VARIANT(128, 128, 16, 4, 4, 2); break;
      case 4: PGF_LAUNCH_VARIANT(128, 128, 16, 2, 4, 2); break;
      case 5: PGF_LAUNCH_VARIANT(128, 64, 16, 2, 2, 2); break;
      case 6: PGF_LAUNCH_VARIANT(128, 64, 16, 4, 2, 2); break;
      case 7: PGF_LAUNCH_VARIANT(128, 128, 32, 2, 4, 2); break;
      case 8: PGF_LAUNCH_VARIANT(64, 128, 16, 2, 2, 2); break;
      case 9: PGF_LAUNCH_VARIANT(64, 64, 64, 2, 2, 0); break;
      case 10: PGF_LAUNCH_VARIANT(64, 64, 32, 2, 2, 0); break;
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
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(f.K);
  (void)hipFree(f.W);
  (void)hipStreamDestroy(s);
  return e;
}
