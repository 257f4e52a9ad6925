// Device helpers shared by the factorisation translation units (pgf_ldlt.hip,
// pgf_factor2.hip).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// plain loads / stores (the trailing int is the retired "coherent access" switch of the
// two-queue experiments; kept so that the call sites read the same)
__device__ __forceinline__ double ld_f64(const double *p, int) { return *p; }
__device__ __forceinline__ double2_t ld_f64x2(const double *p, int) {
  return *reinterpret_cast<const double2_t *>(p);
}
__device__ __forceinline__ void st_f64(double *p, double v, int) { *p = v; }
__device__ __forceinline__ void st_f64x2(double *p, double2_t v, int) {
  *reinterpret_cast<double2_t *>(p) = v;
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
// Batched kernels: workgroup id -> (instance, tile).  Consecutive ids go round-robin over the 8
// XCDs; instance i is pinned to XCD i % 8 (see the batched variants in pgf_ldlt.hip).
__device__ __forceinline__ bool batch_decode_id(int id, int B, int per, int &inst, int &t) {
  const int slot = id >> 3;
  const int il = slot / per;
  t = slot - il * per;
  inst = il * 8 + (id & 7);
  return inst < B;
}
__device__ __forceinline__ bool batch_decode(int B, int per, int &inst, int &t) {
  const int id = blockIdx.x;
  const int slot = id >> 3;
  const int il = slot / per;
  t = slot - il * per;
  inst = il * 8 + (id & 7);
  return inst < B;
}
static inline int batch_grid(int B, int per) { return 8 * ((B + 7) / 8) * per; }

#define UPD_BM 128

// BM x BN = tile (rows x columns); BK = K-chunk staged per barrier pair; 4 wavefronts as
// 2 x 2, each (BM/2) x (BN/2) = TM x TN MFMA tiles.  LDS rows are padded to BK + 2 doubles.
// SCALE: the A operand is L itself (W points into K) and is multiplied by D (dsc[k], k from the
// first column of the K-range) while it is staged: W = L D of ANY earlier block, not only of
// the one whose W panel is still in its buffer (the lazy schedule of pgf_factor2.hip).
// EXP (pgf_bench_update only; results are wrong by construction): 1 no global fetch after the
// first chunk, 2 no LDS operand reads, 4 no LDS stage writes, 8 no MFMA, 16 no barriers
template <int BM, int BN, int BK, int WR = 2, int WC = 2, int DB = 0, bool SCALE = false, int EXP = 0>
__device__ __forceinline__ void update_tile(unsigned char *smem, const int tid, const int i0,
                                            const int j0,
                                            double *__restrict__ K, int64_t ldk,
                                            const double *__restrict__ W, int64_t ldw, int N,
                                            int nrows, int colEnd, int kc0, int KBc,
                                            const double *__restrict__ dsc = nullptr,
                                            int KB1 = 1 << 30, const double *__restrict__ W2 = nullptr,
                                            int64_t ldw2 = 0, const double *__restrict__ dsc2 = nullptr) {
  const int KB = KBc;  // K-depth
  const int coh = 0;
  constexpr int NT = 64 * WR * WC;           // threads per workgroup
  constexpr int WM = BM / WR, WN = BN / WC;  // rows / columns per wavefront
  constexpr int TM = WM / 16, TN = WN / 16;  // MFMA tiles per wavefront
  constexpr int LD = BK + 2;
  constexpr int PPR = BK / 2;                // 16-byte pieces per row
  constexpr int PA = BM * PPR / NT, PB = BN * PPR / NT;
  static_assert(PA >= 1 && PB >= 1, "tile too small for the workgroup");
  constexpr int STAGE = (BM + BN) * LD * 8;  // bytes of one LDS stage (A then B)

  // tid: thread index inside the (64 WR WC)-thread group that owns this tile (the group may
  // be a slice of a larger workgroup: k_chain_update)
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
  // (the D pair of a thread's pieces is the same for all of them: NT is a multiple of PPR)
  static_assert(!SCALE || NT % PPR == 0, "one D pair per thread");
  double2_t pa[PA], pb[PB], pd = (double2_t){1.0, 1.0};
  bool staged_once = false;
  // SCALE: the K-range may consist of two segments -- chunks [0, KB1) from the panel (W, ldw, dsc),
  // chunks [KB1, KB) from (W2, ldw2, dsc2): a pre-eliminated block's panel V followed by columns
  // of K (pgf_factor2.hip; two jobs on the same tiles in one launch would race).  KB1 is a
  // multiple of BK; the choice is uniform over the workgroup.
  auto fetch = [&](int kk, int = 0) {
    if ((EXP & 1) && kk > 0) return;
    const double *__restrict__ Wc = W;
    const double *__restrict__ dc = dsc;
    int64_t ldc = ldw;
    int kq = kk;
    if (SCALE && kk >= KB1) {
      Wc = W2;
      dc = dsc2;
      ldc = ldw2;
      kq = kk - KB1;
    }
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int p = q * NT + tid;
      const int row = p / PPR, kofs = (p % PPR) * 2;
      // rows beyond the region are clamped, not predicated: what they contribute stays in
      // accumulator rows that are never stored, and a predicated load costs the prefetch a
      // branch with a full s_waitcnt at its join
      const int gi = min(i0 + row, nrows - 1);
      pa[q] = ld_f64x2(Wc + (int64_t)gi * ldc + kq + kofs, coh);
    }
    if (SCALE) pd = *reinterpret_cast<const double2_t *>(dc + kq + (tid % PPR) * 2);
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int p = q * NT + tid;
      const int row = p / PPR, kofs = (p % PPR) * 2;
      const int gj = min(j0 + row, colEnd - 1);
      // (SCALE: both operands are rows of the same panel -- K + kc0 for an ordinary block, the
      // pre-eliminated block's panel V for a virtual one)
      pb[q] = SCALE ? ld_f64x2(Wc + (int64_t)gj * ldc + kq + kofs, coh)
                    : ld_f64x2(K + (int64_t)gj * ldk + kc0 + kk + kofs, coh);
    }
  };
  auto stage = [&](int buf, int = 0) {
    if ((EXP & 4) && buf >= 0 && staged_once) return;
    staged_once = true;
    double(*As)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE);
    double(*Bs)[LD] = reinterpret_cast<double(*)[LD]>(smem + buf * STAGE + BM * LD * 8);
    // negate here, not at the fetch: touching the loaded value there would make the
    // wavefront wait for the prefetch before it starts the current chunk's MFMAs
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int p = q * NT + tid;
      *reinterpret_cast<double2_t *>(&As[p / PPR][(p % PPR) * 2]) = SCALE ? -pa[q] * pd : -pa[q];
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
      for (int t = 0; t < TM; ++t) a[t] = (EXP & 2) ? 1e-3 * (ks + t) : As[wr * WM + t * 16 + l15][ks + l4];
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = (EXP & 2) ? 1e-3 * (ks - t) : Bs[wc * WN + t * 16 + l15][ks + l4];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int nj = 0; nj < TN; ++nj) {
          if (EXP & 8) acc[mi][nj][0] += a[mi] + b[nj];
          else acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
        }
    }
  };

  fetch(0);
  if (DB) {
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
      if (!(EXP & 16)) __syncthreads();
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

