// The diagonal chain D(k), third design (default; replaces the LDS-panel chain of
// pgf_factor2.hip's chain_body, which stays behind PGF_CHAIN=2 as the cross-check).
//
// LDL^T of one 256 x 256 diagonal block of the reduced KKT matrix by ONE 16-wavefront
// workgroup: the factorisation's serial pivot chain (reference: SuperLU gstrf behind
// scipy.sparse.linalg.splu, pygradflow/linear_solver/lu_solver.py:14; inertia = negative pivots).
//
//   * The block lives in REGISTERS: its lower 16 x 16 tiles are dealt to wavefronts 1..15 (nine
//     each) as MFMA accumulators, transposed -- register r of lane (l15, l4) holds
//     T[l15][l4 + 4 r] -- so that an accumulator IS the B operand of the next product and nothing
//     is ever staged: LDS carries only the 16-column panel X = L D of the current step (32 KB),
//     the inverse of the pivot tile and the two tiles handed to wavefront 0.
//   * Wavefront 0 runs the serial chain and never waits for a barrier inside it.  Per step:
//       - the 16 x 16 pivot tile is eliminated with lane <-> row in every row of 16 lanes and DPP
//         row broadcasts (v_fmac_f64_dpp ... row_newbcast:k: ONE instruction per rank-1 update
//         entry instead of two v_readlane + FMA); the same row operations applied to the identity
//         give inv(L_kk) from the same instruction stream.  Pivot recurrence per column:
//         a[c+1] += bcast_c(1/d_c) * (-a[c] * a[c+1][c])  ->  rcp  ->  one Newton step.
//       - the panel tile right below, X^T = inv(L_kk) T^T, and with it the update of the NEXT
//         pivot tile: 2 x 4 MFMAs on operands the tile's owner parked in LDS early in the step;
//         the result goes through 2 KB of LDS from the MFMA layout back to lane <-> row.
//   * Everything else is MFMA work of the owners beside the next elimination: the rest of the
//     panel (behind a counter in LDS, not a barrier), the rank-16 update of every live tile --
//     the two tiles wavefront 0 needs next first --, the write-back of the panel and of the
//     factored pivot tile to global memory.  ONE workgroup barrier per step.
//   * inv of the unit-lower 64 x 64 diagonal tiles (the solves and T(k) multiply with them) grows
//     by one block row per step: X_pq = -inv(L_pp) sum_r L_pr X_rq, one wavefront per block.
//     No helper workgroups, no stamps in global memory, no placement requirement.
//
// Measured on MI355X (tools/): a v_fmac_f64_dpp issues every 4 cycles; the elimination of a
// 16 x 16 tile takes 1.45 us with v_readlane, 0.83 us with DPP broadcasts (chain_dpp_test);
// FP64 MFMAs of ANOTHER wavefront on the same SIMD slow the elimination 2-4 x (the FP64 pipe is
// shared and an MFMA holds it for 64 cycles; mfma_share_test), wavefronts are served oldest
// first (mfma_cu_test); an LDS store costs ~20 cycles per 8 bytes of a lane (lds_cost_test) --
// hence no barrier and as few LDS stores as possible on wavefront 0's path.
#pragma once

#include "pgf_ldlt_dev.h"

#define C3_PLD 18  // row stride of the 16 x 16 LDS tiles: conflict-free ds_read_b128 of rows
#define C3_LLD 66  // row stride of the 64 x 64 inverse workspace (= C_LD of pgf_factor2.hip)
#define C3_PT_OFF 0                                  // next pivot tile, MFMA layout -> rows [16][18]
#define C3_NEI 6                                     // inv(L_kk) of the last six steps
#define C3_EI_OFF (C3_PT_OFF + 16 * C3_PLD * 8)      // [C3_NEI][16][18]
#define C3_LK_OFF (C3_EI_OFF + C3_NEI * 16 * C3_PLD * 8)  // factored pivot tile [4][16][18]
#define C3_D_OFF (C3_LK_OFF + 4 * 16 * C3_PLD * 8)   // D[4][16] | 1/D[4][16] | counters
#define C3_ST_OFF (C3_D_OFF + 8 * 16 * 8 + 16)       // parked tiles: B operand [4][64] | C [4][64]
#define C3_XW_OFF (C3_ST_OFF + 2 * 256 * 8)          // wavefront 0's panel tile, as -L [3][4][64]
#define C3_XP_OFF (C3_XW_OFF + 3 * 256 * 8)          // X panels of two steps [2][16 tiles][4][64]
#define C3_LG_OFF (C3_XP_OFF + 2 * 16 * 256 * 8)     // inverse workspace [2 groups][12 blocks][16][18]
#define C3_SMEM (C3_LG_OFF + 2 * 12 * 16 * C3_PLD * 8)

// Tile (i, l) of owner o, slot q: i | l << 4 (0xf0: none); the owners are the wavefronts of SIMDs
// 1..3 (1, 2, 3, 5, 6, 7, 9, ...): wavefronts 4, 8, 12 share SIMD 0 with wavefront 0, whose
// elimination ANY FP64 MFMA on that SIMD stalls for up to 64 cycles per instruction (measured:
// 2-5 x slower) -- they do the write-backs instead.  Tile (0, 0), the first pivot tile, goes from
// global memory to the elimination and has no owner.
//   * The band (j, j - 2), (j, j - 1), (j, j) of tile row j has ONE owner: in step j - 1 it finishes
//     panel tile (j, j - 2), updates (j, j - 1) and (j, j) from its own registers and parks them
//     for wavefront 0 -- without waiting for any other wavefront.
//   * The rest is dealt for an even number of live tiles per owner throughout and few tiles of
//     one column on one owner.
#define C3_NS 12
__device__ static const unsigned char c3_tab[12][C3_NS] = {
    {0x01, 0x11, 0xbd, 0xcd, 0xdd, 0x7b, 0x6c, 0x5e, 0x4e, 0x3d, 0x2c, 0xf0},
    {0x02, 0x12, 0x22, 0xce, 0xde, 0xee, 0x6b, 0x5d, 0x4d, 0x3c, 0xf0, 0xf0},
    {0x13, 0x23, 0x33, 0xdf, 0xef, 0xff, 0x5c, 0x4c, 0x09, 0xf0, 0xf0, 0xf0},
    {0x24, 0x34, 0x44, 0xcf, 0xaf, 0x8c, 0x7d, 0x6e, 0x17, 0x05, 0x0d, 0xf0},
    {0x35, 0x45, 0x55, 0xbe, 0x9c, 0x8d, 0x7e, 0x6f, 0x2a, 0x1a, 0x08, 0xf0},
    {0x46, 0x56, 0x66, 0xbf, 0x9d, 0x8f, 0x38, 0x27, 0x14, 0x1d, 0x03, 0x0c},
    {0x57, 0x67, 0x77, 0xad, 0x9e, 0x48, 0x37, 0x25, 0x2e, 0x15, 0x1e, 0x04},
    {0x68, 0x78, 0x88, 0xae, 0x9f, 0x59, 0x49, 0x39, 0x28, 0x18, 0x06, 0x0e},
    {0x79, 0x89, 0x99, 0x69, 0x58, 0x47, 0x36, 0x3f, 0x26, 0x2f, 0x16, 0x1f},
    {0x8a, 0x9a, 0xaa, 0x7a, 0x6a, 0x5a, 0x4a, 0x3a, 0x29, 0x19, 0x07, 0x0f},
    {0x9b, 0xab, 0xbb, 0x8b, 0x7c, 0x6d, 0x5f, 0x4f, 0x3e, 0x2d, 0x1c, 0x0b},
    {0xac, 0xbc, 0xcc, 0x8e, 0x7f, 0x5b, 0x4b, 0x3b, 0x2b, 0x1b, 0x0a, 0xf0}};

// lane L of the own row of 16 lanes -> every lane of the row
template <int L>
__device__ __forceinline__ double c3_bcast(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xf, 0xf, false);
}
// acc += (lane L's src) * mult
template <int L>
__device__ __forceinline__ void c3_fmac(double &acc, double src, double mult) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mult), "n"(L));
}
// the same, two wait states first: a DPP operand that the PREVIOUS instruction wrote is read
// stale otherwise (inline assembly is invisible to the compiler's hazard recogniser; seen as
// the un-refined reciprocal seed reaching the pivot chain, 1e-11 instead of 1e-16)
template <int L>
__device__ __forceinline__ void c3_fmac_fresh(double &acc, double src, double mult) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mult), "n"(L));
}
// workgroup barrier behind the LDS traffic only: __syncthreads() also waits for the global
// stores in flight (L tiles on their way to memory, 1-2 us), which nobody in this kernel reads
__device__ __forceinline__ void c3_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ int c3_ld(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#define C3_BAD_CLASS (0x1 | 0x2 | 0x4 | 0x20 | 0x40 | 0x200)  // NaNs, -inf, -0, +0, +inf

// Column C of the elimination.  a[]: the lane's row of the tile (entries j <= row valid; what
// sits above the diagonal never reaches a valid result), e[]: its row of inv(L) in the making,
// -L[row][C] (0 on and above the diagonal) goes to LDS at byte address lkrow + 8 C from lanes
// 0..15 (the other lanes hold copies), r: 1 / a[C] of the own lane (meaningful in lane C),
// npc: -a[C] * a[C + 1][C]; dmine <- the pivot of the own row, dsel[s] <- 1 / D[4 s + q] (q = the
// lane's row of 16 lanes: the D^-1 entries its MFMA operands need).
template <int C>
__device__ __forceinline__ void c3_col(double (&a)[16], double (&e)[16], unsigned lkrow, double &r,
                                       double &npc, double &dmine, double (&dsel)[4], int row, int q) {
  constexpr int C1 = (C + 1) & 15, C2 = (C + 2) & 15;
  double rn = 0.0, npn = 0.0;
  if (C + 1 < 16) {
    // the pivot chain: next pivot's column first, its reciprocal in flight behind the rest
    c3_fmac_fresh<C>(a[C1], r, npc);
    rn = __builtin_amdgcn_rcp(a[C1]);
  }
  const double rb = c3_bcast<C>(r);
  const double l = (row > C) ? -a[C] * rb : 0.0;
  asm volatile("s_mov_b64 exec, 0xffff\n\tds_write_b64 %0, %1 offset:%2\n\ts_mov_b64 exec, -1"
               :
               : "v"(lkrow), "v"(l), "n"(C * 8)
               : "memory");
  if (C + 2 < 16) npn = -a[C1] * c3_bcast<C2>(a[C1]);
#define C3_UA(K) \
  if (K > C + 1) c3_fmac<K>(a[K], a[C], l);
  C3_UA(2) C3_UA(3) C3_UA(4) C3_UA(5) C3_UA(6) C3_UA(7) C3_UA(8) C3_UA(9) C3_UA(10) C3_UA(11)
  C3_UA(12) C3_UA(13) C3_UA(14) C3_UA(15)
#undef C3_UA
  // (row operations on the identity: column C of it comes alive here, e[C] = delta + l)
  e[C] = (row == C) ? 1.0 : l;
#define C3_UE(J) \
  if (J < C) c3_fmac<C>(e[J], e[J], l);
  C3_UE(0) C3_UE(1) C3_UE(2) C3_UE(3) C3_UE(4) C3_UE(5) C3_UE(6) C3_UE(7) C3_UE(8) C3_UE(9)
  C3_UE(10) C3_UE(11) C3_UE(12) C3_UE(13) C3_UE(14) C3_UE(15)
#undef C3_UE
  dmine = (row == C) ? a[C] : dmine;
  dsel[C >> 2] = (q == (C & 3)) ? rb : dsel[C >> 2];
  if (C + 1 < 16) {
    rn = fma(rn, fma(-a[C1], rn, 1.0), rn);
    r = rn;
    npc = npn;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// wavefront 0: LDL^T of the tile in a[] (lane <-> row in each row of 16 lanes).  Leaves -L in
// lkrow[] (LDS), inv(L) in e[], the own row's pivot in dmine and 1 / D[4 s + q] in dsel[s] (one
// Newton step).
__device__ __forceinline__ void c3_eliminate(double (&a)[16], unsigned lkrow, double (&e)[16],
                                             double &dmine, double (&dsel)[4], int row, int q) {
  double r = __builtin_amdgcn_rcp(a[0]);
  r = fma(r, fma(-a[0], r, 1.0), r);
  double npc = -a[0] * c3_bcast<1>(a[0]);
  dmine = 1.0;
  c3_col<0>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<1>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<2>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<3>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<4>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<5>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<6>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<7>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<8>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<9>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<10>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<11>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<12>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<13>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<14>(a, e, lkrow, r, npc, dmine, dsel, row, q);
  c3_col<15>(a, e, lkrow, r, npc, dmine, dsel, row, q);
}

// One block of the inverse of the unit-lower 64 x 64 tile of a 64-group.  The group's workspace
// holds twelve 16 x 18 blocks: L block (p, r), r < p, at index p (p - 1) / 2 + r, and the finished
// inverse block X_pq, transposed, at 6 + p (p - 1) / 2 + q; Ep = inv(L_pp) and Eq = inv(L_qq) come
// from the ring of the last steps' inverses (row-major, stride C3_PLD).
//   X_pq = -inv(L_pp) (L_pq inv(L_qq) + sum_{q < r < p} L_pr X_rq),   q < p,
// with MFMA: a 16 x 16 accumulator IS the B operand of the next four k-steps.
// Writes X_pq^T into the workspace and X_pq into inv / inv^T ([row][64] each).
__device__ __forceinline__ double *c3_lblock(double *Lg, int p, int r) { return Lg + (p * (p - 1) / 2 + r) * 16 * C3_PLD; }
__device__ __forceinline__ void c3_inverse_block(double *Lg, const double *Ep, const double *Eq, int p, int q,
                                                 int l15, int l4, double *__restrict__ o,
                                                 double *__restrict__ ot) {
  double4_t S = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int r = q; r < p; ++r) {
    const double *Lpr = c3_lblock(Lg, p, r) + l15 * C3_PLD + l4;
    const double *Xrq = (r == q) ? Eq + l4 * C3_PLD + l15 : c3_lblock(Lg, r, q) + 6 * 16 * C3_PLD + l15 * C3_PLD + l4;
    const int xs = (r == q) ? 4 * C3_PLD : 4;  // k-step stride: rows of E_q, columns of X_rq^T
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) S = __builtin_amdgcn_mfma_f64_16x16x4f64(Lpr[4 * rr], Xrq[xs * rr], S, 0, 0, 0);
  }
  double4_t X = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
    X = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ep[l15 * C3_PLD + 4 * rr + l4], S[rr], X, 0, 0, 0);
  double *Xt = c3_lblock(Lg, p, q) + 6 * 16 * C3_PLD;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int a = l4 + 4 * rr;  // row inside block (p, q), column l15
    Xt[l15 * C3_PLD + a] = X[rr];
    o[(16 * p + a) * 64 + 16 * q + l15] = X[rr];
    ot[(16 * q + l15) * 64 + 16 * p + a] = X[rr];
  }
}

// The chain workgroup (1024 threads).  Rows / columns [c0, c0 + nb) of K, nb <= 256.
//   K      lower triangle, row-major (stride ldk); on exit unit-lower L below, D on the diagonal
//   dvec / dinv   D and 1 / D
//   flags  [0] |= 1 on a zero / non-finite pivot, [1] += negative pivots
//   Linv / LinvT  inverses of the unit-lower 64 x 64 diagonal tiles and their transposes
//   dbg    optional phase stamps (wall clock; shader cycles 64 entries further): [0, 32)
//          wavefront 0, [32, 64) wavefront 5
