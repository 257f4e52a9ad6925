// The diagonal chain D(k), third design -- EXPERIMENTAL, NOT the default (PGF_CHAIN=3 selects it;
// the production chain is chain_body of pgf_factor2.hip).  Correct for every block size
// (tools/chain3_test.hip, tests/test_gpu_schedules.py variant "register_resident_chain"), but
// 78-92 us per block against 64 us for the LDS-panel chain with DPP elimination: the rank-16 MFMA
// work of one block is bound by ONE CU's matrix pipes whatever the layout, MFMAs and the
// elimination's VALU work contend whenever they share a SIMD, and the operand traffic through LDS
// comes back as the critical section.  Kept as the record of that experiment and
// of the hardware measurements at the end of this comment; the description below is the design
// as built (three role loops: elimination wavefront 0, service wavefronts 4 / 8 / 12 for the
// write-backs, twelve owner wavefronts on SIMDs 1-3 with lagged rank-16 updates).
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
#define C3_D_OFF (C3_LK_OFF + 4 * 16 * C3_PLD * 8)   // D[4][16] | 1/D[4][16] | 1/D as [4][row of lanes][k-step] | counters
#define C3_ST_OFF (C3_D_OFF + 12 * 16 * 8 + 16)      // parked tiles: B operand [2][64][2] | C [2][64][2]
#define C3_XW_OFF (C3_ST_OFF + 2 * 256 * 8)          // wavefront 0's panel tile, as -L [3][2][64][2]
#define C3_XP_OFF (C3_XW_OFF + 3 * 256 * 8)          // X panels of two steps [2][16 tiles][2][64][2]
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
__device__ __forceinline__ void chain3_body(unsigned char *smem, double *K, int64_t ldk, int c0, int nb,
                                            double *__restrict__ dvec, double *__restrict__ dinv,
                                            int *__restrict__ flags, double *__restrict__ Linv,
                                            double *__restrict__ LinvT, long long *__restrict__ dbg) {
  double *PT = reinterpret_cast<double *>(smem + C3_PT_OFF);
  double *EI = reinterpret_cast<double *>(smem + C3_EI_OFF);  // [C3_NEI][16][18]
  double *LK = reinterpret_cast<double *>(smem + C3_LK_OFF);  // [4][16][18]: -L_kk, D on the diagonal
  double *Dl = reinterpret_cast<double *>(smem + C3_D_OFF);    // D [4][16], then 1/D [4][16]
  int *stg = reinterpret_cast<int *>(smem + C3_D_OFF + 12 * 16 * 8);  // steps whose two tiles are parked
  double *STB = reinterpret_cast<double *>(smem + C3_ST_OFF), *STC = STB + 256;
  double *XW = reinterpret_cast<double *>(smem + C3_XW_OFF);  // [3][4][64]
  double *XP = reinterpret_cast<double *>(smem + C3_XP_OFF);  // [2][16][4][64]
  double *Lg = reinterpret_cast<double *>(smem + C3_LG_OFF);  // [2][12][16][18]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bend = c0 + nb;
  const int nst = (nb + 15) >> 4;  // 16-column steps
  int dbi = 0;
  (void)dbi;
#ifdef C3_TIMING
#define C3_STAMP()                     \
  do {                                 \
    if (dbg && tid == 0 && dbi < 32) { \
      dbg[64 + dbi] = clock64();       \
      dbg[dbi++] = wall_clock64();     \
    }                                  \
  } while (0)
#else
#define C3_STAMP() \
  do {             \
  } while (0)
#endif
  C3_STAMP();
  int dbo = 32;
  (void)dbo;
#ifdef C3_TIMING
#define C3_OSTAMP(cond)                                                       \
  do {                                                                        \
    if (dbg && tid == 320 && (cond) && dbo < 64) dbg[dbo++] = wall_clock64(); \
  } while (0)
#else
#define C3_OSTAMP(cond) \
  do {                  \
  } while (0)
#endif

  // inv / inv^T of the block's 64 x 64 tiles are written block row by block row as the chain
  // advances (zeros above the diagonal included); where a ragged last tile has no block rows:
  // identity, here
  if (nst & 3) {
    const int g = (nb - 1) >> 6;
    double *o = Linv + (size_t)((c0 >> 6) + g) * 4096, *ot = LinvT + (size_t)((c0 >> 6) + g) * 4096;
    for (int p = tid; p < 4096; p += 1024) {
      const int rr = p >> 6, cc = p & 63;
      if ((rr >> 4) >= (nst & 3)) o[p] = (rr == cc) ? 1.0 : 0.0;
      if ((cc >> 4) >= (nst & 3)) ot[p] = (rr == cc) ? 1.0 : 0.0;
    }
  }
  if (tid == 0) *stg = 0;

  // Three role loops with the same barrier sequence: the roles' register needs differ
  // (elimination: two 16-entry rows; tile owners: twelve accumulators), and in one loop the
  // compiler keeps all sets alive across each other's code and spills.
  if (wave == 0) {
    {
      // pivot tile of step 0: from global memory (identity beyond the block's end) to where every
      // step finds its tile
      const int row = c0 + l15;
      const double *src = K + (int64_t)min(row, bend - 1) * ldk + c0 + l4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = c0 + l4 + 4 * r;
        const double v = src[4 * r];
        PT[l15 * C3_PLD + l4 + 4 * r] = (row < bend) ? ((col <= row) ? v : 0.0) : ((row == col) ? 1.0 : 0.0);
      }
    }
    bool bad = false;
    int neg = 0;
    c3_barrier();
    C3_STAMP();
    for (int k = 0; k < nst; ++k) {
      const int ncol = min(16, bend - (c0 + 16 * k));
      double e[16], dsel[4] = {0.0, 0.0, 0.0, 0.0}, dmine;
      {
        // (lane indices laundered around the elimination: otherwise every LDS address of the
        // step is computed up front and kept alive across it, where a lane has no register to
        // spare, and spilled)
        int ln = threadIdx.x;
        asm volatile("" : "+v"(ln));
        // the pivot tile, lane <-> row (from the MFMA layout through LDS; not carried in
        // registers from the end of the previous step: the compiler then keeps two copies)
        double a[16];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const double2_t v = *reinterpret_cast<const double2_t *>(PT + (ln & 15) * C3_PLD + j);
          a[j] = v.x;
          a[j + 1] = v.y;
        }
        // (-L_kk goes to LDS column by column: a service wavefront takes it to global memory two
        // steps on)
        double *lkp = LK + ((k & 3) * 16 + (ln & 15)) * C3_PLD;
        c3_eliminate(a, (unsigned)(uintptr_t)lkp, e, dmine, dsel, ln & 15, ln >> 4);
        if (ln < 16) lkp[ln] = dmine;
      }
      if (k == 1 || k == 14) C3_STAMP();  // elimination
      int ln = threadIdx.x;
      asm volatile("" : "+v"(ln));
      const int r15 = ln & 15, r4 = ln >> 4;
      const double imine = fast_recip(dmine);
      {
        const bool mine = ln < 16 && r15 < ncol;
        bad |= __ballot(mine && __builtin_amdgcn_class(dmine, C3_BAD_CLASS)) != 0ull;
        neg += __popcll(__ballot(mine && dmine < 0.0));
      }
      // inv(L_kk) leaves the registers as the MFMA operand each lane needs of it -- entries
      // (row, 4 s + its row of 16 lanes) -- and goes to LDS from all 64 lanes in that shape: four
      // 8-byte stores instead of eight 16-byte stores of 16 lanes (an LDS store costs ~20 cycles
      // per 8 bytes of a lane)
      double as[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        as[s] = (r4 == 0) ? e[4 * s] : (r4 == 1) ? e[4 * s + 1] : (r4 == 2) ? e[4 * s + 2] : e[4 * s + 3];
      double *ei = EI + ((k % C3_NEI) * 16 + r15) * C3_PLD + r4;
      if (k + 1 < nst) {
        // ---- panel tile (k + 1, k) and the next pivot tile (k + 1, k + 1), parked by their owner
        for (int it = 0; it < (1 << 22) && c3_ld(stg) < k + 1; ++it) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        // (tiles in MFMA-operand layout live in LDS as [k-step pair][lane][2]: one 16-byte access
        // per two k-steps -- LDS time, not the matrix pipe, bounds the owners' updates otherwise)
        double bt[4];
        double4_t x = (double4_t){0.0, 0.0, 0.0, 0.0}, t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double2_t b2 = *reinterpret_cast<const double2_t *>(STB + 128 * h + 2 * ln);
          const double2_t c2 = *reinterpret_cast<const double2_t *>(STC + 128 * h + 2 * ln);
          bt[2 * h] = b2.x;
          bt[2 * h + 1] = b2.y;
          t[2 * h] = c2.x;
          t[2 * h + 1] = c2.y;
        }
        // (what the owners wait for goes out while the matrix pipe works)
#pragma unroll
        for (int s = 0; s < 4; ++s) ei[4 * s] = as[s];
        if (ln < 16) {
          Dl[16 * (k & 3) + r15] = dmine;
          Dl[64 + 16 * (k & 3) + r15] = imine;
          Dl[128 + 16 * (k & 3) + 4 * (r15 & 3) + (r15 >> 2)] = imine;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(as[s], bt[s], x, 0, 0, 0);
        // the panel tile is published as -L = -X D^-1: every live tile it meets lies below its
        // tile row, where it is the scaled operand (and the write-back wants L)
        double *xo = XW + (k % 3) * 256 + 2 * ln;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double2_t nl2;
          nl2.x = -x[2 * h] * dsel[2 * h];
          nl2.y = -x[2 * h + 1] * dsel[2 * h + 1];
          *reinterpret_cast<double2_t *>(xo + 128 * h) = nl2;
          t = __builtin_amdgcn_mfma_f64_16x16x4f64(nl2.x, x[2 * h], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f64_16x16x4f64(nl2.y, x[2 * h + 1], t, 0, 0, 0);
        }
        // next pivot tile: MFMA layout -> (next step) lane <-> row through LDS
#pragma unroll
        for (int r = 0; r < 4; ++r) PT[r15 * C3_PLD + r4 + 4 * r] = t[r];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) ei[4 * s] = as[s];
        if (ln < 16) {
          Dl[16 * (k & 3) + r15] = dmine;
          Dl[64 + 16 * (k & 3) + r15] = imine;
          Dl[128 + 16 * (k & 3) + 4 * (r15 & 3) + (r15 >> 2)] = imine;
        }
      }
      if (k == 1 || k == 14) C3_STAMP();  // panel tile, next pivot tile
      c3_barrier();  // step k is published: inv(L_kk), D, panel tile (k + 1, k), the factored tile
      C3_STAMP();
    }
    c3_barrier();  // (the last write-back reads the last factored tile)
    if (lane == 0) {
      if (bad) atomicOr(&flags[0], 1);
      if (neg) atomicAdd(&flags[1], neg);
    }
  } else if ((wave & 3) == 0) {
    // ---- wavefronts 4, 8, 12: the write-backs (no FP64 arithmetic to speak of on wavefront 0's SIMD)
    const int sv = (wave >> 2) - 1;  // 0, 1, 2: the tiles with tile row % 3 == sv
    const int wr = lane >> 2, wc4 = lane & 3;
    // 16 x 16 tile in MFMA-operand layout ([4][64] at xs, scaled by dsc[column] or negated) ->
    // rows [row0, row0 + 16), columns [col0, col0 + 16) of K, four lanes per row
    auto write_tile = [&](const double *xs, const double *dsc, int row0, int col0) __attribute__((always_inline)) {
      const double *x4 = xs + (wc4 >> 1) * 128 + 2 * wr + (wc4 & 1);  // entry (k-step wc4, lane t * 16 + wr)
      double2_t lo, hi;
      lo.x = dsc ? x4[0] * dsc[4 * wc4] : -x4[0];
      lo.y = dsc ? x4[32] * dsc[4 * wc4 + 1] : -x4[32];
      hi.x = dsc ? x4[64] * dsc[4 * wc4 + 2] : -x4[64];
      hi.y = dsc ? x4[96] * dsc[4 * wc4 + 3] : -x4[96];
      if (row0 + wr < bend) {
        double *dst = K + (int64_t)(row0 + wr) * ldk + col0 + 4 * wc4;
        *reinterpret_cast<double2_t *>(dst) = lo;
        *reinterpret_cast<double2_t *>(dst + 2) = hi;
      }
    };
    // the factored pivot tile of step ks with D and 1 / D
    auto write_pivot_tile = [&](int ks) __attribute__((always_inline)) {
      const int row = c0 + 16 * ks + wr;
      const double *lk = LK + ((ks & 3) * 16 + wr) * C3_PLD + 4 * wc4;
      if (row < bend) {
        double *dst = K + (int64_t)row * ldk + c0 + 16 * ks + 4 * wc4;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (4 * wc4 + t <= wr) dst[t] = (4 * wc4 + t == wr) ? lk[t] : -lk[t];
        if (wc4 == 0) {
          dvec[row] = Dl[16 * (ks & 3) + wr];
          dinv[row] = Dl[64 + 16 * (ks & 3) + wr];
        }
      }
    };
    c3_barrier();
    for (int k = 1; k <= nst; ++k) {
      c3_barrier();
      const int kk = k - 1, kl = k - 2;
      // panel tile (k, kk), wavefront 0's (-L): global memory, and its copy in the inverse workspace
      if (k < nst && sv == kk % 3) {
        const double *xw = XW + (kk % 3) * 256;
        write_tile(xw, nullptr, c0 + 16 * k, c0 + 16 * kk);
        if ((k >> 2) == (kk >> 2)) {
          double *lb = c3_lblock(Lg + ((kk >> 2) & 1) * 12 * 16 * C3_PLD, k & 3, kk & 3) + l15 * C3_PLD + l4;
#pragma unroll
          for (int s = 0; s < 4; ++s) lb[4 * s] = -xw[(s >> 1) * 128 + 2 * lane + (s & 1)];
        }
      }
      // the owners' tiles of panel kl (complete since the last barrier; its buffer stays through
      // this body), and the factored pivot tile of that step
      if (kl >= 0) {
        for (int it = k + sv; it < nst; it += 3)
          write_tile(XP + (size_t)((kl & 1) * 4096 + it * 256), Dl + 64 + 16 * (kl & 3), c0 + 16 * it, c0 + 16 * kl);
        if (sv == kl % 3) write_pivot_tile(kl);
      }
    }
    c3_barrier();
    if (sv == (nst - 1) % 3) write_pivot_tile(nst - 1);
  } else {
    // ---- the tile owners
    const int oi = wave - 1 - (wave >> 2);
    int ti[C3_NS], tl[C3_NS];
    double4_t acc[C3_NS];
#pragma unroll
    for (int q = 0; q < C3_NS; ++q) {
      const int v = c3_tab[oi][q];
      ti[q] = ((v & 15) >= nst || (v >> 4) > (v & 15)) ? -1 : (v & 15);
      tl[q] = v >> 4;
    }
#pragma unroll
    for (int q = 0; q < C3_NS; ++q) {
      acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
      if (ti[q] >= 0) {
        const int row = c0 + 16 * ti[q] + l15;
        // (loads unconditional, from a clamped row: predicated, every one of them waits for the
        // one before; what they fetch above the diagonal or beyond the block's end is replaced)
        const double *src = K + (int64_t)min(row, bend - 1) * ldk + c0 + 16 * tl[q] + l4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = c0 + 16 * tl[q] + l4 + 4 * r;
          const double v = src[4 * r];
          acc[q][r] = (row < bend) ? ((col <= row) ? v : 0.0) : ((row == col) ? 1.0 : 0.0);  // identity beyond the end
        }
      }
    }
    // the two tiles wavefront 0 needs in step k -- T(k + 1, k) as the B operand of its panel
    // product, pivot tile (k + 1, k + 1) as the accumulator of its update -- go to LDS
    auto park = [&](int k) __attribute__((always_inline)) {
      if (k + 1 >= nst) return;
      bool mine = false;
#pragma unroll
      for (int q = 0; q < C3_NS; ++q) {
        if (ti[q] == k + 1 && tl[q] == k) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            *reinterpret_cast<double2_t *>(STB + 128 * h + 2 * lane) = (double2_t){acc[q][2 * h], acc[q][2 * h + 1]};
          ti[q] = -1;
          mine = true;
        }
        if (ti[q] == k + 1 && tl[q] == k + 1) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            *reinterpret_cast<double2_t *>(STC + 128 * h + 2 * lane) = (double2_t){acc[q][2 * h], acc[q][2 * h + 1]};
          ti[q] = -1;
        }
      }
      if (mine) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(stg, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    park(0);
    c3_barrier();
    // Body k, beside wavefront 0's step k.  The rank-16 updates LAG one step: what body k applies
    // to the bulk of the tiles is step k - 2, whose panel has been complete since the last
    // barrier -- nobody waits for the panel of step kk = k - 1, which this body only writes.  Only
    // the tiles about to turn into operands catch up first: tile column kk (the panel tiles of
    // this body) and the band of tile row k + 1 (parked for wavefront 0).
    for (int k = 1; k <= nst; ++k) {
      c3_barrier();  // wavefront 0 has published step kk: inv(L_kk), D, panel tile (k, kk)
      // (tile indices laundered per step: otherwise every LDS address of every slot is hoisted
      // out of the loop and kept alive beside the accumulators)
#pragma unroll
      for (int q = 0; q < C3_NS; ++q) asm volatile("" : "+s"(ti[q]), "+s"(tl[q]));
      const int kk = k - 1, kl = k - 2;
      const double *xpl = XP + (size_t)((kl & 1) * 4096) + 2 * lane;  // panel kl (complete)
      double *xpn = XP + (size_t)((kk & 1) * 4096) + 2 * lane;        // panel kk (this body writes it)
      const double *xwl = XW + ((kl + 3) % 3) * 256 + 2 * lane;       // -L of panel tile (kk, kl)
      const double *xwn = XW + (kk % 3) * 256 + 2 * lane;             // -L of panel tile (k, kk)
      const double *ei = EI + ((kk % C3_NEI) * 16 + l15) * C3_PLD + l4;
      const double *din = Dl + 128 + 16 * (kk & 3) + 4 * l4;  // 1 / D of step kk, this lane's four
      const double *dil = Dl + 128 + 16 * (kl & 3) + 4 * l4;  // 1 / D of step kl, this lane's four
      double *Lgk = Lg + ((kk >> 2) & 1) * 12 * 16 * C3_PLD;
      C3_OSTAMP(k == 2 || k == 13);
      // step kl's update of tile q (tile rows >= k, tile columns >= kk; panel tile (kk, kl) is
      // wavefront 0's and comes as -L: every live tile it meets has it on the column side).
      // Two k-steps at a time: with all eight operands of a tile in flight at once the twelve
      // accumulators do not fit the 128 registers of a lane.
      auto update_lag = [&](int q) __attribute__((always_inline)) {
        const double *xi = xpl + ti[q] * 256;
        const double *xl = (tl[q] == kk) ? xwl : xpl + tl[q] * 256;
        const bool scale = tl[q] != kk;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double2_t b2 = *reinterpret_cast<const double2_t *>(xi + 128 * h);
          double2_t a2 = *reinterpret_cast<const double2_t *>(xl + 128 * h);
          if (scale) a2 = -a2 * *reinterpret_cast<const double2_t *>(dil + 2 * h);
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b2.x, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b2.y, acc[q], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // panel tile of slot q: X^T = inv(L_kk) T^T -> x (MFMA C layout = operand layout of the
      // updates), panel buffer, inverse workspace
      auto panel_tile = [&](int q, double4_t &x) __attribute__((always_inline)) {
        x = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(ei[4 * s], acc[q][s], x, 0, 0, 0);
        double *xo = xpn + ti[q] * 256;
        const bool ingroup = (ti[q] >> 2) == (kk >> 2);
        double *lb = c3_lblock(Lgk, ti[q] & 3, kk & 3) + l15 * C3_PLD + l4;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<double2_t *>(xo + 128 * h) = (double2_t){x[2 * h], x[2 * h + 1]};
        if (ingroup) {
#pragma unroll
          for (int s = 0; s < 4; ++s) lb[4 * s] = x[s] * din[s];
        }
        ti[q] = -1;  // final
      };
      // ---- the band of tile row k + 1, if it is this wavefront's: catch up with step kl, panel
      // tile (k + 1, kk), then from registers step kk's update of the two tiles wavefront 0
      // takes over in step k
      if (k + 1 < nst) {
        bool band = false;
#pragma unroll
        for (int q = 0; q < C3_NS; ++q) {
          if (ti[q] == k + 1 && tl[q] >= kk) {
            if (kl >= 0) update_lag(q);
            band = true;
          }
        }
        if (band) {
          double4_t xb = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < C3_NS; ++q)
            if (ti[q] == k + 1 && tl[q] == kk) panel_tile(q, xb);
#pragma unroll
          for (int q = 0; q < C3_NS; ++q) {
            if (ti[q] == k + 1 && tl[q] == k) {
#pragma unroll
              for (int s = 0; s < 4; ++s)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xwn[(s >> 1) * 128 + (s & 1)], xb[s], acc[q], 0, 0, 0);
            }
            if (ti[q] == k + 1 && tl[q] == k + 1) {
#pragma unroll
              for (int s = 0; s < 4; ++s)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xb[s] * din[s], xb[s], acc[q], 0, 0, 0);
            }
          }
          park(k);
        }
      }
      C3_OSTAMP(k == 2 || k == 13);
      // ---- the rest of tile column kk: catch up, then the panel tile
#pragma unroll
      for (int q = 0; q < C3_NS; ++q) {
        if (ti[q] > k && tl[q] == kk) {
          if (kl >= 0) update_lag(q);
          double4_t x;
          panel_tile(q, x);
        }
      }
      C3_OSTAMP(k == 2 || k == 13);
      // ---- step kl's update of every other live tile
      if (kl >= 0) {
#pragma unroll
        for (int q = 0; q < C3_NS; ++q)
          if (ti[q] >= 0 && tl[q] > kk) update_lag(q);
      }
      C3_OSTAMP(k == 2 || k == 13);
      // ---- block row kk & 3 of the group's 64 x 64 inverse: one wavefront per block
      {
        const int p = kk & 3, g = kk >> 2;
        double *o = Linv + (size_t)((c0 >> 6) + g) * 4096, *ot = LinvT + (size_t)((c0 >> 6) + g) * 4096;
        const double *Ep = EI + (kk % C3_NEI) * 16 * C3_PLD;
        if (wave >= 13 && wave - 13 < p) {
          const int q = wave - 13;
          c3_inverse_block(Lgk, Ep, EI + ((kk - p + q) % C3_NEI) * 16 * C3_PLD, p, q, l15, l4, o, ot);
        }
        if (wave == 11) {  // the diagonal block, and the zero blocks to its right
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int a = l4 + 4 * rr;
            const double v = Ep[a * C3_PLD + l15];
            o[(16 * p + a) * 64 + 16 * p + l15] = v;
            ot[(16 * p + l15) * 64 + 16 * p + a] = v;
            for (int q = p + 1; q < 4; ++q) {
              o[(16 * p + a) * 64 + 16 * q + l15] = 0.0;
              ot[(16 * q + l15) * 64 + 16 * p + a] = 0.0;
            }
          }
        }
      }
      C3_OSTAMP(k == 2 || k == 13);
    }
    c3_barrier();
  }
#undef C3_STAMP
#undef C3_OSTAMP
}
