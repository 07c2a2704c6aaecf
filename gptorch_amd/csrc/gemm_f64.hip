// fp64 MFMA "NT" contraction for gfx950:  C = alpha * A * B^T + beta * C
//   A[M,K], B[N,K], C[M,N] row-major, K contiguous in both operands.
// This one kernel carries every O(N^3) flop of the path: the SYRK/GEMM trailing
// updates of the blocked Cholesky, the panel solves (as products with the
// inverted 64x64 diagonal blocks), the predict-path TRSM and the K^-1 products
// of the backward.
//
// Design (MI355X-first, not a port of anything):
//  * v_mfma_f64_16x16x4_f64: lane l supplies A[row l&15][k l>>4] and
//    B[k l>>4][col l&15]; result reg r of lane l is C[(l>>4)+4r][l&15].
//  * operand tiles are staged global->LDS by LDS-DMA (global_load_lds_dwordx4):
//    one wave-instruction fills one 1 KiB "fragment block" = 8 rows x 16 k's, eight
//    adjacent lanes per 128-byte line; the wave later reads a 16-row operand tile
//    back with ONE ds_read_b128 per lane and 8-k group (XOR-swizzled slots, bank-
//    conflict free).  A lane's 16 bytes are the pair (k=2g, k=2g+1) of its row,
//    g = l>>4, so one b128 read feeds two MFMAs (first MFMA sums k = {0,2,4,6},
//    second k = {1,3,5,7}: the k order inside a K-step is a permutation shared by
//    A and B, so the product is unchanged).
//  * 128x128 block tile as 8 waves of 32x64 (8 accumulator tiles = 64 VGPRs per wave),
//    BK = 16, two LDS stages (64 KB) => 2 workgroups per CU = 4 waves / SIMD, one barrier
//    / K-step.  (Until late round 2: 4 waves of 64x64, 2 / SIMD -- 3-4 % slower.)
//  * software-pipelined K loop (PIPE): operand fragments double-buffered in registers and
//    requested half a K-step ahead, barrier between the two halves, the LDS-DMA pieces of
//    step t+2 between the MFMA groups -- a wave's MFMA stream waits only for the barrier.
//  * blockIdx -> tile: bijective XCD remap (blocks b, b+8 share an XCD/L2) then
//    grouped ordering (8 tile-rows per group) so the tiles resident on one XCD
//    share operand panels in its 4 MiB L2; `lower` drops tiles above the diagonal.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdlib>
#include "gpn_common.h"

namespace gpn {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int mt, nt;       // tile counts
  int lower;        // 0 full, 1 lower-tile square, 2 trapezoid, 3 staircase (st_* below), 4 quarters of the last big tiles (q_*)
  // lower == 4: this launch computes, as BM x BN = 64 x 64 quarter tiles, the 128 x 128 tiles q_off .. q_off + q_cnt - 1 of a
  // lower-tile launch with q_mt tile rows (its partial last round: gemm_nt_impl); block b = quarter (b & 3) of tile q_off + b/4
  int q_off, q_cnt, q_mt;
  int lds_pad_kb;   // extra dynamic LDS per workgroup: caps the workgroups per CU of a launch that shares the chip (look-ahead)
  int group_h;      // tile rows per group of the grouped tile order (8; A/B of the L2 reuse: tools' build)
  int thin;         // 1: tiles with at most 16 rows of the matrix skip the MFMAs of their empty blocks (0: A/B, tools' build)
  // staircase: C is M x (nb * st_blk); column block b (st_blk columns) only has the rows from b * st_step on, and with
  // st_diag its first st_blk x st_blk square is lower-only -- the local tile columns of one block-cyclic trailing update
  int st_blk, st_step, st_diag;
  int tri;          // GPN_TRI_* structure flags: skip the K range where an operand is known zero
  double alpha, beta;
  // strided batch: problem z uses A + z*sA, B + z*sB, C + z*sC (batch identical shapes in ONE launch)
  int batch;
  int64_t sA, sB, sC;
  // two-level batch (lock-step models x equal nodes of one model): problem z = z1 + inner * z2 uses A + z1*sA + z2*sA2 ...
  // (inner == 0: one level)
  int inner;
  int64_t sA2, sB2, sC2;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective XCD remap: consecutive logical ids land on the same XCD (blocks b, b+8 share one)
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// full rectangle: groups of 8 tile rows, column-major inside a group
// `spread`: K-clipped (triangular-operand) launches have very unequal work per tile row, so
// their tiles are dealt round-robin over the XCDs (plain blockIdx order) instead of in
// contiguous per-XCD chunks: balance beats L2 locality there.
__device__ __forceinline__ void tile_of_block(int bid, int nwg, int mt, int nt, bool spread, int& ti, int& tj, int GH = 8) {
  const int logical = spread ? bid : xcd_remap(bid, nwg);
  const int group = GH * nt;
  const int g = logical / group;
  const int first = g * GH;
  const int gm = min(GH, mt - first);
  const int rem = logical - g * group;
  ti = first + rem % gm;
  tj = rem / gm;
}

// lower triangle only (grid = mt(mt+1)/2 exactly, so every XCD gets the same
// number of real tiles): groups of 8 tile rows; group g holds the 8g full columns
// left of the diagonal super-tile (column-major, 8 per column) followed by the
// 36 tiles of the diagonal super-tile.  Tiles before group g: 32 g^2 + 4 g.
__device__ __forceinline__ void lower_tile_of_index(int q, int mt, int& ti, int& tj, int GH = 8);
__device__ __forceinline__ void tile_of_block_lower(int bid, int nwg, int mt, bool spread, int& ti, int& tj, int GH = 8) {
  lower_tile_of_index(spread ? bid : xcd_remap(bid, nwg), mt, ti, tj, GH);
}
// logical index q of the grouped lower enumeration -> tile.  Group height GH (8 in the product): group g holds the GH g
// full columns left of its diagonal super-tile (column-major, GH per column) followed by the GH (GH + 1) / 2 tiles of the
// super-tile; tiles before group g: GH^2 g (g - 1) / 2 + g GH (GH + 1) / 2  (GH = 8: 32 g^2 + 4 g).
__device__ __forceinline__ void lower_tile_of_index(int q, int mt, int& ti, int& tj, int GH) {
  const int G = mt / GH;
  const int tri = GH * (GH + 1) / 2, sq = GH * GH;
  auto before = [&](int g) { return sq * (g * (g - 1) / 2) + g * tri; };
  const int full_total = before(G);
  int g, h;
  if (q < full_total) {
    // sq/2 g^2 + (tri - sq/2) g - q = 0
    const double a = 0.5 * sq, b = (double)tri - 0.5 * sq;
    g = (int)((sqrt(b * b + 4.0 * a * (double)q) - b) / (2.0 * a));
    while (g > 0 && before(g) > q) --g;
    while (before(g + 1) <= q) ++g;
    h = GH;
  } else {
    g = G;
    h = mt - G * GH;
  }
  int r = q - before(g);
  const int left = h * GH * g;
  if (r < left) {
    tj = r / h;
    ti = GH * g + r - tj * h;
  } else {
    r -= left;
    int c = 0;
    while (r >= h - c) { r -= h - c; ++c; }
    ti = GH * g + c + r;
    tj = GH * g + c;
  }
}

template <int BM, int BN, int WM, int WN, bool DMA, int NS = 2, bool BLOW = false, bool PIPE = false>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), ((BM / WM) * (BN / WN) > 8 ? 1 : 2)) void gemm_nt_kernel(GemmArgs p) {
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int WAVES_N = BN / WN;
  constexpr int BK = 16;
  constexpr int A_BLOCKS = (BM / 16) * 2, B_BLOCKS = (BN / 16) * 2;
  constexpr int NBLK = A_BLOCKS + B_BLOCKS;      // 1 KiB fragment blocks per K-step
  constexpr int NWAVES = (BM / WM) * (BN / WN);  // 4, or 8 (128x128 tile as 64x32 wave tiles: 4 waves / SIMD at 2 workgroups / CU)
  constexpr int PER_WAVE = (NBLK + NWAVES - 1) / NWAVES;
  constexpr int STAGE = NBLK * 1024;             // bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int ti, tj;
  const bool spread = (p.tri & (GPN_TRI_A_UPPER | GPN_TRI_A_LOWER | GPN_TRI_B_UPPER)) != 0;
  int bid = blockIdx.x, nwg = gridDim.x;
  if (p.batch > 1) {                               // strided batch: consecutive blocks = one problem
    nwg = gridDim.x / p.batch;
    const int z = bid / nwg;
    bid -= z * nwg;
    if (p.inner > 0) {
      const int z2 = z / p.inner, z1 = z - z2 * p.inner;
      p.A += z1 * p.sA + z2 * p.sA2; p.B += z1 * p.sB + z2 * p.sB2; p.C += z1 * p.sC + z2 * p.sC2;
    } else {
      p.A += z * p.sA; p.B += z * p.sB; p.C += z * p.sC;
    }
  }
  int diag_off = 0;       // lower launches: the entry (row, col) is on or below ITS diagonal iff col + diag_off <= row
  bool stair_diag_tile = false;
  if (p.lower == 3) {
    int q = xcd_remap(bid, nwg);
    const int bt = p.st_blk / BN, nb = p.nt / bt, stept = p.st_step / BM;
    int b = 0, sbt = 0, rows_t = 0, cnt = 0;
    for (; b < nb; ++b) {
      sbt = b * stept;
      rows_t = max(p.mt - sbt, 0);
      cnt = rows_t * bt - ((p.st_diag && rows_t >= bt) ? bt * (bt - 1) / 2 : 0);
      if (q < cnt) break;
      q -= cnt;
    }
    if (p.st_diag && rows_t >= bt) {
      const int rect = (rows_t - bt) * bt;
      if (q < rect) {
        tile_of_block(q, rect, rows_t - bt, bt, true, ti, tj, p.group_h);
        ti += sbt + bt;
      } else {
        tile_of_block_lower(q - rect, bt * (bt + 1) / 2, bt, true, ti, tj, p.group_h);
        stair_diag_tile = ti == tj;
        ti += sbt;
      }
    } else {
      tile_of_block(q, cnt, rows_t, bt, true, ti, tj, p.group_h);
      ti += sbt;
    }
    tj += b * bt;
    diag_off = sbt * BM - b * bt * BN;
  } else if (p.lower == 2) {
    // trapezoid (M >= N): the N x N top square lower-tile only, the (M - N) x N rectangle below it whole -- one tile
    // column of a block-cyclic trailing update incl. its diagonal tile.  The rectangle's tiles come first (they are
    // the bulk), in the grouped order of the full-rectangle case; the triangle's nt (nt + 1) / 2 tiles last.
    const int rect = (p.mt - p.nt) * p.nt;
    const int q = xcd_remap(bid, nwg);
    if (q < rect) {
      tile_of_block(q, rect, p.mt - p.nt, p.nt, true, ti, tj, p.group_h);
      ti += p.nt;
    } else {
      tile_of_block_lower(q - rect, nwg - rect, p.nt, true, ti, tj, p.group_h);
    }
  } else if (p.lower == 4) {
    // quarters of the big tiles of a partial last round: the four quarters of one parent are consecutive logical ids,
    // i.e. on one XCD (they share the parent's operand panels)
    const int idx = xcd_remap(bid, nwg);
    int pi, pj;
    lower_tile_of_index(p.q_off + (idx >> 2), p.q_mt, pi, pj, p.group_h);
    ti = 2 * pi + ((idx >> 1) & 1);
    tj = 2 * pj + (idx & 1);
    if (tj > ti || ti >= p.mt) return;        // the upper-right quarter of a diagonal parent; a ragged parent's empty half
  } else if (p.lower == 5) {
    // (A/B: 2:1 macro tiles) lower-tile square with BM = 2 BN: the two column halves of the BM x BM parent tiles of the
    // grouped lower order, consecutive logical ids (one XCD: they share the parent's row panel)
    const int idx = xcd_remap(bid, nwg);
    int pi, pj;
    lower_tile_of_index(idx >> 1, p.mt, pi, pj, p.group_h);
    ti = pi;
    tj = 2 * pj + (idx & 1);
    if (tj * BN >= p.N) return;
  } else if (p.lower) tile_of_block_lower(bid, nwg, p.mt, spread, ti, tj, p.group_h);
  else tile_of_block(bid, nwg, p.mt, p.nt, spread, ti, tj, p.group_h);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int m0 = ti * BM, n0 = tj * BN;
  const int mlim = (p.M + 15) & ~15, nlim = (p.N + 15) & ~15;

  // Fragment blocks.  One LDS-DMA wave-instruction moves 1 KiB = 8 rows x 16 k: lanes 8r..8r+7
  // fetch the eight 16-byte segments of ONE 128-byte line of row r (block h of a 16-row group
  // holds its rows 8h..8h+7), so every operand line is requested once per K-step by 8 adjacent
  // lanes -- that is what the DMA's address path coalesces on.  (Until r1x a block was 16 rows x
  // 8 k with lane l on row l&15: 16 different lines per instruction, each line fetched twice.
  // Same-box A/B: 8192^3 65.9 -> 68.7 TFLOP/s with 128x128 tiles, C3 212 -> 203 ms, C4 1569 ->
  // 1493 ms.)  The DMA writes LDS lane-linearly, so the segment index is XORed with (row>>1)&7 to
  // keep the operand read (lane l: row l&15, k pair l>>4) on 16 distinct 16-byte bank groups.
  const int R16 = lane & 15;
  const int roff0 = (R16 >> 3) * 1024 + ((R16 & 7) * 8 + ((lane >> 4) ^ ((R16 >> 1) & 7))) * 16;   // k group 0; group 1: ^ 64

  d4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  d2 stage_regs[DMA ? 1 : PER_WAVE];

  // per-wave staging slots: fragment block idx = wave + 4*i; source pointers are
  // hoisted out of the K loop (only k0 advances)
  const double* src_base[PER_WAVE];
  bool src_ok[PER_WAVE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int idx = wave + NWAVES * i;
    const bool isA = idx < A_BLOCKS;
    const int b = isA ? idx : idx - A_BLOCKS;
    const int rg = b >> 1, half = b & 1;
    const int row = (isA ? m0 : n0) + rg * 16;
    src_ok[i] = (idx < NBLK) && row < (isA ? mlim : nlim);
    // row groups past the operand's end re-read its first row group instead (always
    // readable; the garbage only reaches output rows/cols that are never stored), so
    // every wave issues exactly PER_WAVE DMAs per K-step and the counted waits stay exact
    const int srow = src_ok[i] ? row : 0;
    const int r16 = half * 8 + (lane >> 3);
    const int seg = (lane & 7) ^ ((r16 >> 1) & 7);
    src_base[i] = (isA ? p.A + (int64_t)(srow + r16) * p.lda : p.B + (int64_t)(srow + r16) * p.ldb) + seg * 2;
  }

  // issue the loads of K-step `t` into LDS stage `s` (DMA) or registers (!DMA)
  auto stage_issue = [&](int t, int s) {
    const int k0 = t * BK;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int idx = wave + NWAVES * i;
      const double* src = src_base[i] + k0;
      if constexpr (DMA) {
        if (NBLK % NWAVES == 0 || idx < NBLK) {   // NBLK % NWAVES == 0: every wave has PER_WAVE pieces, no branch in the loop
          char* dst = smem + s * STAGE + idx * 1024;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      } else {
        stage_regs[i] = src_ok[i] ? *reinterpret_cast<const d2*>(src) : d2{0.0, 0.0};
      }
    }
  };
  // one LDS-DMA piece of K-step `t` (PIPE: the pieces go out between the MFMA groups of step t - 2's second half)
  auto stage_issue_piece = [&](int i, int t, int s) {
    if constexpr (DMA) {
      const int idx = wave + NWAVES * i;
      char* dst = smem + s * STAGE + idx * 1024;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src_base[i] + t * BK),
          (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  auto stage_commit = [&](int s) {
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < PER_WAVE; ++i) {
        const int idx = wave + NWAVES * i;
        if (idx < NBLK) *reinterpret_cast<d2*>(smem + s * STAGE + idx * 1024 + lane * 16) = stage_regs[i];
      }
    }
  };

  // BLOW: B is lower-triangular (panel solve against an inverted leaf block, B[j][k] = 0 for
  // k > j): a 16-column tile needs no K beyond its last column -- skipped per (tile, 8-k group)
  auto compute = [&](int s, int k0) {
    const char* base = smem + s * STAGE;
#pragma unroll
    for (int kg8 = 0; kg8 < 2; ++kg8) {
      d2 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const d2*>(base + (wave_m * TM + i) * 2048 + (kg8 ? (roff0 ^ 64) : roff0));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const d2*>(base + (A_BLOCKS + (wave_n * TN + j) * 2) * 1024 + (kg8 ? (roff0 ^ 64) : roff0));
      if constexpr (BLOW) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (k0 + kg8 * 8 > n0 + wave_n * WN + j * 16 + 15) continue;   // wave-uniform
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          }
      }
    }
  };

  // triangular operands: rows of an upper-triangular operand are zero left of the
  // diagonal, rows of a lower-triangular one right of it -> clip the K range per tile
  int k_lo = 0, k_hi = p.K;
  if (p.tri & GPN_TRI_A_UPPER) k_lo = max(k_lo, m0);
  if (p.tri & GPN_TRI_B_UPPER) k_lo = max(k_lo, n0);
  if (p.tri & GPN_TRI_A_LOWER) k_hi = min(k_hi, m0 + BM);
  if (p.tri & GPN_TRI_B_LOWER) k_hi = min(k_hi, n0 + BN);
  const int t0 = k_lo / BK;
  const int nk = max(t0, (k_hi + BK - 1) / BK);
  if constexpr (NS == 2) {
    if constexpr (PIPE) {
      // Software-pipelined K loop (the default for the two big tile shapes).  A K-step is two halves of 8 k's;
      // the operand fragments of each half are read from LDS most of a half (3 of 4 MFMA groups at 128x128) before
      // their MFMAs into a second register set (+32 VGPRs: 212 at 128x128, still 2 workgroups / CU), the barrier
      // sits BETWEEN the halves, and the LDS-DMA pieces of step t + 2 go out between the MFMA groups of the second
      // half -- a wave's MFMA stream waits for nothing but the barrier.  Same summation order as the plain loop
      // (bit-identical results).  Same-box medians, tools/gemm_ab.py: 8192^3 70.1 -> 73.0 TFLOP/s, M = 30720 lower
      // K = 2048 (C3's first trailing update) 68.6 -> 70.3, M = 61440 lower K = 1024 66.7 -> 68.7; 64x64 tiles:
      // 8192^2 x 2048 66.6 -> 68.1, M = 6656 lower K = 1536 (C2's) 62.7 -> 65.1, 30912 x 128 x 1920 (in-panel at C3)
      // 57.0 -> 60.0.  (Only moving the DMA issue between the MFMAs, without the register double-buffering, changed
      // nothing: 68.2 vs 68.2; and with every K-step re-reading L2-resident lines -- a timing-only build -- the
      // pipelined loop gains another 0.5 %: neither the DMA issue slots nor the fabric are what is left.)
      static_assert(DMA && !BLOW && PER_WAVE % TM == 0, "pipelined loop: LDS-DMA staging, whole pieces per MFMA group");
      d2 a0[TM], b0[TN], a1[TM], b1[TN];
      auto read_ops = [&](int s, int kg8, d2 (&a)[TM], d2 (&b)[TN]) {
        const char* base = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const d2*>(base + (wave_m * TM + i) * 2048 + (kg8 ? (roff0 ^ 64) : roff0));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const d2*>(base + (A_BLOCKS + (wave_n * TN + j) * 2) * 1024 + (kg8 ? (roff0 ^ 64) : roff0));
      };
      // MFMA groups [i_lo, i_hi) of one half K-step (a group = one row of 16x16 tiles = 2 TN MFMAs); `issue`: PG
      // LDS-DMA pieces of K-step `tnext` behind each group
      auto mfma_groups = [&](d2 (&a)[TM], d2 (&b)[TN], int i_lo, int i_hi, auto issue_c, int tnext, int snext) {
        constexpr bool issue = decltype(issue_c)::value;
        constexpr int PG = PER_WAVE / TM;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (i < i_lo || i >= i_hi) continue;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          }
          if constexpr (issue) {
#pragma unroll
            for (int q = 0; q < PG; ++q) stage_issue_piece(i * PG + q, tnext, snext);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      // one K-step.  The fragments of a half are requested right after the FIRST MFMA group of the half before, so
      // the wait in front of a half only ever covers reads that are 3 groups old (the compiler's lgkmcnt(0) is free)
      auto kstep = [&](int t, auto issue_c, bool read_next) {
        const int s = t & 1;
        mfma_groups(a0, b0, 0, 1, std::false_type{}, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_ops(s, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_groups(a0, b0, 1, TM, std::false_type{}, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();            // step t + 1 has landed; every wave has its fragments of stage s in registers
        mfma_groups(a1, b1, 0, 1, issue_c, t + 2, s);
        __builtin_amdgcn_sched_barrier(0);
        if (read_next) read_ops(s ^ 1, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_groups(a1, b1, 1, TM, issue_c, t + 2, s);
        __builtin_amdgcn_sched_barrier(0);
      };
      // THIN tiles: at most 16 of the tile's BM rows are rows of the matrix -- the last tile row of a factorisation's
      // trailing update, which holds the e right-hand-side rows carried below the matrix (1 of 57 tile rows of C2's first
      // K = 1024 update).  Only the first 16-row block of the first wave row has anything to compute: the other MFMAs are
      // skipped (plain K loop; the staging and the barriers stay), which makes such a tile ~7x cheaper than a full one.
      const bool thin = p.thin && (p.M - m0) <= 16 && p.lower != 3;
      if (thin) {
        if (t0 < nk) stage_issue(t0, t0 & 1);
        for (int t = t0; t < nk; ++t) {
          const int s = t & 1;
          __syncthreads();
          if (t + 1 < nk) stage_issue(t + 1, s ^ 1);
          if (wave_m == 0) {
            const char* base = smem + s * STAGE;
#pragma unroll
            for (int kg8 = 0; kg8 < 2; ++kg8) {
              const d2 av = *reinterpret_cast<const d2*>(base + (kg8 ? (roff0 ^ 64) : roff0));
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const d2 bv = *reinterpret_cast<const d2*>(base + (A_BLOCKS + (wave_n * TN + j) * 2) * 1024 + (kg8 ? (roff0 ^ 64) : roff0));
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, acc[0][j], 0, 0, 0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (t0 < nk) {
        stage_issue(t0, t0 & 1);
        __syncthreads();
        read_ops(t0 & 1, 0, a0, b0);
        if (t0 + 1 < nk) stage_issue(t0 + 1, (t0 + 1) & 1);
        int t = t0;
        for (; t + 2 < nk; ++t) kstep(t, std::true_type{}, true);
        for (; t < nk; ++t) kstep(t, std::false_type{}, t + 1 < nk);   // last two steps: nothing left to issue
      }
    } else {
      if (t0 < nk) {
        stage_issue(t0, t0 & 1);
        stage_commit(t0 & 1);
      }
      for (int t = t0; t < nk; ++t) {
        const int s = t & 1;
        __syncthreads();  // K-step t has landed (vmcnt(0)); every wave is done reading stage s^1
        if (t + 1 < nk) stage_issue(t + 1, s ^ 1);
        compute(s, t * BK);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) stage_commit(s ^ 1);
      }
    }
  } else {
    // Deep LDS-DMA ring for latency-bound launches (few, small workgroups): NS-1 K-steps
    // stay in flight; the wait for step t is a COUNTED vmcnt that leaves the younger steps
    // outstanding, then a raw s_barrier (a __syncthreads() would drain the ring: vmcnt(0)).
    // Every wave issues exactly PER_WAVE DMAs per K-step, so the count is exact.
    static_assert(DMA, "the ring is LDS-DMA only");
    static_assert(NBLK % NWAVES == 0, "equal DMA count per wave");
    constexpr int AHEAD = NS - 1;
    for (int t = t0; t < min(nk, t0 + AHEAD); ++t) stage_issue(t, (t - t0) % NS);
    for (int t = t0; t < nk; ++t) {
      const int slot = (t - t0) % NS;
      if (t + AHEAD <= nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE * (AHEAD - 1)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (t + AHEAD < nk) stage_issue(t + AHEAD, (t - t0 + AHEAD) % NS);   // slot last read in step t-1
      compute(slot, t * BK);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: reg r of lane l is C[(l>>4) + 4r][l&15] of its 16x16 tile
  const int crow = lane >> 4, ccol = lane & 15;
  const bool diag_tile = p.lower == 3 ? stair_diag_tile : p.lower == 5 ? (ti == (tj >> 1)) : (p.lower && (ti == tj));     // (lower == 4 included)
  // beta != 0: ALL loads of one row of 16x16 tiles are issued before the first use (one HBM
  // round trip per TN*4 elements); element-by-element load -> fma -> store serialises 16+ round
  // trips per tile, which is most of the run time of a K = 128 update
  const bool use_c = p.beta != 0.0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    double cold[TN][4];
    bool ok[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wave_n * WN + j * 16 + ccol;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave_m * WM + i * 16 + crow + 4 * r;
        ok[j][r] = row < p.M && col < p.N && (!diag_tile || col + diag_off <= row);
        cold[j][r] = 0.0;
        if (use_c && ok[j][r]) cold[j][r] = p.C[(int64_t)row * p.ldc + col];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wave_n * WN + j * 16 + ccol;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave_m * WM + i * 16 + crow + 4 * r;
        if (ok[j][r]) p.C[(int64_t)row * p.ldc + col] = fma(p.beta, cold[j][r], p.alpha * acc[i][j][r]);
      }
    }
  }
}

// A/B switches exist only in the tools' build (libgpnative_dbg.so, -DGPN_DEBUG_SWITCHES): per calling thread, so two
// threads of one process can hold different variants and a setter never races another thread's launches.  The product
// library is compiled with the defaults as constants and exports no setter.
#ifdef GPN_DEBUG_SWITCHES
static thread_local int g_thin_tiles = 1;    // (A/B of the thin-tile path; reaches the kernels through GemmArgs)
static thread_local int g_smem_pad = 0;      // debug: extra dynamic LDS per workgroup (KiB) to lower the occupancy
#else
static constexpr int g_thin_tiles = 1;
static constexpr int g_smem_pad = 0;
#endif

// workgroups of a staircase launch with b x b tiles
static int64_t stair_tiles(int64_t M, int64_t N, int st_blk, int st_step, int st_diag, int64_t b) {
  const int64_t mt = (M + b - 1) / b, bt = st_blk / b, nb = N / st_blk, stept = st_step / b;
  int64_t total = 0;
  for (int64_t k = 0; k < nb; ++k) {
    const int64_t rows_t = std::max<int64_t>(mt - k * stept, 0);
    total += rows_t * bt - ((st_diag && rows_t >= bt) ? bt * (bt - 1) / 2 : 0);
  }
  return total;
}

template <int BM, int BN, int WM, int WN, bool DMA, int NS = 2, bool BLOW = false, bool PIPE = false>
static int launch(hipStream_t s, const GemmArgs& a0, int inplace = 0) {
  GemmArgs a = a0;
  a.mt = (a.M + BM - 1) / BM;
  a.nt = (a.N + BN - 1) / BN;
  if (a.lower == 3 && (BM != BN || a.st_blk % BN || a.st_step % BM)) return GPN_E_UNSUPPORTED;
  const int grid = (a.lower == 3   ? (int)stair_tiles(a.M, a.N, a.st_blk, a.st_step, a.st_diag, BM)
                    : a.lower == 2 ? (a.mt - a.nt) * a.nt + a.nt * (a.nt + 1) / 2
                    : a.lower == 4 ? 4 * a.q_cnt
                    : a.lower == 5 ? a.mt * (a.mt + 1)
                    : a.lower      ? (a.q_cnt > 0 ? a.q_cnt : a.mt * (a.mt + 1) / 2) : a.mt * a.nt) * std::max(1, a.batch);
  if (grid <= 0) return GPN_OK;
  const int smem = ((BM + BN) / 16) * 2 * 1024 * NS + (g_smem_pad + a.lds_pad_kb) * 1024;
  auto kern = gemm_nt_kernel<BM, BN, WM, WN, DMA, NS, BLOW, PIPE>;
  static std::atomic<int> attr_set{-1};      // per template instance: the largest size asked for so far (the attribute is a maximum)
  if (attr_set.load(std::memory_order_acquire) < smem) {
    GPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set.store(smem, std::memory_order_release);
  }
  const bool prof = profile_on();
  int rec = -1;
  if (prof) {
    // executed flops: tiles actually computed x 2*BM*BN*K
    const double tiles = (a.lower == 3   ? (double)stair_tiles(a.M, a.N, a.st_blk, a.st_step, a.st_diag, BM)
                          : a.lower == 2 ? (double)(a.mt - a.nt) * a.nt + 0.5 * a.nt * (a.nt + 1.0)
                          : a.lower == 4 ? 4.0 * a.q_cnt
                          : a.lower == 5 ? (double)a.mt * (a.mt + 1.0)
                          : a.lower      ? (a.q_cnt > 0 ? (double)a.q_cnt : 0.5 * a.mt * (a.mt + 1.0)) : (double)a.mt * a.nt) * std::max(1, a.batch);
    const int cls = inplace ? PROF_GEMM_SOLVE : (a.tri ? PROF_GEMM_TRI : ((a.lower == 1 || a.lower == 4 || a.lower == 5) ? PROF_GEMM_SYRK : PROF_GEMM));
    rec = profile_begin(s, tiles * 2.0 * BM * BN * (double)a.K, cls);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (BM / WM) * (BN / WN)), smem, s, a);
  if (prof) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

#ifdef GPN_DEBUG_SWITCHES
static thread_local int g_gemm_variant = 0;  // 0 = LDS-DMA staging, 1 = register staging, 3..6 = forced tile shapes (debug/A-B)
static thread_local int g_group_h = 0;       // 0 = from GPN_GEMM_GROUP_H at first use (default 8): A/B of the grouped tile order's L2 reuse
static thread_local int g_big_tile_min_trapezoid = 4096;   // trapezoid launches (nested panels): 128x128 tiles from this many of them
static thread_local int g_split_tail = 0;    // 1 = the partial last round of a big lower-tile launch as quarter tiles (measured neutral: off)
static thread_local int g_tri_big_k = 0, g_tri_big_tiles = 0;   // A/B: K-clipped launches of a lock-step batch on 128x128 tiles from this K / tile count
#else
static constexpr int g_gemm_variant = 0;
static constexpr int g_big_tile_min_trapezoid = 4096;
static constexpr int g_group_h = 8;
static constexpr int g_split_tail = 0;
static constexpr int g_tri_big_k = 0, g_tri_big_tiles = 0;
#endif

struct Stair { int blk = 0, step = 0, diag = 0; };
struct Outer { int count = 0; int64_t sA = 0, sB = 0, sC = 0; };     // second batch level (count == 0: none)
static int gemm_nt_impl(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb,
                        double beta, double* C, int64_t ldc, int lower, int tri, int inplace,
                        int batch, int64_t sA, int64_t sB, int64_t sC, Stair st = Stair(), int lds_pad_kb = 0, Outer ob = Outer());

int gemm_nt(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb,
            double beta, double* C, int64_t ldc, int lower, int tri, int inplace, int lds_pad_kb) {
  return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, inplace, 1, 0, 0, 0, Stair(), lds_pad_kb);
}

int gemm_nt_batched(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                    const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                    double beta, double* C, int64_t ldc, int64_t sC, int tri, int batch) {
  return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, 0, tri, 0, batch, sA, sB, sC);
}

int gemm_nt_strided(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                    const double* A, int64_t lda, const double* B, int64_t ldb,
                    double beta, double* C, int64_t ldc, int lower, int tri, int inplace,
                    int batch, int64_t sA, int64_t sB, int64_t sC) {
  return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, inplace, batch, sA, sB, sC);
}

int gemm_nt_strided2(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                     const double* A, int64_t lda, const double* B, int64_t ldb,
                     double beta, double* C, int64_t ldc, int lower, int tri,
                     int inner, int64_t sA, int64_t sB, int64_t sC, int outer, int64_t sA2, int64_t sB2, int64_t sC2) {
  if (inner <= 1) return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, 0, outer, sA2, sB2, sC2);
  if (outer <= 1) return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, 0, inner, sA, sB, sC);
  Outer ob;
  ob.count = outer; ob.sA = sA2; ob.sB = sB2; ob.sC = sC2;
  return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, 0, inner, sA, sB, sC, Stair(), 0, ob);
}

int gemm_nt_stair(hipStream_t s, int64_t M, int64_t nblocks, int64_t blk, int64_t K, double alpha,
                  const double* A, int64_t lda, const double* B, int64_t ldb,
                  double beta, double* C, int64_t ldc, int64_t step, int diag) {
  Stair st;
  st.blk = (int)blk; st.step = (int)step; st.diag = diag ? 1 : 0;
  return gemm_nt_impl(s, M, nblocks * blk, K, alpha, A, lda, B, ldb, beta, C, ldc, 3, 0, 0, 1, 0, 0, 0, st);
}

static int gemm_nt_impl(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb,
                        double beta, double* C, int64_t ldc, int lower, int tri, int inplace,
                        int batch, int64_t sA, int64_t sB, int64_t sC, Stair st, int lds_pad_kb, Outer ob) {
  if (M <= 0 || N <= 0 || batch <= 0) return GPN_OK;
  GemmArgs a;
  a.batch = batch; a.sA = sA; a.sB = sB; a.sC = sC;
  a.inner = 0; a.sA2 = a.sB2 = a.sC2 = 0;
  if (ob.count > 0) {
    a.inner = batch; a.sA2 = ob.sA; a.sB2 = ob.sB; a.sC2 = ob.sC;
    a.batch = batch = batch * ob.count;
  }
  a.A = A; a.B = B; a.C = C;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.mt = a.nt = 0;
  a.lower = lower;
  a.st_blk = st.blk; a.st_step = st.step; a.st_diag = st.diag;
  a.q_off = a.q_cnt = a.q_mt = 0;
  a.lds_pad_kb = lds_pad_kb;
#ifdef GPN_DEBUG_SWITCHES
  if (g_group_h == 0) { const char* e = getenv("GPN_GEMM_GROUP_H"); g_group_h = e ? atoi(e) : 8; if (g_group_h < 1 || g_group_h > 64) g_group_h = 8; }
#endif
  a.group_h = g_group_h;
  a.thin = g_thin_tiles;
  a.tri = tri;
  a.alpha = alpha; a.beta = beta;
  // Tile choice (same-box sweeps, tools/gemm_ab.py).  The big tile is 128x128 as EIGHT waves of 32x64 (2 workgroups
  // / CU = 4 waves / SIMD, 126 VGPRs, 64 KB LDS, half the L2->LDS traffic per flop of the small tile): with the
  // pipelined loop 8192^3 72.9 TFLOP/s, M = 30720 lower K = 2048 (C3's first trailing update) 72.0, M = 61440 lower
  // K = 1024 71.1 -- against 72.5 / 69.9 / 68.1 for the same tile as four 64x64 waves (2 / SIMD) and 71.4 / 70.6 for
  // eight 64x32 waves.  It wins once there are >= 8 rounds of its 512 slots; below that the 64x64 tile's (5 / CU,
  // 1280 slots) better tail quantisation wins: lower K = 2048 M = 8192 64.7 vs 67.3, 10240 70.3 vs 68.6, 12288 68.5
  // vs 67.4.  Rules that also priced the partial last round of the 512 slots measured neutral or worse on whole
  // evaluations (tools/workload_ab.py: C3 184.3 vs 184.5 ms, C2 6.85 vs 6.79, C5 600 vs 563 with an earlier form).
  auto tiles = [&](int64_t b) {
    const int64_t mt = (M + b - 1) / b, nt = (N + b - 1) / b;
    if (lower == 3) return stair_tiles(M, N, st.blk, st.step, st.diag, b);
    return (lower == 2 ? (mt - nt) * nt + nt * (nt + 1) / 2 : lower ? mt * (mt + 1) / 2 : mt * nt) * batch;
  };
  const int64_t t128 = tiles(128), t64 = tiles(64);
  // K-clipped launches (tri != 0) have uneven tiles, so the finer grain wins longer.  U U^T (lower):
  // N = 8192 3.02 (64) vs 3.12 ms (128), N = 12288 10.3 vs 9.8, N = 16384 25.3 vs 22.7; the
  // triangular inversion's rectangular products stay on 64x64 tiles up to N = 16384 (28.0 vs 29.2 ms).
  const bool tri_big_ab = g_tri_big_k > 0 && batch > 1 && K >= g_tri_big_k && t128 >= g_tri_big_tiles && M > 64 && N > 64;
  const bool small = tri ? !((K >= 8192 && t128 >= (lower ? 4096 : 8192) && M > 64 && N > 64) || tri_big_ab)
                         : !(K >= 512 && t128 >= (lower == 2 ? g_big_tile_min_trapezoid : 4096) && M > 64 && N > 64);
  if (inplace) {
    // C aliases A (panel solve against an inverted leaf block): one column tile must cover
    // the whole N and K extent of its rows -- a workgroup only stores after its last load
    if (N > 128 || K > 128 || lower) return GPN_E_UNSUPPORTED;
    if (g_gemm_variant == 2) return launch<64, 128, 32, 64, true>(s, a, 1);
    return (tri & GPN_TRI_B_LOWER) ? launch<32, 128, 16, 64, true, 4, true>(s, a, 1) : launch<32, 128, 16, 64, true, 4>(s, a, 1);
  }
  if (g_gemm_variant == 3) return launch<128, 128, 64, 64, true>(s, a);        // A/B: force a tile shape (3..6: the plain K loop)
  if (g_gemm_variant == 4) return launch<64, 64, 32, 32, true>(s, a);
  if (g_gemm_variant == 5) return launch<64, 64, 32, 32, true, 8>(s, a);
  if (g_gemm_variant == 6) return launch<32, 32, 16, 16, true, 8>(s, a);
  if (g_gemm_variant == 7) return launch<128, 128, 64, 64, true, 2, false, true>(s, a);   // A/B: pipelined K loop
  if (g_gemm_variant == 8) return launch<64, 64, 32, 32, true, 2, false, true>(s, a);
  if (g_gemm_variant == 9) return launch<128, 128, 64, 32, true, 2, false, true>(s, a);   // 8 waves x (64x32), pipelined
  if (g_gemm_variant == 10) return launch<128, 128, 64, 32, true, 2, false, false>(s, a); // 8 waves x (64x32), plain loop
  if (g_gemm_variant == 11) return launch<128, 128, 32, 64, true, 2, false, true>(s, a);  // 8 waves x (32x64), pipelined
#ifdef GPN_DEBUG_SWITCHES
  // round-3 review item 8: a 256 x 128 macro tile (16 waves of 32 x 64, ONE workgroup / CU, 96 KB LDS: 25 % fewer operand
  // bytes per flop from L2) against the 128 x 128 tile, both with the plain K loop (the pipelined loop needs whole LDS-DMA
  // pieces per MFMA group: 3 pieces per wave here)
  if (g_gemm_variant == 12) return launch<128, 128, 32, 64, true, 2, false, false>(s, a);
  if (g_gemm_variant == 13) {
    if (a.lower == 1 && !a.tri && a.batch == 1) a.lower = 5;
    else if (a.lower) return GPN_E_UNSUPPORTED;
    return launch<256, 128, 32, 64, true, 2, false, false>(s, a);
  }
#endif
  // skinny products (a handful of rows against a long K, e.g. alpha^T U^T): latency-bound per
  // K-step, so the deep ring and 4x more workgroups pay (131 vs 448 us at 1 x 8192 x 8192)
  // 21: A/B of whole workloads (tools/workload_ab.py): the shipped dispatch with the 4-wave 128x128 kernel
  const bool std_path = g_gemm_variant == 0 || g_gemm_variant == 21;
  if (std_path && (M <= 32 || N <= 32)) return launch<32, 32, 16, 16, true, 8>(s, a);
  if (std_path && (t64 <= 64 || (N <= 128 && t64 <= 128))) {   // (M = 4096, N = 128, K = 128: 8.1 vs 10.9 us)
    // a handful of workgroups: per-CU MFMA rate and DMA latency are the limits -> 4x more,
    // 4x smaller workgroups (32x32 tiles) with 8 K-steps of LDS-DMA in flight
    return launch<32, 32, 16, 16, true, 8>(s, a);
  }
  if (std_path) {
    if (small) return launch<64, 64, 32, 32, true, 2, false, true>(s, a);
    if (g_gemm_variant == 21) return launch<128, 128, 64, 64, true, 2, false, true>(s, a);
    // Partial last round of a lower-tile launch: the 128 x 128 kernel has 512 slots (2 workgroups / CU), so t128 mod 512
    // tiles keep the chip for one whole tile time (0.44 ms at K = 2048) however few they are.  When their 64 x 64 quarters
    // fit ONE round of the small kernel's 1280 slots they go out as a second launch of quarter tiles instead (0.27 ms);
    // every entry keeps its summation order (bit-identical).  C3's ten big trailing updates all qualify (mt a multiple of
    // 16 => t128 mod 512 in 48 .. 248).  Measured NEUTRAL on whole evaluations (C3 186.05 vs 186.28 ms, C4 1363.9 vs 1363.0:
    // the big kernel's last round is not synchronous, tiles are dealt to slots as they free up), so it is OFF in the product
    // and kept behind g_split_tail (tools' build, gemm variant bit 7) with its test.
    const int64_t rem = t128 % 512;
    if (g_split_tail && lower == 1 && batch == 1 && rem > 0 && 4 * rem <= 1280) {
      GemmArgs big = a;
      big.q_cnt = (int)(t128 - rem);
      int rc = launch<128, 128, 32, 64, true, 2, false, true>(s, big);
      if (rc != GPN_OK) return rc;
      GemmArgs q = a;
      q.lower = 4; q.q_off = (int)(t128 - rem); q.q_cnt = (int)rem; q.q_mt = (int)((M + 127) / 128);
      return launch<64, 64, 32, 32, true, 2, false, true>(s, q);
    }
    return launch<128, 128, 32, 64, true, 2, false, true>(s, a);
  }
  return small ? launch<64, 64, 32, 32, false>(s, a) : launch<128, 128, 64, 64, false>(s, a);
}

}  // namespace gpn

extern "C" int gpn_gemm_nt_stair(void* stream, int64_t M, int64_t nblocks, int64_t blk, int64_t K, double alpha,
                                 const double* A, int64_t lda, const double* B, int64_t ldb,
                                 double beta, double* C, int64_t ldc, int64_t step, int diag) {
  if (M < 0) return -2;
  if (nblocks < 0) return -3;
  if (blk <= 0 || (blk % 128)) return -4;
  if (K <= 0 || (K % 16)) return -5;
  if (step < 0 || (step % 128)) return -14;
  if (diag && M < blk) return -15;
  if (diag)      // a lower-only first square needs all blk rows of its block: a block that starts inside the last blk rows
    for (int64_t b = 1; b < nblocks; ++b) {          // would get the rectangular treatment and write above its diagonal
      const int64_t rows = M - b * step;
      if (rows > 0 && rows < blk) return -15;
    }
  if ((lda & 1) || (ldb & 1)) return GPN_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  if (M == 0 || nblocks == 0) return GPN_OK;
  return gpn::gemm_nt_stair(static_cast<hipStream_t>(stream), M, nblocks, blk, K, alpha, A, lda, B, ldb, beta, C, ldc, step, diag);
}

#ifdef GPN_DEBUG_SWITCHES
extern "C" int gpn_debug_set_thin_tiles(int on) { gpn::g_thin_tiles = on; return GPN_OK; }
extern "C" int gpn_debug_set_tri_big(int k, int tiles) { gpn::g_tri_big_k = k; gpn::g_tri_big_tiles = tiles; return GPN_OK; }
extern "C" int gpn_debug_set_big_tile_min_trapezoid(int t) { gpn::g_big_tile_min_trapezoid = t; return GPN_OK; }
extern "C" int gpn_debug_set_gemm_variant(int v) {     // (libgpnative_dbg.so only; the calling thread's launches)
  gpn::g_gemm_variant = v & 0x7f;
  gpn::g_split_tail = (v & 0x80) ? 1 : 0;   // bit 7: quarter-tile launch for the partial last round of big lower-tile launches
  gpn::g_smem_pad = v >> 8;           // bits 8..: KiB of LDS padding per workgroup
  return GPN_OK;
}
#endif

extern "C" int gpn_gemm_nt_batched(void* stream, int64_t M, int64_t N, int64_t K, double alpha,
                                   const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                                   double beta, double* C, int64_t ldc, int64_t sC, int lower, int tri, int batch) {
  if (M < 0) return -2;
  if (N < 0) return -3;
  if (K < 0 || (K % 16) != 0) return -4;
  if (batch < 1) return -18;
  if (M == 0 || N == 0) return GPN_OK;
  if (!A) return -6;
  if (!B) return -9;
  if (!C) return -13;
  if (lower && M != N) return -16;
  if ((lda % 2) || (ldb % 2) || (sA % 2) || (sB % 2)) return GPN_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  return gpn::gemm_nt_impl(static_cast<hipStream_t>(stream), M, N, K, alpha, A, lda, B, ldb, beta, C, ldc,
                           lower ? 1 : 0, tri, 0, batch, sA, sB, sC);
}

extern "C" int gpn_gemm_nt(void* stream, int64_t M, int64_t N, int64_t K, double alpha,
                           const double* A, int64_t lda, const double* B, int64_t ldb,
                           double beta, double* C, int64_t ldc, int lower, int tri) {
  if (M < 0) return -2;
  if (N < 0) return -3;
  if (K < 0 || (K % 16) != 0) return -4;
  if (lower < 0 || lower > 2) return -13;
  if (lower == 1 && M != N) return -13;
  if (lower == 2 && (M < N || tri)) return -13;
  if ((lda & 1) || (ldb & 1)) return GPN_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  if (K == 0) {
    // C = beta*C: degenerate, not on the hot path
    return GPN_E_UNSUPPORTED;
  }
  if (tri < 0 || tri > 15) return -14;
  return gpn::gemm_nt(static_cast<hipStream_t>(stream), M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, 0);
}
