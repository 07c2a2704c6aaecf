// The fp64 MFMA "NT" tile of gemm_f64.hip as a device function (shared with the persistent factorisation, ppotrf.hip).
// See gemm_f64.hip for the design notes; this header holds GemmArgs, the blockIdx -> tile maps and gemm_nt_tile.
#pragma once
#include <type_traits>
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
  int lower;        // 0 full, 1 lower-tile square, 2 trapezoid, 3 staircase (st_* below)
  int group_h;      // tile rows per group of the grouped tile order (8)
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
  // WT instances only (the persistent factorisation, ppotrf.hip): a tile's accumulators carried between two tasks -- acc_in:
  // start from these raw sums instead of zero; acc_out: dump the raw sums and skip the epilogue.  128 KB per 128 x 128 tile,
  // element ((wave * TM + i) * TN + j) * 4 + r of lane l at [...] * 64 + l.  NULL: off.  (Ordinary launches never read them.)
  const double* acc_in;
  double* acc_out;
  // strided batch only: problem z scales its product by alpha_dev[z] (device memory) instead of `alpha` -- lock-step models whose
  // scale is a hyper-parameter of the model (the sparse bound's 1 / noise variance).  NULL: off.
  const double* alpha_dev = nullptr;
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

// One output tile.  `bid` of `nwg`: the workgroup's index in the launch's tile enumeration (blockIdx.x of gridDim.x for an ordinary
// launch; the persistent form below walks several).  `direct`: bid IS the logical index of the grouped order (no XCD remap).
// XW: the workgroup has MORE waves than the tile's (BM / WM) (BN / WN) -- the extra waves only keep the barrier count (the
// persistent factorisation's 12-wave workgroups run the 8-wave tile).  WT: the epilogue's stores are agent-scope write-through
// (`sc1`): the tile is handed to other workgroups of the same launch (ppotrf.hip; MI355X_MICROARCH.md "Valid forms").
template <int BM, int BN, int WM, int WN, bool DMA, int NS = 2, bool BLOW = false, bool PIPE = false, bool XW = false, bool WT = false>
__device__ __forceinline__ void gemm_nt_tile(GemmArgs p, int bid, int nwg, const bool direct, const int tid_in = -1) {
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
  const bool spread = direct || (p.tri & (GPN_TRI_A_UPPER | GPN_TRI_A_LOWER | GPN_TRI_B_UPPER)) != 0;
  if (p.batch > 1) {                               // strided batch: consecutive blocks = one problem
    nwg = nwg / p.batch;
    const int z = bid / nwg;
    bid -= z * nwg;
    if (p.inner > 0) {
      const int z2 = z / p.inner, z1 = z - z2 * p.inner;
      p.A += z1 * p.sA + z2 * p.sA2; p.B += z1 * p.sB + z2 * p.sB2; p.C += z1 * p.sC + z2 * p.sC2;
    } else {
      p.A += z * p.sA; p.B += z * p.sB; p.C += z * p.sC;
    }
    if (p.alpha_dev) p.alpha = p.alpha_dev[z];
  } else if (p.alpha_dev) {
    p.alpha = p.alpha_dev[0];
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
  } else if (p.lower) tile_of_block_lower(bid, nwg, p.mt, spread, ti, tj, p.group_h);
  else tile_of_block(bid, nwg, p.mt, p.nt, spread, ti, tj, p.group_h);

  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;       // (tid_in: see leaf16_body.h)
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
  if constexpr (XW) {
    static_assert(NS == 2 && PIPE, "extra waves: the pipelined two-stage loop only");
    if (wave >= NWAVES) {            // same barriers as the tile's waves, nothing else
      const bool thin_x = p.thin && (p.M - m0) <= 16 && p.lower != 3;
      if (thin_x) {
        for (int t = t0; t < nk; ++t) __syncthreads();
      } else if (t0 < nk) {
        __syncthreads();
        for (int t = t0; t < nk; ++t) __syncthreads();
      }
      return;
    }
  }
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
        if constexpr (WT) {
          // the sums so far (see GemmArgs::acc_in), requested behind the first K-step's operands: one round trip for both
          if (p.acc_in) {
            const double* ai = p.acc_in + (int64_t)wave * (TM * TN * 4 * 64) + lane;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = ai[((i * TN + j) * 4 + r) * 64];
          }
        }
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
  const bool diag_tile = p.lower == 3 ? stair_diag_tile : (p.lower && (ti == tj));
  // beta != 0: ALL loads of one row of 16x16 tiles are issued before the first use (one HBM
  // round trip per TN*4 elements); element-by-element load -> fma -> store serialises 16+ round
  // trips per tile, which is most of the run time of a K = 128 update
  const bool use_c = p.beta != 0.0;
  if constexpr (WT) {
    if (p.acc_out) {               // raw sums to scratch (write-through): the tile is finished by a later task (acc_in)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(&p.acc_out[((((int64_t)wave * TM + i) * TN + j) * 4 + r) * 64 + lane], acc[i][j][r], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
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
        if (ok[j][r]) {
          const double v = fma(p.beta, cold[j][r], p.alpha * acc[i][j][r]);
          if constexpr (WT) __hip_atomic_store(&p.C[(int64_t)row * p.ldc + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else p.C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

}  // namespace gpn
