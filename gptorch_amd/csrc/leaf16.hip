// 128x128 diagonal leaf, second generation (round 4): factor + inverse in one workgroup with 16-pivot blocks.
// Replaces torch.cholesky on the diagonal blocks of functions.cholesky (functions.py:46-47); same contract as
// potrf_leaf_kernel<true> (potrf.hip): L = chol(A[0:kb,0:kb]) in place (lower), W = L^-1 -> winv (128 x 128, ld 128, zero
// outside the kb x kb lower triangle), rows / columns >= kb act as identity, LAPACK-style info (first non-positive pivot).
//
// Why a second kernel.  The first leaf hands over between its pivot wave and its tile waves twice per EIGHT pivots and
// keeps LDS-crossbar round trips (ds_swizzle / ds_bpermute) on the pivot chain: ~725 cycles per pivot, 40.6 us per leaf,
// 38 % of a C2 evaluation on one CU.  Here:
//   * the 128 x 128 block is 8 x 8 tiles of 16 x 16, stacked on the identity ([A ; I], 16 x 8 tiles) so that the
//     identity rows come out as I L^-T = W^T (as before);
//   * every tile lives in registers in TRANSPOSED storage: reg r of lane (g = lane >> 4, c = lane & 15) = T[c][g + 4 r].
//     In that storage a tile's four registers ARE the v_mfma_f64_16x16x4_f64 operand fragments of the tile (k index
//     g + 4 r: a permutation of 0..15 shared by both operands), and the product W T^T of two such operands comes out in
//     the same storage again -- so  solve  X = T W_k^T  (4 MFMAs)  and  update  T_ij -= X_ik X_jk^T  (4 MFMAs)  chain
//     with NO layout change; tiles only cross waves as plain register dumps through LDS;
//   * the PIVOT wave (wave 8) factors one 16 x 16 diagonal block per step with ONE ROW PER LANE (lanes 0..15 the rows
//     of D_k, lanes 16..31 the identity rows -> W_k^T): per pivot a readlane broadcast of the pivot, v_rsq_f64 + one
//     cubic correction, and per remaining column one readlane pair + one FMA -- nothing on the chain goes through LDS;
//   * the pivot wave runs ahead by itself: after block k it forms the next diagonal block from two RAW tiles the tile
//     waves published one panel earlier,  X = A(k+1,k) W_k^T,  D(k+1) -= X X^T  (8 MFMAs),  so it never waits for the
//     tile waves' update;
//   * ONE hardware barrier per 16 pivots (W_k out, raw tiles in) + one software barrier of the 8 tile waves.
// Deterministic (fixed summation order).  FACTOR=false (inverse of a given L) stays on the first kernel.
#include "gpn_common.h"

#ifndef L16_COALESCED_STORE
#define L16_COALESCED_STORE 1
#endif
// Lanes of ONE wave exchange data through LDS without a barrier: the hardware executes a wave's LDS instructions in
// order, but the COMPILER reasons per thread -- it may prove that a lane's own stores and loads never overlap and hoist
// the loads above the stores (it did, in the prologue's layout change: lanes then read the previous tile).  This fence
// pins the program order of memory instructions; it emits no code.
#define L16_WAVE_FENCE() asm volatile("" ::: "memory")

namespace gpn {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// 12 waves: the hardware deals a workgroup's waves round-robin over the CU's 4 SIMDs, so waves {0,4,8} share one.  Wave 0 is
// the PIVOT wave and waves 4 and 8 stay idle (barriers only): fp64 MFMAs and fp64 vector instructions of one SIMD share
// the DP pipe, and a tile wave's 64-cycle MFMAs next to the pivot wave stretched its dependent chain from ~100 to ~370
// cycles per pivot (measured: tools/lat_bench.hip alone vs the s_memtime stamps of the 9-wave version).  The 8 tile rows
// go to the other three SIMDs by cost: {0,7} | {1,3,5} | {2,4,6} (56 tile updates each).
constexpr int L16_THREADS = 768;
constexpr int TS = 18;                // row stride of the prologue's staging tiles (16-B aligned rows, conflict-free transposed reads)
constexpr int RS = 18;                // row stride (doubles) of the row-major 16 x 16 blocks in LDS (144 B: 16-B aligned rows)

struct Leaf16Args {
  double* A;
  int64_t lda;
  int kb, col0;
  double* winv;
  int32_t* info;
  int64_t sA, sW, sInfo;              // per-workgroup strides (elements): blockIdx.x-th problem of a batch
};

// DIAG build: a timeline -- diag[(wave * 8 + k) * 8 + ev] = s_memtime at event ev of block k, pinned behind the value `tie`
// (the stamp waits for every outstanding LDS operation of the wave: it perturbs what it measures a little)
#define L16_TU(k, j, tie)                                                                                    \
  if constexpr (DIAG) {                                                                                      \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(tie) :: "memory");                  \
    if (lane == 0) diag[768 + 18432 + (wave * 8 + (k)) * 8 + (j)] = t_;                                      \
  }
#define L16_TL(k, ev, tie)                                                                                   \
  if constexpr (DIAG) {                                                                                      \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(tie) :: "memory");                  \
    if (lane == 0) diag[(wave * 8 + (k)) * 8 + (ev)] = t_;                                                   \
  }

template <bool DIAG>
__global__ __launch_bounds__(L16_THREADS) void potrf_leaf16_kernel(Leaf16Args p, unsigned long long* diag) {
  int tie0 = 0;
  if constexpr (DIAG) {
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    if ((threadIdx.x & 63) == 0) diag[((threadIdx.x >> 6) * 8 + 0) * 8 + 6] = t_;          // kernel entry
  }
  double* A = p.A + (int64_t)blockIdx.x * p.sA;
  double* winv = p.winv + (int64_t)blockIdx.x * p.sW;
  int32_t* info = p.info ? p.info + (int64_t)blockIdx.x * p.sInfo : nullptr;
  const int64_t lda = p.lda;
  const int kb = p.kb;

  __shared__ double Xbuf[8][4][64];       // solved panel tiles X(j,k), j > k (A part): register dumps, read as A operands
  __shared__ double Raw[2][2][4][64];     // [parity][0: A(k+1,k), 1: D(k+1,k+1)][reg][lane]: raw tiles for the pivot wave
  __shared__ double Wf[2][16 * RS];       // W_k, fragment-ready: Wf[RS * m + c] = W_k[c][m]
  __shared__ double Lrow[2][16 * RS];     // L_k, row-major
  __shared__ double Drow[32 * RS];        // rows 0..15: the pivot wave's next block (row-major); rows 16..31: identity
  __shared__ __attribute__((aligned(16))) double Tb[8][16 * TS];   // per tile wave: staging tile of the prologue's layout change
  __shared__ double Lcol[64];              // the pivot wave's current column, for the broadcast reads
  __shared__ int tb_count;
  __shared__ int failflag;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> role: 0 pivot; 4, 8, 11 idle; tile row of the others
  const int rowmap = (wave == 1) ? 1 : (wave == 5) ? 3 : (wave == 9) ? 5 : (wave == 2) ? 2 : (wave == 6) ? 4 : (wave == 10) ? 6 :
                     (wave == 3) ? 0 : (wave == 7) ? 7 : -1;
  const bool tilewave = rowmap >= 0;
  const bool pivotwave = wave == 0;
  const int w = rowmap & 7;
  const int g = lane >> 4, lc = lane & 15;
  if (tid == 0) { failflag = 0; tb_count = 0; }
  for (int idx = tid; idx < 16 * RS; idx += L16_THREADS) Drow[16 * RS + idx] = ((idx / RS) == (idx % RS)) ? 1.0 : 0.0;

  auto bcast = [](double v, int src) -> double {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  };

  if (tilewave) {
    // ======================================= tile waves =======================================
    // slot J (J <= w): A tile (w, J); slot J + 1 (J >= w): identity tile (8 + w, J).  Transposed storage.
    d4 acc[9];
    L16_TL(1, 6, tie0)
    {
      // One memory round trip, COALESCED and 16 bytes per lane: a tile is fetched row-major by two instructions (lane l:
      // row (l >> 3) + 8 i, columns 2 (l & 7), + 1 -- 8 consecutive lanes = 128 consecutive bytes), only the tiles on and left
      // of the diagonal, every load issued before the first use; then turned into the transposed storage through a per-wave
      // 16 x 18 LDS tile.  (Fetching the transposed storage directly puts consecutive lanes on different rows: 64
      // transactions per load instruction, 13-21 k cycles of prologue; the memory pipe takes 16 cycles per wave
      // instruction whatever its width, so 8-byte loads of all 8 tile columns still cost 4 k cycles CU-wide.)
      double* tb = Tb[w];
      const int pr = lane & 7;
      d2 ld[8][2];
      bool rok[2];
      int rl[2];
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        rl[i2] = (lane >> 3) + 8 * i2;
        const int row = 16 * w + rl[i2];
        rok[i2] = row < kb;
        const double* rowp = A + (int64_t)(rok[i2] ? row : 0) * lda + 2 * pr;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (q <= w) ld[q][i2] = *reinterpret_cast<const d2*>(rowp + 16 * q);      // (uniform branch, no use inside)
        }
      }
      L16_TL(2, 6, tie0)
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q < 8 && q < w) {                                         // A tile left of the diagonal (uniform branch)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const d2 v = rok[i2] ? ld[q < 8 ? q : 0][i2] : d2{0.0, 0.0};
            *reinterpret_cast<d2*>(&tb[rl[i2] * TS + 2 * pr]) = v;
          }
          L16_WAVE_FENCE();
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[q][r] = tb[lc * TS + g + 4 * r];
          L16_WAVE_FENCE();
        } else if (q < 8 && q == w) {                                 // diagonal tile: symmetric fill from the lower triangle
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const d2 vv = ld[q < 8 ? q : 0][i2];
            const int a_ = rl[i2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int b_ = 2 * pr + h;
              if (b_ <= a_) {
                const double v = rok[i2] ? vv[h] : (a_ == b_ ? 1.0 : 0.0);                   // identity beyond kb
                tb[a_ * TS + b_] = v;
                tb[b_ * TS + a_] = v;
              }
            }
          }
          L16_WAVE_FENCE();
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[q][r] = tb[lc * TS + g + 4 * r];
          L16_WAVE_FENCE();
        } else {                                                      // identity tiles: (8 + w, w) = I, the others 0
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[q][r] = (q == w + 1 && lc == g + 4 * r) ? 1.0 : 0.0;
        }
      }
    }
    L16_TL(3, 6, acc[0])
    if constexpr (DIAG) {                                             // debug: the tiles as loaded (after the 768 stamps)
      double* dbg = reinterpret_cast<double*>(diag + 768);
#pragma unroll
      for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbg[((w * 9 + q) * 4 + r) * 64 + lane] = acc[q][r];
    }
    auto dump = [&](double* dst, const d4& t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r * 64 + lane] = t[r];
    };
    // raw tiles for the pivot wave's first steps: D(0,0) from wave 0, A(1,0) and D(1,1) from wave 1
    if (w == 0) dump(&Raw[0][1][0][0], acc[0]);
    if (w == 1) { dump(&Raw[1][0][0][0], acc[0]); dump(&Raw[1][1][0][0], acc[1]); }
    L16_TL(4, 6, tie0)
    __syncthreads();                                                  // P

    // software barrier of the 8 tile waves, split: ARRIVE right after the wave's X tile is in LDS, WAIT after its global
    // stores -- the slowest wave's stores (the diagonal tile's masked rows) used to hold everybody for ~1.6 k cycles
    int epoch = 0;
    auto tb_arrive = [&]() {
      ++epoch;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(&tb_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto tb_wait = [&]() {
      int spins = 0;
      while (__hip_atomic_load(&tb_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { failflag = LEAF + 1; break; }
      }
      asm volatile("" ::: "memory");
    };
    // T(i,j) -= X(i,k) X(j,k)^T:  xa = X(j,k) fragments, nx = -X(i,k) (own registers)
    auto update = [&](d4& t, const d4& xa, const d4& nx) {
#pragma unroll
      for (int r = 0; r < 4; ++r) t = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[r], nx[r], t, 0, 0, 0);
    };
    auto load_frag = [&](const double* src) -> d4 {
      d4 f;
#pragma unroll
      for (int r = 0; r < 4; ++r) f[r] = src[r * 64 + lane];
      return f;
    };

    // The panel loop is fully UNROLLED: the tile registers are reached through wave-uniform branches on static slots, and with
    // k a run-time value those branches turn every accumulator update into MFMA-to-temporary + copy-back behind a pipeline
    // drain (measured: the rolled loop ran the update phase at 58 % of the MFMA rate, the unrolled one at 78 %).  The code
    // is 20+ KB of straight line, run once per wave: fine with a warm instruction cache.
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      L16_TL(k, 0, tie0)
      __syncthreads();                                                // B(k): W_k and L_k are out
      L16_TL(k, 1, tie0)
      if (failflag) break;                                            // uniform
      d4 wf;
#pragma unroll
      for (int r = 0; r < 4; ++r) wf[r] = Wf[k & 1][RS * (g + 4 * r) + lc];
      const bool apart = w > k;                                       // my tile of column k: A tile (w, k) or identity tile (8 + w, k)
      const int slot = apart ? k : k + 1;
      // X = T W_k^T (transposed storage both sides); the tile's registers are dead afterwards
      d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (q == slot) {
#pragma unroll
          for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[r], acc[q][r], x, 0, 0, 0);
        }
      }
      L16_TL(k, 2, x)
      if (apart && k < 7) dump(&Xbuf[w][0][0], x);
      if (k < 7) tb_arrive();
      if (apart) {
        // L tile (w, k), stored row-major (coalesced): element X[g + 4 r][lc], read back from my register dump in Xbuf
        // (or, for the last panel, straight from a dump made for the purpose)
#if L16_COALESCED_STORE
        if (k == 7) dump(&Xbuf[w][0][0], x);
        L16_WAVE_FENCE();
        const double* xd = &Xbuf[w][0][0] + (lc >> 2) * 64 + (lc & 3) * 16 + g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + g + 4 * r;
          const double v = xd[4 * r];
          if (row < kb) A[(int64_t)row * lda + 16 * k + lc] = v;
        }
#else
        const int row = 16 * w + lc;
        if (row < kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) A[(int64_t)row * lda + 16 * k + g + 4 * r] = x[r];
        }
#endif
      } else {                                                        // W^T tile (w, k): X[a][b] = W[16 k + b][16 w + a]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wr = 16 * k + g + 4 * r, wc = 16 * w + lc;
          winv[(int64_t)wr * LEAF + wc] = (wr < kb && wc < kb) ? x[r] : 0.0;
        }
        if (w == k) {                                                 // the diagonal tile L_k comes from the pivot wave
          const double* Lr = Lrow[k & 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r, col = lc;
            if (col <= row && 16 * k + row < kb) A[(int64_t)(16 * k + row) * lda + 16 * k + col] = Lr[row * RS + col];
          }
        }
      }
      const d4 nx = d4{-x[0], -x[1], -x[2], -x[3]};
      L16_TL(k, 3, tie0)
      if (k == 7) break;
      tb_wait();                                                      // T(k): the X tiles of this panel are in LDS
      L16_TL(k, 4, tie0)
      {
        // A operand X(j,k) of the NEXT tile requested before the current tile's MFMAs (rolling prefetch)
        d4 xa_n = load_frag(&Xbuf[k + 1][0][0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) {
          if (j > k) {                                                // uniform
            const d4 xa = xa_n;
            if (j < 7) xa_n = load_frag(&Xbuf[j + 1][0][0]);
            if (apart) {                                              // A tiles (w, j), j = k + 1 .. w
              if (j < w) {
                update(acc[j], xa, nx);
                L16_TU(k, j, acc[j])
              } else if (j == w) {
                update(acc[j], x, nx);
                L16_TU(k, j, acc[j])
              }
              // the raw tiles the pivot wave needs after its NEXT block: A(k+2, k+1) and D(k+2, k+2), from wave k + 2
              if (w == k + 2) {
                if (j == k + 1) dump(&Raw[k & 1][0][0][0], acc[j]);
                if (j == k + 2) dump(&Raw[k & 1][1][0][0], acc[j]);
              }
            } else {                                                  // identity tiles (8 + w, j), j = k + 1 .. 7
              update(acc[j + 1], xa, nx);
              L16_TU(k, j, acc[j + 1])
            }
          }
        }
      }
      if constexpr (DIAG) {
        double chk = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) chk += acc[q][0];
        L16_TL(k, 5, chk)
        if (chk == 1.2345e300) tie0 += 1;
      }
    }
  } else if (!pivotwave) {
    // idle waves (they share the pivot wave's SIMD): the barrier sequence, and -- off everybody's critical path -- the zero
    // fill of winv above the diagonal tiles (28 tiles W[16 k + ..][16 j + ..], k < j; 16 bytes per lane)
    __syncthreads();                                                  // P
    {
      const int me = (wave == 4) ? 0 : (wave == 8) ? 1 : 2;
      int t = 0;
      for (int k = 0; k < 7; ++k)
        for (int j2 = k + 1; j2 < 8; ++j2, ++t) {
          if (t % 3 != me) continue;
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            *reinterpret_cast<d2*>(&winv[(int64_t)(16 * k + (lane >> 3) + 8 * i2) * LEAF + 16 * j2 + 2 * (lane & 7)]) = d2{0.0, 0.0};
        }
    }
    for (int k = 0; k < 8; ++k) {
      __syncthreads();                                                // B(k)
      if (failflag) break;
    }
  } else {
    // ======================================= pivot wave =======================================
    __builtin_amdgcn_s_setprio(3);
    L16_TL(4, 6, tie0)
    __syncthreads();                                                  // P
    L16_TL(5, 6, tie0)
    double a[16];
    const int myrow = lane & 31;
    // block 0: D(0,0) as dumped by wave 0 (transposed storage of a symmetric tile) -> one row per lane
    {
      d4 dacc;
#pragma unroll
      for (int r = 0; r < 4; ++r) dacc[r] = Raw[0][1][r][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) Drow[lc * RS + g + 4 * r] = dacc[r];
      L16_WAVE_FENCE();
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = Drow[myrow * RS + c];
      L16_WAVE_FENCE();
    }
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      L16_TL(k, 0, a[0])
      // ---- 16 pivots, one row per lane; lanes 16..31 carry the identity rows.  Per pivot J only the NEXT column is updated
      // at once (readlane broadcast: it feeds the next pivot); the columns after it take pivot J's rank-1 update one pivot
      // LATER, from an LDS broadcast of the column (one ds_write_b64 + uniform ds_read_b128s issued here, consumed during
      // pivot J + 1): two readlanes per column were what bound the first version (370 cycles per pivot, issue-bound).
      // Every entry still receives its updates in pivot order, so the results are bit-identical to the eager form.
      double sb[16], lprev = 0.0;
#pragma unroll
      for (int J = 0; J < 16; ++J) {
        const double d = bcast(a[J], J);
        // y = d^-1/2 = y0 (1 + e p), e = 1 - d y0^2, p = 1/2 + 3 e / 8; products are formed as x y0 (1 + e p) so that
        // nothing waits for the refined y
        const double y0 = __builtin_amdgcn_rsq(d);
        const double ay0 = a[J] * y0;
        const double e = fma(-d * y0, y0, 1.0);
        const double pp = fma(e, 0.375, 0.5);
        const double l = fma(ay0 * e, pp, ay0);                       // column J: L[i][J] (lane J: sqrt(d))
        a[J] = l;
        if (J >= 1) {                                                 // pivot J - 1's update of the columns J + 1 ..
#pragma unroll
          for (int c = J + 1; c < 16; ++c) a[c] = fma(-lprev, sb[c], a[c]);
        }
        if (J < 15) a[J + 1] = fma(-l, bcast(l, J + 1), a[J + 1]);
        if (J < 14) {
          L16_WAVE_FENCE();
          Lcol[lane] = l;
          L16_WAVE_FENCE();
#pragma unroll
          for (int c = J + 2; c < 16; ++c) sb[c] = Lcol[c];
        }
        lprev = l;
      }
      L16_TL(k, 1, a[15])
      // a failed pivot (d <= 0 or NaN) turns everything after it into NaN, a[15] of row 15 included
      {
        const double last = bcast(a[15], 15);
        if (!(last == last) || fabs(last) > 1.7e308) {
          int first = 0;
#pragma unroll
          for (int J = 15; J >= 0; --J) {
            const double dj = bcast(a[J], J);
            if (!(dj == dj) || fabs(dj) > 1.7e308) first = J + 1;
          }
          if (lane == 0) failflag = 16 * k + first;
        }
      }
      // ---- publish L_k (rows, lanes 0..15) and W_k (lanes 16..31 hold the rows of W_k^T)
      if (lane < 32) {
        double* dst = lane < 16 ? &Lrow[k & 1][lane * RS] : &Wf[k & 1][(lane - 16) * RS];
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c] = a[c];
      }
      L16_TL(k, 2, tie0)
      __syncthreads();                                                // B(k)
      L16_TL(k, 3, tie0)
      if (failflag || k == 7) break;
      // ---- next diagonal block from the raw tiles:  X = A(k+1,k) W_k^T,  D(k+1) -= X X^T
      {
        const int par = (k + 1) & 1;
        d4 wf, ar, dacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          wf[r] = Wf[k & 1][RS * (g + 4 * r) + lc];
          ar[r] = Raw[par][0][r][lane];
          dacc[r] = Raw[par][1][r][lane];
        }
        d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[r], ar[r], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[r], x[r], dacc, 0, 0, 0);
        L16_TL(k, 4, dacc)
#pragma unroll
        for (int r = 0; r < 4; ++r) Drow[lc * RS + g + 4 * r] = dacc[r];
        L16_WAVE_FENCE();
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = Drow[myrow * RS + c];
        L16_WAVE_FENCE();
      }
      L16_TL(k, 5, a[0])
    }
  }
  if constexpr (DIAG) {
    L16_TL(0, 7, tie0)                                                 // end of the wave's work
    if (lane == 0) diag[(wave * 8 + 1) * 8 + 7] = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);   // HW_ID
  }
  __syncthreads();
  if (failflag) {
    if (tid == 0 && info) {
      if (failflag > LEAF) *info = GPN_INFO_INTERNAL;
      else if (*info == 0) *info = p.col0 + failflag;
    }
    for (int idx = tid; idx < LEAF * LEAF; idx += L16_THREADS) winv[idx] = 0.0;
  }
}

// `batch` leaves in one launch: problem b at A + b sA, winv + b sW, info + b sInfo (batch = 1: strides ignored)
int leaf16(hipStream_t s, double* A, int64_t lda, int kb, int col0, double* winv, int32_t* info, int batch, int64_t sA,
           int64_t sW, int64_t sInfo) {
  Leaf16Args a;
  a.A = A; a.lda = lda; a.kb = kb; a.col0 = col0; a.winv = winv; a.info = info; a.sA = sA; a.sW = sW; a.sInfo = sInfo;
  hipLaunchKernelGGL(potrf_leaf16_kernel<false>, dim3((unsigned)batch), dim3(L16_THREADS), 0, s, a, nullptr);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

int leaf16_timing(hipStream_t s, double* A, int64_t lda, double* winv, int32_t* info, unsigned long long* diag768) {
  Leaf16Args a;
  a.A = A; a.lda = lda; a.kb = LEAF; a.col0 = 0; a.winv = winv; a.info = info; a.sA = 0; a.sW = 0; a.sInfo = 0;
  hipLaunchKernelGGL(potrf_leaf16_kernel<true>, dim3(1), dim3(L16_THREADS), 0, s, a, diag768);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn
