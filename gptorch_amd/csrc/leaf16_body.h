// The 128 x 128 diagonal leaf of leaf16.hip as a device function (shared with the persistent factorisation, ppotrf.hip).
// Design notes: leaf16.hip.
#pragma once
#include <type_traits>
#include "gpn_common.h"

namespace gpn {

#ifndef L16_COALESCED_STORE
#define L16_COALESCED_STORE 1
#endif
// Lanes of ONE wave exchange data through LDS without a barrier: the hardware executes a wave's LDS instructions in
// order, but the COMPILER reasons per thread -- it may prove that a lane's own stores and loads never overlap and hoist
// the loads above the stores (it did, in the prologue's layout change: lanes then read the previous tile).  This fence
// pins the program order of memory instructions; it emits no code.
#define L16_WAVE_FENCE() asm volatile("" ::: "memory")


typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// 12 waves: the hardware deals a workgroup's waves round-robin over the CU's 4 SIMDs, so waves {0,4,8} share one.  Wave 0 is
// the PIVOT wave and waves 4 and 8 stay idle (barriers only): fp64 MFMAs and fp64 vector instructions of one SIMD share
// the DP pipe, and a tile wave's 64-cycle MFMAs next to the pivot wave stretched its dependent chain from ~100 to ~370
// cycles per pivot (measured: tools/lat_bench.hip alone vs the s_memtime stamps of the 9-wave version).  The 8 tile rows
// go to the other three SIMDs by cost: {0,7} | {1,3,5} | {2,4,6} (56 tile updates each).
constexpr int L16_THREADS = 768;
constexpr int TS = 18;                // row stride of the prologue's staging tiles (16-B aligned rows, conflict-free transposed reads)
constexpr int L16_LDS_DOUBLES = 28 * 256 + 2 * 2 * 256 + 2 * 8 * 16 * 18 + 32 * 18 + 64 + 8 * 16 * 18;
// The workgroup asks for 132 KB although it uses 123: with less than 32 KB of the CU's 160 KB left, no workgroup of the
// contraction kernel (32 / 64 KB) or of the column kernel (33 KB) running on another stream can move in beside the leaf --
// a co-resident wave on the pivot wave's SIMD would stall its fp64 chain (see the note on the 12 waves below).
constexpr int L16_LDS_BYTES = (L16_LDS_DOUBLES * 8 > 132 * 1024) ? L16_LDS_DOUBLES * 8 : 132 * 1024;
constexpr int RS = 18;                // row stride (doubles) of the row-major 16 x 16 blocks in LDS (144 B: 16-B aligned rows)

struct Leaf16Args {
  double* A;
  int64_t lda;
  int kb, col0;
  double* winv;
  int32_t* info;
  int64_t sA, sW, sInfo;              // per-workgroup strides (elements): blockIdx.x-th problem of a batch
};

// WT: every global store is an agent-scope write-through (`sc1`) store -- the leaf as a task of the persistent factorisation
// (ppotrf.hip), whose results other workgroups of the same launch read (MI355X_MICROARCH.md "Valid forms").
template <bool WT = false>
__device__ __forceinline__ void leaf16_body(const Leaf16Args& p, const int prob, const int tid_in = -1) {
  auto gst = [](double* q, double v) {
    if constexpr (WT) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *q = v;
  };
  double* A = p.A + (int64_t)prob * p.sA;
  double* winv = p.winv + (int64_t)prob * p.sW;
  int32_t* info = p.info ? p.info + (int64_t)prob * p.sInfo : nullptr;
  const int64_t lda = p.lda;
  const int kb = p.kb;

  // LDS (dynamic: L16_LDS_BYTES = 123 KB): everything that crosses waves lives for the whole kernel
  extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
  double* const Xall = lds_dyn;                         // [28][4][64]: X(j,k), j > k, at slot j (j - 1) / 2 + k: register dumps, read as A operands
  double* const Raw = Xall + 28 * 256;                  // [2 parity][2: A(k+1,k), D(k+1,k+1)][4][64]: raw tiles for the pivot wave
  double* const Wf = Raw + 2 * 2 * 256;                 // [8 blocks][16 * RS]: W_k, fragment-ready: Wf[RS * m + c] = W_k[c][m]
  double* const Lrow = Wf + 8 * 16 * RS;                // [8 blocks][16 * RS]: L_k, row-major
  double* const Drow = Lrow + 8 * 16 * RS;              // [32 * RS]: rows 0..15 the pivot wave's next block; rows 16..31 identity
  double* const Lcol = Drow + 32 * RS;                  // [64]: the pivot wave's current column, for the broadcast reads
  double* const Tb0 = Lcol + 64;                        // prologue only: per tile wave a 16 x TS staging tile
  __shared__ int wready;                                // number of diagonal blocks whose W / L are out
  __shared__ int xready[8];                             // xready[j]: number of panels whose X(j, .) is out
  __shared__ int rawflag[8];                            // rawflag[k] != 0: the raw tiles for diagonal block k are out
  __shared__ int failflag;

  // (tid_in: the caller's thread index made opaque per call -- inside a persistent task loop the optimiser otherwise hoists
  //  every lane-derived address of this body out of the loop and spills them: ppotrf.hip)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> role: 0 pivot; 4, 8, 11 idle; tile row of the others.  Waves {1,5,9}, {2,6,10}, {3,7,11} share a SIMD each, and
  // on one SIMD the OLDER wave wins the matrix pipe whenever both are ready -- s_setprio does not reorder fp64 MFMAs, and a
  // dependent MFMA chain holds the pipe (tools/prio_bench.hip: two waves with 64 dependent MFMAs each run one after the
  // other, the older first, whatever their priorities).  Age is therefore the only priority there is: the oldest wave of a
  // SIMD gets the HIGHEST tile row, which stays in the factor's own (critical) part longest, and the rows that turn to the
  // identity part (-> W, needed by nobody inside the kernel) first sit on the youngest waves and fill what is left.
  const int rowmap = (wave == 1) ? 5 : (wave == 5) ? 3 : (wave == 9) ? 1 : (wave == 2) ? 6 : (wave == 6) ? 4 : (wave == 10) ? 2 :
                     (wave == 3) ? 7 : (wave == 7) ? 0 : -1;
  const bool tilewave = rowmap >= 0;
  const bool pivotwave = wave == 0;
  const int wrow = rowmap & 7;
  const int g = lane >> 4, lc = lane & 15;
  if (tid == 0) failflag = 0;
  if (tid < 8) { xready[tid] = 0; rawflag[tid] = 0; }
  if (tid == 8) wready = 0;
  for (int idx = tid; idx < 16 * RS; idx += L16_THREADS) Drow[16 * RS + idx] = ((idx / RS) == (idx % RS)) ? 1.0 : 0.0;
  __syncthreads();                                      // the flags and the identity rows are initialised: the ONLY barrier before the end

  auto bcast = [](double v, int src) -> double {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  };

  // flags: the writer's LDS stores are issued before its flag store and a wave's LDS operations execute in order; the
  // reader polls, then reads.  Spins are bounded: on a lost flag the leaf reports an internal failure instead of hanging.
  auto publish_flag = [&](int* flag, int value) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_flag = [&](int* flag, int above) -> bool {               // until *flag > above; false: give up (failure somewhere)
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= above) {
      if (__hip_atomic_load(&failflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { failflag = LEAF + 1; return false; }
    }
    asm volatile("" ::: "memory");
    return true;
  };
  auto xslot = [&](int j, int k) -> double* { return Xall + (j * (j - 1) / 2 + k) * 256; };

  // ======================================= tile waves =======================================
  // The code of a tile wave is instantiated once per tile ROW (w a compile-time constant, selected by a switch): with w a
  // run-time value every "is this tile mine" test is a wave-uniform branch around MFMAs, and at each join the register
  // allocator copied whole accumulators behind a pipeline drain (s_nop 14 + 8 v_mov per tile, out-of-place MFMAs): the
  // update phase ran at 40-50 % of the matrix pipe.  Per-row code is straight line; only the flag spins branch.
  auto tile_wave = [&](auto wconst) {
    constexpr int w = decltype(wconst)::value;
    // slot J (J <= w): A tile (w, J); slot J + 1 (J >= w): identity tile (8 + w, J).  Transposed storage.
    d4 acc[9];
    {
      // One memory round trip, COALESCED and 16 bytes per lane: a tile is fetched row-major by two instructions (lane l:
      // row (l >> 3) + 8 i, columns 2 (l & 7), + 1 -- 8 consecutive lanes = 128 consecutive bytes), only the tiles on and left
      // of the diagonal, every load issued before the first use; then turned into the transposed storage through a per-wave
      // 16 x 18 LDS tile.  (Fetching the transposed storage directly puts consecutive lanes on different rows: 64
      // transactions per load instruction, 13-21 k cycles of prologue; the memory pipe takes 16 cycles per wave
      // instruction whatever its width, so 8-byte loads of all 8 tile columns still cost 4 k cycles CU-wide.)
      double* tb = Tb0 + w * (16 * TS);
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
    auto dump = [&](double* dst, const d4& t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r * 64 + lane] = t[r];
    };
    // raw tiles for the pivot wave's first steps: D(0,0) from tile row 0, A(1,0) and D(1,1) from tile row 1 -- announced by
    // flags, not by a barrier: the pivot wave starts when row 0's ONE tile is in, not when row 7's eight are
    if (w == 0) { dump(Raw + (0 * 2 + 1) * 256, acc[0]); publish_flag(&rawflag[0], 1); }
    if (w == 1) { dump(Raw + (1 * 2 + 0) * 256, acc[0]); dump(Raw + (1 * 2 + 1) * 256, acc[1]); publish_flag(&rawflag[1], 1); }

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
    // drain (measured: the rolled loop ran the update phase at 58 % of the MFMA rate, the unrolled one at 78 %).
    bool ok = true;                                                   // false: a failure somewhere -- no more global stores
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ok = ok && wait_flag(&wready, k);                               // W_k and L_k are out
      if (!ok) break;                                                 // uniform
      d4 wf;
#pragma unroll
      for (int r = 0; r < 4; ++r) wf[r] = Wf[k * (16 * RS) + RS * (g + 4 * r) + lc];
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
      double* myx = xslot(w > 0 ? w : 1, k < w ? k : 0);              // (only used when apart)
      if (apart) {
        dump(myx, x);
        publish_flag(&xready[w], k + 1);                              // X(w, k) is out
      }
      const d4 nx = d4{-x[0], -x[1], -x[2], -x[3]};
      if (k < 7) {
        // ---- updates with panel k, BEFORE this panel's global stores (the stores are fire-and-forget, the next blocks wait
        // for these tiles).  The A operand X(j, k) of the NEXT tile is requested between the first and the second MFMA of
        // the current one (a non-blocking look at its flag: by now nearly every X(., k) is out), so a tile costs its four
        // dependent MFMAs and not flag poll + LDS round trip + MFMAs in a row.
        const int jlast = apart ? w - 1 : 7;                          // last tile column whose operand comes from another wave
        d4 xa = d4{0.0, 0.0, 0.0, 0.0};
        if (k + 1 <= jlast) {
          ok = ok && wait_flag(&xready[k + 1], k);
          xa = load_frag(xslot(k + 1, k));
        }
#pragma unroll
        for (int j = 1; j < 8; ++j) {
          if (j > k) {                                                // (static)
            if (j <= jlast) {                                         // (uniform) operand from tile row j
              auto op = [&](d4& t) {
                const d4 cur = xa;
                bool got = true;
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0], nx[0], t, 0, 0, 0);
                if (j + 1 <= jlast && j + 1 < 8) {
                  got = __hip_atomic_load(&xready[j + 1 < 8 ? j + 1 : 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > k;
                  if (got) xa = load_frag(xslot(j + 1 < 8 ? j + 1 : 7, k));
                }
#pragma unroll
                for (int r = 1; r < 4; ++r) t = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[r], nx[r], t, 0, 0, 0);
                if (!got) {
                  ok = ok && wait_flag(&xready[j + 1 < 8 ? j + 1 : 7], k);
                  xa = load_frag(xslot(j + 1 < 8 ? j + 1 : 7, k));
                }
              };
              if (apart) op(acc[j]);                                  // A tile (w, j)
              else op(acc[j + 1]);                                    // identity tile (8 + w, j)
            } else if (apart && j == w) {                             // my diagonal tile: the operand is my own X
              update(acc[j], x, nx);
            }
            // the raw tiles the pivot wave needs for diagonal block k + 2: A(k+2, k+1) and D(k+2, k+2), from tile row k + 2
            if (apart && w == k + 2) {
              if (j == k + 1) dump(Raw + ((k & 1) * 2 + 0) * 256, acc[j]);
              if (j == k + 2) { dump(Raw + ((k & 1) * 2 + 1) * 256, acc[j]); publish_flag(&rawflag[k + 2], 1); }
            }
          }
        }
      }
      // ---- this panel's final tiles to global memory
      if (!ok) break;
      if (apart) {
        // L tile (w, k), stored row-major (coalesced): element X[g + 4 r][lc], read back from my register dump
        L16_WAVE_FENCE();
        const double* xd = myx + (lc >> 2) * 64 + (lc & 3) * 16 + g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * w + g + 4 * r;
          const double v = xd[4 * r];
          if (row < kb) gst(&A[(int64_t)row * lda + 16 * k + lc], v);
        }
      } else {                                                        // W^T tile (w, k): X[a][b] = W[16 k + b][16 w + a]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wr = 16 * k + g + 4 * r, wc = 16 * w + lc;
          gst(&winv[(int64_t)wr * LEAF + wc], (wr < kb && wc < kb) ? x[r] : 0.0);
        }
        if (w == k) {                                                 // the diagonal tile L_k comes from the pivot wave
          const double* Lr = Lrow + k * (16 * RS);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r, col = lc;
            if (col <= row && 16 * k + row < kb) gst(&A[(int64_t)(16 * k + row) * lda + 16 * k + col], Lr[row * RS + col]);
          }
        }
      }
    }
  };
  if (tilewave) {
    switch (wrow) {
      case 0: tile_wave(std::integral_constant<int, 0>{}); break;
      case 1: tile_wave(std::integral_constant<int, 1>{}); break;
      case 2: tile_wave(std::integral_constant<int, 2>{}); break;
      case 3: tile_wave(std::integral_constant<int, 3>{}); break;
      case 4: tile_wave(std::integral_constant<int, 4>{}); break;
      case 5: tile_wave(std::integral_constant<int, 5>{}); break;
      case 6: tile_wave(std::integral_constant<int, 6>{}); break;
      default: tile_wave(std::integral_constant<int, 7>{}); break;
    }
  } else if (!pivotwave) {
    // idle waves (they share the pivot wave's SIMD): off everybody's critical path, the zero
    // fill of winv above the diagonal tiles (28 tiles W[16 k + ..][16 j + ..], k < j; 16 bytes per lane)
    {
      const int me = (wave == 4) ? 0 : (wave == 8) ? 1 : 2;
      int t = 0;
      for (int k = 0; k < 7; ++k)
        for (int j2 = k + 1; j2 < 8; ++j2, ++t) {
          if (t % 3 != me) continue;
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
          {
            double* zq = &winv[(int64_t)(16 * k + (lane >> 3) + 8 * i2) * LEAF + 16 * j2 + 2 * (lane & 7)];
            if constexpr (WT) { gst(zq, 0.0); gst(zq + 1, 0.0); }
            else *reinterpret_cast<d2*>(zq) = d2{0.0, 0.0};
          }
        }
    }
  } else {
    // ======================================= pivot wave =======================================
    __builtin_amdgcn_s_setprio(3);
    wait_flag(&rawflag[0], 0);                                        // D(0,0) is out (a failure leaves through the loop below)
    double a[16];
    const int myrow = lane & 31;
    // block 0: D(0,0) as dumped by wave 0 (transposed storage of a symmetric tile) -> one row per lane
    {
      d4 dacc;
#pragma unroll
      for (int r = 0; r < 4; ++r) dacc[r] = Raw[(0 * 2 + 1) * 256 + r * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) Drow[lc * RS + g + 4 * r] = dacc[r];
      L16_WAVE_FENCE();
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = Drow[myrow * RS + c];
      L16_WAVE_FENCE();
    }
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
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
        double* dst = lane < 16 ? Lrow + k * (16 * RS) + lane * RS : Wf + k * (16 * RS) + (lane - 16) * RS;
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c] = a[c];
      }
      // (after a failed pivot nothing is announced: the waiters see failflag in their spin and leave without storing)
      if (__hip_atomic_load(&failflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      // ONE LDS round trip between two blocks: the raw tiles for block k + 1 were published a panel ago, so they are read
      // speculatively together with their flag and with my own W_k (in fragment order); the flag is checked afterwards
      // and the reads repeated in the rare case it was not up yet.
      const int par = (k + 1) & 1;
      d4 wf, ar, dacc;
      int rf = 1;
      if (k < 7) {
        L16_WAVE_FENCE();
        rf = __hip_atomic_load(&rawflag[k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          wf[r] = Wf[k * (16 * RS) + RS * (g + 4 * r) + lc];
          ar[r] = Raw[(par * 2 + 0) * 256 + r * 64 + lane];
          dacc[r] = Raw[(par * 2 + 1) * 256 + r * 64 + lane];
        }
      }
      publish_flag(&wready, k + 1);                                   // W_k and L_k are out (waits for the LDS operations above)
      if (k == 7) break;
      if (rf == 0) {
        if (!wait_flag(&rawflag[k + 1], 0)) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ar[r] = Raw[(par * 2 + 0) * 256 + r * 64 + lane];
          dacc[r] = Raw[(par * 2 + 1) * 256 + r * 64 + lane];
        }
      }
      // ---- next diagonal block from the raw tiles:  X = A(k+1,k) W_k^T,  D(k+1) -= X X^T
      {
        d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[r], ar[r], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[r], x[r], dacc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Drow[lc * RS + g + 4 * r] = dacc[r];
        L16_WAVE_FENCE();
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = Drow[myrow * RS + c];
        L16_WAVE_FENCE();
      }
    }
  }
  __syncthreads();
  if (failflag) {
    if (tid == 0 && info) {
      if constexpr (WT) {          // another workgroup of the same launch may have reported before: first report wins, atomically
        if (failflag > LEAF) __hip_atomic_store(info, (int32_t)GPN_INFO_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { int32_t expect = 0; __hip_atomic_compare_exchange_strong(info, &expect, (int32_t)(p.col0 + failflag), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      } else {
        if (failflag > LEAF) *info = GPN_INFO_INTERNAL;
        else if (*info == 0) *info = p.col0 + failflag;
      }
    }
    for (int idx = tid; idx < LEAF * LEAF; idx += L16_THREADS) gst(&winv[idx], 0.0);
  }
}

}  // namespace gpn
