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

namespace gpn {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int L16_THREADS = 576;      // 8 tile waves + the pivot wave
constexpr int RS = 18;                // row stride (doubles) of the row-major 16 x 16 blocks in LDS (144 B: 16-B aligned rows)

struct Leaf16Args {
  double* A;
  int64_t lda;
  int kb, col0;
  double* winv;
  int32_t* info;
  int64_t sA, sW, sInfo;              // per-workgroup strides (elements): blockIdx.x-th problem of a batch
};

#define L16_STAMP(k)                                                            \
  if constexpr (DIAG) {                                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                 \
    acc_t[k] += t_ - last_t;                                                    \
    last_t = t_;                                                                \
  }

template <bool DIAG>
__global__ __launch_bounds__(L16_THREADS) void potrf_leaf16_kernel(Leaf16Args p, unsigned long long* diag) {
  unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t = 0;
  if constexpr (DIAG) last_t = __builtin_amdgcn_s_memtime();
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
  __shared__ int tb_count;
  __shared__ int failflag;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool tilewave = wave < 8;
  const int w = wave & 7;
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
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const bool isA = q <= w;
      const int J = isA ? q : q - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + lc, col = 16 * J + g + 4 * r;     // element T[lc][g + 4 r] of tile (w, J)
        double v = (row == col) ? 1.0 : 0.0;
        if (isA) {
          const int rr = row > col ? row : col, cc = row > col ? col : row;   // diagonal tile: symmetric fill
          if (q < w) v = 0.0;
          if (rr < kb) v = A[(int64_t)rr * lda + cc];
        } else if (J != w) {
          v = 0.0;
        }
        acc[q][r] = v;
      }
    }
    // the part of winv above the diagonal tiles is zero: W[16 k + ..][16 w + ..], k < w
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (k < w) {
#pragma unroll
        for (int r = 0; r < 4; ++r) winv[(int64_t)(16 * k + g + 4 * r) * LEAF + 16 * w + lc] = 0.0;
      }
    }
    auto dump = [&](double* dst, const d4& t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r * 64 + lane] = t[r];
    };
    // raw tiles for the pivot wave's first steps: D(0,0) from wave 0, A(1,0) and D(1,1) from wave 1
    if (w == 0) dump(&Raw[0][1][0][0], acc[0]);
    if (w == 1) { dump(&Raw[1][0][0][0], acc[0]); dump(&Raw[1][1][0][0], acc[1]); }
    __syncthreads();                                                  // P

    int epoch = 0;
    auto tile_barrier = [&]() {
      ++epoch;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(&tb_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int spins = 0;
      while (__hip_atomic_load(&tb_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { failflag = LEAF + 1; break; }
      }
      asm volatile("" ::: "memory");
    };
    // X = T W_k^T in place (transposed storage both sides)
    auto solve = [&](d4& t, const d4& wf) {
      d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[r], t[r], x, 0, 0, 0);
      t = x;
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

#pragma unroll
    for (int k = 0; k < 8; ++k) {
      L16_STAMP(0)
      __syncthreads();                                                // B(k): W_k and L_k are out
      L16_STAMP(1)
      if (failflag) break;                                            // uniform
      d4 wf;
#pragma unroll
      for (int r = 0; r < 4; ++r) wf[r] = Wf[k & 1][RS * (g + 4 * r) + lc];
      d4 nx;                                                          // -X of my tile in column k
      if (w > k) {
        solve(acc[k], wf);
        const d4& x = acc[k];
        if (k < 7) dump(&Xbuf[w][0][0], x);
        const int row = 16 * w + lc;                                  // L tile (w, k)
        if (row < kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) A[(int64_t)row * lda + 16 * k + g + 4 * r] = x[r];
        }
        nx = d4{-x[0], -x[1], -x[2], -x[3]};
      } else {
        solve(acc[k + 1], wf);
        const d4& x = acc[k + 1];                                     // W^T tile (w, k): X[a][b] = W[16 k + b][16 w + a]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wr = 16 * k + g + 4 * r, wc = 16 * w + lc;
          winv[(int64_t)wr * LEAF + wc] = (wr < kb && wc < kb) ? x[r] : 0.0;
        }
        nx = d4{-x[0], -x[1], -x[2], -x[3]};
        if (w == k) {                                                 // the diagonal tile L_k comes from the pivot wave
          const double* Lr = Lrow[k & 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r, col = lc;
            if (col <= row && 16 * k + row < kb) A[(int64_t)(16 * k + row) * lda + 16 * k + col] = Lr[row * RS + col];
          }
        }
      }
      L16_STAMP(2)
      if (k == 7) break;
      tile_barrier();                                                 // T(k): the X tiles of this panel are in LDS
      L16_STAMP(3)
      if (w > k) {
        // A tiles (w, j), j = k + 1 .. w
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
          if (j < w) {
            const d4 xa = load_frag(&Xbuf[j][0][0]);
            update(acc[j], xa, nx);
          } else if (j == w) {
            const d4 xa = d4{-nx[0], -nx[1], -nx[2], -nx[3]};
            update(acc[j], xa, nx);
          }
          // the raw tiles the pivot wave needs after its NEXT block: A(k+2, k+1) and D(k+2, k+2), from wave k + 2
          if (w == k + 2) {
            if (j == k + 1) dump(&Raw[k & 1][0][0][0], acc[j]);
            if (j == k + 2) dump(&Raw[k & 1][1][0][0], acc[j]);
          }
        }
      } else {
        // identity tiles (8 + w, j), j = k + 1 .. 7
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
          const d4 xa = load_frag(&Xbuf[j][0][0]);
          update(acc[j + 1], xa, nx);
        }
      }
      L16_STAMP(4)
    }
  } else {
    // ======================================= pivot wave =======================================
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                                  // P
    double a[16];
    const int myrow = lane & 31;
    // block 0: D(0,0) as dumped by wave 0 (transposed storage of a symmetric tile) -> one row per lane
    {
      d4 dacc;
#pragma unroll
      for (int r = 0; r < 4; ++r) dacc[r] = Raw[0][1][r][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) Drow[lc * RS + g + 4 * r] = dacc[r];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = Drow[myrow * RS + c];
    }
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      L16_STAMP(0)
      // ---- 16 pivots, one row per lane; lanes 16..31 carry the identity rows
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
#pragma unroll
        for (int c = J + 1; c < 16; ++c) a[c] = fma(-l, bcast(l, c), a[c]);
      }
      L16_STAMP(1)
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
      L16_STAMP(2)
      __syncthreads();                                                // B(k)
      L16_STAMP(3)
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
#pragma unroll
        for (int r = 0; r < 4; ++r) Drow[lc * RS + g + 4 * r] = dacc[r];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = Drow[myrow * RS + c];
      }
      L16_STAMP(4)
    }
  }
  if constexpr (DIAG) {
    L16_STAMP(7)
    if (lane == 0) for (int q = 0; q < 8; ++q) diag[wave * 8 + q] = acc_t[q];
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

int leaf16_timing(hipStream_t s, double* A, int64_t lda, double* winv, int32_t* info, unsigned long long* diag72) {
  Leaf16Args a;
  a.A = A; a.lda = lda; a.kb = LEAF; a.col0 = 0; a.winv = winv; a.info = info; a.sA = 0; a.sW = 0; a.sInfo = 0;
  hipLaunchKernelGGL(potrf_leaf16_kernel<true>, dim3(1), dim3(L16_THREADS), 0, s, a, diag72);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn
