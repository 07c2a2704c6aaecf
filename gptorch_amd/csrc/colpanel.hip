// The two column passes of the factorisation's in-panel chain as a dedicated kernel.
//
// Per 128-wide column block k the look-ahead driver (potrf.hip) has, on its critical path,
//     leaf(k)  ->  X = B W_k^T            all m rows below the block, in place   ("solve")
//              ->  C -= X X_top^T         the next column block                  ("update")
//              ->  leaf(k+1)
// i.e. two products  [m x 128] x [128 x 128]^T  with K = 128 (functions.py:46-47 is what the whole chain replaces).
// Through the generic contraction kernel (gemm_f64.hip) these cost 29 + 30 us per step at N = 32768 -- 18 TFLOP/s,
// 1-2 TB/s -- because a 32 x 128 tile walks its 8 K-steps through a 3-deep LDS ring: one memory round trip per
// K-step, nothing else to do in between.  Here K is the whole problem, so nothing is pipelined over K at all:
//   * the 128 x 128 B operand (W_k, or the 128 solved rows X_top) goes straight into REGISTERS, once per workgroup:
//     a wave owns 32 of the 128 columns, i.e. 32 x 128 doubles = 64 per lane, in the MFMA B-fragment layout
//     (lane l: column l & 15 of a 16-column tile, the k pair 8 j + 2 (l >> 4) .. + 1 for j = 0..15);
//   * the workgroup's 32 x 128 A tile is fetched with ALL its loads in flight at once (8 x 16 B per thread: one
//     memory round trip), parked in LDS (row stride 1 KiB + 16 B: the 16 rows x 4 k-pairs of an operand read fall on
//     16 distinct 16-byte bank groups per 16 lanes), and read back as MFMA A fragments;
//   * update mode loads the old C tile into the accumulators up front (the same round trip) and negates B, so the
//     MFMAs produce C - X X_top^T directly; solve mode skips the k range above a column tile's last column (W is
//     lower triangular) and deals the 16-column tiles {w, 7 - w} to wave w so that the four waves carry equal work.
// v_mfma_f64_16x16x4_f64 fragment maps as in gemm_f64.hip; the k order inside a group of 8 is {0,2,4,6},{1,3,5,7} for
// both operands (a permutation shared by A and B).  Deterministic (fixed summation order per entry).
#include "gpn_common.h"

namespace gpn {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int CP_ROWS = 32;                   // rows per workgroup
constexpr int CP_K = LEAF;                    // K = the block width
constexpr int CP_LDS_ROW = CP_K * 8 + 16;     // bytes per A row in LDS

struct ColPanelArgs {
  const double* A;      // [m, 128]   left operand rows
  const double* B;      // [nb, 128]  right operand (W_k lower triangular, or X_top)
  double* C;            // [m, nb]
  int64_t lda, ldb, ldc;
  int m, nb;
  int kvalid;           // K columns of A that hold data: the rest of the 128-wide K range is read as zero (a ragged last block's
                        // padding columns may hold anything, NaN included: 0 * NaN inside an MFMA would poison the row)
  int64_t sA, sB, sC;   // per-problem strides (elements) of a batch: blockIdx.y-th problem
};

// MODE 0: C = A B^T with B lower triangular (C may alias A: every row tile is read completely before it is written)
// MODE 1: C = C - A B^T
template <int MODE>
__global__ __launch_bounds__(256, 2) void colpanel_kernel(ColPanelArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[CP_ROWS * CP_LDS_ROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * CP_ROWS;
  const int lr = lane & 15, lq = lane >> 4;
  p.A += (int64_t)blockIdx.y * p.sA;
  p.B += (int64_t)blockIdx.y * p.sB;
  p.C += (int64_t)blockIdx.y * p.sC;

  // ---- A tile: all loads of the workgroup in flight at once
  d2 areg[8];
  {
    const int row = tid >> 3, seg0 = tid & 7;
    const bool ok = r0 + row < p.m;
    const double* src = p.A + (int64_t)(r0 + row) * p.lda + seg0 * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d2 v = ok ? *reinterpret_cast<const d2*>(src + i * 16) : d2{0.0, 0.0};
      const int k0 = seg0 * 2 + i * 16;
      if (k0 >= p.kvalid) v.x = 0.0;
      if (k0 + 1 >= p.kvalid) v.y = 0.0;
      areg[i] = v;
    }
  }
  // ---- old C tile into the accumulators (update), or zeros (solve)
  const int ct[2] = {MODE == 0 ? wave : 2 * wave, MODE == 0 ? 7 - wave : 2 * wave + 1};   // my two 16-column tiles
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc[i][c] = d4{0.0, 0.0, 0.0, 0.0};
      if constexpr (MODE == 1) {
        const int col = ct[c] * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + i * 16 + lq + 4 * r;
          if (row < p.m && col < p.nb) acc[i][c][r] = p.C[(int64_t)row * p.ldc + col];
        }
      }
    }
  // ---- B fragments: 2 column tiles x 16 k-groups of 8, the k pair (8 j + 2 lq, + 1) of column lr
  d2 b[2][16];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = ct[c] * 16 + lr;
    const bool ok = col < p.nb;
    const double* src = p.B + (int64_t)(ok ? col : 0) * p.ldb + 2 * lq;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      d2 v = d2{0.0, 0.0};
      if (MODE == 1 || 8 * j <= ct[c] * 16 + 15) v = *reinterpret_cast<const d2*>(src + 8 * j);   // (wave-uniform skip)
      if (!ok) v = d2{0.0, 0.0};
      if constexpr (MODE == 1) v = d2{-v.x, -v.y};
      b[c][j] = v;
    }
  }
  // ---- park the A tile in LDS
  {
    const int row = tid >> 3, seg0 = tid & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<d2*>(lds + row * CP_LDS_ROW + (seg0 + 8 * i) * 16) = areg[i];
  }
  __syncthreads();
  // ---- MFMAs: 2 row tiles x 16 k-groups x 2 column tiles x 2
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    d2 a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const d2*>(lds + (i * 16 + lr) * CP_LDS_ROW + (4 * j + lq) * 16);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (MODE == 0 && 8 * j > ct[c] * 16 + 15) continue;     // W[col][k] = 0 for k > col
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[c][j].x, acc[i][c], 0, 0, 0);
        acc[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[c][j].y, acc[i][c], 0, 0, 0);
      }
    }
  }
  // ---- store: reg r of lane l is C[lq + 4 r][lr] of its 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = ct[c] * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + i * 16 + lq + 4 * r;
        if (row < p.m && col < p.nb) p.C[(int64_t)row * p.ldc + col] = acc[i][c][r];
      }
    }
}

int colpanel(hipStream_t s, int mode, int64_t m, int64_t nb, const double* A, int64_t lda, const double* B, int64_t ldb,
             double* C, int64_t ldc, int batch, int64_t sA, int64_t sB, int64_t sC) {
  if (m <= 0 || nb <= 0 || batch <= 0) return GPN_OK;
  ColPanelArgs a;
  a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.m = (int)m; a.nb = (int)nb;
  a.sA = sA; a.sB = sB; a.sC = sC;
  a.kvalid = mode == 0 ? (int)nb : CP_K;
  const dim3 grid((unsigned)((m + CP_ROWS - 1) / CP_ROWS), (unsigned)batch);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, 2.0 * batch * (double)m * (double)nb * CP_K * (mode == 0 ? 0.5 : 1.0), mode == 0 ? PROF_GEMM_SOLVE : PROF_GEMM);
  if (mode == 0) hipLaunchKernelGGL(colpanel_kernel<0>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(colpanel_kernel<1>, grid, dim3(256), 0, s, a);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn
