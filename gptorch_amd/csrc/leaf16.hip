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
//   * NO barrier inside the panel loop: the kernel is a dataflow over LDS flags.  Every tile that crosses waves is kept
//     for the whole kernel (28 X tiles, 8 W blocks, 8 L blocks: nothing is ever overwritten, so a lagging wave cannot
//     lose its input) and announced by a flag its readers poll: wready (W_k, L_k out), xready[j] (X(j,k) of tile row j
//     out), rawflag[k] (the two raw tiles for block k out).  A wave waits for exactly the tiles it reads: the pivot wave
//     only for the light rows next to the diagonal, never for the heavy rows far below it or for the identity part.
//     (The first version had one hardware barrier per 16 pivots + a barrier of the 8 tile waves: the pivot wave stood
//     still for 2.8 / 2.7 / 2.2 / 1.7 k cycles in panels 1-4 waiting for the SLOWEST tile wave -- profiles/r4_leaf16_timeline.txt.)
// Deterministic (fixed summation order).  FACTOR=false (inverse of a given L) stays on the first kernel.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "gpn_common.h"

#include "leaf16_body.h"

namespace gpn {

template <bool DIAG>
__global__ __launch_bounds__(L16_THREADS) void potrf_leaf16_kernel(Leaf16Args p, unsigned long long* diag) {
  leaf16_body<DIAG>(p, diag, (int)blockIdx.x);
}

#ifdef GPN_DEBUG_SWITCHES
// ---- one step of the in-panel chain as ONE launch: the NEXT leaf next to this step's column work ------------------------
// (A/B, tools' build only: measured slower than the separate launches -- potrf.hip g_fused_steps, LAB.md 10.)
// After leaf(k) the driver solves the 128 rows under the diagonal block (X_top = B_top W_k^T) and updates the next
// diagonal block (D' = D - X_top X_top^T) with two four-workgroup launches of colpanel.hip; what is left of step k -- the
// solve of all rows below those 128 and the update of the next column block by them -- needs nothing from leaf(k+1) and
// leaf(k+1) nothing from it.  Two streams cannot use that (a just-in-time event between streams costs 12-20 us, LAB 8-1e),
// one launch can: workgroup 0 of each problem is the leaf, the others walk 32-row tiles of the rows below:
//     X = B W_k^T (in place, and kept in LDS)  ->  C -= X X_top^T      with W_k, then X_top, as register B fragments.
// Same fragment maps and k order as colpanel.hip (eight waves own 16 columns each; waves 8..11 only load and wait).
struct ChainStepArgs {
  Leaf16Args leaf;      // the next diagonal block
  double* B;            // rows below the next diagonal block, columns of block k: [m, 128], solved in place
  const double* W;      // W_k [128, 128] lower triangular (winv block k)
  const double* Xtop;   // the 128 solved rows under diagonal block k: [128, 128] at the matrix's leading dimension
  double* C;            // the same rows, columns of block k+1: [m, 128]
  int64_t lda;
  int m;
  int64_t sA, sW;       // per-problem strides (blockIdx.y)
};
constexpr int CS_ROWS = 32;
constexpr int CS_LDS_ROW = LEAF * 8 + 16;      // bytes per tile row in LDS (colpanel.hip's layout)

__device__ __forceinline__ void chain_colwork_body(const ChainStepArgs& a, const int prob, const int first, const int stride) {
  extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
  char* const At = reinterpret_cast<char*>(lds_dyn);
  char* const Xt = At + CS_ROWS * CS_LDS_ROW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const bool mm = wave < 8;                              // the matrix waves
  // waves w and w + 4 share a SIMD: column tiles {s, 7 - s} per SIMD carry equal work in the triangular solve
  const int ct = wave < 4 ? wave : 11 - wave;
  double* B = a.B + (int64_t)prob * a.sA;
  double* C = a.C + (int64_t)prob * a.sA;
  const double* W = a.W + (int64_t)prob * a.sW;
  const double* Xtop = a.Xtop + (int64_t)prob * a.sA;
  const int64_t lda = a.lda;
  const int ntiles = (a.m + CS_ROWS - 1) / CS_ROWS;
  for (int tile = first; tile < ntiles; tile += stride) {
    const int r0 = tile * CS_ROWS;
    d2 areg[4];
    d4 accu[2];
    d2 b[16];
    if (mm) {
      const int t = tid, row = t >> 4, seg = t & 15;     // 512 threads: 32 rows x 16 segments, 4 x 16 B each
      const bool ok = r0 + row < a.m;
      const double* src = B + (int64_t)(r0 + row) * lda + seg * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) areg[i] = ok ? *reinterpret_cast<const d2*>(src + i * 32) : d2{0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row2 = r0 + i * 16 + lq + 4 * r;
          accu[i][r] = row2 < a.m ? C[(int64_t)row2 * lda + ct * 16 + lr] : 0.0;
        }
      const double* wsrc = W + (int64_t)(ct * 16 + lr) * LEAF + 2 * lq;
#pragma unroll
      for (int j = 0; j < 16; ++j) b[j] = (8 * j <= ct * 16 + 15) ? *reinterpret_cast<const d2*>(wsrc + 8 * j) : d2{0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<d2*>(At + row * CS_LDS_ROW + (seg + 16 * i) * 16) = areg[i];
    }
    __syncthreads();
    if (mm) {
      d4 accs[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (8 * j > ct * 16 + 15) continue;              // W[col][k] = 0 for k > col (wave-uniform)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const d2 av = *reinterpret_cast<const d2*>(At + (i * 16 + lr) * CS_LDS_ROW + (4 * j + lq) * 16);
          accs[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b[j].x, accs[i], 0, 0, 0);
          accs[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b[j].y, accs[i], 0, 0, 0);
        }
      }
      // X: in place to the matrix, and to LDS as the update's left operand
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = i * 16 + lq + 4 * r;
          if (r0 + lrow < a.m) B[(int64_t)(r0 + lrow) * lda + ct * 16 + lr] = accs[i][r];
          *reinterpret_cast<double*>(Xt + lrow * CS_LDS_ROW + (ct * 16 + lr) * 8) = accs[i][r];
        }
      const double* xsrc = Xtop + (int64_t)(ct * 16 + lr) * lda + 2 * lq;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const d2 v = *reinterpret_cast<const d2*>(xsrc + 8 * j);
        b[j] = d2{-v.x, -v.y};
      }
    }
    __syncthreads();
    if (mm) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const d2 av = *reinterpret_cast<const d2*>(Xt + (i * 16 + lr) * CS_LDS_ROW + (4 * j + lq) * 16);
          accu[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b[j].x, accu[i], 0, 0, 0);
          accu[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b[j].y, accu[i], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row2 = r0 + i * 16 + lq + 4 * r;
          if (row2 < a.m) C[(int64_t)row2 * lda + ct * 16 + lr] = accu[i][r];
        }
    }
  }
}

__global__ __launch_bounds__(L16_THREADS) void chain_step_kernel(ChainStepArgs a, int batch, int cw) {
  // the leaves are the first `batch` workgroups (dispatched first); column worker q of problem q % batch walks tiles
  // q / batch, q / batch + cw, ...
  const int bid = blockIdx.x;
  if (bid < batch) leaf16_body<false>(a.leaf, nullptr, bid);
  else chain_colwork_body(a, (bid - batch) % batch, (bid - batch) / batch, cw);
}

#endif

// `batch` leaves in one launch: problem b at A + b sA, winv + b sW, info + b sInfo (batch = 1: strides ignored)
int leaf16(hipStream_t s, double* A, int64_t lda, int kb, int col0, double* winv, int32_t* info, int batch, int64_t sA,
           int64_t sW, int64_t sInfo) {
  Leaf16Args a;
  a.A = A; a.lda = lda; a.kb = kb; a.col0 = col0; a.winv = winv; a.info = info; a.sA = sA; a.sW = sW; a.sInfo = sInfo;
  static std::atomic<int> attr_done{0};
  if (!attr_done.load(std::memory_order_acquire)) {
    GPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS_BYTES));
    attr_done.store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL(potrf_leaf16_kernel<false>, dim3((unsigned)batch), dim3(L16_THREADS), L16_LDS_BYTES, s, a, nullptr);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

#ifdef GPN_DEBUG_SWITCHES
// leaf(k+1) on the diagonal block at Anext (its inverse -> Wnext) next to step k's column work on the m rows below it:
// B (columns of block k; solved in place against W), C (columns of block k+1) -= X Xtop^T.  All of one matrix (lda).
int chain_step(hipStream_t s, double* Anext, int64_t lda, int col0, double* Wnext, int32_t* info, double* B, const double* W,
               const double* Xtop, double* C, int64_t m, int batch, int64_t sA, int64_t sW, int64_t sInfo) {
  ChainStepArgs a;
  a.leaf.A = Anext; a.leaf.lda = lda; a.leaf.kb = LEAF; a.leaf.col0 = col0; a.leaf.winv = Wnext; a.leaf.info = info;
  a.leaf.sA = sA; a.leaf.sW = sW; a.leaf.sInfo = sInfo;
  a.B = B; a.W = W; a.Xtop = Xtop; a.C = C; a.lda = lda; a.m = (int)m; a.sA = sA; a.sW = sW;
  static std::atomic<int> attr_done{0};
  if (!attr_done.load(std::memory_order_acquire)) {
    GPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS_BYTES));
    attr_done.store(1, std::memory_order_release);
  }
  const int64_t tiles = (m + CS_ROWS - 1) / CS_ROWS;
  const int64_t cw = std::max<int64_t>(1, std::min<int64_t>(tiles, std::max<int64_t>(1, 255 / batch)));   // one workgroup per CU: at most a round
  hipLaunchKernelGGL(chain_step_kernel, dim3((unsigned)((1 + cw) * batch)), dim3(L16_THREADS), L16_LDS_BYTES, s, a, batch, (int)cw);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

#endif

int leaf16_timing(hipStream_t s, double* A, int64_t lda, double* winv, int32_t* info, unsigned long long* diag768) {
  Leaf16Args a;
  a.A = A; a.lda = lda; a.kb = LEAF; a.col0 = 0; a.winv = winv; a.info = info; a.sA = 0; a.sW = 0; a.sInfo = 0;
  GPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS_BYTES));
  hipLaunchKernelGGL(potrf_leaf16_kernel<true>, dim3(1), dim3(L16_THREADS), L16_LDS_BYTES, s, a, diag768);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn
