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

__global__ __launch_bounds__(L16_THREADS) void potrf_leaf16_kernel(Leaf16Args p) {
  leaf16_body<false>(p, (int)blockIdx.x);
}

// `batch` leaves in one launch: problem b at A + b sA, winv + b sW, info + b sInfo (batch = 1: strides ignored)
int leaf16(hipStream_t s, double* A, int64_t lda, int kb, int col0, double* winv, int32_t* info, int batch, int64_t sA,
           int64_t sW, int64_t sInfo) {
  Leaf16Args a;
  a.A = A; a.lda = lda; a.kb = kb; a.col0 = col0; a.winv = winv; a.info = info; a.sA = sA; a.sW = sW; a.sInfo = sInfo;
  static std::atomic<int> attr_done{0};
  if (!attr_done.load(std::memory_order_acquire)) {
    GPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS_BYTES));
    attr_done.store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL(potrf_leaf16_kernel, dim3((unsigned)batch), dim3(L16_THREADS), L16_LDS_BYTES, s, a);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn
