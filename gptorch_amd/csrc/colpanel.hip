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

// ---- both column passes of a chain step in ONE launch (A/B, tools' build only: measured SLOWER, potrf.hip g_fused_colstep) --
#ifdef GPN_DEBUG_SWITCHES
// Step k of the chain solves all m rows below diagonal block k (X = B W_k^T, in place) and then updates the next column
// block by them (C -= X X_top^T, X_top = the first 128 solved rows).  As two launches the second re-reads X from HBM and
// pays its own ramp (13.5 us at C2 for 0.27 GFLOP through the generic contraction); here a workgroup keeps its 32 x 128
// X tile in LDS and goes straight on to its rows of C.  The only thing it needs from OTHER workgroups is X_top: the four
// workgroups that own those rows (blockIdx.x < 4: dispatched first) store their tiles with agent-scope stores, drain them
// and count themselves in on a device counter; every workgroup polls that counter after its own solve (by then the four
// are done or about to be), then reads X_top with agent-scope loads (MI355X_MICROARCH.md "Valid forms": sc1 payload, sc1
// flag, no fence needed; tools/handoff_bench.hip measured the pattern).  The poll is bounded: a counter that never
// arrives sets info = GPN_INFO_INTERNAL instead of hanging the queue.  Same fragment maps and k order as the two kernels
// above, so X is bit-identical to colpanel_kernel<0>'s.
struct ColStepArgs {
  double* B;            // [m, 128] rows below the diagonal block, columns of block k (solved in place)
  const double* W;      // W_k
  double* C;            // [m, 128] the same rows, columns of block k+1
  int64_t lda;
  int m;
  int* flag;            // one zeroed counter per problem of the batch
  int32_t* info;        // per problem
  int64_t sA, sW;
};

__global__ __launch_bounds__(256, 2) void colstep_kernel(ColStepArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[CP_ROWS * CP_LDS_ROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * CP_ROWS;
  const int lr = lane & 15, lq = lane >> 4;
  const bool top = blockIdx.x < LEAF / CP_ROWS;          // my rows are rows of X_top
  double* B = p.B + (int64_t)blockIdx.y * p.sA;
  double* C = p.C + (int64_t)blockIdx.y * p.sA;
  const double* W = p.W + (int64_t)blockIdx.y * p.sW;
  int* flag = p.flag + blockIdx.y;
  const int64_t lda = p.lda;

  // ---- A tile and old C tile: all loads in flight at once
  d2 areg[8];
  {
    const int row = tid >> 3, seg0 = tid & 7;
    const bool ok = r0 + row < p.m;
    const double* src = B + (int64_t)(r0 + row) * lda + seg0 * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) areg[i] = ok ? *reinterpret_cast<const d2*>(src + i * 16) : d2{0.0, 0.0};
  }
  const int cu[2] = {2 * wave, 2 * wave + 1};            // update: my two 16-column tiles of C
  d4 accu[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + i * 16 + lq + 4 * r;
        accu[i][c][r] = row < p.m ? C[(int64_t)row * lda + cu[c] * 16 + lr] : 0.0;
      }
  // ---- solve: B fragments of W_k for the column tiles {w, 7 - w}
  const int cs[2] = {wave, 7 - wave};
  d2 b[2][16];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const double* src = W + (int64_t)(cs[c] * 16 + lr) * CP_K + 2 * lq;
#pragma unroll
    for (int j = 0; j < 16; ++j) b[c][j] = (8 * j <= cs[c] * 16 + 15) ? *reinterpret_cast<const d2*>(src + 8 * j) : d2{0.0, 0.0};
  }
  {
    const int row = tid >> 3, seg0 = tid & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<d2*>(lds + row * CP_LDS_ROW + (seg0 + 8 * i) * 16) = areg[i];
  }
  __syncthreads();
  d4 accs[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) accs[i][c] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    d2 a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const d2*>(lds + (i * 16 + lr) * CP_LDS_ROW + (4 * j + lq) * 16);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (8 * j > cs[c] * 16 + 15) continue;             // W[col][k] = 0 for k > col
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        accs[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[c][j].x, accs[i][c], 0, 0, 0);
        accs[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[c][j].y, accs[i][c], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                       // every wave is done with the A tile: X takes its place
  // ---- X: in place to the matrix (agent scope for the rows of X_top), and to LDS as the update's left operand
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = i * 16 + lq + 4 * r, col = cs[c] * 16 + lr;
        double* dst = B + (int64_t)(r0 + lrow) * lda + col;
        if (r0 + lrow < p.m) {
          if (top) __hip_atomic_store(dst, accs[i][c][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dst = accs[i][c][r];
        }
        *reinterpret_cast<double*>(lds + lrow * CP_LDS_ROW + col * 8) = accs[i][c][r];
      }
  if (top) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my rows of X_top have left this wave
  __syncthreads();
  if (tid == 0) {
    if (top) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int need = min(LEAF / CP_ROWS, (int)gridDim.x);
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) {                         // (seconds: a lost counter, not a slow producer)
        if (p.info) atomicCAS(p.info + blockIdx.y, 0, GPN_INFO_INTERNAL);
        break;
      }
    }
  }
  __syncthreads();
  // ---- update: B fragments = -X_top rows {2w, 2w+1} tiles (agent-scope loads: written by other workgroups of this launch)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const double* src = B + (int64_t)(cu[c] * 16 + lr) * lda + 2 * lq;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double x = __hip_atomic_load(src + 8 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double y = __hip_atomic_load(src + 8 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b[c][j] = d2{-x, -y};
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    d2 a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const d2*>(lds + (i * 16 + lr) * CP_LDS_ROW + (4 * j + lq) * 16);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        accu[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[c][j].x, accu[i][c], 0, 0, 0);
        accu[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[c][j].y, accu[i][c], 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + i * 16 + lq + 4 * r;
        if (row < p.m) C[(int64_t)row * lda + cu[c] * 16 + lr] = accu[i][c][r];
      }
}

// one chain step's two column passes in one launch: B [m, 128] <- B W^T in place, C [m, 128] -= X X_top^T with X_top = the
// first 128 rows of the result (m >= 128); flag: `batch` zeroed ints; everything inside one matrix (leading dimension lda)
int colstep(hipStream_t s, int64_t m, double* B, const double* W, double* C, int64_t lda, int* flag, int32_t* info, int batch,
            int64_t sA, int64_t sW) {
  if (m < LEAF || batch <= 0) return GPN_E_UNSUPPORTED;
  ColStepArgs a;
  a.B = B; a.W = W; a.C = C; a.lda = lda; a.m = (int)m; a.flag = flag; a.info = info; a.sA = sA; a.sW = sW;
  const dim3 grid((unsigned)((m + CP_ROWS - 1) / CP_ROWS), (unsigned)batch);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, 3.0 * batch * (double)m * LEAF * CP_K, PROF_GEMM_SOLVE);
  hipLaunchKernelGGL(colstep_kernel, grid, dim3(256), 0, s, a);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

#endif

// C[m, nb] = A[m, 128] B[nb, 128]^T (mode 0, B lower triangular, C may be A) or C -= A B^T (mode 1); nb <= 128; `batch`
// problems at constant strides in one launch;
// operands 16-byte aligned with even leading dimensions; mode 0 reads only the first nb of A's 128 K columns (the padding
// columns of a ragged last block may hold anything).
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
