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
#include "gemm_tile.h"

namespace gpn {

template <int BM, int BN, int WM, int WN, bool DMA, int NS = 2, bool BLOW = false, bool PIPE = false>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), ((BM / WM) * (BN / WN) > 8 ? 1 : 2)) void gemm_nt_kernel(GemmArgs p) {
  gemm_nt_tile<BM, BN, WM, WN, DMA, NS, BLOW, PIPE>(p, (int)blockIdx.x, (int)gridDim.x, false);
}

// Switches (gpn_common.h: constants in the product library, per-thread variables behind gpn_debug_set_gemm_variant /
// gpn_debug_set_thin_tiles in the tools' build) -- forcing one tile shape for a launch is how tests/test_gpu_gemm.py holds every
// shape against the reference product (and against each other: bit-identical).
GPN_SWITCH int g_gemm_variant = 0;  // 0 = the shipped dispatch, 1 = register staging, 3..11 = forced tile shapes
GPN_SWITCH int g_thin_tiles = 1;    // the thin-tile path (reaches the kernels through GemmArgs)
GPN_SWITCH int g_smem_pad = 0;      // extra dynamic LDS per workgroup (KiB): lowers the occupancy
static constexpr int g_group_h = 8;
static constexpr int g_big_tile_min_trapezoid = 4096;   // trapezoid launches (nested panels): 128 x 128 tiles from this many of them

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
                    : a.lower      ? a.mt * (a.mt + 1) / 2 : a.mt * a.nt) * std::max(1, a.batch);
  if (grid <= 0) return GPN_OK;
  const int smem = ((BM + BN) / 16) * 2 * 1024 * NS + g_smem_pad * 1024;
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
                          : a.lower      ? 0.5 * a.mt * (a.mt + 1.0) : (double)a.mt * a.nt) * std::max(1, a.batch);
    const int cls = inplace ? PROF_GEMM_SOLVE : (a.tri ? PROF_GEMM_TRI : (a.lower == 1 ? PROF_GEMM_SYRK : PROF_GEMM));
    rec = profile_begin(s, tiles * 2.0 * BM * BN * (double)a.K, cls);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (BM / WM) * (BN / WN)), smem, s, a);
  if (prof) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

struct Stair { int blk = 0, step = 0, diag = 0; };
struct Outer { int count = 0; int64_t sA = 0, sB = 0, sC = 0; };     // second batch level (count == 0: none)
static int gemm_nt_impl(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb,
                        double beta, double* C, int64_t ldc, int lower, int tri, int inplace,
                        int batch, int64_t sA, int64_t sB, int64_t sC, Stair st = Stair(), Outer ob = Outer(),
                        const double* alpha_dev = nullptr);

int gemm_nt(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb,
            double beta, double* C, int64_t ldc, int lower, int tri, int inplace) {
  return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, inplace, 1, 0, 0, 0);
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
  return gemm_nt_impl(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, 0, inner, sA, sB, sC, Stair(), ob);
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
                        int batch, int64_t sA, int64_t sB, int64_t sC, Stair st, Outer ob, const double* alpha_dev) {
  if (M <= 0 || N <= 0 || batch <= 0) return GPN_OK;
  GemmArgs a;
  a.alpha_dev = alpha_dev;
  a.batch = batch; a.sA = sA; a.sB = sB; a.sC = sC;
  a.inner = 0; a.sA2 = a.sB2 = a.sC2 = 0;
  a.acc_in = nullptr; a.acc_out = nullptr;
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
  const bool small = tri ? !(K >= 8192 && t128 >= (lower ? 4096 : 8192) && M > 64 && N > 64)
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

// (libgpnative_dbg.so only; the calling thread's launches)
GPN_DEBUG_ONLY(
extern "C" int gpn_debug_set_thin_tiles(int on) { gpn::g_thin_tiles = on; return GPN_OK; }
extern "C" int gpn_debug_set_gemm_variant(int v) {
  gpn::g_gemm_variant = v & 0x7f;     // forced tile shape (see gemm_nt_impl)
  gpn::g_smem_pad = v >> 8;           // bits 8..: KiB of LDS padding per workgroup
  return GPN_OK;
})

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

// gpn_gemm_nt_batched with one scale per problem, read from device memory: C_z = alphas[z] A_z B_z^T + beta C_z.  Per problem
// bit-identical to gpn_gemm_nt(alpha = alphas[z]) -- lock-step sparse models scale by their own 1 / noise variance
// (sparse_gpr.py:131-135 divides by sigma per model).
extern "C" int gpn_gemm_nt_batched_scaled(void* stream, int64_t M, int64_t N, int64_t K, const double* alphas,
                                          const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                                          double beta, double* C, int64_t ldc, int64_t sC, int lower, int tri, int batch) {
  if (M < 0) return -2;
  if (N < 0) return -3;
  if (K <= 0 || (K % 16) != 0) return -4;
  if (!alphas) return -5;
  if (batch < 1) return -18;
  if (M == 0 || N == 0) return GPN_OK;
  if (!A) return -6;
  if (!B) return -9;
  if (!C) return -13;
  if (lower && M != N) return -16;
  if (tri < 0 || tri > 15) return -17;
  if ((lda % 2) || (ldb % 2) || (sA % 2) || (sB % 2)) return GPN_E_ALIGN;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  return gpn::gemm_nt_impl(static_cast<hipStream_t>(stream), M, N, K, 0.0, A, lda, B, ldb, beta, C, ldc,
                           lower ? 1 : 0, tri, 0, batch, sA, sB, sC, gpn::Stair(), gpn::Outer(), alphas);
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
