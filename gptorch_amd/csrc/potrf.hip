// Blocked fp64 Cholesky for gfx950 (lower, in place, row-major) + the triangular
// solves built on it.  Replaces torch.cholesky / torch.triangular_solve under
// functions.cholesky / functions.trtrs (functions.py:46-47, 71-76).
//
// Structure: recursive blocking down to a 128x128 leaf.
//   potrf(A):  A11 = potrf(A11);  A21 <- A21 * L11^-T;  A22 -= A21 A21^T;  potrf(A22)
//   trsm(B,L): B1 <- B1 * L11^-T; B2 -= B1 * L21^T;     B2 <- B2 * L22^-T
// Every flop outside the 64x64 leaves is an "NT" fp64-MFMA contraction
// (gemm_f64.hip) whose K extent is as large as the recursion allows, so the
// N^2 matrix is streamed O(log N) times instead of N/nb times and the trailing
// updates stay MFMA-bound rather than HBM-bound.  The leaf kernel factors its
// block AND inverts it in one workgroup (blocked elimination on [A ; I], see
// below); panel solves are then products with the stored inverses, done
// IN PLACE: a LEAF-column-wide tile covers the whole K and N extent of its rows,
// and a workgroup only stores after its last load.
//
// "Extra rows" (gpnative.h): rows n..n+e-1 ride along in every panel solve and
// trailing update of the right spine of the recursion; on exit they hold
// (L^-1 R)^T -- alpha^T of gpr.py:62 -- for free.
#include <type_traits>
#include "gpn_common.h"

namespace gpn {

// ---------------------------------------------------------------------------------
// Leaf: one workgroup (512 threads, 8 waves) factors a 128x128 diagonal block AND
// forms its inverse:  L = chol(A[0:kb,0:kb]) in place,  W = L^-1 -> winv (128x128,
// ld 128, zero outside the kb x kb lower triangle).  Rows/cols >= kb act as identity.
// FACTOR=false: A already holds a lower-triangular L; only W is formed (blockIdx.x
// selects the diagonal block).
//
// Right-looking, blocked by 8 columns, on the stacked matrix [A ; I] (256 x 128): the
// identity rows are "extra rows" exactly as in the outer algorithm, so they come out as
// I * L^-T = W^T and the inverse needs no pass of its own.  The whole trailing matrix
// lives in REGISTERS as 16x16 MFMA accumulator tiles (72 tiles, 9 per wave); per block of
// 8 pivots:
//   P0  owners of the current tile column publish the raw 8-column panel to LDS
//   P1  wave 0 factors the 8x8 diagonal block in-lane (no cross-lane traffic on the
//       serial pivot chain: rsq + Newton per pivot), publishes L8 and 1/diag
//   P2  one thread per row solves its 8 panel entries against L8 (forward substitution),
//       stores them to global (final L / W values) and back to LDS
//   P3  rank-8 update of every live tile: 2 x v_mfma_f64_16x16x4_f64 per tile
// 3 barriers per 8 pivots; the panel buffer is double-buffered.
// ---------------------------------------------------------------------------------
constexpr int XPS = 9;                 // padded row of the panel buffer (doubles)

template <bool FACTOR>
__global__ __launch_bounds__(512) void potrf_leaf_kernel(double* A, int64_t lda, int kb_, int col0_,
                                                         double* winv_, int32_t* info, int n_total) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  int kb = kb_, col0 = col0_;
  double* winv = winv_;
  if constexpr (!FACTOR) {
    col0 = blockIdx.x * LEAF;
    kb = min(LEAF, n_total - col0);
    A += (int64_t)col0 * lda + col0;
    winv += (int64_t)blockIdx.x * LEAF * LEAF;
  }
  __shared__ double Xp[2][256 * XPS];   // panel rows 0..127: A part, 128..255: identity (-> W^T) part
  __shared__ double Dg[64 + 8];         // L8 (row-major 8x8) + reciprocal diagonal
  __shared__ int failflag;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;      // D-layout: rows lr + 4r, column lc
  if (tid == 0) failflag = 0;

  // tile slots: global slot s = wave + 8k (k = 0..8); J = s / 9, idx = s % 9;
  // idx < 8-J: A-part tile row I = J + idx;  else identity-part tile row I = 8 + (idx - (8-J))
  int TI[9], TJ[9];
  d4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int s = wave + 8 * k;
    const int J = s / 9, idx = s - 9 * J;
    const int I = (idx < 8 - J) ? J + idx : 8 + (idx - (8 - J));
    TI[k] = I;
    TJ[k] = J;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + lr + 4 * r, col = 16 * J + lc;
      double v;
      if (I < 8) {
        v = (row == col) ? 1.0 : 0.0;
        if (row < kb && col <= row) v = A[(int64_t)row * lda + col];
      } else {
        v = (row - 128 == col) ? 1.0 : 0.0;
      }
      acc[k][r] = v;
    }
  }
  __syncthreads();

  for (int kb8 = 0; kb8 < 16; ++kb8) {
    const int c0 = kb8 * 8;
    const int J0 = kb8 >> 1, half = kb8 & 1;
    double* xp = Xp[kb8 & 1];
    // ---- P0: publish the raw panel (8 columns of tile column J0) --------------------
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (TJ[k] == J0 && (lc >> 3) == half) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xp[(16 * TI[k] + lr + 4 * r) * XPS + (lc & 7)] = acc[k][r];
      }
    }
    __syncthreads();
    // ---- P1: 8x8 diagonal block, in-lane, wave 0 ------------------------------------
    if (wave == 0) {
      double a[8][8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c) a[i][c] = xp[(c0 + i) * XPS + c];
      double invd[8];
      int fail = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double d = a[j][j];
        if (FACTOR) {
          if (!(d > 0.0)) {              // LAPACK dpotrf: ajj <= 0 or NaN
            if (!fail) fail = c0 + j + 1;
            d = 1.0;
          }
          double y = __builtin_amdgcn_rsq(d);
          const double hd = 0.5 * d;
          y = fma(y, fma(-hd * y, y, 0.5), y);
          y = fma(y, fma(-hd * y, y, 0.5), y);
          double sq = d * y;
          sq = fma(fma(-sq, sq, d), 0.5 * y, sq);
          a[j][j] = sq;
          invd[j] = y;
#pragma unroll
          for (int i = j + 1; i < 8; ++i) a[i][j] *= y;
#pragma unroll
          for (int c = j + 1; c < 8; ++c)
#pragma unroll
            for (int i = c; i < 8; ++i) a[i][c] = fma(-a[i][j], a[c][j], a[i][c]);
        } else {
          if (d == 0.0) {                // dtrtri: zero pivot
            if (!fail) fail = c0 + j + 1;
            d = 1.0;
          }
          invd[j] = 1.0 / d;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c <= i; ++c) Dg[i * 8 + c] = a[i][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) Dg[64 + j] = invd[j];
        if (fail) failflag = fail;
      }
    }
    __syncthreads();
    if (failflag) break;                 // uniform
    // ---- P2: one thread per panel row: forward substitution against L8 ----------------
    if (tid < 256) {
      const bool apart = tid < 128;
      const int rho = apart ? tid : tid - 128;
      const bool solve = apart ? (FACTOR && tid >= c0 + 8) : (rho <= c0 + 7);
      const bool diagrow = apart && tid >= c0 && tid < c0 + 8;
      double x[8];
      if (solve) {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = xp[tid * XPS + c];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          double v = x[c];
#pragma unroll
          for (int k2 = 0; k2 < c; ++k2) v = fma(-x[k2], Dg[c * 8 + k2], v);
          x[c] = v * Dg[64 + c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) xp[tid * XPS + c] = x[c];
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = 0.0;
      }
      if (apart) {
        if (FACTOR && tid < kb) {
          if (solve) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (c0 + c < kb) A[(int64_t)tid * lda + c0 + c] = x[c];
          } else if (diagrow) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (c <= tid - c0) A[(int64_t)tid * lda + c0 + c] = Dg[(tid - c0) * 8 + c];
          }
        }
      } else {
        // W[c0+c][rho] = (W^T)[rho][c0+c]; zeros everywhere else of the 128x128 block
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bool in = (c0 + c < kb) && (rho < kb);
          winv[(int64_t)(c0 + c) * LEAF + rho] = in ? x[c] : 0.0;
        }
      }
    }
    __syncthreads();
    // ---- P3: rank-8 update of the live tiles -------------------------------------------
    const int jact = (c0 + 8) >> 4;      // first tile column that still has unfinished columns
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int I = TI[k], J = TJ[k];
      const bool live = (J >= jact) && (I < 8 ? FACTOR : (I - 8 <= J0));
      if (live) {                        // wave-uniform
        const double* pa = xp + (16 * I + lc) * XPS + lr;
        const double* pb = xp + (16 * J + lc) * XPS + lr;
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[0], pb[0], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4], pb[4], acc[k], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  if (failflag) {
    if (tid == 0 && info && *info == 0) *info = col0 + failflag;
    // leave the rest of A untouched; publish a finite (zero) winv so later kernels stay finite
    for (int idx = tid; idx < LEAF * LEAF; idx += 512) winv[idx] = 0.0;
  }
}


struct Ctx {
  hipStream_t s;
  int64_t lda;
  double* winv;
  int32_t* info;
  int rc;
};

static inline int64_t split_point(int64_t n) {
  // largest power-of-two multiple of LEAF strictly below n
  int64_t h = LEAF;
  while (h * 2 < n) h *= 2;
  return h;
}

// B[m, kb] <- B * L^-T, L = kb x kb lower block whose first column is global column diag0
static void trsm_rec(Ctx& c, double* B, int64_t m, int64_t ldb, const double* L, int64_t ldl,
                     int64_t kb, int64_t diag0, const double* winv) {
  if (c.rc != GPN_OK || m <= 0 || kb <= 0) return;
  if (kb <= LEAF) {
    const double* W = winv + (diag0 / LEAF) * (LEAF * LEAF);
    // in place: one LEAF-wide column tile per row block (see file header)
    c.rc = gemm_nt(c.s, m, kb, LEAF, 1.0, B, ldb, W, LEAF, 0.0, B, ldb, 0, GPN_TRI_B_LOWER, /*inplace=*/1);
    return;
  }
  const int64_t h = split_point(kb);
  trsm_rec(c, B, m, ldb, L, ldl, h, diag0, winv);
  if (c.rc != GPN_OK) return;
  c.rc = gemm_nt(c.s, m, kb - h, h, -1.0, B, ldb, L + h * ldl, ldl, 1.0, B + h, ldb, 0);
  trsm_rec(c, B + h, m, ldb, L + h * ldl + h, ldl, kb - h, diag0 + h, winv);
}

static void potrf_rec(Ctx& c, double* A, int64_t n, int64_t e, int64_t col0) {
  if (c.rc != GPN_OK || n <= 0) return;
  if (n <= LEAF) {
    hipLaunchKernelGGL(potrf_leaf_kernel<true>, dim3(1), dim3(512), 0, c.s, A, c.lda, (int)n, (int)col0,
                       c.winv + (col0 / LEAF) * (LEAF * LEAF), c.info, 0);
    if (hipGetLastError() != hipSuccess) { c.rc = GPN_E_HIP; return; }
    if (e > 0) trsm_rec(c, A + n * c.lda, e, c.lda, A, c.lda, n, col0, c.winv);
    return;
  }
  const int64_t h = split_point(n);
  potrf_rec(c, A, h, 0, col0);
  double* A21 = A + h * c.lda;
  const int64_t m = n - h + e;
  trsm_rec(c, A21, m, c.lda, A, c.lda, h, col0, c.winv);
  if (c.rc != GPN_OK) return;
  c.rc = gemm_nt(c.s, m, m, h, -1.0, A21, c.lda, A21, c.lda, 1.0, A21 + h, c.lda, 1);
  potrf_rec(c, A21 + h, n - h, e, col0 + h);
}

// U_ii <- W_ii^T for every LEAF x LEAF diagonal block
__global__ __launch_bounds__(256) void diag_transpose_kernel(const double* winv, double* U, int64_t ldu, int n) {
  // 32x32 sub-tiles through LDS: blockIdx.y enumerates the (LEAF/32)^2 sub-tiles of W
  __shared__ double t[32][33];
  const int blk = blockIdx.x, tid = threadIdx.x;
  const int si = blockIdx.y / (LEAF / 32), sj = blockIdx.y % (LEAF / 32);
  const double* W = winv + (int64_t)blk * LEAF * LEAF;
  const int tx = tid & 31, ty = tid >> 5;
  for (int k = ty; k < 32; k += 8) t[k][tx] = W[(si * 32 + k) * LEAF + sj * 32 + tx];
  __syncthreads();
  const int kb = min(LEAF, n - blk * LEAF);
  double* Ub = U + ((int64_t)blk * LEAF) * ldu + blk * LEAF;
  for (int k = ty; k < 32; k += 8) {
    const int i = sj * 32 + k, c = si * 32 + tx;     // U[i][c] = W[c][i]
    if (i < kb && c < kb) Ub[(int64_t)i * ldu + c] = t[tx][k];
  }
}

// U (upper, row-major) <- L^-T by recursion on the block structure:
//   U12 = -U11 * L21^T * L22^-T : one NT contraction (A = U11 upper: K range clipped)
//   followed by the in-place right solve with L22 (trsm_rec) -- no new primitive.
static void trtri_rec(Ctx& c, const double* L, int64_t ldl, double* U, int64_t ldu, int64_t n, int64_t diag0) {
  if (c.rc != GPN_OK || n <= LEAF) return;
  const int64_t h = split_point(n);
  trtri_rec(c, L, ldl, U, ldu, h, diag0);
  trtri_rec(c, L + h * ldl + h, ldl, U + h * ldu + h, ldu, n - h, diag0 + h);
  if (c.rc != GPN_OK) return;
  c.rc = gemm_nt(c.s, h, n - h, h, -1.0, U, ldu, L + h * ldl, ldl, 0.0, U + h, ldu, 0, GPN_TRI_A_UPPER);
  trsm_rec(c, U + h, h, ldu, L + h * ldl + h, ldl, n - h, diag0 + h, c.winv);
}

// ---- reductions / utilities -------------------------------------------------
__global__ __launch_bounds__(256) void lml_reduce_kernel(const double* A, int64_t n, int64_t e, int64_t lda,
                                                         double* out3) {
  // single workgroup: sums are O(N) work
  __shared__ double red[2][256];
  const int tid = threadIdx.x;
  double ld = 0.0, sq = 0.0;
  for (int64_t i = tid; i < n; i += 256) ld += log(A[i * lda + i]);
  for (int64_t c = 0; c < e; ++c) {
    const double* row = A + (n + c) * lda;
    for (int64_t i = tid; i < n; i += 256) sq = fma(row[i], row[i], sq);
  }
  red[0][tid] = ld;
  red[1][tid] = sq;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s];
      red[1][tid] += red[1][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double logdet = red[0][0], quad = red[1][0];
    out3[0] = logdet;
    out3[1] = quad;
    // gpr.py:63-67
    out3[2] = -0.5 * quad - (double)e * logdet - 0.5 * (double)e * (double)n * 1.8378770664093454836;
  }
}

__global__ void transpose_kernel(const double* src, int64_t rows, int64_t cols, int64_t lds,
                                 double* dst, int64_t ldd) {
  __shared__ double t[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int64_t r = r0 + k, cc = c0 + tx;
    t[k][tx] = (r < rows && cc < cols) ? src[r * lds + cc] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int64_t cc = c0 + k, r = r0 + tx;   // dst[cc, r]
    if (cc < cols && r < rows) dst[cc * ldd + r] = t[tx][k];
  }
}

__global__ void copy_matrix_kernel(const double* src, int64_t rows, int64_t cols, int64_t lds,
                                   double* dst, int64_t ldd, int tril) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    double v = src[r * lds + c];
    if (tril && c > r) v = 0.0;
    dst[r * ldd + c] = v;
  }
}

__global__ __launch_bounds__(256) void row_sumsq_kernel(const double* A, int64_t rows, int64_t cols,
                                                        int64_t lda, double* out) {
  __shared__ double red[256];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  const double* row = A + r * lda;
  double s = 0.0;
  for (int64_t c = tid; c < cols; c += 256) s = fma(row[c], row[c], s);
  red[tid] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) red[tid] += red[tid + k];
    __syncthreads();
  }
  if (tid == 0) out[r] = red[0];
}

}  // namespace gpn

using namespace gpn;

extern "C" int64_t gpn_factor_ld(int64_t n, int64_t e) { return round_up(n + e, LEAF); }
// +16 zero rows: a contraction whose operand starts at row n (the extra rows) reads whole 16-row groups
extern "C" int64_t gpn_factor_rows(int64_t n, int64_t e) { return round_up(n + e, LEAF) + 16; }
extern "C" int64_t gpn_winv_bytes(int64_t n) {
  return (round_up(n, LEAF) / LEAF) * LEAF * LEAF * (int64_t)sizeof(double);
}

extern "C" int gpn_potrf_lower(void* stream, double* A, int64_t n, int64_t e, int64_t lda,
                               double* winv, int32_t* info) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < round_up(n + e, LEAF) || (lda % LEAF) != 0) return -5;
  if (!winv) return -6;
  if (!info) return -7;
  if (reinterpret_cast<uintptr_t>(A) & 15) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), lda, winv, info, GPN_OK};
  potrf_rec(c, A, n, e, 0);
  return c.rc;
}

extern "C" int gpn_trtri_diag(void* stream, const double* L, int64_t n, int64_t ldl, double* winv, int32_t* info) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < n) return -4;
  if (!winv) return -5;
  if (n == 0) return GPN_OK;
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL(potrf_leaf_kernel<false>, dim3(nb), dim3(512), 0, static_cast<hipStream_t>(stream),
                     const_cast<double*>(L), ldl, 0, 0, winv, info, (int)n);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_trsm_right_lt(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                                 double* B, int64_t m, int64_t ldb) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!winv) return -5;
  if (!B) return -6;
  if (m < 0) return -7;
  if (ldb < round_up(n, LEAF) || (ldb % LEAF) != 0) return -8;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  if (n == 0 || m == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), ldl, const_cast<double*>(winv), nullptr, GPN_OK};
  trsm_rec(c, B, m, ldb, L, ldl, n, 0, winv);
  return c.rc;
}

extern "C" int gpn_trtri_upper(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                               double* U, int64_t ldu) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!winv) return -5;
  if (!U) return -6;
  if (ldu < round_up(n, LEAF) || (ldu % LEAF) != 0) return -7;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(U) & 15)) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL(diag_transpose_kernel, dim3(nb, (LEAF / 32) * (LEAF / 32)), dim3(256), 0, s, winv, U, ldu, (int)n);
  GPN_LAUNCH_CHECK();
  Ctx c{s, ldl, const_cast<double*>(winv), nullptr, GPN_OK};
  trtri_rec(c, L, ldl, U, ldu, n, 0);
  return c.rc;
}

extern "C" int gpn_lml_reduce(void* stream, const double* A, int64_t n, int64_t e, int64_t lda, double* out3) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < n) return -5;
  if (!out3) return -6;
  hipLaunchKernelGGL(lml_reduce_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), A, n, e, lda, out3);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_transpose(void* stream, const double* src, int64_t rows, int64_t cols, int64_t lds,
                             double* dst, int64_t ldd) {
  if (!src) return -2;
  if (rows < 0) return -3;
  if (cols < 0) return -4;
  if (lds < cols) return -5;
  if (!dst) return -6;
  if (ldd < rows) return -7;
  if (rows == 0 || cols == 0) return GPN_OK;
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, rows, cols, lds, dst, ldd);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_copy_matrix(void* stream, const double* src, int64_t rows, int64_t cols, int64_t lds,
                               double* dst, int64_t ldd, int tril) {
  if (!src) return -2;
  if (rows < 0) return -3;
  if (cols < 0) return -4;
  if (lds < cols) return -5;
  if (!dst) return -6;
  if (ldd < cols) return -7;
  if (rows == 0 || cols == 0) return GPN_OK;
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535));
  hipLaunchKernelGGL(copy_matrix_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, rows, cols, lds, dst, ldd, tril);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_row_sumsq(void* stream, const double* A, int64_t rows, int64_t cols, int64_t lda, double* out) {
  if (!A) return -2;
  if (rows < 0) return -3;
  if (cols < 0) return -4;
  if (lda < cols) return -5;
  if (!out) return -6;
  if (rows == 0) return GPN_OK;
  hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream), A, rows, cols, lda, out);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}
