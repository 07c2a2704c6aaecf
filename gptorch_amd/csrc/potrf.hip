// Blocked fp64 Cholesky for gfx950 (lower, in place, row-major) + the triangular
// solves built on it.  Replaces torch.cholesky / torch.triangular_solve under
// functions.cholesky / functions.trtrs (functions.py:46-47, 71-76).
//
// Structure: recursive blocking down to a 64x64 leaf.
//   potrf(A):  A11 = potrf(A11);  A21 <- A21 * L11^-T;  A22 -= A21 A21^T;  potrf(A22)
//   trsm(B,L): B1 <- B1 * L11^-T; B2 -= B1 * L21^T;     B2 <- B2 * L22^-T
// Every flop outside the 64x64 leaves is an "NT" fp64-MFMA contraction
// (gemm_f64.hip) whose K extent is as large as the recursion allows, so the
// N^2 matrix is streamed O(log N) times instead of N/nb times and the trailing
// updates stay MFMA-bound rather than HBM-bound.  The leaf kernel factors its
// block AND inverts it in one workgroup (fused right-looking elimination on
// [A | I] in LDS); panel solves are then products with the stored inverses, done
// IN PLACE: a 64-column-wide tile covers the whole K and N extent of its rows,
// and a workgroup only stores after its last load.
//
// "Extra rows" (gpnative.h): rows n..n+e-1 ride along in every panel solve and
// trailing update of the right spine of the recursion; on exit they hold
// (L^-1 R)^T -- alpha^T of gpr.py:62 -- for free.
#include <type_traits>
#include "gpn_common.h"

namespace gpn {

// One workgroup: L = chol(A[0:kb,0:kb]) in place, W = L^-1 -> winv (64x64, ld 64,
// zero outside the kb x kb lower triangle).  col0 = global index of column 0
// (for info).  Rows/cols >= kb are treated as identity.
// FACTOR=false: A already holds a lower-triangular L; only the inverse is formed
// (one workgroup per 64-block: blockIdx.x selects the diagonal block).
//
// Right-looking elimination on [A | I] with the whole working set in REGISTERS:
// the 256 threads form a 16x16 grid, thread (ty,tx) owns A[ty+16a][tx+16b] and
// W[ty+16a][tx+16b], a,b = 0..3.  Per pivot column j only the (unscaled) column
// j of A and row j of W cross threads, through a double-buffered 2x64-double LDS
// broadcast: one barrier per column; every thread derives 1/sqrt(d) itself.
// Finished columns of L / rows of W stay in registers and are stored once at the end.
#define GPN_STAMP(k)                                                                       \
  if constexpr (DIAG) {                                                                    \
    unsigned long long t_;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    acc_t[k] += t_ - last_t;                                                               \
    last_t = t_;                                                                           \
  }

template <bool FACTOR, bool DIAG = false>
__global__ __launch_bounds__(256) void potrf_leaf_kernel(double* A, int64_t lda, int kb_, int col0_,
                                                         double* winv_, int32_t* info, int n_total,
                                                         unsigned long long* diag = nullptr) {
  unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, last_t = 0;
  if constexpr (DIAG) last_t = __builtin_amdgcn_s_memtime();
  int kb = kb_, col0 = col0_;
  double* winv = winv_;
  if constexpr (!FACTOR) {
    col0 = blockIdx.x * LEAF;
    kb = min(LEAF, n_total - col0);
    A += (int64_t)col0 * lda + col0;
    winv += (int64_t)blockIdx.x * LEAF * LEAF;
  }
  __shared__ double colbuf[2][LEAF];
  __shared__ double rowbuf[2][LEAF];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;

  double ar[4][4], wr_[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = ty + 16 * a, c = tx + 16 * b;
      double v = (i == c) ? 1.0 : 0.0;
      if (i < kb && c <= i) v = A[(int64_t)i * lda + c];
      ar[a][b] = v;
      wr_[a][b] = (i == c) ? 1.0 : 0.0;
    }

  int fail = 0;
  // broadcast of column 0 / row 0
  if (tx == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) colbuf[0][ty + 16 * a] = ar[a][0];
  }
  if (ty == 0) {
#pragma unroll
    for (int b = 0; b < 4; ++b) rowbuf[0][tx + 16 * b] = wr_[0][b];
  }

  // 4 blocks of 16 pivots; the block index is a compile-time constant so that every
  // register-array subscript below is static (no selects, no scratch)
  auto pivots16 = [&](auto jb_c) {
    constexpr int jb = decltype(jb_c)::value;
    for (int jt = 0; jt < 16; ++jt) {
      const int j = jb * 16 + jt;
      const int p = j & 1;
      GPN_STAMP(4)
      __syncthreads();
      GPN_STAMP(0)
      const double d = colbuf[p][j];
      if (FACTOR ? !(d > 0.0) : (d == 0.0)) {   // LAPACK dpotrf: ajj <= 0 or NaN; dtrtri: zero pivot
        fail = j + 1;                            // d comes from LDS: uniform across the workgroup
        return;
      }
      // values every thread needs from column j / row j (all LDS reads issued together)
      double cl[4], cc[4], rw[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) cl[a] = colbuf[p][ty + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        cc[b] = colbuf[p][tx + 16 * b];
        rw[b] = rowbuf[p][tx + 16 * b];
      }
      // 1/sqrt(d) by v_rsq_f64 + two Newton steps (the serial chain of the whole
      // factorisation runs through here: ~6 dependent ops instead of sqrt + divide);
      // s = d * inv_s is then within 1 ulp of sqrt(d).  trtri mode: inv_s = 1/d.
      double s, inv_s;
      if (FACTOR) {
        double y = __builtin_amdgcn_rsq(d);
        const double hd = 0.5 * d;
        y = fma(y, fma(-hd * y, y, 0.5), y);
        y = fma(y, fma(-hd * y, y, 0.5), y);
        inv_s = y;
        s = d * y;
        s = fma(fma(-s, s, d), 0.5 * y, s);   // one correction of sqrt(d) itself
      } else {
        s = d;
        inv_s = 1.0 / d;
      }
      // rows/cols at or before the pivot take no part: blocks a < jb (b < jb) are dead
      // for the A part, blocks b > jb are dead for the W part -- all decided statically
      double li[4], lc[4], wj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a < jb) li[a] = 0.0;
        else if (a == jb) li[a] = (ty > jt) ? (FACTOR ? cl[a] * inv_s : cl[a]) : 0.0;
        else li[a] = FACTOR ? cl[a] * inv_s : cl[a];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (!FACTOR || b < jb) lc[b] = 0.0;
        else if (b == jb) lc[b] = (tx > jt) ? cc[b] * inv_s : 0.0;
        else lc[b] = cc[b] * inv_s;
        if (b > jb) wj[b] = 0.0;
        else if (b == jb) wj[b] = (tx <= jt) ? rw[b] * inv_s : 0.0;
        else wj[b] = rw[b] * inv_s;
      }
      GPN_STAMP(1)
      // look-ahead: bring column j+1 of A / row j+1 of W up to date FIRST and publish
      // them through the other buffer, so the next pivot's chain starts while the bulk
      // of this rank-1 update is still being issued
      if (jt < 15) {
        if (tx == jt + 1) {
#pragma unroll
          for (int a = jb; a < 4; ++a)
            colbuf[p ^ 1][ty + 16 * a] = FACTOR ? fma(-li[a], lc[jb], ar[a][jb]) : ar[a][jb];
        }
        if (ty == jt + 1) {
#pragma unroll
          for (int b = 0; b <= jb; ++b) rowbuf[p ^ 1][tx + 16 * b] = fma(-li[jb], wj[b], wr_[jb][b]);
        }
      } else if constexpr (jb < 3) {
        if (tx == 0) {
#pragma unroll
          for (int a = jb + 1; a < 4; ++a)
            colbuf[p ^ 1][ty + 16 * a] = FACTOR ? fma(-li[a], lc[jb + 1], ar[a][jb + 1]) : ar[a][jb + 1];
        }
        if (ty == 0) {
#pragma unroll
          for (int b = 0; b <= jb + 1; ++b) rowbuf[p ^ 1][tx + 16 * b] = fma(-li[jb + 1], wj[b], wr_[jb + 1][b]);
        }
      }
      GPN_STAMP(2)
      // finished column j of L / row j of W stay in their owners' registers (the bulk
      // update below leaves them alone: lc = 0 for c <= j, li = 0 for i <= j)
      if (FACTOR && tx == jt) {
        ar[jb][jb] = (ty == jt) ? s : (ty > jt ? li[jb] : ar[jb][jb]);
#pragma unroll
        for (int a = jb + 1; a < 4; ++a) ar[a][jb] = li[a];
      }
      if (ty == jt) {
#pragma unroll
        for (int b = 0; b < jb; ++b) wr_[jb][b] = wj[b];
        if (tx <= jt) wr_[jb][jb] = wj[jb];
      }
      GPN_STAMP(3)
      // rank-1 update of the trailing rows (registers only; dead blocks skipped statically)
#pragma unroll
      for (int a = jb; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (FACTOR && b >= jb) ar[a][b] = fma(-li[a], lc[b], ar[a][b]);
          if (b <= jb) wr_[a][b] = fma(-li[a], wj[b], wr_[a][b]);
        }
    }
  };
  pivots16(std::integral_constant<int, 0>{});
  if (!fail) pivots16(std::integral_constant<int, 1>{});
  if (!fail) pivots16(std::integral_constant<int, 2>{});
  if (!fail) pivots16(std::integral_constant<int, 3>{});
  __syncthreads();
  if constexpr (DIAG) {
    GPN_STAMP(5)
    if ((tid & 63) == 0) for (int k = 0; k < 6; ++k) diag[(tid >> 6) * 6 + k] = acc_t[k];
  }
  if (fail) {
    if (tid == 0 && info && *info == 0) *info = col0 + fail;
    // leave A untouched; still publish a finite winv so later kernels stay finite
    for (int idx = tid; idx < LEAF * LEAF; idx += 256) winv[idx] = 0.0;
    return;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = ty + 16 * a, c = tx + 16 * b;
      const bool in = (i < kb && c <= i);
      if (FACTOR && in) A[(int64_t)i * lda + c] = ar[a][b];
      winv[i * LEAF + c] = in ? wr_[a][b] : 0.0;
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
    // in place: one 64-wide column tile per row block (see file header)
    c.rc = gemm_nt(c.s, m, kb, LEAF, 1.0, B, ldb, W, LEAF, 0.0, B, ldb, 0);
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
    hipLaunchKernelGGL((potrf_leaf_kernel<true, false>), dim3(1), dim3(256), 0, c.s, A, c.lda, (int)n, (int)col0,
                       c.winv + (col0 / LEAF) * (LEAF * LEAF), c.info, 0, nullptr);
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

// U_ii <- W_ii^T for every 64x64 diagonal block (one workgroup per block)
__global__ __launch_bounds__(256) void diag_transpose_kernel(const double* winv, double* U, int64_t ldu, int n) {
  __shared__ double t[LEAF][LEAF + 1];
  const int blk = blockIdx.x, tid = threadIdx.x;
  const double* W = winv + (int64_t)blk * LEAF * LEAF;
  for (int idx = tid; idx < LEAF * LEAF; idx += 256) t[idx >> 6][idx & 63] = W[idx];
  __syncthreads();
  const int kb = min(LEAF, n - blk * LEAF);
  double* Ub = U + ((int64_t)blk * LEAF) * ldu + blk * LEAF;
  for (int idx = tid; idx < LEAF * LEAF; idx += 256) {
    const int i = idx >> 6, c = idx & 63;
    if (i < kb && c < kb) Ub[(int64_t)i * ldu + c] = t[c][i];
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

// diagnostic build of the leaf with s_memtime stamps: diag[wave*6 + k] = cycles summed
// over the 64 columns in segment k (0 barrier, 1 pivot+reads, 2 look-ahead publish,
// 3 park, 4 bulk update, 5 tail).  Not part of the public header.
extern "C" int gpn_debug_leaf_timing(void* stream, double* A, int64_t lda, double* winv, int32_t* info,
                                     unsigned long long* diag24) {
  hipLaunchKernelGGL((potrf_leaf_kernel<true, true>), dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                     A, lda, LEAF, 0, winv, info, 0, diag24);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_trtri_diag(void* stream, const double* L, int64_t n, int64_t ldl, double* winv, int32_t* info) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < n) return -4;
  if (!winv) return -5;
  if (n == 0) return GPN_OK;
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL((potrf_leaf_kernel<false, false>), dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream),
                     const_cast<double*>(L), ldl, 0, 0, winv, info, (int)n, nullptr);
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
  hipLaunchKernelGGL(diag_transpose_kernel, dim3(nb), dim3(256), 0, s, winv, U, ldu, (int)n);
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
