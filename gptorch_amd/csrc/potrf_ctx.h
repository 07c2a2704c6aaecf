// The factorisation context and the right-solve recursion shared by the factorisation drivers (potrf.hip) and the
// triangular solves / inversions built on the factor (trisolve.hip).  Included by those two sources only.
#pragma once
#include "gpn_common.h"

namespace gpn {

struct Ctx {
  hipStream_t s;
  int64_t lda;
  double* winv;
  int32_t* info;
  int rc;
  // corner = true (factor buffers): the trailing update is one lower-tile square over the matrix rows AND the
  // extra rows, so the e x e corner right of column n accumulates -R R^T garbage (the buffer has room for it).
  // corner = false (a tile column of a larger matrix, gpn_potrf_lower_panel): nothing right of column n is
  // touched -- the extra rows get a rectangular update of their own.
  bool corner = true;
  // `batch` independent factorisations of identical shape in lock step (gpn_potrf_lower_batched): problem b lives at
  // A + b sA, winv + b sW, info + b; every launch of the drivers below covers all of them
  int batch = 1;
  int64_t sA = 0, sW = 0;
  // right-solves of a batch whose right-hand sides live OUTSIDE the factor buffers (gpn_trsm_right_lt_batched): their stride
  // (-1: inside the factor buffers, sA)
  int64_t sRhs = -1;
};

// the drivers' contraction / column-pass launches, batched when the context is
static inline int cgemm(const Ctx& c, hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int lower, int tri = 0, int inplace = 0) {
  if (c.batch == 1) return gemm_nt(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, inplace);
  return gemm_nt_strided(s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lower, tri, inplace, c.batch, c.sA, c.sA, c.sA);
}
// mode 0: B = an inverted leaf block of winv; mode 1: everything inside the factor buffers
static inline int ccolpanel(const Ctx& c, hipStream_t s, int mode, int64_t m, int64_t nb, const double* A, int64_t lda, const double* B,
                            int64_t ldb, double* C, int64_t ldc) {
  return colpanel(s, mode, m, nb, A, lda, B, ldb, C, ldc, c.batch, c.sA, mode == 0 ? c.sW : c.sA, c.sA);
}

// one factor leaf (or `batch` of them at constant strides) on stream s
// one factor leaf (or `batch` of them at constant strides) on stream s
static int launch_leaf(hipStream_t s, double* A, int64_t lda, int kb, int col0, double* W, int32_t* info, int batch = 1,
                       int64_t sA = 0, int64_t sW = 0, int64_t sInfo = 0) {
  return leaf16(s, A, lda, kb, col0, W, info, batch, sA, sW, sInfo);
}

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
  const int64_t sB = c.sRhs >= 0 ? c.sRhs : c.sA;
  if (kb <= LEAF) {
    const double* W = winv + (diag0 / LEAF) * (LEAF * LEAF);
    // in place: one LEAF-wide column tile per row block (see file header)
    c.rc = colpanel(c.s, 0, m, kb, B, ldb, W, LEAF, B, ldb, c.batch, sB, c.sW, sB);
    return;
  }
  const int64_t h = split_point(kb);
  trsm_rec(c, B, m, ldb, L, ldl, h, diag0, winv);
  if (c.rc != GPN_OK) return;
  c.rc = c.batch == 1 ? gemm_nt(c.s, m, kb - h, h, -1.0, B, ldb, L + h * ldl, ldl, 1.0, B + h, ldb, 0, 0, 0)
                      : gemm_nt_strided(c.s, m, kb - h, h, -1.0, B, ldb, L + h * ldl, ldl, 1.0, B + h, ldb, 0, 0, 0, c.batch, sB, c.sA, sB);
  trsm_rec(c, B + h, m, ldb, L + h * ldl + h, ldl, kb - h, diag0 + h, winv);
}

}  // namespace gpn
