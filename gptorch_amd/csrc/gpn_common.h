// Shared host-side helpers for libgpnative (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gpnative.h"

// A/B switches.  In the product library they are compile-time constants; in the tools' build (libgpnative_dbg.so, compiled with
// -DGPN_DEBUG_SWITCHES by build.sh) they are per-thread variables behind the gpn_debug_* entry points -- other PARAMETRISATIONS of
// the shipped code (forced tile shapes, panel widths, one launch per block instead of the persistent back-substitution ...)
// that the cross-check tests hold against each other.  This is the only conditional region: sources declare a switch with
// GPN_SWITCH and wrap what only the tools' build has in GPN_DEBUG_ONLY(...).
#ifdef GPN_DEBUG_SWITCHES
#define GPN_SWITCH static thread_local
#define GPN_DEBUG_ONLY(...) __VA_ARGS__
#else
#define GPN_SWITCH static constexpr
#define GPN_DEBUG_ONLY(...)
#endif

namespace gpn {

constexpr int LEAF = 128;  // diagonal leaf block (potrf + inverse in one workgroup) = padding granule of factor buffers

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

void set_hip_error(hipError_t e, const char* where);

#define GPN_HIP_CHECK(expr)                          \
  do {                                               \
    hipError_t _e = (expr);                          \
    if (_e != hipSuccess) {                          \
      gpn::set_hip_error(_e, #expr);                 \
      return GPN_E_HIP;                              \
    }                                                \
  } while (0)

#define GPN_LAUNCH_CHECK() GPN_HIP_CHECK(hipGetLastError())

// C = alpha*A*B^T + beta*C, see gpnative.h gpn_gemm_nt.  Internal entry used by
// the factorisation drivers (no argument validation).
int gemm_nt(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
            const double* A, int64_t lda, const double* B, int64_t ldb,
            double beta, double* C, int64_t ldc, int lower, int tri = 0, int inplace = 0);

// gpn_potrf_lower as one persistent launch (ppotrf.hip); GPN_E_UNSUPPORTED = not this size / not under this capture
int potrf_persistent(hipStream_t s, double* A, int64_t n, int64_t e, int64_t lda, double* winv, int32_t* info);
void potrf_persistent_release(hipStream_t s);

// staircase: C is M x (nblocks * blk); column block b has the rows from b * step on; diag: its first blk x blk square
// is lower-only (gpnative.h gpn_gemm_nt_stair)
int gemm_nt_stair(hipStream_t s, int64_t M, int64_t nblocks, int64_t blk, int64_t K, double alpha,
                  const double* A, int64_t lda, const double* B, int64_t ldb,
                  double beta, double* C, int64_t ldc, int64_t step, int diag);

// the in-panel chain's column passes (colpanel.hip): C[m, nb] = A[m, 128] B[nb, 128]^T with B lower triangular (mode 0; C may
// be A) or C -= A B^T (mode 1); nb <= 128
int colpanel(hipStream_t s, int mode, int64_t m, int64_t nb, const double* A, int64_t lda, const double* B, int64_t ldb,
             double* C, int64_t ldc, int batch = 1, int64_t sA = 0, int64_t sB = 0, int64_t sC = 0);

// the 128 x 128 factor leaf, second generation (leaf16.hip): `batch` independent leaves in one launch, problem b at
// A + b sA, winv + b sW, info + b sInfo
int leaf16(hipStream_t s, double* A, int64_t lda, int kb, int col0, double* winv, int32_t* info, int batch, int64_t sA,
           int64_t sW, int64_t sInfo);
// `batch` problems of identical shape at constant strides (elements) in one launch
int gemm_nt_batched(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                    const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                    double beta, double* C, int64_t ldc, int64_t sC, int tri, int batch);

// gemm_nt for `batch` problems at constant strides, with the lower-tile forms (the batched factorisation's updates)
int gemm_nt_strided(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                    const double* A, int64_t lda, const double* B, int64_t ldb,
                    double beta, double* C, int64_t ldc, int lower, int tri, int inplace,
                    int batch, int64_t sA, int64_t sB, int64_t sC);

// two-level strided batch: `outer` lock-step models (strides s*2) x `inner` equal problems inside each model (strides s*)
int gemm_nt_strided2(hipStream_t s, int64_t M, int64_t N, int64_t K, double alpha,
                     const double* A, int64_t lda, const double* B, int64_t ldb,
                     double beta, double* C, int64_t ldc, int lower, int tri,
                     int inner, int64_t sA, int64_t sB, int64_t sC, int outer, int64_t sA2, int64_t sB2, int64_t sC2);

// gpn_lml_grad_batched with the models' partial-sum regions sWork doubles apart (grad.hip)
int lml_grad_batched(hipStream_t s, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                     const double* variance, const double* length_scales, int nls,
                     const double* Kinv, int64_t ldk, int64_t sK, const double* at, int64_t ldat, int64_t sAt, int dy,
                     double* work, int64_t sWork, double* out, const int32_t* n_of = nullptr);
// U_b = L_b^-T of `batch` lock-step models (trisolve.hip)
int trtri_upper_ws_batched(hipStream_t s, const double* L, int64_t n, int64_t ldl, int64_t sL, const double* winv, int64_t sW,
                           double* U, int64_t ldu, int64_t sU, double* S, int64_t lds, int64_t sS, int batch);

// gpn_lml_reduce_batched with one point count per model (matutil.hip; gpn_lml_forward_ragged)
int lml_reduce_ragged(hipStream_t s, const double* A, int64_t n, int64_t e, int64_t lda, int64_t sA, double* out3, int batch,
                      const int32_t* n_of);
// extra rows <- (Y - M)^T, corner right of them <- 0, *info <- 0 (kmat.hip; used by gpn_lml_forward)
int pack_rhs_full(hipStream_t s, const double* Y, const double* M, int64_t n, int dy, double* E, int64_t lde,
                  int32_t* info);

// K(X) + noise I (lower tiles) into A and into Ksave (kmat.hip; gpn_lml_forward_saving)
int assemble_lower_saving(hipStream_t s, int kind, const double* X, int64_t n, int d, const double* variance, const double* length_scales,
                          int nls, const double* noise, double* A, double* Ksave, int64_t lda);
// K(X_b) + noise_b I (lower tiles) + right-hand sides + info words of `batch` models (kmat.hip; gpn_lml_forward_batched)
int assemble_batched(hipStream_t s, int kind, int batch, const double* X, int64_t sX, int64_t n, int d, const double* Y, int64_t sY,
                     const double* M, int64_t sM, int dy, const double* variance, const double* length_scales, int nls,
                     const double* noise, double* A, int64_t lda, int64_t sA, int32_t* info, const int32_t* n_of = nullptr);

// launch classes of the optional HIP-event profiler (profile.hip; bench.py's roofline legs)
enum { PROF_GEMM = 0,         // rectangular contraction (in-panel updates, predict, VFE ...)
       PROF_GEMM_SYRK = 1,    // lower-tile contraction: the SYRK trailing updates of the factorisation
       PROF_GEMM_SOLVE = 2,   // in-place panel solve against an inverted leaf block
       PROF_GEMM_TRI = 3,     // K-clipped contraction (triangular inversion, Kyy^-1 = U U^T)
       PROF_KMAT = 4,         // K assembly (work = algorithmic bytes)
       PROF_GRAD = 5,         // gradient sweep (work = algorithmic bytes)
       PROF_LEAF = 6,         // 128x128 diagonal leaf
       PROF_NCLASSES = 7 };
// release the exchange streams/events gpn_dist_lml_forward keeps for a caller stream (dist.hip)
void dist_release(hipStream_t s);

// the residual pass of the refinement step for a covariance expression (kexpr.hip; refine.hip gpn_lml_refine_expr): tiles
// q0 <= q < q0 + cnt of the lower 64 x 64 tiles times a [dy][lds] into prow / pcol (slot q - q0), double-double
int expr_resid(hipStream_t s, const gpn_expr_term* terms, int nterms, const int* gstart, int ngroups, const double* theta,
               const double* X, int64_t n, int d, const double* noise, const double* a, int dy, int64_t lds, int64_t q0, int64_t cnt,
               double* prow, double* pcol);

bool profile_on();
int profile_begin(hipStream_t s, double work, int cls);
void profile_end(hipStream_t s, int idx);

}  // namespace gpn
