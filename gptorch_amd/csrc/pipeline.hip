// Whole-path entry points of libgpnative: one C call per reference method.
//   gpn_lml_forward   = GPR.log_likelihood            (gpr.py:47-67)
//   gpn_lml_backward  = its autograd backward          (CholeskyBackward0 + TriangularSolveBackward0
//                                                        + the elementwise chain)
//   gpn_predict       = GPR._predict                   (gpr.py:88-117), given the factor
// Each is the fixed sequence of the single-purpose entry points of gpnative.h on ONE stream --
// what gptorch_amd/_ops.py and _backward.py issue call by call -- for callers that are not
// Python (or do not want ~10 FFI crossings per evaluation).  No host synchronisation, no
// allocation: the caller owns the factor buffer and the workspaces.
#include "gpn_common.h"

namespace gpn {

__global__ void var_minus_kernel(double* v, const double* variance, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = variance[0] - v[i];            // Kdiag - colsumsq(A), gpr.py:109-113
}

__global__ void neg_transpose_small_kernel(const double* at, int64_t ldat, int64_t n, int dy, double* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < dy; ++c) out[i * dy + c] = -at[(int64_t)c * ldat + i];   // dLML/d(y-m) = -a
}

struct BackwardLayout {
  int64_t rows, ld, u, s, at, sweep, total;        // offsets in doubles
};

static BackwardLayout backward_layout(int64_t n, int dy, int nls) {
  BackwardLayout b;
  b.ld = gpn_factor_ld(n, 0);
  b.rows = gpn_factor_rows(n, 0);
  b.u = 0;                                          // U = L^-T
  b.s = b.u + b.rows * b.ld;                        // scratch of the inversion, then Kyy^-1 (lower)
  b.at = b.s + b.rows * b.ld;                       // a^T [round_up(dy,16), ld]
  b.sweep = b.at + round_up(dy, 16) * b.ld;
  b.total = b.sweep + (gpn_grad_work_bytes(n, n, nls, 1) + 7) / 8;
  return b;
}

}  // namespace gpn

using namespace gpn;

extern "C" int gpn_lml_forward(void* stream, int kind, const double* X, int64_t n, int d,
                               const double* Y, const double* M, int dy,
                               const double* variance, const double* length_scales, int nls,
                               const double* noise, double* A, int64_t lda, double* winv,
                               int32_t* info, double* out3) {
  if (n < 0) return -4;
  if (!Y) return -6;
  if (dy <= 0) return -8;
  if (!A) return -13;
  if (lda != gpn_factor_ld(n, dy)) return -14;
  if (!winv) return -15;
  if (!info) return -16;
  if (!out3) return -17;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = gpn_kernel_matrix(stream, kind, X, n, nullptr, n, d, variance, length_scales, nls, noise, GPN_LOWER, A, lda);
  if (rc != GPN_OK) return rc;
  // (Y - M)^T into the extra rows; the corner right of them (it accumulates -alpha alpha^T during
  // the factorisation) and the info word are cleared by the same launch
  if (n > 0) {
    rc = pack_rhs_full(s, Y, M, n, dy, A + n * lda, lda, info);
    if (rc != GPN_OK) return rc;
  } else {
    GPN_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int32_t), s));
  }
  rc = gpn_potrf_lower(stream, A, n, dy, lda, winv, info);
  if (rc != GPN_OK) return rc;
  return gpn_lml_reduce(stream, A, n, dy, lda, out3);
}

// `batch` evaluations of GPR.log_likelihood in lock step (hyper-parameter restarts, one model per entry): the reference
// can only evaluate them one after the other (gptorch/models/base.py:260-269).  Model b: points X + b sX (sX = 0: shared),
// targets Y + b sY, mean values M + b sM (or NULL), hyper-parameters variance[b], length_scales[b nls ..], noise[b];
// factor buffer A + b sA, leaf inverses winv + b sW, info[b], out3[3 b ..].  Every model's numbers are bit-identical to
// gpn_lml_forward on that model alone.
extern "C" int gpn_lml_forward_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                                       const double* Y, int64_t sY, const double* M, int64_t sM, int dy,
                                       const double* variance, const double* length_scales, int nls, const double* noise,
                                       double* A, int64_t lda, int64_t sA, double* winv, int64_t sW, int32_t* info, double* out3) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (batch < 1) return -3;
  if (!X) return -4;
  if (n < 0) return -6;
  if (d <= 0) return -7;
  if (!Y) return -8;
  if (dy <= 0) return -12;
  if (!variance) return -13;
  if (!length_scales) return -14;
  if (nls != 1 && nls != d) return -15;
  if (!noise) return -16;
  if (!A) return -17;
  if (lda != gpn_factor_ld(n, dy)) return -18;
  if (batch > 1 && (sA < gpn_factor_rows(n, dy) * lda || (sA & 1))) return -19;
  if (!winv) return -20;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -21;
  if (!info) return -22;
  if (!out3) return -23;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    GPN_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int32_t) * batch, s));
    return gpn_lml_reduce_batched(stream, A, n, dy, lda, sA, out3, batch);
  }
  int rc = assemble_batched(s, kind, batch, X, sX, n, d, Y, sY, M, sM, dy, variance, length_scales, nls, noise, A, lda, sA, info);
  if (rc != GPN_OK) return rc;
  rc = gpn_potrf_lower_batched(stream, A, n, dy, lda, sA, winv, sW, info, batch);
  if (rc != GPN_OK) return rc;
  return gpn_lml_reduce_batched(stream, A, n, dy, lda, sA, out3, batch);
}

extern "C" int64_t gpn_lml_backward_work_bytes(int64_t n, int dy, int nls) {
  if (n < 0 || dy <= 0 || nls <= 0) return 0;
  return backward_layout(n, dy, nls).total * (int64_t)sizeof(double);
}

extern "C" int gpn_lml_backward(void* stream, int kind, const double* X, int64_t n, int d,
                                const double* variance, const double* length_scales, int nls,
                                const double* A, int64_t lda, const double* winv, int dy,
                                double* work, double* grads, double* grad_resid) {
  if (n < 0) return -4;
  if (!A) return -9;
  if (dy <= 0) return -12;
  if (lda != gpn_factor_ld(n, dy)) return -10;
  if (!winv) return -11;
  if (!work) return -13;
  if (!grads) return -14;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const BackwardLayout b = backward_layout(n, dy, nls);
  double* U = work + b.u;
  double* S = work + b.s;
  double* at = work + b.at;
  GPN_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)(b.at - b.u) * sizeof(double), s));     // U and S start as zero
  int rc = (n > 256) ? gpn_trtri_upper_ws(stream, A, n, lda, winv, U, b.ld, S, b.ld)
                     : gpn_trtri_upper(stream, A, n, lda, winv, U, b.ld);
  if (rc != GPN_OK) return rc;
  const int64_t kp = round_up(n, 16);
  double* Kinv = S;                                 // the scratch is free again: Kyy^-1 = U U^T (lower)
  rc = gpn_gemm_nt(stream, n, n, kp, 1.0, U, b.ld, U, b.ld, 0.0, Kinv, b.ld, 1, GPN_TRI_A_UPPER | GPN_TRI_B_UPPER);
  if (rc != GPN_OK) return rc;
  // a^T = alpha^T U^T (alpha^T = the extra rows of the factor buffer)
  rc = gpn_gemm_nt(stream, dy, n, kp, 1.0, A + n * lda, lda, U, b.ld, 0.0, at, b.ld, 0, GPN_TRI_B_UPPER);
  if (rc != GPN_OK) return rc;
  rc = gpn_lml_grad(stream, kind, X, n, d, variance, length_scales, nls, Kinv, b.ld, at, b.ld, dy, work + b.sweep, grads);
  if (rc != GPN_OK) return rc;
  if (grad_resid) {
    hipLaunchKernelGGL(neg_transpose_small_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, at, b.ld, n, dy, grad_resid);
    GPN_LAUNCH_CHECK();
  }
  return GPN_OK;
}

extern "C" int64_t gpn_predict_work_bytes(int64_t n, int64_t ns, int dy) {
  if (n < 0 || ns < 0 || dy <= 0) return 0;
  return round_up(ns > 0 ? ns : 1, 128) * gpn_factor_ld(n, dy) * (int64_t)sizeof(double);
}

static int predict_impl(void* stream, int kind, const double* X, int64_t n, int d,
                        const double* Xs, int64_t ns, const double* Ms,
                        const double* variance, const double* length_scales, int nls,
                        const double* A, int64_t lda, const double* winv, const double* wb, int dy, int full_cov,
                        double* work, double* mean, double* var) {
  if (n < 0) return -4;
  if (!Xs) return -6;
  if (ns < 0) return -7;
  if (!A) return -11;
  if (lda != gpn_factor_ld(n, dy)) return -12;
  if (!winv) return -13;
  if (dy <= 0) return -14;
  if (!work) return -16;
  if (!mean) return -17;
  if (!var) return -18;
  if (ns == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t one = gpn_predict_work_bytes(n, ns, dy);
  double* Bt = work;                                // [round_up(ns,128), lda], zero padded
  GPN_HIP_CHECK(hipMemsetAsync(Bt, 0, (size_t)((wb && (n % 16)) ? 2 * one : one), s));    // (the second buffer: only its K padding must be zero)
  int rc = GPN_OK;
  if (n > 0) {
    rc = gpn_kernel_matrix(stream, kind, Xs, ns, X, n, d, variance, length_scales, nls, nullptr, GPN_FULL, Bt, lda);  // K(x*, X)
    if (rc != GPN_OK) return rc;
    if (wb) {                                       // A^T = K(x*, X) L^-T through the big inverted blocks, into the second buffer
      double* At = work + one / (int64_t)sizeof(double);
      rc = gpn_trsm_right_lt_blocked(stream, A, n, lda, wb, Bt, ns, lda, At, lda);
      Bt = At;
    } else {
      rc = gpn_trsm_right_lt(stream, A, n, lda, winv, Bt, ns, lda);                    // A^T = K(x*, X) L^-T
    }
    if (rc != GPN_OK) return rc;
  }
  const int64_t kp = round_up(n, 16);
  // mean = m(x*) + A^T V (gpr.py:107-108): the mean function's values at the test points, if any, are the C operand
  if (Ms) GPN_HIP_CHECK(hipMemcpyAsync(mean, Ms, (size_t)ns * dy * sizeof(double), hipMemcpyDeviceToDevice, s));
  rc = gpn_gemm_nt(stream, ns, dy, kp, 1.0, Bt, lda, A + n * lda, lda, Ms ? 1.0 : 0.0, mean, dy, 0, 0);   // (+) A^T V
  if (rc != GPN_OK) return rc;
  if (!full_cov) {
    rc = gpn_row_sumsq(stream, Bt, ns, n, lda, var);
    if (rc != GPN_OK) return rc;
    hipLaunchKernelGGL(var_minus_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, var, variance, ns);
    GPN_LAUNCH_CHECK();
    return GPN_OK;
  }
  rc = gpn_kernel_matrix(stream, kind, Xs, ns, nullptr, ns, d, variance, length_scales, nls, nullptr, GPN_FULL, var, ns);
  if (rc != GPN_OK) return rc;
  return gpn_gemm_nt(stream, ns, ns, kp, -1.0, Bt, lda, Bt, lda, 1.0, var, ns, 0, 0);  // K(x*) - A^T A
}

extern "C" int gpn_predict(void* stream, int kind, const double* X, int64_t n, int d,
                           const double* Xs, int64_t ns, const double* Ms,
                           const double* variance, const double* length_scales, int nls,
                           const double* A, int64_t lda, const double* winv, int dy, int full_cov,
                           double* work, double* mean, double* var) {
  return predict_impl(stream, kind, X, n, d, Xs, ns, Ms, variance, length_scales, nls, A, lda, winv, nullptr, dy, full_cov, work, mean, var);
}

// gpn_predict with the right-solve through the inverted 1024 x 1024 diagonal blocks (gpn_block_inverse: formed ONCE per
// factor by a caller that predicts more than once with it).  work: 2 * gpn_predict_work_bytes(n, ns, dy).
extern "C" int gpn_predict_blocked(void* stream, int kind, const double* X, int64_t n, int d,
                                   const double* Xs, int64_t ns, const double* Ms,
                                   const double* variance, const double* length_scales, int nls,
                                   const double* A, int64_t lda, const double* winv, const double* wb, int dy, int full_cov,
                                   double* work, double* mean, double* var) {
  if (!wb) return -15;
  return predict_impl(stream, kind, X, n, d, Xs, ns, Ms, variance, length_scales, nls, A, lda, winv, wb, dy, full_cov, work, mean, var);
}
