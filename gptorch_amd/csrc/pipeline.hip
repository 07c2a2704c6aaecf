// Whole-path entry points of libgpnative: one C call per reference method.
//   gpn_lml_forward   = GPR.log_likelihood            (gpr.py:47-67)
//   gpn_lml_backward  = its autograd backward          (CholeskyBackward0 + TriangularSolveBackward0
//                                                        + the elementwise chain)
//   gpn_predict       = GPR._predict                   (gpr.py:88-117), given the factor
// Each is the fixed sequence of the single-purpose entry points of gpnative.h on ONE stream --
// what gptorch_amd/_ops.py and _backward.py issue call by call -- for callers that are not
// Python (or do not want ~10 FFI crossings per evaluation).  No host synchronisation, no
// allocation: the caller owns the factor buffer and the workspaces.
#include <algorithm>
#include "gpn_common.h"

namespace gpn {

// The tail of a prediction in ONE pass over A^T = K(x*, X) L^-T [ns, n]: per test point (one workgroup per row)
//   mean[i, c] = (m(x*)[i, c]) + sum_k A^T[i, k] V[c][k]   (V = the factor's extra rows, gpr.py:107-108)
//   var[i]     = variance - sum_k A^T[i, k]^2              (diagonal case, gpr.py:109-113; `var` NULL: mean only)
// As two launches the mean was a 1-column contraction (32 workgroups walking K = N in 16-wide steps: 127 us at C2) and the
// sum of squares a second pass over the same 68 MB.  Up to PT_DY right-hand sides per pass; fixed summation order.
typedef double pd2 __attribute__((ext_vector_type(2)));
constexpr int PT_DY = 4;
__global__ __launch_bounds__(256) void predict_tail_kernel(const double* __restrict__ At, int64_t lda, int64_t n, const double* __restrict__ V,
                                                           int dy, int c0, const double* __restrict__ Ms, const double* __restrict__ variance,
                                                           double* __restrict__ mean, double* __restrict__ var) {
  __shared__ double red[PT_DY + 1][256];
  const int t = threadIdx.x;
  const int64_t i = blockIdx.x;
  const double* row = At + i * lda;
  const int nc = min(PT_DY, dy - c0);
  double acc[PT_DY], sq = 0.0;
#pragma unroll
  for (int c = 0; c < PT_DY; ++c) acc[c] = 0.0;
  const int64_t n2 = n & ~(int64_t)1;
  for (int64_t k = 2 * t; k < n2; k += 512) {
    const pd2 a = *reinterpret_cast<const pd2*>(row + k);
    sq = fma(a.y, a.y, fma(a.x, a.x, sq));
#pragma unroll
    for (int c = 0; c < PT_DY; ++c)
      if (c < nc) {
        const pd2 v = *reinterpret_cast<const pd2*>(V + (int64_t)(c0 + c) * lda + k);
        acc[c] = fma(a.y, v.y, fma(a.x, v.x, acc[c]));
      }
  }
  if (t == 0 && n2 < n) {                                 // odd n: the last entry
    const double a = row[n2];
    sq = fma(a, a, sq);
    for (int c = 0; c < nc; ++c) acc[c] = fma(a, V[(int64_t)(c0 + c) * lda + n2], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < PT_DY; ++c) red[c][t] = acc[c];
  red[PT_DY][t] = sq;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (t < h) {
#pragma unroll
      for (int c = 0; c <= PT_DY; ++c) red[c][t] += red[c][t + h];
    }
    __syncthreads();
  }
  if (t < nc) mean[i * dy + c0 + t] = (Ms ? Ms[i * dy + c0 + t] : 0.0) + red[t][0];
  if (t == 0 && var && c0 == 0) var[i] = variance[0] - red[PT_DY][0];
}

// zero what the assembly / the solves do not write of a [rows_padded, ld] operand buffer: the columns [n, ld) of its `rows`
// data rows (K padding of the contractions) and the padding rows whole -- instead of a memset of the whole buffer (68 MB at C2)
__global__ __launch_bounds__(256) void zero_padding_kernel(double* B, int64_t ld, int64_t rows, int64_t n) {
  double* row = B + (int64_t)blockIdx.x * ld;
  for (int64_t k = ((int64_t)blockIdx.x < rows ? n : 0) + threadIdx.x; k < ld; k += 256) row[k] = 0.0;
}

__global__ void neg_transpose_small_kernel(const double* at, int64_t ldat, int64_t n, int dy, double* out, int64_t sAt = 0) {
  at += (int64_t)blockIdx.y * sAt;                  // gridDim.y models of a lock-step batch, out [batch, n, dy]
  out += (int64_t)blockIdx.y * n * dy;
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

// gpn_lml_forward that also keeps a pristine copy of the lower triangle of Kyy in Ksave [n, lda] (the factorisation overwrites it
// in A): gpn_lml_refine_dense(stream, Ksave, lda, 0.0, ...) then READS the matrix for its residual pass instead of re-computing
// every kernel entry -- the refinement step of large evaluations (gpn_lml_refine) spends two thirds of its time there.
extern "C" int gpn_lml_forward_saving(void* stream, int kind, const double* X, int64_t n, int d,
                                      const double* Y, const double* M, int dy,
                                      const double* variance, const double* length_scales, int nls,
                                      const double* noise, double* A, int64_t lda, double* winv,
                                      int32_t* info, double* out3, double* Ksave) {
  if (!X) return -3;
  if (n <= 0) return -4;
  if (d <= 0) return -5;
  if (!Y) return -6;
  if (dy <= 0) return -8;
  if (!variance) return -9;
  if (!length_scales) return -10;
  if (nls != 1 && nls != d) return -11;
  if (!A) return -13;
  if (lda != gpn_factor_ld(n, dy)) return -14;
  if (!winv) return -15;
  if (!info) return -16;
  if (!out3) return -17;
  if (!Ksave) return -18;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = assemble_lower_saving(s, kind, X, n, d, variance, length_scales, nls, noise, A, Ksave, lda);
  if (rc != GPN_OK) return rc;
  rc = pack_rhs_full(s, Y, M, n, dy, A + n * lda, lda, info);
  if (rc != GPN_OK) return rc;
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

// gpn_lml_forward_batched for models of DIFFERENT sizes (k-fold folds of unequal length, learning curves): model b has n_of[b] <= n
// points (256 < n_of[b]; its points X + b sX, its right-hand sides Y + b sY, both padded to n rows) and is evaluated as the n x n
// problem [Kyy_b 0; 0 I] -- identity rows, zero right-hand sides -- by the SAME launches as the equal-size batch.  The identity
// block factors to itself, adds log 1 to the log-determinant and 0 to the quadratic form, and every launch of the drivers treats
// an entry by its POSITION (panel boundaries are multiples of the panel widths from the top-left corner): each model's factor,
// alpha and out3 are BIT-IDENTICAL to gpn_lml_forward on its own n_of[b] points, provided n and every n_of[b] select the same
// panel levels (gpn_potrf_panel_levels) -- the caller groups accordingly.  n_of: device array of int32.
extern "C" int gpn_lml_forward_ragged(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, const int32_t* n_of, int d,
                                      const double* Y, int64_t sY, int dy,
                                      const double* variance, const double* length_scales, int nls, const double* noise,
                                      double* A, int64_t lda, int64_t sA, double* winv, int64_t sW, int32_t* info, double* out3) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (batch < 1) return -3;
  if (!X) return -4;
  if (n <= 2 * 128) return -6;
  if (!n_of) return -7;
  if (d <= 0) return -8;
  if (!Y) return -9;
  if (dy <= 0) return -11;
  if (!variance) return -12;
  if (!length_scales) return -13;
  if (nls != 1 && nls != d) return -14;
  if (!noise) return -15;
  if (!A) return -16;
  if (lda != gpn_factor_ld(n, dy)) return -17;
  if (batch > 1 && (sA < gpn_factor_rows(n, dy) * lda || (sA & 1))) return -18;
  if (!winv) return -19;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -20;
  if (!info) return -21;
  if (!out3) return -22;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = assemble_batched(s, kind, batch, X, sX, n, d, Y, sY, nullptr, 0, dy, variance, length_scales, nls, noise, A, lda, sA, info, n_of);
  if (rc != GPN_OK) return rc;
  rc = gpn_potrf_lower_batched(stream, A, n, dy, lda, sA, winv, sW, info, batch);
  if (rc != GPN_OK) return rc;
  return lml_reduce_ragged(s, A, n, dy, lda, sA, out3, batch, n_of);
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
  // U and S start as zero -- where anything reads what the schedule does not write.  For n a multiple of the 128-wide leaf
  // (no ragged block, no K padding) and the level-parallel inversion, every entry a contraction reads has been written by a
  // launch before it: the leaf transposes write whole diagonal blocks (zeros below the diagonal included), every product is
  // K-clipped to the blocks on or above the diagonal, the scratch is written (beta = 0 / transposes) before it is read.  The
  // two clears are 2 x 8.6 GB at N = 32768 (3.2 ms of a 545 ms step) and 1.5 ms of a 53 ms lock-step backward of 8 x C2;
  // tests/test_gpu_lockstep_fit.py runs both entry points on NaN-poisoned workspaces.
  if (!(n > 256 && n % 128 == 0))
    GPN_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)(b.at - b.u) * sizeof(double), s));
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
    hipLaunchKernelGGL(neg_transpose_small_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, at, b.ld, n, dy, grad_resid, (int64_t)0);
    GPN_LAUNCH_CHECK();
  }
  return GPN_OK;
}

// The KERNEL-INDEPENDENT half of the lock-step backward: for each of `batch` factors (model z at A + z sA, winv + z sW) U = L^-T,
// Kyy^-1 = U U^T (lower) at work + z stride + b.s and a^T = alpha^T U^T at work + z stride + b.at -- every launch once over all
// models; per model the launches of gpn_lml_backward, bit for bit.  (batch == 1 or n <= 256: model by model, the single-model form.)
static int kinv_batched_impl(hipStream_t s, int batch, int64_t n, const double* A, int64_t lda, int64_t sA, const double* winv, int64_t sW,
                             int dy, double* work, int64_t stride, const BackwardLayout& b) {
  const int64_t kp = round_up(n, 16);
  if (batch == 1 || n <= 256) {
    for (int z = 0; z < batch; ++z) {
      double* U = work + z * stride + b.u;
      double* S = work + z * stride + b.s;
      double* at = work + z * stride + b.at;
      const double* Az = A + z * sA;
      const double* wz = winv + z * sW;
      if (!(n > 256 && n % 128 == 0)) GPN_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)(b.at - b.u) * sizeof(double), s));
      int rc = (n > 256) ? gpn_trtri_upper_ws(s, Az, n, lda, wz, U, b.ld, S, b.ld) : gpn_trtri_upper(s, Az, n, lda, wz, U, b.ld);
      if (rc != GPN_OK) return rc;
      rc = gpn_gemm_nt(s, n, n, kp, 1.0, U, b.ld, U, b.ld, 0.0, S, b.ld, 1, GPN_TRI_A_UPPER | GPN_TRI_B_UPPER);
      if (rc != GPN_OK) return rc;
      rc = gpn_gemm_nt(s, dy, n, kp, 1.0, Az + n * lda, lda, U, b.ld, 0.0, at, b.ld, 0, GPN_TRI_B_UPPER);
      if (rc != GPN_OK) return rc;
    }
    return GPN_OK;
  }
  double* U = work + b.u;
  double* S = work + b.s;
  double* at = work + b.at;
  // U and S of every model start as zero where anything unwritten is read (see gpn_lml_backward: only with a ragged block)
  if (n % 128 != 0)
    for (int z = 0; z < batch; ++z)
      GPN_HIP_CHECK(hipMemsetAsync(U + z * stride, 0, (size_t)(b.at - b.u) * sizeof(double), s));
  int rc = trtri_upper_ws_batched(s, A, n, lda, sA, winv, sW, U, b.ld, stride, S, b.ld, stride, batch);
  if (rc != GPN_OK) return rc;
  // the scratch is free again: Kyy^-1 = U U^T (lower)
  rc = gemm_nt_strided(s, n, n, kp, 1.0, U, b.ld, U, b.ld, 0.0, S, b.ld, 1, GPN_TRI_A_UPPER | GPN_TRI_B_UPPER, 0, batch, stride, stride, stride);
  if (rc != GPN_OK) return rc;
  // a^T = alpha^T U^T (alpha^T = the extra rows of the factor buffers)
  return gemm_nt_strided(s, dy, n, kp, 1.0, A + n * lda, lda, U, b.ld, 0.0, at, b.ld, 0, GPN_TRI_B_UPPER, 0, batch, sA, stride, stride);
}

// gpn_lml_kinv_batched: that half as an entry point of its own -- for callers whose dKyy/dtheta is not one of the native
// stationary kinds (gptorch_amd/_expr.py: composite kernels sweep their own expression against Kyy^-1 and a, model by model).
// Layout (gpn_lml_kinv_layout): out4 = {ld, offset of Kyy^-1 (lower, [n, ld]), offset of a^T ([dy, ld]), doubles from one model to the next}.
extern "C" int gpn_lml_kinv_layout(int64_t n, int dy, int64_t* out4) {
  if (n < 0) return -1;
  if (dy <= 0) return -2;
  if (!out4) return -3;
  const BackwardLayout b = backward_layout(n, dy, 1);
  out4[0] = b.ld; out4[1] = b.s; out4[2] = b.at; out4[3] = round_up(b.sweep, 2);
  return GPN_OK;
}
extern "C" int64_t gpn_lml_kinv_batched_work_bytes(int64_t n, int dy, int batch) {
  if (n < 0 || dy <= 0 || batch < 1) return 0;
  return (int64_t)batch * round_up(backward_layout(n, dy, 1).sweep, 2) * (int64_t)sizeof(double);
}
extern "C" int gpn_lml_kinv_batched(void* stream, int batch, int64_t n, const double* A, int64_t lda, int64_t sA, const double* winv, int64_t sW,
                                    int dy, double* work) {
  if (batch < 1) return -2;
  if (n < 0) return -3;
  if (!A) return -4;
  if (dy <= 0) return -9;
  if (lda != gpn_factor_ld(n, dy)) return -5;
  if (batch > 1 && (sA < gpn_factor_rows(n, dy) * lda || (sA & 1))) return -6;
  if (!winv) return -7;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -8;
  if (!work) return -10;
  if (n == 0) return GPN_OK;
  const BackwardLayout b = backward_layout(n, dy, 1);
  const int64_t stride = round_up(b.sweep, 2);
  const int64_t max_models = std::max<int64_t>(1, 65535 / std::max<int64_t>(1, n / 256 + 1));
  for (int z0 = 0; z0 < batch; z0 += (int)max_models) {
    const int nb = (int)std::min<int64_t>(max_models, batch - z0);
    const int rc = kinv_batched_impl(static_cast<hipStream_t>(stream), nb, n, A + z0 * sA, lda, sA, winv + z0 * sW, sW, dy, work + z0 * stride, stride, b);
    if (rc != GPN_OK) return rc;
  }
  return GPN_OK;
}

// The backward of `batch` lock-step models (gpn_lml_forward_batched's factors, FactorBatch layout: model b at A + b sA,
// winv + b sW) in lock step: the reference runs loss(); backward(); step() one model at a time (gptorch/models/base.py:260-269).
// Every launch of gpn_lml_backward's schedule -- leaf transposes, the level-parallel triangular inversion, Kyy^-1 = U U^T,
// a^T = alpha^T U^T, the gradient sweep and its reduction -- goes out ONCE over all models (strided batches; equal nodes of one
// inversion level x models as a two-level batch).  Per model the same kernels in the same per-entry summation order:
// grads + b (2 + nls) and grad_resid + b n dy are BIT-IDENTICAL to gpn_lml_backward on model b alone.
// X + b sX (sX = 0: shared points), variance[b], length_scales + b nls.  work: gpn_lml_backward_batched_work_bytes.
extern "C" int64_t gpn_lml_backward_batched_work_bytes(int64_t n, int dy, int nls, int batch) {
  if (n < 0 || dy <= 0 || nls <= 0 || batch < 1) return 0;
  return (int64_t)batch * round_up(backward_layout(n, dy, nls).total, 2) * (int64_t)sizeof(double);
}

extern "C" int gpn_lml_backward_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                                        const double* variance, const double* length_scales, int nls,
                                        const double* A, int64_t lda, int64_t sA, const double* winv, int64_t sW, int dy,
                                        double* work, double* grads, double* grad_resid) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (batch < 1) return -3;
  if (!X) return -4;
  if (n < 0) return -6;
  if (d <= 0) return -7;
  if (!variance) return -8;
  if (!length_scales) return -9;
  if (nls != 1 && nls != d) return -10;
  if (!A) return -11;
  if (lda != gpn_factor_ld(n, dy)) return -12;
  if (batch > 1 && (sA < gpn_factor_rows(n, dy) * lda || (sA & 1))) return -13;
  if (!winv) return -14;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -15;
  if (dy <= 0) return -16;
  if (!work) return -17;
  if (!grads) return -18;
  if (n == 0) return GPN_OK;
  const BackwardLayout b = backward_layout(n, dy, nls);
  const int64_t sWk = round_up(b.total, 2);         // one model's workspace (doubles)
  if (batch == 1 || n <= 256) {
    // (the recursive inversion below 257 rows has no lock-step form: a few microseconds per model)
    for (int z = 0; z < batch; ++z) {
      const int rc = gpn_lml_backward(stream, kind, X + z * sX, n, d, variance + z, length_scales + (int64_t)z * nls, nls, A + z * sA, lda,
                                      winv + z * sW, dy, work + z * sWk, grads + (int64_t)z * (2 + nls),
                                      grad_resid ? grad_resid + (int64_t)z * n * dy : nullptr);
      if (rc != GPN_OK) return rc;
    }
    return GPN_OK;
  }
  // (a grid dimension holds 65535 blocks: the inversion's transposes put equal nodes x models into gridDim.z, the sweep the
  //  models into gridDim.y -- larger batches go out in chunks of models)
  const int64_t max_models = std::max<int64_t>(1, 65535 / std::max<int64_t>(1, n / 256 + 1));
  if (batch > max_models) {
    for (int z0 = 0; z0 < batch; z0 += (int)max_models) {
      const int nb = (int)std::min<int64_t>(max_models, batch - z0);
      const int rc = gpn_lml_backward_batched(stream, kind, nb, X + z0 * sX, sX, n, d, variance + z0, length_scales + (int64_t)z0 * nls, nls,
                                              A + z0 * sA, lda, sA, winv + z0 * sW, sW, dy, work + z0 * sWk, grads + (int64_t)z0 * (2 + nls),
                                              grad_resid ? grad_resid + (int64_t)z0 * n * dy : nullptr);
      if (rc != GPN_OK) return rc;
    }
    return GPN_OK;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* Kinv = work + b.s;
  double* at = work + b.at;
  int rc = kinv_batched_impl(s, batch, n, A, lda, sA, winv, sW, dy, work, sWk, b);
  if (rc != GPN_OK) return rc;
  rc = lml_grad_batched(s, kind, batch, X, sX, n, d, variance, length_scales, nls, Kinv, b.ld, sWk, at, b.ld, sWk, dy,
                        work + b.sweep, sWk, grads);
  if (rc != GPN_OK) return rc;
  if (grad_resid) {
    hipLaunchKernelGGL(neg_transpose_small_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, s, at, b.ld, n, dy,
                       grad_resid, sWk);
    GPN_LAUNCH_CHECK();
  }
  return GPN_OK;
}

// The backward of a gpn_lml_forward_ragged call: gpn_lml_backward_batched on the padded factors, the gradient sweep masked to every
// model's own points.  The inversion tree of n restricted to a model's points IS the tree of n_of[b] (a node splits at the largest
// power-of-two multiple of 128 below its size: the same point whether the node ends at n or earlier), the identity block inverts to
// itself and adds zeros to every sum over k: grads + b (2 + nls) is BIT-IDENTICAL to gpn_lml_backward on model b's own points.
// work: gpn_lml_backward_batched_work_bytes(n, dy, nls, batch).
extern "C" int gpn_lml_backward_ragged(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, const int32_t* n_of, int d,
                                       const double* variance, const double* length_scales, int nls,
                                       const double* A, int64_t lda, int64_t sA, const double* winv, int64_t sW, int dy,
                                       double* work, double* grads) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (batch < 1) return -3;
  if (!X) return -4;
  if (n <= 2 * 128) return -6;
  if (!n_of) return -7;
  if (d <= 0) return -8;
  if (!variance) return -9;
  if (!length_scales) return -10;
  if (nls != 1 && nls != d) return -11;
  if (!A) return -12;
  if (lda != gpn_factor_ld(n, dy)) return -13;
  if (batch > 1 && (sA < gpn_factor_rows(n, dy) * lda || (sA & 1))) return -14;
  if (!winv) return -15;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -16;
  if (dy <= 0) return -17;
  if (!work) return -18;
  if (!grads) return -19;
  const BackwardLayout b = backward_layout(n, dy, nls);
  const int64_t sWk = round_up(b.total, 2);
  const int64_t max_models = std::max<int64_t>(2, 65535 / std::max<int64_t>(1, n / 256 + 1));
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int z0 = 0; z0 < batch; z0 += (int)max_models) {
    int nb = (int)std::min<int64_t>(max_models, batch - z0);
    double* wk = work + z0 * sWk;
    // (kinv_batched_impl runs a single model through the single-model entry points: the same launches)
    int rc = kinv_batched_impl(s, nb, n, A + z0 * sA, lda, sA, winv + z0 * sW, sW, dy, wk, sWk, b);
    if (rc != GPN_OK) return rc;
    rc = lml_grad_batched(s, kind, nb, X + z0 * sX, sX, n, d, variance + z0, length_scales + (int64_t)z0 * nls, nls, wk + b.s, b.ld, sWk,
                          wk + b.at, b.ld, sWk, dy, wk + b.sweep, sWk, grads + (int64_t)z0 * (2 + nls), n_of + z0);
    if (rc != GPN_OK) return rc;
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
  {
    const int64_t rp = round_up(ns, 128);
    hipLaunchKernelGGL(zero_padding_kernel, dim3((unsigned)rp), dim3(256), 0, s, Bt, lda, ns, n);
    if (wb) hipLaunchKernelGGL(zero_padding_kernel, dim3((unsigned)rp), dim3(256), 0, s, work + one / (int64_t)sizeof(double), lda, ns, n);
    GPN_LAUNCH_CHECK();
  }
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
  // mean = m(x*) + A^T V (gpr.py:107-108) and, for the diagonal case, var = Kdiag - rowsumsq(A^T) (gpr.py:109-113): one pass
  for (int c0 = 0; c0 < dy; c0 += PT_DY) {
    hipLaunchKernelGGL(predict_tail_kernel, dim3((unsigned)ns), dim3(256), 0, s, Bt, lda, n, A + n * lda, dy, c0, Ms, variance, mean,
                       full_cov ? nullptr : var);
    GPN_LAUNCH_CHECK();
  }
  if (!full_cov) return GPN_OK;
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
