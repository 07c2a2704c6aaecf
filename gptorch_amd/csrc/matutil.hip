// Small reductions and copies around the factor buffers: the LML terms (functions.lt_log_determinant, functions.py:61-68, and
// |alpha|^2 of gpr.py:63-67), the scalar sums of the sparse bound (sparse_gpr.py:139-151), copies and row sums of squares.
#include "gpn_common.h"

namespace gpn {

// ---- reductions / utilities -------------------------------------------------
__global__ __launch_bounds__(1024) void lml_reduce_kernel(const double* A, int64_t n, int64_t e, int64_t lda,
                                                          double* out3, int64_t sA, const int32_t* n_of = nullptr) {
  A += (int64_t)blockIdx.x * sA;                 // `gridDim.x` problems at stride sA, results 3 apart
  out3 += 3 * blockIdx.x;
  // single workgroup: sums are O(N) work.  The diagonal is one cache line per element, so the
  // loads go out in batches of 8 per thread before the first log() needs one (issued one by
  // one behind a log() each they cost a full memory round trip per element: 30 us at N = 8192)
  constexpr int NT = 1024;
  __shared__ double red[2][NT];
  const int tid = threadIdx.x;
  double ld = 0.0, sq = 0.0;
  for (int64_t base = tid; base < n; base += (int64_t)NT * 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = base + (int64_t)k * NT;
      v[k] = i < n ? A[i * lda + i] : 1.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) ld += log(v[k]);
  }
  for (int64_t c = 0; c < e; ++c) {
    const double* row = A + (n + c) * lda;
    for (int64_t base = tid; base < n; base += (int64_t)NT * 8) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t i = base + (int64_t)k * NT;
        v[k] = i < n ? row[i] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) sq = fma(v[k], v[k], sq);
    }
  }
  red[0][tid] = ld;
  red[1][tid] = sq;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
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
    // (ragged batch: the identity rows of a padded model add log 1 and 0^2 to the sums above; its constant counts its own points)
    const double npts = n_of ? (double)n_of[blockIdx.x] : (double)n;
    out3[2] = -0.5 * quad - (double)e * logdet - 0.5 * (double)e * npts * 1.8378770664093454836;
  }
}

// out[z] = sum_{i < rows, j < cols} x_z[i ldx + j] * y_z[i ldy + j]  (y == NULL: the plain sum of x), problem z at x + z sx,
// y + z sy.  One workgroup per problem; thread t adds the entries t, t + 256, ... of the row-major index order, then a fixed
// tree: the value does not depend on how many problems share the launch.
__global__ __launch_bounds__(256) void dot2d_kernel(const double* x, int64_t ldx, int64_t sx, const double* y, int64_t ldy, int64_t sy,
                                                    int64_t rows, int64_t cols, double* out) {
  __shared__ double red[256];
  x += (int64_t)blockIdx.x * sx;
  if (y) y += (int64_t)blockIdx.x * sy;
  const int tid = threadIdx.x;
  double s = 0.0;
  const int64_t total = rows * cols;
  for (int64_t k = tid; k < total; k += 256) {
    const int64_t i = k / cols, j = k - i * cols;
    const double a = x[i * ldx + j];
    s += y ? a * y[i * ldy + j] : a;
  }
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) out[blockIdx.x] = red[0];
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

extern "C" int gpn_lml_reduce(void* stream, const double* A, int64_t n, int64_t e, int64_t lda, double* out3) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < n) return -5;
  if (!out3) return -6;
  hipLaunchKernelGGL(lml_reduce_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), A, n, e, lda, out3, (int64_t)0);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_lml_reduce_batched(void* stream, const double* A, int64_t n, int64_t e, int64_t lda, int64_t sA, double* out3,
                                      int batch) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < n) return -5;
  if (!out3) return -7;
  if (batch < 1) return -8;
  hipLaunchKernelGGL(lml_reduce_kernel, dim3((unsigned)batch), dim3(1024), 0, static_cast<hipStream_t>(stream), A, n, e, lda, out3, sA);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_lml_reduce_batched for a ragged batch (gpn_lml_forward_ragged): the padded factors' sums, each model's constant with its own n_of[b]
int gpn::lml_reduce_ragged(hipStream_t s, const double* A, int64_t n, int64_t e, int64_t lda, int64_t sA, double* out3, int batch,
                           const int32_t* n_of) {
  hipLaunchKernelGGL(lml_reduce_kernel, dim3((unsigned)batch), dim3(1024), 0, s, A, n, e, lda, out3, sA, n_of);
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

// The small scalar sums of the sparse bound (sparse_gpr.py:139-151: tr(AA^T), |err|^2, ...) for `batch` models in one launch,
// each value independent of the batch size (see dot2d_kernel): out[z] = <x_z, y_z> over a rows x cols view (y NULL: sum of x).
extern "C" int gpn_dot2d_batched(void* stream, const double* x, int64_t ldx, int64_t sx, const double* y, int64_t ldy, int64_t sy,
                                 int64_t rows, int64_t cols, double* out, int batch) {
  if (!x) return -2;
  if (rows < 0) return -8;
  if (cols < 0) return -9;
  if (ldx < 0 || ldy < 0) return -3;
  if (!out) return -10;
  if (batch < 1) return -11;
  hipLaunchKernelGGL(dot2d_kernel, dim3((unsigned)batch), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, sx, y, ldy, sy, rows, cols,
                     out);
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
