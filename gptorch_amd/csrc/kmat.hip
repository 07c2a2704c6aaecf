// K assembly: pairwise scaled distances + stationary-kernel epilogue, fused.
// Replaces util.squared_distance (util.py:73-88), Stationary.squared_dist/dist
// (kernels.py:149-172), Rbf/Matern52/Matern32/Exp .K (kernels.py:182-222) and the
// "+ sigma_n^2 I" of GPR._compute_kyy (gpr.py:69-86): the reference makes >= 12
// N x N passes for this; here the N x N matrix is written exactly once.
//
// HBM-bound for small D (8 B written per entry, X tiles come from L2), fp64
// vector-ALU bound for large D.  64x64 output tile per 256-thread workgroup:
// the two point blocks (64 x DC coordinates each, pre-divided by ell) are staged
// in LDS in [coordinate][point] order so that a wave reads 4 row points as one
// broadcast b128 pair and its 2+2 column points as conflict-free b128s; every
// thread owns a 4x4 micro-tile laid out so each store instruction writes 256 B
// contiguous per row (16 lanes x double2).
#include "gpn_common.h"
#include "kernel_fn.h"

namespace gpn {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int KT = 64;   // output tile edge
constexpr int DC = 16;   // coordinates staged per pass

struct KmatArgs {
  const double* X;
  const double* X2;   // == X for the symmetric case
  const double* variance;
  const double* ls;
  const double* noise;  // nullptr => no diagonal add
  double* K;
  int64_t ldk;
  int n, m, d, nls;
  int symmetric, lower, vec_ok;
  int64_t sX, sK;     // strided batch (gridDim.z problems): points and output at these strides, hyper-parameters consecutive
  int64_t sX2 = -1;   // ... of the second point set (-1: the first set's stride -- the symmetric case)
  // ragged lock-step batch (gpn_lml_forward_ragged; symmetric lower only): model z has n_of[z] <= n real points; rows and columns
  // from there to n are written as IDENTITY rows, so that the padded matrix factors to [L_z 0; 0 I]
  const int32_t* n_of = nullptr;
  // gpn_lml_forward_saving: every entry is also stored here (same leading dimension) -- a pristine copy of Kyy for the refinement
  // step's residual pass, which otherwise re-computes every entry (the factorisation overwrites K in place)
  double* K2 = nullptr;
};

template <int KIND>
__global__ __launch_bounds__(256) void kmat_kernel(KmatArgs p) {
  __shared__ __attribute__((aligned(16))) double xs[DC][KT];   // row points
  __shared__ __attribute__((aligned(16))) double ys[DC][KT];   // column points

  __shared__ double inv_ell[DC];
  int nreal = p.n;                           // real points of this model (ragged batches: < p.n)
  if (gridDim.z > 1) {                       // problem z of a strided batch (gpn_lml_forward_batched)
    const int z = blockIdx.z;
    p.X2 += z * (p.sX2 < 0 ? p.sX : p.sX2); p.X += z * p.sX; p.K += z * p.sK;
    if (p.K2) p.K2 += z * p.sK;
    p.variance += z; p.ls += z * p.nls;
    if (p.noise) p.noise += z;
  }
  if (p.n_of) nreal = p.n_of[blockIdx.z];
  int tj, ti;
  if (p.lower) {
    // only the tiles on/below the diagonal are launched (row-major over the triangle): no
    // empty workgroups, equal work for every XCD
    const int q = blockIdx.x;
    ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > q) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
    tj = q - ti * (ti + 1) / 2;
  } else {
    tj = blockIdx.x;
    ti = blockIdx.y;
  }
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = ti * KT, j0 = tj * KT;

  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

  for (int d0 = 0; d0 < p.d; d0 += DC) {
    // stage: thread -> (point = tid/4, 4 coordinates), scaled by 1/ell (kernels.py:154-158).
    // One reciprocal per coordinate and workgroup; the points are multiplied by it -- an fp64
    // division per staged coordinate would be a quarter of the kernel's instructions at D = 8.
    if (tid < DC) {
      const int dd = d0 + tid;
      inv_ell[tid] = dd < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : dd] : 0.0;
    }
    __syncthreads();
    {
      const int pt = tid >> 2, c4 = (tid & 3) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = d0 + c4 + c;
        double vx = 0.0, vy = 0.0;
        if (dd < p.d) {
          const double ie = inv_ell[c4 + c];
          if (i0 + pt < p.n) vx = p.X[(int64_t)(i0 + pt) * p.d + dd] * ie;
          if (j0 + pt < p.m) vy = p.X2[(int64_t)(j0 + pt) * p.d + dd] * ie;
        }
        xs[c4 + c][pt] = vx;
        ys[c4 + c][pt] = vy;
      }
    }
    __syncthreads();
    const int dmax = min(DC, p.d - d0);
    for (int dd = 0; dd < dmax; ++dd) {
      const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][ty * 4]);
      const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][ty * 4 + 2]);
      const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][tx * 2]);
      const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + tx * 2]);
      const double xr[4] = {xa.x, xa.y, xb.x, xb.y};
      const double yc[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = xr[a] - yc[b];
          acc[a][b] = fma(df, df, acc[a][b]);
        }
    }
    __syncthreads();
  }

  const double var = p.variance[0];
  const double noise = p.noise ? p.noise[0] : 0.0;
  const bool add_diag = p.noise != nullptr && p.symmetric;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = i0 + ty * 4 + a;
    if (row >= p.n) continue;
    double* krow = p.K + (int64_t)row * p.ldk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = j0 + h * 32 + tx * 2;
      double v0 = kernel_of_r2<KIND>(acc[a][2 * h], var);
      double v1 = kernel_of_r2<KIND>(acc[a][2 * h + 1], var);
      if (add_diag) {
        if (row == col) v0 += noise;
        if (row == col + 1) v1 += noise;
      }
      if (p.n_of) {                            // ragged batch (symmetric, lower tiles)
        if (row >= nreal) {                    // an identity row (col <= row in the tiles that matter)
          v0 = row == col ? 1.0 : 0.0;
          v1 = row == col + 1 ? 1.0 : 0.0;
        } else {
          if (col >= nreal) v0 = 0.0;          // (upper part of a diagonal tile: never read)
          if (col + 1 >= nreal) v1 = 0.0;
        }
      }
      if (p.vec_ok && col + 1 < p.m) {
        *reinterpret_cast<d2*>(krow + col) = d2{v0, v1};
        if (p.K2) *reinterpret_cast<d2*>(p.K2 + (int64_t)row * p.ldk + col) = d2{v0, v1};
      } else {
        if (col < p.m) krow[col] = v0;
        if (col + 1 < p.m) krow[col + 1] = v1;
        if (p.K2) {
          if (col < p.m) p.K2[(int64_t)row * p.ldk + col] = v0;
          if (col + 1 < p.m) p.K2[(int64_t)row * p.ldk + col + 1] = v1;
        }
      }
    }
  }
}

// corner != 0: also clear columns n..lde-1 of the extra rows (the corner of the factor buffer that
// accumulates -alpha alpha^T during the factorisation) and the info word -- one launch instead of
// this one plus two fills in gpn_lml_forward
__global__ void pack_rhs_kernel(const double* Y, const double* M, int64_t n, int dy, double* E, int64_t lde,
                                int corner, int32_t* info, int64_t sY = 0, int64_t sM = 0, int64_t sE = 0, const int32_t* n_of = nullptr) {
  if (gridDim.y > 1) {                       // problem y of a strided batch
    Y += blockIdx.y * sY;
    if (M) M += blockIdx.y * sM;
    E += blockIdx.y * sE;
    if (info) info += blockIdx.y;
  }
  if (n_of) n = n_of[blockIdx.y];            // ragged batch: the right-hand sides of the identity rows are zero (corner != 0 clears them)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (corner) {
    if (i == 0 && info) *info = 0;
    if (i >= n && i < lde)
      for (int c = 0; c < dy; ++c) E[(int64_t)c * lde + i] = 0.0;
  }
  if (i >= n) return;
  for (int c = 0; c < dy; ++c) {
    double v = Y[i * dy + c];
    if (M) v -= M[i * dy + c];
    E[(int64_t)c * lde + i] = v;
  }
}

int pack_rhs_full(hipStream_t s, const double* Y, const double* M, int64_t n, int dy, double* E, int64_t lde,
                  int32_t* info) {
  hipLaunchKernelGGL(pack_rhs_kernel, dim3((unsigned)((lde + 255) / 256)), dim3(256), 0, s, Y, M, n, dy, E, lde, 1, info);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// K(X) + noise I, lower tiles, into A AND into Ksave (same leading dimension): gpn_lml_forward_saving
int assemble_lower_saving(hipStream_t s, int kind, const double* X, int64_t n, int d, const double* variance, const double* length_scales,
                          int nls, const double* noise, double* A, double* Ksave, int64_t lda) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  KmatArgs a;
  a.X = X; a.X2 = X;
  a.variance = variance; a.ls = length_scales; a.noise = noise;
  a.K = A; a.K2 = Ksave; a.ldk = lda;
  a.n = (int)n; a.m = (int)n; a.d = d; a.nls = nls;
  a.symmetric = 1; a.lower = 1;
  a.vec_ok = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(Ksave) & 15) == 0);
  a.sX = 0; a.sK = 0;
  const unsigned tm = (unsigned)((n + KT - 1) / KT);
  const dim3 grid(tm * (tm + 1) / 2);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, 8.0 * (0.5 * n * (n + 1.0) + (double)n * d), PROF_KMAT);     // (algorithmic bytes: the copy earns nothing)
  switch (kind) {
    case GPN_RBF: hipLaunchKernelGGL(kmat_kernel<GPN_RBF>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN52: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN52>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN32: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN32>, grid, dim3(256), 0, s, a); break;
    case GPN_EXP: hipLaunchKernelGGL(kmat_kernel<GPN_EXP>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(kmat_kernel<GPN_PERIODIC>, grid, dim3(256), 0, s, a); break;
  }
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// symmetric K(X_b) + noise_b I, lower tiles, for `batch` models in one launch (+ their right-hand sides and info words)
int assemble_batched(hipStream_t s, int kind, int batch, const double* X, int64_t sX, int64_t n, int d, const double* Y, int64_t sY,
                     const double* M, int64_t sM, int dy, const double* variance, const double* length_scales, int nls,
                     const double* noise, double* A, int64_t lda, int64_t sA, int32_t* info, const int32_t* n_of) {
  KmatArgs a;
  a.n_of = n_of;
  a.X = X; a.X2 = X;
  a.variance = variance; a.ls = length_scales; a.noise = noise;
  a.K = A; a.ldk = lda;
  a.n = (int)n; a.m = (int)n; a.d = d; a.nls = nls;
  a.symmetric = 1; a.lower = 1;
  a.vec_ok = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((sA & 1) == 0);
  a.sX = sX; a.sK = sA;
  const unsigned tm = (unsigned)((n + KT - 1) / KT);
  const dim3 grid(tm * (tm + 1) / 2, 1, (unsigned)batch);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, batch * 8.0 * (0.5 * n * (n + 1.0) + (double)n * d), PROF_KMAT);
  switch (kind) {
    case GPN_RBF: hipLaunchKernelGGL(kmat_kernel<GPN_RBF>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN52: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN52>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN32: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN32>, grid, dim3(256), 0, s, a); break;
    case GPN_EXP: hipLaunchKernelGGL(kmat_kernel<GPN_EXP>, grid, dim3(256), 0, s, a); break;
    case GPN_PERIODIC: hipLaunchKernelGGL(kmat_kernel<GPN_PERIODIC>, grid, dim3(256), 0, s, a); break;
    default: return -2;
  }
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  hipLaunchKernelGGL(pack_rhs_kernel, dim3((unsigned)((lda + 255) / 256), (unsigned)batch), dim3(256), 0, s, Y, M, n, dy,
                     A + n * lda, lda, 1, info, sY, sM, sA, n_of);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn

extern "C" int gpn_kernel_matrix(void* stream, int kind, const double* X, int64_t n, const double* X2,
                                 int64_t m, int d, const double* variance, const double* length_scales,
                                 int nls, const double* noise, int uplo, double* K, int64_t ldk) {
  using namespace gpn;
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (!X) return -3;
  if (n < 0) return -4;
  const bool symmetric = (X2 == nullptr);
  if (symmetric) m = n;
  if (m < 0) return -6;
  if (d <= 0) return -7;
  if (!variance) return -8;
  if (!length_scales) return -9;
  if (nls != 1 && nls != d) return -10;
  if (uplo != GPN_FULL && uplo != GPN_LOWER) return -12;
  if (uplo == GPN_LOWER && !symmetric) return -12;
  if (!K) return -13;
  if (ldk < m) return -14;
  if (n == 0 || m == 0) return GPN_OK;
  KmatArgs a;
  a.X = X; a.X2 = symmetric ? X : X2;
  a.variance = variance; a.ls = length_scales; a.noise = noise;
  a.K = K; a.ldk = ldk;
  a.n = (int)n; a.m = (int)m; a.d = d; a.nls = nls;
  a.symmetric = symmetric; a.lower = (uplo == GPN_LOWER);
  a.vec_ok = ((ldk & 1) == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
  a.sX = 0; a.sK = 0;
  const unsigned tn = (unsigned)((m + KT - 1) / KT), tm = (unsigned)((n + KT - 1) / KT);
  dim3 grid = a.lower ? dim3(tm * (tm + 1) / 2) : dim3(tn, tm);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rec = -1;
  if (profile_on())   // algorithmic bytes: the entries written once + the points read once
    rec = profile_begin(s, a.lower ? 8.0 * (0.5 * n * (n + 1.0) + (double)n * d) : 8.0 * ((double)n * m + (double)(n + (symmetric ? 0 : m)) * d),
                        PROF_KMAT);
  switch (kind) {
    case GPN_RBF: hipLaunchKernelGGL(kmat_kernel<GPN_RBF>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN52: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN52>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN32: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN32>, grid, dim3(256), 0, s, a); break;
    case GPN_EXP: hipLaunchKernelGGL(kmat_kernel<GPN_EXP>, grid, dim3(256), 0, s, a); break;
    case GPN_PERIODIC: hipLaunchKernelGGL(kmat_kernel<GPN_PERIODIC>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(kmat_kernel<GPN_SQDIST>, grid, dim3(256), 0, s, a); break;
  }
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_kernel_matrix for `batch` models in ONE launch (gridDim.z): model b reads X + b sX (sX = 0: shared points), X2 + b sX2,
// variance[b], length_scales + b nls, noise[b] and writes K + b sK.  Per model the same kernel, entry by entry: bit-identical to
// gpn_kernel_matrix.  (Lock-step sparse models: K(Z_b) and K(x, Z_b), sparse_gpr.py:126-129, for every restart at once.)
extern "C" int gpn_kernel_matrix_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n,
                                         const double* X2, int64_t sX2, int64_t m, int d,
                                         const double* variance, const double* length_scales, int nls, const double* noise,
                                         int uplo, double* K, int64_t ldk, int64_t sK) {
  using namespace gpn;
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (batch < 1 || batch > 65535) return -3;
  if (!X) return -4;
  if (n < 0) return -6;
  const bool symmetric = (X2 == nullptr);
  if (symmetric) m = n;
  if (m < 0) return -9;
  if (d <= 0) return -10;
  if (!variance) return -11;
  if (!length_scales) return -12;
  if (nls != 1 && nls != d) return -13;
  if (uplo != GPN_FULL && uplo != GPN_LOWER) return -15;
  if (uplo == GPN_LOWER && !symmetric) return -15;
  if (!K) return -16;
  if (ldk < m) return -17;
  if (batch > 1 && sK < (n - 1) * ldk + m) return -18;
  if (n == 0 || m == 0) return GPN_OK;
  KmatArgs a;
  a.X = X; a.X2 = symmetric ? X : X2;
  a.variance = variance; a.ls = length_scales; a.noise = noise;
  a.K = K; a.ldk = ldk;
  a.n = (int)n; a.m = (int)m; a.d = d; a.nls = nls;
  a.symmetric = symmetric; a.lower = (uplo == GPN_LOWER);
  a.vec_ok = ((ldk & 1) == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0) && ((sK & 1) == 0);
  a.sX = sX; a.sK = sK; a.sX2 = symmetric ? sX : sX2;
  const unsigned tn = (unsigned)((m + KT - 1) / KT), tm = (unsigned)((n + KT - 1) / KT);
  if (!a.lower && tm > 65535) return GPN_E_UNSUPPORTED;
  dim3 grid = a.lower ? dim3(tm * (tm + 1) / 2, 1, (unsigned)batch) : dim3(tn, tm, (unsigned)batch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rec = -1;
  if (profile_on())
    rec = profile_begin(s, batch * (a.lower ? 8.0 * (0.5 * n * (n + 1.0) + (double)n * d) : 8.0 * ((double)n * m + (double)(n + (symmetric ? 0 : m)) * d)),
                        PROF_KMAT);
  switch (kind) {
    case GPN_RBF: hipLaunchKernelGGL(kmat_kernel<GPN_RBF>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN52: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN52>, grid, dim3(256), 0, s, a); break;
    case GPN_MATERN32: hipLaunchKernelGGL(kmat_kernel<GPN_MATERN32>, grid, dim3(256), 0, s, a); break;
    case GPN_EXP: hipLaunchKernelGGL(kmat_kernel<GPN_EXP>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(kmat_kernel<GPN_PERIODIC>, grid, dim3(256), 0, s, a); break;
  }
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_pack_rhs(void* stream, const double* Y, const double* M, int64_t n, int dy,
                            double* E, int64_t lde) {
  if (!Y) return -2;
  if (n < 0) return -4;
  if (dy <= 0) return -5;
  if (!E) return -6;
  if (lde < n) return -7;
  if (n == 0) return GPN_OK;
  hipLaunchKernelGGL(gpn::pack_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), Y, M, n, dy, E, lde, 0, (int32_t*)nullptr);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}
