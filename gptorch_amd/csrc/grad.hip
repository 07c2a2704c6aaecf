// Hyper-parameter gradients of the kernel matrix contracted with a weight matrix G:
//     out[0]      = sum_ij G_ij * dK_ij/d(variance)
//     out[1+d]    = sum_ij G_ij * dK_ij/d(ell_d)        (isotropic: one entry, summed over d)
//     out[1+nls]  = trace(G)                            (LML mode only: d/d(noise))
// Two sources of G:
//   LML mode   -- G = 1/2 (a a^T - dy * Kyy^-1) formed on the fly from the lower
//                 triangle of Kyy^-1 and a^T [dy, n]  (closed-form backward of
//                 gpr.py:47-67; the reference gets it from PyTorch's CholeskyBackward /
//                 TriangularSolveBackward chain at ~2 N^3 flops + dozens of N x N passes);
//                 only tiles on/below the diagonal are visited, off-diagonal pairs count twice.
//   dense mode -- G is a given [n, m] matrix (autograd backward of Kernel.K).
// K and dK/d(theta) are RE-COMPUTED from the points (same staging as kmat.hip), so the
// sweep reads each G entry exactly once: HBM-bound at 8 B per pair for small D.
//
// dK/d(ell_d) = B(r) * s_d / ell_d with s_d = ((x_id - x_jd)/ell_d)^2 and
//   Rbf: B = K;  Matern52: B = var*(5/3)(1+sqrt5 r)exp(-sqrt5 r);  Matern32: B = 3 var exp(-sqrt3 r);
//   Exp: B = var*exp(-r)/r (0 where the reference's clamp at 1e-40 kills the gradient);
//   Periodic (var*cos r): B = var*sin(r)/r.
// Per-workgroup partial sums go to a workspace and are reduced by a second tiny kernel
// (deterministic; no float atomics).
#include "gpn_common.h"

namespace gpn {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int GT = 64;     // tile edge
constexpr int GDC = 16;    // coordinates per staged chunk
constexpr int GMAXD = 64;  // gpn_kernel_grad_x2: up to this input dimension one accumulator per coordinate in registers, the chunked kernel above it

struct GradArgs {
  const double* X;
  const double* X2;
  const double* variance;
  const double* ls;
  const double* G;      // dense mode: G [n, m]; LML mode: Kinv (lower) [n, n]
  int64_t ldg;
  const double* at;     // LML mode: a^T [dy, n]
  int64_t ldat;
  double* partial;      // [nblocks, nout]
  int n, m, d, nls, dy, nout;
  int tiles_m, tiles_n;
  // lock-step batch (gridDim.y models, gpn_lml_grad_batched): model z reads X + z sX, variance[z], ls + z nls, G + z sG,
  // at + z sAt and writes partial + z sPartial
  int64_t sX = 0, sG = 0, sAt = 0, sPartial = 0;
  int64_t sX2 = -1;     // stride of the second point set (-1: sX -- the symmetric case)
  // ragged lock-step batch (gpn_lml_backward_ragged, LML mode): model z has n_of[z] <= n real points; the launch still covers the
  // tiles of n, the tiles beyond a model's points write zero partial sums
  const int32_t* n_of = nullptr;
};

__device__ __forceinline__ void select_model(GradArgs& p) {
  if (gridDim.y > 1) {
    const int64_t z = blockIdx.y;
    p.X2 += z * (p.sX2 < 0 ? p.sX : p.sX2); p.X += z * p.sX; p.variance += z; p.ls += z * p.nls;
    p.G += z * p.sG; p.at += z * p.sAt; p.partial += z * p.sPartial;
  }
  if (p.n_of) p.n = p.m = p.n_of[blockIdx.y];
}

template <int KIND>
__device__ __forceinline__ void k_and_base(double r2, double var, double& K, double& B) {
  if constexpr (KIND == GPN_RBF) {
    K = var * exp(-0.5 * r2);
    B = K;
  } else if constexpr (KIND == GPN_SQDIST) {
    // util.squared_distance itself (util.py:73-88): K = r^2, no variance.  B = -2 dK/d(r^2) = -2
    // everywhere, also at r = 0 (direct differences: nothing is clamped, so the second derivative
    // the reference guards with its detach() trick, test_util.py:78-106, is the plain 2)
    K = 0.0;
    B = -2.0;
  } else {
    const bool dead = r2 < 1e-40;                 // kernels.py:172 clamp: no gradient below it
    const double r = sqrt(fmax(r2, 1e-40));
    if constexpr (KIND == GPN_MATERN52) {
      const double s5 = 2.23606797749978969641;
      const double e = exp(-s5 * r);
      K = var * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * e;
      B = dead ? 0.0 : var * (5.0 / 3.0) * (1.0 + s5 * r) * e;
    } else if constexpr (KIND == GPN_MATERN32) {
      const double s3 = 1.73205080756887729353;
      const double e = exp(-s3 * r);
      K = var * (1.0 + s3 * r) * e;
      B = dead ? 0.0 : 3.0 * var * e;
    } else if constexpr (KIND == GPN_PERIODIC) {
      K = var * cos(r);
      B = dead ? 0.0 : var * sin(r) / r;
    } else {
      const double e = exp(-r);
      K = var * e;
      B = dead ? 0.0 : var * e / r;
    }
  }
}

// NCH = number of 16-coordinate chunks (d <= 16*NCH); LML = G formed from Kinv + a;
// ISO = one shared length-scale: sum_d S_d = sum_ij G'_ij r2_ij, so the second pass over the coordinates
// (and its one accumulator per dimension: 32-128 VGPRs) is not needed at all
template <int KIND, int NCH, bool LML, bool ISO>
__global__ __launch_bounds__(256) void grad_sweep_kernel(GradArgs p) {
  select_model(p);
  __shared__ __attribute__((aligned(16))) double xs[NCH * GDC][GT];
  __shared__ __attribute__((aligned(16))) double ys[NCH * GDC][GT];
  __shared__ double red[256];

  int ti, tj;
  if (LML) {
    // compact lower-triangle enumeration (row-major over the triangle)
    const int q = blockIdx.x;
    ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > q) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
    tj = q - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x / p.tiles_n;
    tj = blockIdx.x - ti * p.tiles_n;
  }
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = ti * GT, j0 = tj * GT;

  // LML mode: the weight entries G_ij = 1/2 (a_i . a_j - dy Kinv_ij) need Kinv (one 16-byte load per column pair, requested
  // HERE so that the round trip runs under the distance pass) and a^T at the tile's rows and columns (staged in LDS: read
  // per entry from global they were 2 dy dependent loads per entry and left the sweep latency-bound -- 50 % VALU-issue, 16 %
  // of HBM peak, profiles/r2_valu_bound_c3.json)
  constexpr int ADY = 4;                      // right-hand sides staged at a time
  __shared__ double a_rows[LML ? ADY : 1][GT], a_cols[LML ? ADY : 1][GT];
  d2 gk[4][2];
  const bool gvec = LML && ((p.ldg & 1) == 0) && ((reinterpret_cast<uintptr_t>(p.G) & 15) == 0);
  if constexpr (LML) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = i0 + ty * 4 + a;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = j0 + h * 32 + tx * 2;
        gk[a][h] = d2{0.0, 0.0};
        if (row < p.n && col <= row) {          // (col + 1 may lie above the diagonal: loaded, never used)
          if (gvec) gk[a][h] = *reinterpret_cast<const d2*>(p.G + (int64_t)row * p.ldg + col);
          else { gk[a][h].x = p.G[(int64_t)row * p.ldg + col]; if (col + 1 <= row) gk[a][h].y = p.G[(int64_t)row * p.ldg + col + 1]; }
        }
      }
    }
    if (p.dy <= ADY && tid < 2 * GT) {
      const int pt = tid & (GT - 1);
      const int idx = (tid < GT ? i0 : j0) + pt;
      for (int c = 0; c < p.dy; ++c) (tid < GT ? a_rows : a_cols)[c][pt] = idx < p.n ? p.at[(int64_t)c * p.ldat + idx] : 0.0;
    }
  }

  // stage both point blocks, scaled by the reciprocal length-scales (kernels.py:154-158; one
  // reciprocal per coordinate and workgroup, as in kmat.hip)
  __shared__ double inv_ell[NCH * GDC];
  if (tid < NCH * GDC) inv_ell[tid] = tid < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : tid] : 0.0;
  __syncthreads();
  for (int idx = tid; idx < NCH * GDC * GT; idx += 256) {
    const int dd = idx / GT, pt = idx - dd * GT;   // pt fastest: conflict-free LDS writes
    double vx = 0.0, vy = 0.0;
    if (dd < p.d) {
      const double ie = inv_ell[dd];
      if (i0 + pt < p.n) vx = p.X[(int64_t)(i0 + pt) * p.d + dd] * ie;   // scaling as kmat.hip
      if (j0 + pt < p.m) vy = p.X2[(int64_t)(j0 + pt) * p.d + dd] * ie;
    }
    xs[dd][pt] = vx;
    ys[dd][pt] = vy;
  }
  __syncthreads();

  // pass 1: squared distances of the 4x4 micro-tile (rows ty*4+a, cols {2tx,2tx+1,32+2tx,33+2tx})
  double r2[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
  for (int dd = 0; dd < p.d; ++dd) {
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
        r2[a][b] = fma(df, df, r2[a][b]);
      }
  }

  // G' = weight * G * B, plus the variance / noise sums
  const double var = p.variance[0];
  double gb[ISO ? 1 : 4][ISO ? 1 : 4];
  double s_var = 0.0, s_tr = 0.0, s_iso = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = i0 + ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = j0 + (b >> 1) * 32 + tx * 2 + (b & 1);
      double g = 0.0;
      if (row < p.n && col < p.m) {
        if constexpr (LML) {
          if (col <= row) {
            double aa = 0.0;
            if (p.dy <= ADY) {
              const int rl = ty * 4 + a, cl = (b >> 1) * 32 + tx * 2 + (b & 1);
              for (int c = 0; c < p.dy; ++c) aa = fma(a_rows[c][rl], a_cols[c][cl], aa);
            } else {
              for (int c = 0; c < p.dy; ++c) aa = fma(p.at[(int64_t)c * p.ldat + row], p.at[(int64_t)c * p.ldat + col], aa);
            }
            const double kinv = (b & 1) ? gk[a][b >> 1].y : gk[a][b >> 1].x;
            g = 0.5 * (aa - (double)p.dy * kinv);
            if (col == row) s_tr += g;
            else g *= 2.0;   // symmetric partner (col, row)
          }
        } else {
          g = p.G[(int64_t)row * p.ldg + col];
        }
      }
      double K, B;
      k_and_base<KIND>(r2[a][b], var, K, B);
      s_var = fma(g, K, s_var);
      if constexpr (ISO) s_iso = fma(g * B, r2[a][b], s_iso);
      else gb[a][b] = g * B;
    }
  }

  // pass 2: per-dimension sums  S_d = sum G'_ij * s_d
  double acc[ISO ? 1 : NCH * GDC];
#pragma unroll
  for (int dd = 0; dd < (ISO ? 1 : NCH * GDC); ++dd) acc[dd] = 0.0;
#pragma unroll
  for (int dd = 0; dd < (ISO ? 0 : NCH * GDC); ++dd) {
    if (dd < p.d) {
      const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][ty * 4]);
      const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][ty * 4 + 2]);
      const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][tx * 2]);
      const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + tx * 2]);
      const double xr[4] = {xa.x, xa.y, xb.x, xb.y};
      const double yc[4] = {ya.x, ya.y, yb.x, yb.y};
      double sacc = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = xr[a] - yc[b];
          sacc = fma(gb[a][b], df * df, sacc);
        }
      acc[dd] = sacc;
    }
  }

  // workgroup reduction of (1 + nls' + 1) values; isotropic sums over d first
  auto block_sum = [&](double v) -> double {
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double* out = p.partial + (int64_t)blockIdx.x * p.nout;
  const double tv = block_sum(s_var);
  if (tid == 0) out[0] = tv / var;                         // dK/dvar = K / var
  if constexpr (ISO) {
    const double t = block_sum(s_iso);
    if (tid == 0) out[1] = t / p.ls[0];
  } else if (p.nls == 1) {
    double s = 0.0;
#pragma unroll
    for (int dd = 0; dd < NCH * GDC; ++dd) s += acc[dd];
    const double t = block_sum(s);
    if (tid == 0) out[1] = t / p.ls[0];
  } else {
#pragma unroll
    for (int dd = 0; dd < NCH * GDC; ++dd) {
      if (dd < p.d) {                                       // uniform
        const double t = block_sum(acc[dd]);
        if (tid == 0) out[1 + dd] = t / p.ls[dd];
      }
    }
  }
  if (LML) {
    const double t = block_sum(s_tr);
    if (tid == 0) out[1 + p.nls] = t;
  }
}

// Any input dimension: the same sweep with the coordinates staged 16 at a time -- once for the
// squared distances and once more for the per-dimension sums (the resident variant above keeps up
// to 64 coordinates of both point blocks in LDS and one accumulator per dimension in registers).
template <int KIND, bool LML>
__global__ __launch_bounds__(256) void grad_sweep_chunked_kernel(GradArgs p) {
  select_model(p);
  __shared__ __attribute__((aligned(16))) double xs[GDC][GT];
  __shared__ __attribute__((aligned(16))) double ys[GDC][GT];
  __shared__ double inv_ell[GDC];
  __shared__ double red[256];

  int ti, tj;
  if (LML) {
    const int q = blockIdx.x;
    ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > q) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
    tj = q - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x / p.tiles_n;
    tj = blockIdx.x - ti * p.tiles_n;
  }
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = ti * GT, j0 = tj * GT;

  auto stage = [&](int c0) {                       // coordinates c0 .. c0+15 of both point blocks
    __syncthreads();                               // previous chunk fully consumed
    if (tid < GDC) inv_ell[tid] = c0 + tid < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : c0 + tid] : 0.0;
    __syncthreads();
    for (int idx = tid; idx < GDC * GT; idx += 256) {
      const int dd = idx / GT, pt = idx - dd * GT;
      double vx = 0.0, vy = 0.0;
      if (c0 + dd < p.d) {
        const double ie = inv_ell[dd];
        if (i0 + pt < p.n) vx = p.X[(int64_t)(i0 + pt) * p.d + c0 + dd] * ie;
        if (j0 + pt < p.m) vy = p.X2[(int64_t)(j0 + pt) * p.d + c0 + dd] * ie;
      }
      xs[dd][pt] = vx;
      ys[dd][pt] = vy;
    }
    __syncthreads();
  };
  auto fragments = [&](int dd, double (&xr)[4], double (&yc)[4]) {
    const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][ty * 4]);
    const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][ty * 4 + 2]);
    const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][tx * 2]);
    const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + tx * 2]);
    xr[0] = xa.x; xr[1] = xa.y; xr[2] = xb.x; xr[3] = xb.y;
    yc[0] = ya.x; yc[1] = ya.y; yc[2] = yb.x; yc[3] = yb.y;
  };

  // pass 1: squared distances
  double r2[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
  for (int c0 = 0; c0 < p.d; c0 += GDC) {
    stage(c0);
    const int lim = min(GDC, p.d - c0);
    for (int dd = 0; dd < lim; ++dd) {
      double xr[4], yc[4];
      fragments(dd, xr, yc);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = xr[a] - yc[b];
          r2[a][b] = fma(df, df, r2[a][b]);
        }
    }
  }

  const double var = p.variance[0];
  double gb[4][4];
  double s_var = 0.0, s_tr = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = i0 + ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = j0 + (b >> 1) * 32 + tx * 2 + (b & 1);
      double g = 0.0;
      if (row < p.n && col < p.m) {
        if constexpr (LML) {
          if (col <= row) {
            double aa = 0.0;
            for (int c = 0; c < p.dy; ++c) aa = fma(p.at[(int64_t)c * p.ldat + row], p.at[(int64_t)c * p.ldat + col], aa);
            g = 0.5 * (aa - (double)p.dy * p.G[(int64_t)row * p.ldg + col]);
            if (col == row) s_tr += g;
            else g *= 2.0;
          }
        } else {
          g = p.G[(int64_t)row * p.ldg + col];
        }
      }
      double K, B;
      k_and_base<KIND>(r2[a][b], var, K, B);
      s_var = fma(g, K, s_var);
      gb[a][b] = g * B;
    }
  }

  auto block_sum = [&](double v) -> double {
    red[tid] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double* out = p.partial + (int64_t)blockIdx.x * p.nout;
  const double tv = block_sum(s_var);
  if (tid == 0) out[0] = tv / var;

  // pass 2: per-dimension sums, chunk by chunk
  double s_iso = 0.0;
  for (int c0 = 0; c0 < p.d; c0 += GDC) {
    stage(c0);
    const int lim = min(GDC, p.d - c0);
    for (int dd = 0; dd < lim; ++dd) {               // uniform
      double xr[4], yc[4];
      fragments(dd, xr, yc);
      double sacc = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = xr[a] - yc[b];
          sacc = fma(gb[a][b], df * df, sacc);
        }
      if (p.nls == 1) {
        s_iso += sacc;
      } else {
        const double t = block_sum(sacc);
        if (tid == 0) out[1 + c0 + dd] = t / p.ls[c0 + dd];
      }
    }
  }
  if (p.nls == 1) {
    const double t = block_sum(s_iso);
    if (tid == 0) out[1] = t / p.ls[0];
  }
  if (LML) {
    const double t = block_sum(s_tr);
    if (tid == 0) out[1 + p.nls] = t;
  }
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(const double* partial, int64_t nblocks, int nout, double* out,
                                                          int64_t sPartial = 0, int sOut = 0) {
  __shared__ double red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  partial += (int64_t)blockIdx.y * sPartial;       // gridDim.y models of a lock-step batch
  out += (int64_t)blockIdx.y * sOut;
  double s = 0.0;
  for (int64_t b = tid; b < nblocks; b += 256) s += partial[b * nout + k];
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) out[k] = red[0];
}

template <int KIND, bool LML>
static int launch_sweep(hipStream_t s, const GradArgs& a, int64_t nblocks, int batch = 1) {
  const int nch = (a.d + GDC - 1) / GDC;
  int rec = -1;
  if (profile_on())   // algorithmic bytes: the gradient matrix is read once (lower triangle for the LML sweep)
    rec = profile_begin(s, batch * (8.0 * (LML ? 0.5 * a.n * (a.n + 1.0) : (double)a.n * a.m) + 8.0 * (a.n + (LML ? 0 : a.m)) * a.d), PROF_GRAD);
  const bool iso = a.nls == 1;
#define GPN_SWEEP(N_) \
  if (iso) hipLaunchKernelGGL((grad_sweep_kernel<KIND, N_, LML, true>), dim3((unsigned)nblocks, (unsigned)batch), dim3(256), 0, s, a); \
  else hipLaunchKernelGGL((grad_sweep_kernel<KIND, N_, LML, false>), dim3((unsigned)nblocks, (unsigned)batch), dim3(256), 0, s, a);
  switch (nch) {
    case 1: GPN_SWEEP(1) break;
    case 2: GPN_SWEEP(2) break;
    case 3: GPN_SWEEP(3) break;
    case 4: GPN_SWEEP(4) break;
    default: hipLaunchKernelGGL((grad_sweep_chunked_kernel<KIND, LML>), dim3((unsigned)nblocks, (unsigned)batch), dim3(256), 0, s, a); break;
  }
#undef GPN_SWEEP
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

template <bool LML>
static int dispatch_kind(hipStream_t s, int kind, const GradArgs& a, int64_t nblocks, int batch = 1) {
  switch (kind) {
    case GPN_RBF: return launch_sweep<GPN_RBF, LML>(s, a, nblocks, batch);
    case GPN_MATERN52: return launch_sweep<GPN_MATERN52, LML>(s, a, nblocks, batch);
    case GPN_MATERN32: return launch_sweep<GPN_MATERN32, LML>(s, a, nblocks, batch);
    case GPN_EXP: return launch_sweep<GPN_EXP, LML>(s, a, nblocks, batch);
    case GPN_PERIODIC: return launch_sweep<GPN_PERIODIC, LML>(s, a, nblocks, batch);
    case GPN_SQDIST: return launch_sweep<GPN_SQDIST, LML>(s, a, nblocks, batch);
    default: return -2;
  }
}

// ---- gradient w.r.t. the points of the second argument --------------------------------
//     out[j, c] = scale * sum_i G[i, j] * dK(x_i, z_j)/dz_jc ,   dK/dz_c = B(r) (x_c - z_c)/ell_c^2
// (what autograd returns for X2 through kernels.py:149-222; used for the inducing points Z
// of sparse_gpr.py:126-129).  One workgroup owns 64 columns j and a slab of rows i: the
// wave index is the row phase, the lane is the column, so every G row segment is one
// coalesced 512-byte read and x_i is an LDS broadcast.  Slab partial sums go to the
// workspace [slabs, m, d]; a second kernel adds them up in a fixed order (deterministic).
struct GradX2Args {
  const double* X;
  const double* X2;
  const double* variance;
  const double* ls;
  const double* G;
  int64_t ldg;
  double* partial;   // [slabs, m, d]
  int n, m, d, nls, slab_rows;
  // lock-step batch (gridDim.z models; the register kernel only): model z reads X + z sX, X2 + z sX2, variance[z], ls + z nls,
  // G + z sG and writes partial + z sPartial
  int batch = 1;
  int64_t sX = 0, sX2 = 0, sG = 0, sPartial = 0;
};

template <int KIND, int DMAX>
__global__ __launch_bounds__(256) void grad_x2_kernel(GradX2Args p) {
  __shared__ double xs[GT][DMAX];          // x_i / ell for the current 64 rows (read as a broadcast)
  __shared__ double red[4][GDC][GT];       // phase reduction, 16 coordinates at a time
  const int tid = threadIdx.x;
  const int lane = tid & 63, ph = tid >> 6;
  const int j0 = blockIdx.x * GT, col = j0 + lane;
  const int r_begin = blockIdx.y * p.slab_rows;
  const int r_end = min(p.n, r_begin + p.slab_rows);
  if (p.batch > 1) {
    const int64_t zb = blockIdx.z;
    p.X += zb * p.sX; p.X2 += zb * p.sX2; p.variance += zb; p.ls += zb * p.nls; p.G += zb * p.sG; p.partial += zb * p.sPartial;
  }
  const double var = p.variance[0];

  __shared__ double inv_ell[DMAX];
  if (tid < DMAX) inv_ell[tid] = tid < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : tid] : 0.0;
  __syncthreads();
  double z[DMAX], acc[DMAX];
#pragma unroll
  for (int c = 0; c < DMAX; ++c) {
    acc[c] = 0.0;
    z[c] = (c < p.d && col < p.m) ? p.X2[(int64_t)col * p.d + c] * inv_ell[c] : 0.0;
  }

  for (int i0 = r_begin; i0 < r_end; i0 += GT) {
    __syncthreads();
    for (int idx = tid; idx < GT * DMAX; idx += 256) {
      const int pt = idx / DMAX, c = idx - pt * DMAX;
      double v = 0.0;
      if (c < p.d && i0 + pt < r_end) v = p.X[(int64_t)(i0 + pt) * p.d + c] * inv_ell[c];
      xs[pt][c] = v;
    }
    __syncthreads();
    const int lim = min(GT, r_end - i0);
    for (int ii = ph; ii < lim; ii += 4) {
      const double g = (col < p.m) ? p.G[(int64_t)(i0 + ii) * p.ldg + col] : 0.0;
      double df[DMAX];
      double r2 = 0.0;
#pragma unroll
      for (int c = 0; c < DMAX; ++c) {
        df[c] = xs[ii][c] - z[c];
        r2 = fma(df[c], df[c], r2);
      }
      double K, B;
      k_and_base<KIND>(r2, var, K, B);
      const double w = g * B;
#pragma unroll
      for (int c = 0; c < DMAX; ++c) acc[c] = fma(w, df[c], acc[c]);
    }
  }

  // add the four row phases, 16 coordinates per round
  for (int cb = 0; cb < DMAX; cb += GDC) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < GDC; ++c) red[ph][c][lane] = acc[cb + c];   // cb uniform, unrolled
    __syncthreads();
#pragma unroll
    for (int q = 0; q < GDC / 4; ++q) {
      const int c = ph * (GDC / 4) + q;
      if (cb + c < p.d && col < p.m) {
        const double t = (red[0][c][lane] + red[1][c][lane]) + (red[2][c][lane] + red[3][c][lane]);
        p.partial[((int64_t)blockIdx.y * p.m + col) * p.d + cb + c] = t * inv_ell[cb + c];
      }
    }
  }
}

// Any input dimension (round 6; the kernel above keeps one accumulator per coordinate in registers and stops at GMAXD = 64):
// gridDim.z enumerates blocks of GDC = 16 OUTPUT coordinates.  Per 64 x 64 tile of (rows, columns) a thread first forms the
// squared distances of its 16 rows over ALL coordinates, 16 at a time through LDS (the direct differences of util.py:73-88's
// result, as everywhere else on the path), turns them into the weights g B(r), and then sweeps its block's 16 coordinates
// once more for the sums.  The distances are recomputed per output block: d / 16 times the arithmetic of the kernel above --
// this is the rarely used wide-input case, and the reference's autograd (util.py:73-88 through kernels.py:149-222) has no limit.
template <int KIND>
__global__ __launch_bounds__(256) void grad_x2_chunked_kernel(GradX2Args p) {
  __shared__ double xs[GT][GDC];
  __shared__ double red[4][GDC][GT];
  __shared__ double inv_ell[GDC];
  const int tid = threadIdx.x;
  const int lane = tid & 63, ph = tid >> 6;
  const int j0 = blockIdx.x * GT, col = j0 + lane;
  const int r_begin = blockIdx.y * p.slab_rows;
  const int r_end = min(p.n, r_begin + p.slab_rows);
  const int c0 = blockIdx.z * GDC;
  const double var = p.variance[0];
  double acc[GDC];
#pragma unroll
  for (int c = 0; c < GDC; ++c) acc[c] = 0.0;
  // stage the coordinates [ch, ch + 16) of the rows i0 .. (scaled by 1 / ell) and this thread's column point
  auto stage = [&](int i0, int ch, double (&z)[GDC]) {
    __syncthreads();
    if (tid < GDC) inv_ell[tid] = ch + tid < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : ch + tid] : 0.0;
    __syncthreads();
    for (int idx = tid; idx < GT * GDC; idx += 256) {
      const int pt = idx / GDC, c = idx - pt * GDC;
      double v = 0.0;
      if (ch + c < p.d && i0 + pt < r_end) v = p.X[(int64_t)(i0 + pt) * p.d + ch + c] * inv_ell[c];
      xs[pt][c] = v;
    }
#pragma unroll
    for (int c = 0; c < GDC; ++c) z[c] = (ch + c < p.d && col < p.m) ? p.X2[(int64_t)col * p.d + ch + c] * inv_ell[c] : 0.0;
    __syncthreads();
  };
  for (int i0 = r_begin; i0 < r_end; i0 += GT) {
    const int lim = min(GT, r_end - i0);
    double r2[GT / 4];
#pragma unroll
    for (int q = 0; q < GT / 4; ++q) r2[q] = 0.0;
    double z[GDC];
    for (int ch = 0; ch < p.d; ch += GDC) {
      stage(i0, ch, z);
#pragma unroll
      for (int q = 0; q < GT / 4; ++q) {
        const int ii = ph + 4 * q;
        if (ii < lim) {
#pragma unroll
          for (int c = 0; c < GDC; ++c) {
            const double df = xs[ii][c] - z[c];
            r2[q] = fma(df, df, r2[q]);
          }
        }
      }
    }
    stage(i0, c0, z);
#pragma unroll
    for (int q = 0; q < GT / 4; ++q) {
      const int ii = ph + 4 * q;
      if (ii < lim) {
        const double g = (col < p.m) ? p.G[(int64_t)(i0 + ii) * p.ldg + col] : 0.0;
        double K, B;
        k_and_base<KIND>(r2[q], var, K, B);
        const double w = g * B;
#pragma unroll
        for (int c = 0; c < GDC; ++c) acc[c] = fma(w, xs[ii][c] - z[c], acc[c]);
      }
    }
  }
  __syncthreads();
  if (tid < GDC) inv_ell[tid] = c0 + tid < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : c0 + tid] : 0.0;
#pragma unroll
  for (int c = 0; c < GDC; ++c) red[ph][c][lane] = acc[c];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < GDC / 4; ++q) {
    const int c = ph * (GDC / 4) + q;
    if (c0 + c < p.d && col < p.m) {
      const double t = (red[0][c][lane] + red[1][c][lane]) + (red[2][c][lane] + red[3][c][lane]);
      p.partial[((int64_t)blockIdx.y * p.m + col) * p.d + c0 + c] = t * inv_ell[c];
    }
  }
}

__global__ __launch_bounds__(256) void grad_x2_reduce_kernel(const double* partial, int slabs, int64_t md, double scale,
                                                             int accumulate, double* out, int64_t sPartial = 0) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= md) return;
  partial += (int64_t)blockIdx.y * sPartial;        // gridDim.y models of a lock-step batch, outputs back to back
  out += (int64_t)blockIdx.y * md;
  double s = 0.0;
  for (int b = 0; b < slabs; ++b) s += partial[(int64_t)b * md + k];
  out[k] = (accumulate ? out[k] : 0.0) + scale * s;
}

static int x2_slabs(int64_t n, int64_t m) {
  const int64_t col_tiles = (m + GT - 1) / GT, row_tiles = (n + GT - 1) / GT;
  int64_t slabs = (1024 + col_tiles - 1) / col_tiles;   // aim for >= 1024 workgroups
  if (slabs > row_tiles) slabs = row_tiles;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

template <int KIND>
static int launch_x2(hipStream_t s, const GradX2Args& a, dim3 grid) {
  if (a.batch > 1 && a.d > GMAXD) return GPN_E_UNSUPPORTED;     // (the chunked kernel's gridDim.z = coordinate blocks)
  if (a.d <= 16) hipLaunchKernelGGL((grad_x2_kernel<KIND, 16>), grid, dim3(256), 0, s, a);
  else if (a.d <= 32) hipLaunchKernelGGL((grad_x2_kernel<KIND, 32>), grid, dim3(256), 0, s, a);
  else if (a.d <= GMAXD) hipLaunchKernelGGL((grad_x2_kernel<KIND, 64>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((grad_x2_chunked_kernel<KIND>), dim3(grid.x, grid.y, (unsigned)((a.d + GDC - 1) / GDC)), dim3(256), 0, s, a);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn

using namespace gpn;

extern "C" int64_t gpn_grad_x2_work_bytes(int64_t n, int64_t m, int d) {
  if (n <= 0 || m <= 0 || d <= 0) return 0;
  return (int64_t)x2_slabs(n, m) * m * d * (int64_t)sizeof(double);
}

extern "C" int gpn_kernel_grad_x2(void* stream, int kind, const double* X, int64_t n, const double* X2, int64_t m, int d,
                                  const double* variance, const double* length_scales, int nls,
                                  const double* G, int64_t ldg, double scale, int accumulate,
                                  double* work, double* out) {
  if (!X) return -3;
  if (n <= 0) return -4;
  if (!X2) return -5;
  if (m <= 0) return -6;
  if (d <= 0 || d > 65535 * GDC) return -7;       // (above GMAXD: the chunked kernel, one grid layer per 16 coordinates)
  if (!variance) return -8;
  if (!length_scales) return -9;
  if (nls != 1 && nls != d) return -10;
  if (!G) return -11;
  if (ldg < m) return -12;
  if (!work) return -15;
  if (!out) return -16;
  GradX2Args a;
  a.X = X; a.X2 = X2; a.variance = variance; a.ls = length_scales; a.G = G; a.ldg = ldg; a.partial = work;
  a.n = (int)n; a.m = (int)m; a.d = d; a.nls = nls;
  const int slabs = x2_slabs(n, m);
  const int64_t row_tiles = (n + GT - 1) / GT;
  a.slab_rows = (int)((row_tiles + slabs - 1) / slabs) * GT;
  const int used = (int)((n + a.slab_rows - 1) / a.slab_rows);   // <= slabs
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((m + GT - 1) / GT), (unsigned)used);
  int rc;
  switch (kind) {
    case GPN_RBF: rc = launch_x2<GPN_RBF>(s, a, grid); break;
    case GPN_MATERN52: rc = launch_x2<GPN_MATERN52>(s, a, grid); break;
    case GPN_MATERN32: rc = launch_x2<GPN_MATERN32>(s, a, grid); break;
    case GPN_EXP: rc = launch_x2<GPN_EXP>(s, a, grid); break;
    case GPN_PERIODIC: rc = launch_x2<GPN_PERIODIC>(s, a, grid); break;
    case GPN_SQDIST: rc = launch_x2<GPN_SQDIST>(s, a, grid); break;
    default: return -2;
  }
  if (rc != GPN_OK) return rc;
  const int64_t md = m * (int64_t)d;
  hipLaunchKernelGGL(grad_x2_reduce_kernel, dim3((unsigned)((md + 255) / 256)), dim3(256), 0, s, work, used, md, scale,
                     accumulate, out);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}


extern "C" int64_t gpn_grad_work_bytes(int64_t n, int64_t m, int nls, int lml) {
  const int64_t tm = (n + GT - 1) / GT, tn = (m + GT - 1) / GT;
  const int64_t blocks = lml ? tm * (tm + 1) / 2 : tm * tn;
  return blocks * (int64_t)(2 + nls) * (int64_t)sizeof(double);
}

extern "C" int gpn_lml_grad(void* stream, int kind, const double* X, int64_t n, int d,
                            const double* variance, const double* length_scales, int nls,
                            const double* Kinv, int64_t ldk, const double* at, int64_t ldat, int dy,
                            double* work, double* out) {
  if (!X) return -3;
  if (n <= 0) return -4;
  if (d <= 0) return -5;
  if (!variance) return -6;
  if (!length_scales) return -7;
  if (nls != 1 && nls != d) return -8;
  if (!Kinv) return -9;
  if (ldk < n) return -10;
  if (!at) return -11;
  if (ldat < n) return -12;
  if (dy <= 0) return -13;
  if (!work) return -14;
  if (!out) return -15;
  GradArgs a;
  a.X = X; a.X2 = X; a.variance = variance; a.ls = length_scales;
  a.G = Kinv; a.ldg = ldk; a.at = at; a.ldat = ldat; a.partial = work;
  a.n = (int)n; a.m = (int)n; a.d = d; a.nls = nls; a.dy = dy; a.nout = 2 + nls;
  a.tiles_m = a.tiles_n = (int)((n + GT - 1) / GT);
  const int64_t nblocks = (int64_t)a.tiles_m * (a.tiles_m + 1) / 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rc = dispatch_kind<true>(s, kind, a, nblocks);
  if (rc != GPN_OK) return rc;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)a.nout), dim3(256), 0, s, work, nblocks, a.nout, out, (int64_t)0, 0);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_lml_grad for `batch` lock-step models in ONE sweep launch (gridDim.y = batch) + one reduction launch: model b reads
// X + b sX (sX = 0: shared points), variance[b], length_scales + b nls, Kinv + b sK, at + b sAt and gets out + b (2 + nls);
// work: batch * gpn_grad_work_bytes(n, n, nls, 1).  Per model the same blocks, the same partial sums in the same order:
// bit-identical to gpn_lml_grad on that model alone.
extern "C" int gpn_lml_grad_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                                    const double* variance, const double* length_scales, int nls,
                                    const double* Kinv, int64_t ldk, int64_t sK, const double* at, int64_t ldat, int64_t sAt, int dy,
                                    double* work, double* out) {
  return gpn::lml_grad_batched(static_cast<hipStream_t>(stream), kind, batch, X, sX, n, d, variance, length_scales, nls, Kinv, ldk, sK,
                               at, ldat, sAt, dy, work, 0, out);
}

// (sWork: doubles between the partial-sum regions of consecutive models; 0 = back to back)
int gpn::lml_grad_batched(hipStream_t s, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                          const double* variance, const double* length_scales, int nls,
                          const double* Kinv, int64_t ldk, int64_t sK, const double* at, int64_t ldat, int64_t sAt, int dy,
                          double* work, int64_t sWork, double* out, const int32_t* n_of) {
  if (batch < 1) return -3;
  if (!X) return -4;
  if (n <= 0) return -6;
  if (d <= 0) return -7;
  if (!variance) return -8;
  if (!length_scales) return -9;
  if (nls != 1 && nls != d) return -10;
  if (!Kinv) return -11;
  if (ldk < n) return -12;
  if (!at) return -14;
  if (ldat < n) return -15;
  if (dy <= 0) return -17;
  if (!work) return -18;
  if (!out) return -19;
  if (batch > 65535) return GPN_E_UNSUPPORTED;
  GradArgs a;
  a.X = X; a.X2 = X; a.variance = variance; a.ls = length_scales;
  a.G = Kinv; a.ldg = ldk; a.at = at; a.ldat = ldat; a.partial = work;
  a.n = (int)n; a.m = (int)n; a.d = d; a.nls = nls; a.dy = dy; a.nout = 2 + nls;
  a.tiles_m = a.tiles_n = (int)((n + GT - 1) / GT);
  const int64_t nblocks = (int64_t)a.tiles_m * (a.tiles_m + 1) / 2;
  a.sX = sX; a.sG = sK; a.sAt = sAt; a.sPartial = sWork > 0 ? sWork : nblocks * a.nout;
  a.n_of = n_of;
  if (sWork > 0 && sWork < nblocks * a.nout) return -18;
  const int rc = dispatch_kind<true>(s, kind, a, nblocks, batch);
  if (rc != GPN_OK) return rc;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)a.nout, (unsigned)batch), dim3(256), 0, s, work, nblocks, a.nout, out,
                     a.sPartial, a.nout);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_kernel_grad(void* stream, int kind, const double* X, int64_t n, const double* X2, int64_t m, int d,
                               const double* variance, const double* length_scales, int nls,
                               const double* G, int64_t ldg, double* work, double* out) {
  if (!X) return -3;
  if (n <= 0) return -4;
  const bool symmetric = (X2 == nullptr);
  if (symmetric) m = n;
  if (m <= 0) return -6;
  if (d <= 0) return -7;
  if (!variance) return -8;
  if (!length_scales) return -9;
  if (nls != 1 && nls != d) return -10;
  if (!G) return -11;
  if (ldg < m) return -12;
  if (!work) return -13;
  if (!out) return -14;
  GradArgs a;
  a.X = X; a.X2 = symmetric ? X : X2; a.variance = variance; a.ls = length_scales;
  a.G = G; a.ldg = ldg; a.at = nullptr; a.ldat = 0; a.partial = work;
  a.n = (int)n; a.m = (int)m; a.d = d; a.nls = nls; a.dy = 0; a.nout = 2 + nls;
  a.tiles_m = (int)((n + GT - 1) / GT);
  a.tiles_n = (int)((m + GT - 1) / GT);
  const int64_t nblocks = (int64_t)a.tiles_m * a.tiles_n;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rc = dispatch_kind<false>(s, kind, a, nblocks);
  if (rc != GPN_OK) return rc;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)(1 + nls)), dim3(256), 0, s, work, nblocks, a.nout, out, (int64_t)0, 0);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_kernel_grad for `batch` models in ONE sweep launch (gridDim.y) + one reduction launch: model b reads X + b sX (0: shared
// points), X2 + b sX2, variance[b], length_scales + b nls, G + b sG and writes out + b (1 + nls).  work: batch *
// gpn_grad_work_bytes(n, m, nls, 0).  Per model bit-identical to gpn_kernel_grad.  (Lock-step sparse models: the sweeps against
// dF/dKuu and dF/dKuf of sparse_gpr.py:126-129 for every restart at once.)
extern "C" int gpn_kernel_grad_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n,
                                       const double* X2, int64_t sX2, int64_t m, int d,
                                       const double* variance, const double* length_scales, int nls,
                                       const double* G, int64_t ldg, int64_t sG, double* work, double* out) {
  if (batch < 1 || batch > 65535) return -3;
  if (!X) return -4;
  if (n <= 0) return -6;
  const bool symmetric = (X2 == nullptr);
  if (symmetric) m = n;
  if (m <= 0) return -9;
  if (d <= 0) return -10;
  if (!variance) return -11;
  if (!length_scales) return -12;
  if (nls != 1 && nls != d) return -13;
  if (!G) return -14;
  if (ldg < m) return -15;
  if (!work) return -17;
  if (!out) return -18;
  GradArgs a;
  a.X = X; a.X2 = symmetric ? X : X2; a.variance = variance; a.ls = length_scales;
  a.G = G; a.ldg = ldg; a.at = nullptr; a.ldat = 0; a.partial = work;
  a.n = (int)n; a.m = (int)m; a.d = d; a.nls = nls; a.dy = 0; a.nout = 2 + nls;
  a.tiles_m = (int)((n + GT - 1) / GT);
  a.tiles_n = (int)((m + GT - 1) / GT);
  const int64_t nblocks = (int64_t)a.tiles_m * a.tiles_n;
  a.sX = sX; a.sX2 = symmetric ? sX : sX2; a.sG = sG; a.sAt = 0; a.sPartial = nblocks * a.nout;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rc = dispatch_kind<false>(s, kind, a, nblocks, batch);
  if (rc != GPN_OK) return rc;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)(1 + nls), (unsigned)batch), dim3(256), 0, s, work, nblocks, a.nout, out,
                     a.sPartial, 1 + nls);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_kernel_grad_x2 for `batch` models: out + b m d (+)= scale * ...; work: batch * gpn_grad_x2_work_bytes(n, m, d).  Up to 64
// input dimensions ONE launch over all models (+ one reduction launch), above that model by model; per model bit-identical to
// gpn_kernel_grad_x2.
extern "C" int gpn_kernel_grad_x2_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n,
                                          const double* X2, int64_t sX2, int64_t m, int d,
                                          const double* variance, const double* length_scales, int nls,
                                          const double* G, int64_t ldg, int64_t sG, double scale, int accumulate,
                                          double* work, double* out) {
  if (batch < 1 || batch > 65535) return -3;
  if (!X) return -4;
  if (n <= 0) return -6;
  if (!X2) return -7;
  if (m <= 0) return -9;
  if (d <= 0 || d > 65535 * GDC) return -10;
  if (!variance) return -11;
  if (!length_scales) return -12;
  if (nls != 1 && nls != d) return -13;
  if (!G) return -14;
  if (ldg < m) return -15;
  if (!work) return -19;
  if (!out) return -20;
  const int64_t md = m * (int64_t)d;
  const int64_t one = gpn_grad_x2_work_bytes(n, m, d) / (int64_t)sizeof(double);
  if (batch == 1 || d > GMAXD) {
    for (int z = 0; z < batch; ++z) {
      const int rc = gpn_kernel_grad_x2(stream, kind, X + z * sX, n, X2 + z * sX2, m, d, variance + z, length_scales + (int64_t)z * nls, nls,
                                        G + z * sG, ldg, scale, accumulate, work + z * one, out + z * md);
      if (rc != GPN_OK) return rc;
    }
    return GPN_OK;
  }
  GradX2Args a;
  a.X = X; a.X2 = X2; a.variance = variance; a.ls = length_scales; a.G = G; a.ldg = ldg; a.partial = work;
  a.n = (int)n; a.m = (int)m; a.d = d; a.nls = nls;
  const int slabs = x2_slabs(n, m);
  const int64_t row_tiles = (n + GT - 1) / GT;
  a.slab_rows = (int)((row_tiles + slabs - 1) / slabs) * GT;
  const int used = (int)((n + a.slab_rows - 1) / a.slab_rows);
  a.batch = batch; a.sX = sX; a.sX2 = sX2; a.sG = sG; a.sPartial = one;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((m + GT - 1) / GT), (unsigned)used, (unsigned)batch);
  int rc;
  switch (kind) {
    case GPN_RBF: rc = launch_x2<GPN_RBF>(s, a, grid); break;
    case GPN_MATERN52: rc = launch_x2<GPN_MATERN52>(s, a, grid); break;
    case GPN_MATERN32: rc = launch_x2<GPN_MATERN32>(s, a, grid); break;
    case GPN_EXP: rc = launch_x2<GPN_EXP>(s, a, grid); break;
    case GPN_PERIODIC: rc = launch_x2<GPN_PERIODIC>(s, a, grid); break;
    case GPN_SQDIST: rc = launch_x2<GPN_SQDIST>(s, a, grid); break;
    default: return -2;
  }
  if (rc != GPN_OK) return rc;
  hipLaunchKernelGGL(grad_x2_reduce_kernel, dim3((unsigned)((md + 255) / 256), (unsigned)batch), dim3(256), 0, s, work, used, md, scale,
                     accumulate, out, one);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}
