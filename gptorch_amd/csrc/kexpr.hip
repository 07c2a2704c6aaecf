// Composite covariance functions in ONE pass: gptorch's Sum / Product combinators (kernels.py:286-306) over
// stationary kernels (kernels.py:108-235), Linear (238-265), Constant / Bias (95-105) and White (83-92), e.g. the
// reference's own example model `Linear + Rbf + Constant` (examples/regression_1d.py:34-53).  The reference composes
// them with elementwise torch ops on dense N x M matrices (one assembly chain per leaf, then + / *); here the
// expression is evaluated per 64 x 64 tile and the N x M matrix is written once -- for GPR straight into the factor
// buffer (lower tiles, noise on the diagonal), exactly like the single-kernel assembly of kmat.hip.
//
// Expression form: the Sum / Product tree is expanded by the caller into a SUM OF PRODUCTS of leaf terms
// (distributivity), which needs no evaluation stack in registers:   K = sum_g prod_{t in g} term_t(x, x').
// Leaf terms read their constrained parameter values from one packed device array `theta`.
//
// gpn_kernel_expr_grad is the matching backward sweep: for ONE target term instance it contracts the weight matrix
// G (given densely, or formed on the fly from Kyy^-1 and a as 1/2 (a a^T - dy Kyy^-1) -- the closed-form dLML/dK) with
// the product of the other terms of its group and with d term / d theta, re-computing everything from the points, so
// each launch reads G once and nothing N x N is written.  A Sum of T leaves costs T such sweeps.
#include "gpn_common.h"
#include "refine_tail.h"
#include "kernel_fn.h"

namespace gpn {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int ET = 64;    // tile edge
constexpr int EDC = 16;   // coordinates staged per pass

struct ExprProgram {      // passed by value (kernel argument)
  int ngroups;
  int gstart[GPN_EXPR_MAX_GROUPS + 1];   // terms of group g: gstart[g] .. gstart[g+1]-1
  gpn_expr_term terms[GPN_EXPR_MAX_TERMS];
};

__device__ __forceinline__ double stationary_value(int kind, double r2, double var) {
  switch (kind) {
    case GPN_RBF: return kernel_of_r2<GPN_RBF>(r2, var);
    case GPN_MATERN52: return kernel_of_r2<GPN_MATERN52>(r2, var);
    case GPN_MATERN32: return kernel_of_r2<GPN_MATERN32>(r2, var);
    case GPN_PERIODIC: return kernel_of_r2<GPN_PERIODIC>(r2, var);
    default: return kernel_of_r2<GPN_EXP>(r2, var);
  }
}

// K and B = the factor of dK/d ell_d = B s_d / ell_d (grad.hip k_and_base), kind at run time
__device__ __forceinline__ void stationary_k_b(int kind, double r2, double var, double& K, double& B) {
  if (kind == GPN_RBF) { K = var * exp(-0.5 * r2); B = K; return; }
  const bool dead = r2 < 1e-40;                  // kernels.py:172 clamp: no gradient below it
  const double r = sqrt(fmax(r2, 1e-40));
  if (kind == GPN_MATERN52) {
    const double s5 = 2.23606797749978969641, e = exp(-s5 * r);
    K = var * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * e;
    B = dead ? 0.0 : var * (5.0 / 3.0) * (1.0 + s5 * r) * e;
  } else if (kind == GPN_MATERN32) {
    const double s3 = 1.73205080756887729353, e = exp(-s3 * r);
    K = var * (1.0 + s3 * r) * e;
    B = dead ? 0.0 : 3.0 * var * e;
  } else if (kind == GPN_PERIODIC) {
    K = var * cos(r);
    B = dead ? 0.0 : var * sin(r) / r;
  } else {
    const double e = exp(-r);
    K = var * e;
    B = dead ? 0.0 : var * e / r;
  }
}

struct TileCtx {
  const double* X;
  const double* X2;
  const double* theta;
  int n, m, d, symmetric;
  int i0, j0, tid, tx, ty;
};

// acc[a][b] <- sum_d f(x_ad, y_bd) for the thread's 4 x 4 micro-tile (rows ty*4+a, cols {2tx, 2tx+1, 32+2tx, 33+2tx}):
// stationary: ((x - y) / ell_d)^2 (points pre-scaled while staging, as kmat.hip); linear: (x v_d) y.
// Called by all 256 threads of the workgroup (barriers inside).
__device__ __forceinline__ void term_accumulate(const TileCtx& c, const gpn_expr_term& t, double (*xs)[ET], double (*ys)[ET],
                                                double* scale, double (&acc)[4][4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  const bool lin = t.type == GPN_TERM_LINEAR;
  for (int d0 = 0; d0 < c.d; d0 += EDC) {
    __syncthreads();                                     // the previous chunk / term is fully consumed
    if (c.tid < EDC) {
      const int dd = d0 + c.tid;
      double s = 0.0;
      if (dd < c.d) {
        const double p = c.theta[lin ? t.var_off + (t.nvar == 1 ? 0 : dd) : t.ls_off + (t.nls == 1 ? 0 : dd)];
        s = lin ? p : 1.0 / p;
      }
      scale[c.tid] = s;
    }
    __syncthreads();
    {
      const int pt = c.tid >> 2, c4 = (c.tid & 3) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dd = d0 + c4 + q;
        double vx = 0.0, vy = 0.0;
        if (dd < c.d) {
          const double sc = scale[c4 + q];
          if (c.i0 + pt < c.n) vx = c.X[(int64_t)(c.i0 + pt) * c.d + dd] * sc;
          if (c.j0 + pt < c.m) vy = c.X2[(int64_t)(c.j0 + pt) * c.d + dd] * (lin ? 1.0 : sc);
        }
        xs[c4 + q][pt] = vx;
        ys[c4 + q][pt] = vy;
      }
    }
    __syncthreads();
    const int dmax = min(EDC, c.d - d0);
    for (int dd = 0; dd < dmax; ++dd) {
      const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4]);
      const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4 + 2]);
      const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][c.tx * 2]);
      const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + c.tx * 2]);
      const double xr[4] = {xa.x, xa.y, xb.x, xb.y};
      const double yc[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (lin) acc[a][b] = fma(xr[a], yc[b], acc[a][b]);
          else { const double df = xr[a] - yc[b]; acc[a][b] = fma(df, df, acc[a][b]); }
        }
    }
  }
}

// val[a][b] <- term value on the thread's micro-tile
__device__ __forceinline__ void term_value(const TileCtx& c, const gpn_expr_term& t, double (*xs)[ET], double (*ys)[ET],
                                           double* scale, double (&val)[4][4]) {
  if (t.type == GPN_TERM_STATIONARY || t.type == GPN_TERM_LINEAR) {
    term_accumulate(c, t, xs, ys, scale, val);
    if (t.type == GPN_TERM_STATIONARY) {
      const double var = c.theta[t.var_off];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) val[a][b] = stationary_value(t.kind, val[a][b], var);
    }
    return;
  }
  const double var = c.theta[t.var_off];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = c.i0 + c.ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = c.j0 + (b >> 1) * 32 + c.tx * 2 + (b & 1);
      // White (kernels.py:83-92): variance on the diagonal of K(X), zero for K(X, X2)
      val[a][b] = t.type == GPN_TERM_CONSTANT ? var : ((c.symmetric && row == col) ? var : 0.0);
    }
  }
}

__device__ __forceinline__ void tile_of_block(int q, bool lower, int tiles_n, int& ti, int& tj) {
  if (lower) {
    ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > q) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
    tj = q - ti * (ti + 1) / 2;
  } else {
    ti = q / tiles_n;
    tj = q - ti * tiles_n;
  }
}

struct ExprArgs {
  const double* X;
  const double* X2;
  const double* theta;
  const double* noise;      // nullptr: no diagonal add
  double* K;
  int64_t ldk;
  int n, m, d, symmetric, lower, tiles_n;
  // lock-step batch (gridDim.y models of ONE program structure, gpn_kernel_matrix_expr_batched): model z reads X + z sX,
  // theta + z sTheta, noise[z] and writes K + z sK
  int64_t sX = 0, sTheta = 0, sK = 0;
};

__global__ __launch_bounds__(256) void kexpr_kernel(ExprArgs p, ExprProgram prog) {
  __shared__ __attribute__((aligned(16))) double xs[EDC][ET];
  __shared__ __attribute__((aligned(16))) double ys[EDC][ET];
  __shared__ double scale[EDC];
  if (gridDim.y > 1) {
    const int64_t z = blockIdx.y;
    p.X += z * p.sX; p.X2 += z * p.sX; p.theta += z * p.sTheta; p.K += z * p.sK;
    if (p.noise) p.noise += z;
  }
  TileCtx c;
  c.X = p.X; c.X2 = p.X2; c.theta = p.theta; c.n = p.n; c.m = p.m; c.d = p.d; c.symmetric = p.symmetric;
  c.tid = threadIdx.x; c.tx = c.tid & 15; c.ty = c.tid >> 4;
  int ti, tj;
  tile_of_block(blockIdx.x, p.lower != 0, p.tiles_n, ti, tj);
  c.i0 = ti * ET; c.j0 = tj * ET;
  double total[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) total[a][b] = 0.0;
  for (int g = 0; g < prog.ngroups; ++g) {
    double prod[4][4], val[4][4];
    for (int t = prog.gstart[g]; t < prog.gstart[g + 1]; ++t) {
      term_value(c, prog.terms[t], xs, ys, scale, val);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) prod[a][b] = t == prog.gstart[g] ? val[a][b] : prod[a][b] * val[a][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) total[a][b] += prod[a][b];
  }
  const double noise = p.noise ? p.noise[0] : 0.0;
  const bool add_diag = p.noise != nullptr && p.symmetric;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = c.i0 + c.ty * 4 + a;
    if (row >= p.n) continue;
    double* krow = p.K + (int64_t)row * p.ldk;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = c.j0 + (b >> 1) * 32 + c.tx * 2 + (b & 1);
      if (col >= p.m) continue;
      double v = total[a][b];
      if (add_diag && row == col) v += noise;
      krow[col] = v;
    }
  }
}

// The residual pass of the refinement step (refine.hip) for an expression: tile q0 + blockIdx.x of the lower 64 x 64
// tiles of Kyy = expression(X, X) + noise I, evaluated exactly as kexpr_kernel assembled it, times a_hat in double-double -- row
// partial and (off-diagonal tiles) mirror column partial, the layout and reduction of refine.hip's stationary kernel.
struct ExprResidArgs {
  const double* X;
  const double* theta;
  const double* noise;
  RefineTailArgs tail;
  int n, d, q_off;
};

template <int NRHS>
__global__ __launch_bounds__(256) void kexpr_resid_kernel(ExprResidArgs p, ExprProgram prog) {
  static_assert(ET == RT, "the expression tile and the residual tile are the same 64 x 64 tile");
  __shared__ __attribute__((aligned(16))) double xs[EDC][ET];
  __shared__ __attribute__((aligned(16))) double ys[EDC][ET];
  __shared__ double scale[EDC];
  __shared__ double arow[NRHS][RT], acol[NRHS][RT];
  __shared__ double red[2][RT][17];
  TileCtx c;
  c.X = p.X; c.X2 = p.X; c.theta = p.theta; c.n = p.n; c.m = p.n; c.d = p.d; c.symmetric = 1;
  c.tid = threadIdx.x; c.tx = c.tid & 15; c.ty = c.tid >> 4;
  int ti, tj;
  tile_of_block((int)blockIdx.x + p.q_off, true, 0, ti, tj);
  c.i0 = ti * ET; c.j0 = tj * ET;
  double total[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) total[a][b] = 0.0;
  for (int g = 0; g < prog.ngroups; ++g) {
    double prod[4][4], val[4][4];
    for (int t = prog.gstart[g]; t < prog.gstart[g + 1]; ++t) {
      term_value(c, prog.terms[t], xs, ys, scale, val);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) prod[a][b] = t == prog.gstart[g] ? val[a][b] : prod[a][b] * val[a][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) total[a][b] += prod[a][b];
  }
  const double noise = p.noise[0];
  double v[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = c.i0 + c.ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = c.j0 + (b >> 1) * 32 + c.tx * 2 + (b & 1);
      double e = total[a][b];
      if (row == col) e += noise;
      v[a][b] = (row < p.n && col < p.n) ? e : 0.0;
    }
  }
  refine_tile_tail<NRHS>(p.tail, v, ti != tj, (int)blockIdx.x, c.i0, c.j0, arow, acol, red);
}

struct ExprGradArgs {
  const double* X;
  const double* X2;
  const double* theta;
  const double* G;          // dense mode: G [n, m]; LML mode: Kyy^-1 (lower) [n, n]
  int64_t ldg;
  const double* at;         // LML mode: a^T [dy, n]
  int64_t ldat;
  double* partial;          // [nblocks, nout]
  int n, m, d, dy, symmetric, lml, tiles_n, nout;
  int target;               // index into prog.terms
  int group;                // its group: the other terms of the group multiply the weight
  int want_trace;           // LML mode: out[nout-1] = trace(G)  (d/d noise)
  // lock-step batch (gridDim.y models of one program structure, gpn_kernel_expr_grad_batched): model z reads X + z sX,
  // theta + z sTheta, G + z sG, at + z sAt and writes partial + z sPartial
  int64_t sX = 0, sTheta = 0, sG = 0, sAt = 0, sPartial = 0;
};

// nout = (target's parameter count) + want_trace.  Parameter order of a target: stationary: variance, then the
// length-scale(s); linear: its variance(s); constant / white: variance.  ARD accumulators live in registers: d <= 16.
__global__ __launch_bounds__(256) void kexpr_grad_kernel(ExprGradArgs p, ExprProgram prog) {
  __shared__ __attribute__((aligned(16))) double xs[EDC][ET];
  __shared__ __attribute__((aligned(16))) double ys[EDC][ET];
  __shared__ double scale[EDC];
  __shared__ double red[256];
  if (gridDim.y > 1) {
    const int64_t z = blockIdx.y;
    p.X += z * p.sX; p.X2 += z * p.sX; p.theta += z * p.sTheta; p.G += z * p.sG; p.partial += z * p.sPartial;
    if (p.at) p.at += z * p.sAt;
  }
  TileCtx c;
  c.X = p.X; c.X2 = p.X2; c.theta = p.theta; c.n = p.n; c.m = p.m; c.d = p.d; c.symmetric = p.symmetric;
  c.tid = threadIdx.x; c.tx = c.tid & 15; c.ty = c.tid >> 4;
  int ti, tj;
  tile_of_block(blockIdx.x, p.lml != 0, p.tiles_n, ti, tj);
  c.i0 = ti * ET; c.j0 = tj * ET;
  const int tid = c.tid;

  // weight: G (x 2 for the symmetric partner of an off-diagonal entry in LML mode), times the other terms of the group
  double g[4][4];
  double s_tr = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = c.i0 + c.ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = c.j0 + (b >> 1) * 32 + c.tx * 2 + (b & 1);
      double v = 0.0;
      if (row < p.n && col < p.m) {
        if (p.lml) {
          if (col <= row) {
            double aa = 0.0;
            for (int q = 0; q < p.dy; ++q) aa = fma(p.at[(int64_t)q * p.ldat + row], p.at[(int64_t)q * p.ldat + col], aa);
            v = 0.5 * (aa - (double)p.dy * p.G[(int64_t)row * p.ldg + col]);
            if (col == row) s_tr += v;
            else v *= 2.0;
          }
        } else {
          v = p.G[(int64_t)row * p.ldg + col];
        }
      }
      g[a][b] = v;
    }
  }
  for (int t = prog.gstart[p.group]; t < prog.gstart[p.group + 1]; ++t) {
    if (t == p.target) continue;
    double val[4][4];
    term_value(c, prog.terms[t], xs, ys, scale, val);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) g[a][b] *= val[a][b];
  }

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
  const gpn_expr_term T = prog.terms[p.target];
  int nparam = 0;
  if (T.type == GPN_TERM_STATIONARY) {
    double r2[4][4];
    term_accumulate(c, T, xs, ys, scale, r2);
    const double var = p.theta[T.var_off];
    double s_var = 0.0, s_iso = 0.0;
    double gb[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double K, B;
        stationary_k_b(T.kind, r2[a][b], var, K, B);
        s_var = fma(g[a][b], K, s_var);
        gb[a][b] = g[a][b] * B;
        s_iso = fma(gb[a][b], r2[a][b], s_iso);
      }
    const double tv = block_sum(s_var);
    if (tid == 0) out[0] = tv / var;                           // dK/dvar = K / var
    if (T.nls == 1) {
      const double tl = block_sum(s_iso);                      // sum_d s_d = r^2
      if (tid == 0) out[1] = tl / p.theta[T.ls_off];
    } else {
      // ARD (d <= 16, checked by the host): the staged chunk of term_accumulate's last (only) pass is still in LDS
      for (int dd = 0; dd < c.d; ++dd) {
        const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4]);
        const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4 + 2]);
        const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][c.tx * 2]);
        const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + c.tx * 2]);
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
        const double td = block_sum(sacc);
        if (tid == 0) out[1 + dd] = td / p.theta[T.ls_off + dd];
      }
    }
    nparam = 1 + T.nls;
  } else if (T.type == GPN_TERM_LINEAR) {
    // d (sum_d x_d v_d y_d) / d v_d = x_d y_d  (one shared variance: the sum over d)
    if (T.nvar == 1) {
      // (x . y from the value pass divided by v would lose the v = 0 case: accumulate the raw coordinates instead)
      double acc1[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc1[a][b] = 0.0;
      for (int d0 = 0; d0 < c.d; d0 += EDC) {
        __syncthreads();
        {
          const int pt = tid >> 2, c4 = (tid & 3) * 4;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int dd = d0 + c4 + q;
            double vx = 0.0, vy = 0.0;
            if (dd < c.d) {
              if (c.i0 + pt < c.n) vx = c.X[(int64_t)(c.i0 + pt) * c.d + dd];
              if (c.j0 + pt < c.m) vy = c.X2[(int64_t)(c.j0 + pt) * c.d + dd];
            }
            xs[c4 + q][pt] = vx;
            ys[c4 + q][pt] = vy;
          }
        }
        __syncthreads();
        const int dmax = min(EDC, c.d - d0);
        for (int dd = 0; dd < dmax; ++dd) {
          const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4]);
          const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4 + 2]);
          const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][c.tx * 2]);
          const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + c.tx * 2]);
          const double xr[4] = {xa.x, xa.y, xb.x, xb.y};
          const double yc[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc1[a][b] = fma(xr[a], yc[b], acc1[a][b]);
        }
      }
      double sv = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) sv = fma(g[a][b], acc1[a][b], sv);
      const double tvv = block_sum(sv);
      if (tid == 0) out[0] = tvv;
      nparam = 1;
    } else {
      // one variance per input (d <= 16, checked by the host): stage the raw coordinates once
      __syncthreads();
      {
        const int pt = tid >> 2, c4 = (tid & 3) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dd = c4 + q;
          double vx = 0.0, vy = 0.0;
          if (dd < c.d) {
            if (c.i0 + pt < c.n) vx = c.X[(int64_t)(c.i0 + pt) * c.d + dd];
            if (c.j0 + pt < c.m) vy = c.X2[(int64_t)(c.j0 + pt) * c.d + dd];
          }
          xs[c4 + q][pt] = vx;
          ys[c4 + q][pt] = vy;
        }
      }
      __syncthreads();
      for (int dd = 0; dd < c.d; ++dd) {
        const d2 xa = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4]);
        const d2 xb = *reinterpret_cast<const d2*>(&xs[dd][c.ty * 4 + 2]);
        const d2 ya = *reinterpret_cast<const d2*>(&ys[dd][c.tx * 2]);
        const d2 yb = *reinterpret_cast<const d2*>(&ys[dd][32 + c.tx * 2]);
        const double xr[4] = {xa.x, xa.y, xb.x, xb.y};
        const double yc[4] = {ya.x, ya.y, yb.x, yb.y};
        double sacc = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) sacc = fma(g[a][b], xr[a] * yc[b], sacc);
        const double td = block_sum(sacc);
        if (tid == 0) out[dd] = td;
      }
      nparam = T.nvar;
    }
  } else {
    // constant: sum of the weights; white: their sum over the diagonal of K(X)
    double sv = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = c.i0 + c.ty * 4 + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = c.j0 + (b >> 1) * 32 + c.tx * 2 + (b & 1);
        if (T.type == GPN_TERM_CONSTANT || (c.symmetric && row == col)) sv += g[a][b];
      }
    }
    const double tvv = block_sum(sv);
    if (tid == 0) out[0] = tvv;
    nparam = 1;
  }
  if (p.want_trace) {
    const double tt = block_sum(s_tr);
    if (tid == 0) out[nparam] = tt;
  }
}

// out[k] = sum over blocks of partial[b * nout + k], in a fixed order
__global__ __launch_bounds__(256) void kexpr_reduce_kernel(const double* partial, int64_t nblocks, int nout, double* out,
                                                           int64_t sPartial = 0) {
  __shared__ double red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  partial += (int64_t)blockIdx.y * sPartial;       // gridDim.y models of a lock-step batch, outputs nout apart
  out += (int64_t)blockIdx.y * nout;
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

static int check_program(const gpn_expr_term* terms, int nterms, const int* gstart, int ngroups, int d, bool for_grad) {
  if (!terms || nterms < 1 || nterms > GPN_EXPR_MAX_TERMS) return -2;
  if (!gstart || ngroups < 1 || ngroups > GPN_EXPR_MAX_GROUPS) return -3;
  if (gstart[0] != 0 || gstart[ngroups] != nterms) return -3;
  for (int g = 0; g < ngroups; ++g) if (gstart[g + 1] <= gstart[g]) return -3;
  for (int t = 0; t < nterms; ++t) {
    const gpn_expr_term& T = terms[t];
    if (T.type < GPN_TERM_STATIONARY || T.type > GPN_TERM_WHITE) return -2;
    if (T.type == GPN_TERM_STATIONARY) {
      if (T.kind < GPN_RBF || T.kind > GPN_PERIODIC || T.kind == GPN_SQDIST) return -2;
      if (T.nls != 1 && T.nls != d) return -2;
      if (for_grad && T.nls != 1 && d > EDC) return GPN_E_UNSUPPORTED;
    }
    if (T.type == GPN_TERM_LINEAR) {
      if (T.nvar != 1 && T.nvar != d) return -2;
      if (for_grad && T.nvar != 1 && d > EDC) return GPN_E_UNSUPPORTED;
    }
    if (T.var_off < 0 || T.ls_off < 0) return -2;
  }
  return GPN_OK;
}

static void fill_program(ExprProgram& P, const gpn_expr_term* terms, int nterms, const int* gstart, int ngroups) {
  P.ngroups = ngroups;
  for (int g = 0; g <= GPN_EXPR_MAX_GROUPS; ++g) P.gstart[g] = g <= ngroups ? gstart[g] : nterms;
  for (int t = 0; t < GPN_EXPR_MAX_TERMS; ++t) {
    if (t < nterms) P.terms[t] = terms[t];
    else P.terms[t] = gpn_expr_term{GPN_TERM_CONSTANT, 0, 0, 0, 1, 1};
  }
}

int expr_resid(hipStream_t s, const gpn_expr_term* terms, int nterms, const int* gstart, int ngroups, const double* theta,
               const double* X, int64_t n, int d, const double* noise, const double* a, int dy, int64_t lds, int64_t q0, int64_t cnt,
               double* prow, double* pcol) {
  int rc = check_program(terms, nterms, gstart, ngroups, d, false);
  if (rc != GPN_OK) return rc;
  if (cnt <= 0) return GPN_OK;
  ExprResidArgs p;
  p.X = X; p.theta = theta; p.noise = noise;
  p.tail.a = a; p.tail.prow = prow; p.tail.pcol = pcol; p.tail.lds = lds; p.tail.n = (int)n; p.tail.dy = dy;
  p.n = (int)n; p.d = d; p.q_off = (int)q0;
  ExprProgram P;
  fill_program(P, terms, nterms, gstart, ngroups);
  if (dy == 1) hipLaunchKernelGGL(kexpr_resid_kernel<1>, dim3((unsigned)cnt), dim3(256), 0, s, p, P);
  else hipLaunchKernelGGL(kexpr_resid_kernel<RDY>, dim3((unsigned)cnt), dim3(256), 0, s, p, P);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

}  // namespace gpn

using namespace gpn;

extern "C" int gpn_kernel_matrix_expr(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                      const double* theta, const double* X, int64_t n, const double* X2, int64_t m, int d,
                                      const double* noise, int uplo, double* K, int64_t ldk) {
  if (d <= 0) return -11;
  int rc = check_program(terms, nterms, group_start, ngroups, d, false);
  if (rc != GPN_OK) return rc;
  if (!theta) return -6;
  if (!X) return -7;
  if (n < 0) return -8;
  const bool symmetric = (X2 == nullptr);
  if (symmetric) m = n;
  if (m < 0) return -10;
  if (uplo != GPN_FULL && uplo != GPN_LOWER) return -13;
  if (uplo == GPN_LOWER && !symmetric) return -13;
  if (!K) return -14;
  if (ldk < m) return -15;
  if (n == 0 || m == 0) return GPN_OK;
  ExprArgs a;
  a.X = X; a.X2 = symmetric ? X : X2; a.theta = theta; a.noise = noise; a.K = K; a.ldk = ldk;
  a.n = (int)n; a.m = (int)m; a.d = d; a.symmetric = symmetric; a.lower = uplo == GPN_LOWER;
  const unsigned tn = (unsigned)((m + ET - 1) / ET), tm = (unsigned)((n + ET - 1) / ET);
  a.tiles_n = (int)tn;
  ExprProgram P;
  fill_program(P, terms, nterms, group_start, ngroups);
  const unsigned blocks = a.lower ? tm * (tm + 1) / 2 : tm * tn;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rec = -1;
  if (profile_on())
    rec = profile_begin(s, a.lower ? 8.0 * (0.5 * n * (n + 1.0) + (double)n * d) : 8.0 * ((double)n * m + (double)(n + (symmetric ? 0 : m)) * d),
                        PROF_KMAT);
  hipLaunchKernelGGL(kexpr_kernel, dim3(blocks), dim3(256), 0, s, a, P);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int64_t gpn_kernel_expr_grad_work_bytes(int64_t n, int64_t m, int d, int lml) {
  if (n <= 0 || m <= 0 || d <= 0) return 0;
  const int64_t tm = (n + ET - 1) / ET, tn = (m + ET - 1) / ET;
  const int64_t blocks = lml ? tm * (tm + 1) / 2 : tm * tn;
  return blocks * (int64_t)(d + 3) * (int64_t)sizeof(double);
}

extern "C" int gpn_kernel_expr_grad(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                    const double* theta, int target, const double* X, int64_t n, const double* X2, int64_t m, int d,
                                    const double* G, int64_t ldg, const double* at, int64_t ldat, int dy, int want_trace,
                                    double* work, double* out) {
  if (d <= 0) return -12;
  int rc = check_program(terms, nterms, group_start, ngroups, d, true);
  if (rc != GPN_OK) return rc;
  if (!theta) return -6;
  if (target < 0 || target >= nterms) return -7;
  if (!X) return -8;
  if (n <= 0) return -9;
  const bool symmetric = (X2 == nullptr);
  if (symmetric) m = n;
  if (m <= 0) return -11;
  if (!G) return -13;
  if (ldg < m) return -14;
  const bool lml = at != nullptr;
  if (lml && (!symmetric || ldat < n || dy <= 0)) return -15;
  if (want_trace && !lml) return -18;
  if (!work) return -19;
  if (!out) return -20;
  ExprGradArgs a;
  a.X = X; a.X2 = symmetric ? X : X2; a.theta = theta; a.G = G; a.ldg = ldg; a.at = at; a.ldat = ldat; a.partial = work;
  a.n = (int)n; a.m = (int)m; a.d = d; a.dy = dy; a.symmetric = symmetric; a.lml = lml; a.target = target; a.want_trace = want_trace ? 1 : 0;
  a.group = 0;
  for (int g = 0; g < ngroups; ++g) if (target >= group_start[g] && target < group_start[g + 1]) a.group = g;
  const gpn_expr_term& T = terms[target];
  const int nparam = T.type == GPN_TERM_STATIONARY ? 1 + T.nls : (T.type == GPN_TERM_LINEAR ? T.nvar : 1);
  a.nout = nparam + a.want_trace;
  const int64_t tm = (n + ET - 1) / ET, tn = (m + ET - 1) / ET;
  a.tiles_n = (int)tn;
  const int64_t nblocks = lml ? tm * (tm + 1) / 2 : tm * tn;
  ExprProgram P;
  fill_program(P, terms, nterms, group_start, ngroups);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, lml ? 8.0 * 0.5 * n * (n + 1.0) : 8.0 * (double)n * m, PROF_GRAD);
  hipLaunchKernelGGL(kexpr_grad_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, a, P);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  hipLaunchKernelGGL(kexpr_reduce_kernel, dim3((unsigned)a.nout), dim3(256), 0, s, work, nblocks, a.nout, out);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_kernel_matrix_expr (symmetric, lower tiles, noise on the diagonal: Kyy straight into the factor buffers) for `batch` models
// whose expressions have ONE structure -- the same term table, every model its own parameter values theta + b sTheta -- in one
// launch (gridDim.y): the reference's example model Linear + Rbf + Constant (examples/regression_1d.py:34-53) in a multi-start
// search.  Model b reads X + b sX (0: shared points), noise[b] and writes K + b sK; per model bit-identical to
// gpn_kernel_matrix_expr.
extern "C" int gpn_kernel_matrix_expr_batched(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                              int batch, const double* theta, int64_t sTheta, const double* X, int64_t sX, int64_t n, int d,
                                              const double* noise, double* K, int64_t ldk, int64_t sK) {
  if (d <= 0) return -12;
  int rc = check_program(terms, nterms, group_start, ngroups, d, false);
  if (rc != GPN_OK) return rc;
  if (batch < 1 || batch > 65535) return -6;
  if (!theta) return -7;
  if (!X) return -9;
  if (n < 0) return -11;
  if (!K) return -14;
  if (ldk < n) return -15;
  if (batch > 1 && sK < n * ldk) return -16;
  if (n == 0) return GPN_OK;
  ExprArgs a;
  a.X = X; a.X2 = X; a.theta = theta; a.noise = noise; a.K = K; a.ldk = ldk;
  a.n = (int)n; a.m = (int)n; a.d = d; a.symmetric = 1; a.lower = 1;
  const unsigned tm = (unsigned)((n + ET - 1) / ET);
  a.tiles_n = (int)tm;
  a.sX = sX; a.sTheta = sTheta; a.sK = sK;
  ExprProgram P;
  fill_program(P, terms, nterms, group_start, ngroups);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, batch * 8.0 * (0.5 * n * (n + 1.0) + (double)n * d), PROF_KMAT);
  hipLaunchKernelGGL(kexpr_kernel, dim3(tm * (tm + 1) / 2, (unsigned)batch), dim3(256), 0, s, a, P);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_kernel_expr_grad in LML mode (weights from Kyy^-1 and a^T) for `batch` models of one structure: ONE sweep launch + one
// reduction launch per leaf instance instead of one pair per model.  Model b reads theta + b sTheta, X + b sX, G + b sG,
// at + b sAt; out [batch, nout]; work: batch * gpn_kernel_expr_grad_work_bytes(n, n, d, 1).  Per model bit-identical to
// gpn_kernel_expr_grad.
extern "C" int gpn_kernel_expr_grad_batched(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                            int batch, const double* theta, int64_t sTheta, int target, const double* X, int64_t sX,
                                            int64_t n, int d, const double* G, int64_t ldg, int64_t sG, const double* at, int64_t ldat,
                                            int64_t sAt, int dy, int want_trace, double* work, double* out) {
  if (d <= 0) return -13;
  int rc = check_program(terms, nterms, group_start, ngroups, d, true);
  if (rc != GPN_OK) return rc;
  if (batch < 1 || batch > 65535) return -6;
  if (!theta) return -7;
  if (target < 0 || target >= nterms) return -9;
  if (!X) return -10;
  if (n <= 0) return -12;
  if (!G) return -14;
  if (ldg < n) return -15;
  if (!at || ldat < n || dy <= 0) return -17;
  if (!work) return -22;
  if (!out) return -23;
  ExprGradArgs a;
  a.X = X; a.X2 = X; a.theta = theta; a.G = G; a.ldg = ldg; a.at = at; a.ldat = ldat; a.partial = work;
  a.n = (int)n; a.m = (int)n; a.d = d; a.dy = dy; a.symmetric = 1; a.lml = 1; a.target = target; a.want_trace = want_trace ? 1 : 0;
  a.group = 0;
  for (int g = 0; g < ngroups; ++g) if (target >= group_start[g] && target < group_start[g + 1]) a.group = g;
  const gpn_expr_term& T = terms[target];
  const int nparam = T.type == GPN_TERM_STATIONARY ? 1 + T.nls : (T.type == GPN_TERM_LINEAR ? T.nvar : 1);
  a.nout = nparam + a.want_trace;
  const int64_t tm = (n + ET - 1) / ET;
  a.tiles_n = (int)tm;
  const int64_t nblocks = tm * (tm + 1) / 2;
  a.sX = sX; a.sTheta = sTheta; a.sG = sG; a.sAt = sAt;
  a.sPartial = gpn_kernel_expr_grad_work_bytes(n, n, d, 1) / (int64_t)sizeof(double);
  ExprProgram P;
  fill_program(P, terms, nterms, group_start, ngroups);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rec = -1;
  if (profile_on()) rec = profile_begin(s, batch * 8.0 * 0.5 * n * (n + 1.0), PROF_GRAD);
  hipLaunchKernelGGL(kexpr_grad_kernel, dim3((unsigned)nblocks, (unsigned)batch), dim3(256), 0, s, a, P);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  hipLaunchKernelGGL(kexpr_reduce_kernel, dim3((unsigned)a.nout, (unsigned)batch), dim3(256), 0, s, work, nblocks, a.nout, out, a.sPartial);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}
