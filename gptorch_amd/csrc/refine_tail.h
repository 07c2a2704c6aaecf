// Shared by refine.hip and kexpr.hip: double-double helpers and the reduction tail of the residual kernels of the refinement step
//  -- a 64 x 64 tile of Kyy in the threads' 4 x 4 micro-tiles (rows ty*4+a, columns {2tx, 2tx+1, 32+2tx, 33+2tx}: the
// assembly kernels' layout) times a_hat, into the tile's row partial and, for an off-diagonal tile, its mirror column partial.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpn {

// ---- double-double accumulation (explicitly rounded intrinsics: never contracted into FMAs by the compiler) ----
struct dd { double hi, lo; };
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
  s = __dadd_rn(a, b);
  const double bb = __dsub_rn(s, a);
  e = __dadd_rn(__dsub_rn(a, __dsub_rn(s, bb)), __dsub_rn(b, bb));
}
__device__ __forceinline__ void dd_fma(dd& acc, double a, double b) {        // acc += a * b, error-free product and sum
  const double p = __dmul_rn(a, b);
  const double ep = __fma_rn(a, b, -p);
  double s, es;
  two_sum(acc.hi, p, s, es);
  acc.hi = s;
  acc.lo = __dadd_rn(acc.lo, __dadd_rn(es, ep));
}
__device__ __forceinline__ void dd_add(dd& acc, const dd x) {
  double s, e;
  two_sum(acc.hi, x.hi, s, e);
  acc.hi = s;
  acc.lo = __dadd_rn(acc.lo, __dadd_rn(e, x.lo));
}

constexpr int RT = 64;        // tile edge of the residual kernels (= kmat.hip's KT, kexpr.hip's ET)
constexpr int RDY = 2;        // right-hand sides per pass when dy > 1

struct RefineTailArgs {
  const double* a;       // [dy][lds]
  double* prow;          // [tiles of the launch][dy][64][2]
  double* pcol;
  int64_t lds;
  int n, dy;
};

// v: the tile's entries (zero outside the matrix); q: the tile's slot in prow / pcol; called by all 256 threads (barriers inside)
template <int NRHS>
__device__ __forceinline__ void refine_tile_tail(const RefineTailArgs& p, const double (&v)[4][4], bool offdiag, int q, int i0, int j0,
                                                 double (*arow)[RT], double (*acol)[RT], double (*red)[RT][17]) {
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  for (int c0 = 0; c0 < p.dy; c0 += NRHS) {
    const int nc = min(NRHS, p.dy - c0);
    __syncthreads();
    if (tid < 2 * RT) {
      const int pt = tid & (RT - 1);
      const int idx = (tid < RT ? i0 : j0) + pt;
      for (int c = 0; c < nc; ++c) (tid < RT ? arow : acol)[c][pt] = idx < p.n ? p.a[(int64_t)(c0 + c) * p.lds + idx] : 0.0;
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
      // rows of I: sum over my 4 columns of v * a_J[col], then over the 16 tx lanes (fixed order)
      dd r[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        r[a] = dd{0.0, 0.0};
#pragma unroll
        for (int b = 0; b < 4; ++b) dd_fma(r[a], v[a][b], acol[c][(b >> 1) * 32 + tx * 2 + (b & 1)]);
        red[0][ty * 4 + a][tx] = r[a].hi;
        red[1][ty * 4 + a][tx] = r[a].lo;
      }
      __syncthreads();
      if (tid < RT) {
        dd sum{red[0][tid][0], red[1][tid][0]};
        for (int k = 1; k < 16; ++k) dd_add(sum, dd{red[0][tid][k], red[1][tid][k]});
        double* out = p.prow + (((int64_t)q * p.dy + (c0 + c)) * RT + tid) * 2;
        out[0] = sum.hi;
        out[1] = sum.lo;
      }
      __syncthreads();
      if (offdiag) {
        // columns of J (the mirror entries): sum over my 4 rows of v * a_I[row], then over the 16 ty lanes
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          dd cc{0.0, 0.0};
#pragma unroll
          for (int a = 0; a < 4; ++a) dd_fma(cc, v[a][b], arow[c][ty * 4 + a]);
          const int cl = (b >> 1) * 32 + tx * 2 + (b & 1);
          red[0][cl][ty] = cc.hi;
          red[1][cl][ty] = cc.lo;
        }
        __syncthreads();
        if (tid < RT) {
          dd sum{red[0][tid][0], red[1][tid][0]};
          for (int k = 1; k < 16; ++k) dd_add(sum, dd{red[0][tid][k], red[1][tid][k]});
          double* out = p.pcol + (((int64_t)q * p.dy + (c0 + c)) * RT + tid) * 2;
          out[0] = sum.hi;
          out[1] = sum.lo;
        }
        __syncthreads();
      }
    }
  }
}

}  // namespace gpn
