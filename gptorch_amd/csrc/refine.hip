// gpn_lml_refine: one step of iterative refinement of the quadratic form y^T Kyy^-1 y of
// GPR.log_likelihood (gpr.py:61-67: alpha = L^-1 (y - m), LML = -1/2 |alpha|^2 - ...).
//
// Why.  north_star asks for the LML within 1e-8 ABSOLUTE of the reference's CPU value.  At BASELINE config 3
// (N = 32768, |LML| = 1.5e5) that is 7e-14 relative -- the rounding level of any fp64 factorisation: with
// L L^T = Kyy + E the computed |alpha|^2 is y^T (Kyy + E)^-1 y = exact - a^T E a, |a|^2 = 9.5e6, and both MKL's
// factor (3.4e-9 off) and ours (7.6e-9 off, the other way) are a few 1e-9 away from the exact value
// (tests/golden/make_c3_extended.py).  One refinement step removes the factor's error from the value:
//     a_hat = L^-T alpha                       (back-substitution: any approximate solution will do)
//     r     = (y - m) - Kyy a_hat              (Kyy RE-COMPUTED from the points, entry for entry the
//                                               arithmetic of the assembly; accumulated in double-double)
//     y^T Kyy^-1 y = y^T a_hat + a_hat^T r + r^T Kyy^-1 r,   the last term is O(|E|^2) and dropped
// so the result depends on the factorisation only to second order (and no longer on its summation order).
// Cost: a pass over L (HBM) in N/128 dependent launches + one pass of kernel evaluations (vector ALU), about 3 %
// of an evaluation at N = 32768; the shell applies it from GPN_REFINE_MIN_N rows on (gptorch_amd/_ops.py).
//
// Kernels: backsub_step_kernel (one launch per 128-column block, right-looking: s_j -= L_kj^T a_k for all columns left
// of block k, and the workgroup that owns block k-1 goes on to a_{k-1} = W_{k-1}^T s_{k-1} with the stored leaf inverse),
// refine_resid_sym_kernel<KIND> (the 64 x 64 tiles of Kyy on / below the diagonal times a_hat, each used for its rows and for its
// mirror columns; error-free products and sums: Ogita-Rump-Oishi Dot2) + refine_gather_kernel (per-row sums of the tile partials),
// refine_finish_kernel (r, the two dot products and the LML, one workgroup, double-double throughout).
#include <algorithm>
#include "gpn_common.h"
#include "kernel_fn.h"
#include "refine_tail.h"

namespace gpn {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int RDC = 16;       // coordinates staged per pass (= kmat.hip's DC: the same summation order over d)
constexpr int BS_COLS = 128;  // columns per workgroup of the back-substitution update = one leaf block

// s <- alpha^T (the first n entries of the factor buffer's extra rows; what is right of them is the corner that
// accumulated -alpha alpha^T), zero padded to lds
__global__ void refine_init_kernel(const double* A, int64_t lda, int64_t n, int dy, double* s, int64_t lds) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lds) return;
  for (int c = 0; c < dy; ++c) s[(int64_t)c * lds + i] = i < n ? A[(n + c) * lda + i] : 0.0;
}

// One step of the back-substitution L^T a = alpha, right-looking over 128-wide blocks, ONE launch per block and nothing
// else on the chain:
//   * every workgroup owns one leaf block j < k of columns and applies  s_j -= L[block k rows, block j cols]^T a_k
//     (a_k was stored by the launch before; all 32 x 16-byte loads of a thread are issued before the first use, so the
//     launch costs one memory round trip);
//   * workgroup 0 owns block k-1, whose s is FINAL after this update (blocks k+1 ... nb-1 were applied by the earlier
//     launches): it goes on to a_{k-1} = W_{k-1}^T s_{k-1} with the stored leaf inverse W = L_{k-1,k-1}^-1 and stores it
//     for the next launch -- the 128 x 128 product never gets a launch (or a redundant copy per workgroup) of its own.
// k = nb (first launch, one workgroup): no update, a_{nb-1} from alpha.  NRHS right-hand sides c0 .. c0+NRHS-1.
template <int NRHS>
__global__ __launch_bounds__(256) void backsub_step_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ winv,
                                                           int k, int nb, int64_t n, int dy, int c0, double* __restrict__ s,
                                                           double* __restrict__ a, int64_t lds) {
  __shared__ double ak[NRHS][LEAF];
  __shared__ double part[4][NRHS][BS_COLS];
  __shared__ double sj[NRHS][LEAF];
  const int t = threadIdx.x;
  const int nc = min(NRHS, dy - c0);
  const int jb = k - 1 - (int)blockIdx.x;                  // my column block (workgroup 0: block k-1)
  const int64_t col0 = (int64_t)jb * LEAF;
  const int cp = t & 63, g = t >> 6;                       // column pair (2 cp, 2 cp + 1), row group g: rows 32 g .. 32 g + 31
  double upd0[NRHS], upd1[NRHS];
#pragma unroll
  for (int c = 0; c < NRHS; ++c) upd0[c] = upd1[c] = 0.0;
  if (k < nb) {
    const int64_t r0 = (int64_t)k * LEAF;
    const int kb = (int)min((int64_t)LEAF, n - r0);        // ragged last block: the rows behind it are the extra rows, not L
    const d2* Lp = reinterpret_cast<const d2*>(A + (r0 + g * 32) * lda + col0 + 2 * cp);
    d2 l[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) l[rr] = (g * 32 + rr < kb) ? Lp[(int64_t)rr * (lda / 2)] : d2{0.0, 0.0};
    if (t < LEAF)
      for (int c = 0; c < nc; ++c) ak[c][t] = a[(int64_t)(c0 + c) * lds + r0 + t];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 32; ++rr)
#pragma unroll
      for (int c = 0; c < NRHS; ++c) {
        const double av = ak[c][g * 32 + rr];
        upd0[c] = fma(l[rr].x, av, upd0[c]);
        upd1[c] = fma(l[rr].y, av, upd1[c]);
      }
  }
#pragma unroll
  for (int c = 0; c < NRHS; ++c) {
    part[g][c][2 * cp] = upd0[c];
    part[g][c][2 * cp + 1] = upd1[c];
  }
  __syncthreads();
  if (t < LEAF)
    for (int c = 0; c < nc; ++c) {
      const int64_t idx = (int64_t)(c0 + c) * lds + col0 + t;
      const double v = s[idx] - ((part[0][c][t] + part[1][c][t]) + (part[2][c][t] + part[3][c][t]));
      if (k < nb) s[idx] = v;
      sj[c][t] = v;
    }
  if (blockIdx.x != 0) return;
  __syncthreads();
  // a_j[i] = sum_{r >= i} W[r][i] s_j[r]  (W lower triangular; rows past a ragged block's end are padding)
  const double* W = winv + (int64_t)jb * LEAF * LEAF;
  const int jbn = (int)min((int64_t)LEAF, n - col0);
  d2 w[32];
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const int r = g * 32 + rr;
    w[rr] = (r < jbn && r >= 2 * cp) ? *reinterpret_cast<const d2*>(W + (int64_t)r * LEAF + 2 * cp) : d2{0.0, 0.0};
  }
  double acc0[NRHS], acc1[NRHS];
#pragma unroll
  for (int c = 0; c < NRHS; ++c) acc0[c] = acc1[c] = 0.0;
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const int r = g * 32 + rr;
#pragma unroll
    for (int c = 0; c < NRHS; ++c) {
      const double sv = sj[c][r];
      acc0[c] = fma(w[rr].x, sv, acc0[c]);
      acc1[c] = (r >= 2 * cp + 1) ? fma(w[rr].y, sv, acc1[c]) : acc1[c];   // (W[2cp][2cp+1] is above the diagonal)
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NRHS; ++c) {
    part[g][c][2 * cp] = acc0[c];
    part[g][c][2 * cp + 1] = acc1[c];
  }
  __syncthreads();
  if (t < LEAF)
    for (int c = 0; c < nc; ++c)
      a[(int64_t)(c0 + c) * lds + col0 + t] = t < jbn ? (part[0][c][t] + part[1][c][t]) + (part[2][c][t] + part[3][c][t]) : 0.0;
}

// The whole back-substitution as ONE launch (round 5): the chain of nb dependent steps above costs a kernel launch each
// (8.1 us x 256 = 2.1 ms at N = 32768, for a pass over L that HBM serves in under 1 ms).  Here every column block has its
// workgroup for the whole solve, and the steps are ordered through the DATA instead of launches:
//   workgroup b owns column block j = nb - 1 - b (so that a workgroup only ever waits for workgroups with a SMALLER index:
//   dispatched before it, hence resident or finished -- no deadlock however many fit the chip at once);
//   a_hat starts as a SENTINEL bit pattern (a NaN payload no arithmetic produces); the owner of block k publishes a_k with
//   agent-scope stores, and a reader's 128 threads each spin on THEIR entry until it is no sentinel -- no flag, no fence, no
//   second round trip: the values validate themselves (first version: flag + release fence + reload of a_k, 12 us per step);
//   for k = nb - 1 ... j + 1:  [L(k, j) is already in registers]  wait for a_k, s_j -= L(k, j)^T a_k  (s_j stays in LDS),
//       and the loads of L(k - 1, j) go out before the next wait;
//   then a_j = W_j^T s_j with the stored leaf inverse (its loads in flight under the last reduction), published.
// The arithmetic per block -- four row groups, their partial sums added as (p0 + p1) + (p2 + p3), the blocks applied in
// descending k -- is backsub_step_kernel's: a_hat is bit-identical.  A bounded spin that runs out publishes NaN: the
// evaluation ends with a NaN value instead of hanging.
constexpr unsigned BS_SPIN_LIMIT = 1u << 26;
constexpr unsigned long long BS_SENTINEL = 0x7FF4DEADBEEF5A5Aull;        // a signalling-NaN payload: never a computed value
__global__ void backsub_sentinel_kernel(unsigned long long* a, int64_t count, int* tickets, int ntickets) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) a[i] = BS_SENTINEL;
  if (i < ntickets) tickets[i] = 0;
}
template <int NRHS>
__global__ __launch_bounds__(256) void backsub_persistent_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ winv,
                                                                 int nb, int64_t n, int dy, int c0, double* a, int64_t lds, int* ticket) {
  __shared__ double ak[NRHS][LEAF];
  __shared__ double part[4][NRHS][BS_COLS];
  __shared__ double sj[NRHS][LEAF];
  __shared__ int failed, my_turn;
  const int t = threadIdx.x;
  const int nc = min(NRHS, dy - c0);
  // column blocks are handed out in the order workgroups START (a ticket, not blockIdx): whoever a workgroup waits for
  // drew its ticket earlier, so it is running or done whatever order the hardware dispatches in
  if (t == 0) my_turn = atomicAdd(ticket, 1);
  __syncthreads();
  const int jb = nb - 1 - my_turn;
  const int64_t col0 = (int64_t)jb * LEAF;
  const int cp = t & 63, g = t >> 6;
  const int jbn = (int)min((int64_t)LEAF, n - col0);
  const double* W = winv + (int64_t)jb * LEAF * LEAF;
  if (t < LEAF)
    for (int c = 0; c < nc; ++c) sj[c][t] = col0 + t < n ? A[(n + c0 + c) * lda + col0 + t] : 0.0;   // s_j <- alpha_j (the factor's extra rows)
  if (t == 0) failed = 0;
  d2 l[32];
  auto fetch = [&](int k) {                                   // rows of block k, my columns
    const int64_t r0 = (int64_t)k * LEAF;
    const int kb = (int)min((int64_t)LEAF, n - r0);
    const d2* Lp = reinterpret_cast<const d2*>(A + (r0 + g * 32) * lda + col0 + 2 * cp);
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) l[rr] = (g * 32 + rr < kb) ? Lp[(int64_t)rr * (lda / 2)] : d2{0.0, 0.0};
  };
  // the leaf inverse of my block takes the tile registers once the last tile of L has been used: its loads are in flight
  // under the last reduction, like a tile's
  auto fetch_w = [&]() {
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
      const int r = g * 32 + rr;
      l[rr] = (r < jbn && r >= 2 * cp) ? *reinterpret_cast<const d2*>(W + (int64_t)r * LEAF + 2 * cp) : d2{0.0, 0.0};
    }
  };
  if (jb + 1 < nb) fetch(nb - 1);
  else fetch_w();
  __syncthreads();
  for (int k = nb - 1; k > jb; --k) {
    if (t < LEAF) {
      for (int c = 0; c < nc; ++c) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a + (int64_t)(c0 + c) * lds + (int64_t)k * LEAF + t);
        unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (v == BS_SENTINEL) {
          if (++spins > BS_SPIN_LIMIT) { failed = 1; v = 0x7FF8000000000000ull; break; }
          if (k == jb + 1) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(16);      // the chain's next link polls hardest
          v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ak[c][t] = __longlong_as_double((long long)v);
      }
    }
    __syncthreads();
    double upd0[NRHS], upd1[NRHS];
#pragma unroll
    for (int c = 0; c < NRHS; ++c) upd0[c] = upd1[c] = 0.0;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr)
#pragma unroll
      for (int c = 0; c < NRHS; ++c) {
        const double av = ak[c][g * 32 + rr];
        upd0[c] = fma(l[rr].x, av, upd0[c]);
        upd1[c] = fma(l[rr].y, av, upd1[c]);
      }
    if (k - 1 > jb) fetch(k - 1);                             // in flight under the reduction and the next wait
    else fetch_w();
#pragma unroll
    for (int c = 0; c < NRHS; ++c) {
      part[g][c][2 * cp] = upd0[c];
      part[g][c][2 * cp + 1] = upd1[c];
    }
    __syncthreads();
    if (t < LEAF)
      for (int c = 0; c < nc; ++c)
        sj[c][t] = sj[c][t] - ((part[0][c][t] + part[1][c][t]) + (part[2][c][t] + part[3][c][t]));
    __syncthreads();
  }
  // a_j[i] = sum_{r >= i} W[r][i] s_j[r]
  double acc0[NRHS], acc1[NRHS];
#pragma unroll
  for (int c = 0; c < NRHS; ++c) acc0[c] = acc1[c] = 0.0;
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const int r = g * 32 + rr;
#pragma unroll
    for (int c = 0; c < NRHS; ++c) {
      const double sv = sj[c][r];
      acc0[c] = fma(l[rr].x, sv, acc0[c]);
      acc1[c] = (r >= 2 * cp + 1) ? fma(l[rr].y, sv, acc1[c]) : acc1[c];
    }
  }
#pragma unroll
  for (int c = 0; c < NRHS; ++c) {
    part[g][c][2 * cp] = acc0[c];
    part[g][c][2 * cp + 1] = acc1[c];
  }
  __syncthreads();
  const bool bad = failed != 0;
  if (t < LEAF)
    for (int c = 0; c < nc; ++c) {
      const double v = t < jbn ? (part[0][c][t] + part[1][c][t]) + (part[2][c][t] + part[3][c][t]) : 0.0;
      __hip_atomic_store(a + (int64_t)(c0 + c) * lds + col0 + t, bad ? __builtin_nan("") : v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Kyy a_hat in double-double, Kyy re-computed from the points: ONE workgroup per 64 x 64 tile on or below the diagonal (the
// assembly's own enumeration; tile structure, staging and the order of the sum over coordinates are kmat_kernel's, so that
// every entry is bit for bit the one the assembly wrote into the factor buffer).  Tile (I, J), I > J, is evaluated once and used twice: its rows times a_J go to the row partial of tile
// row I, its columns times a_I (the mirror entries K[j, i] = K[i, j]) to the column partial of tile row J; a diagonal
// tile is computed whole and only feeds its row partial.  prow / pcol[tile q][c][64] double-double; refine_gather_kernel
// adds them up per row in a fixed order.  (The first version evaluated all N^2 entries, one row strip per workgroup: 3.7 ms at
// C3 against the 1.7 ms of the assembly it mirrors.)
struct RefineSymArgs {
  const double* X;
  const double* variance;
  const double* ls;
  const double* noise;
  const double* a;       // [dy][lds]
  double* prow;          // [ntri][dy][64][2]
  double* pcol;          // [ntri][dy][64][2]
  int64_t lds;
  int n, d, nls, dy;
  int q_off;             // first tile of this launch in the enumeration of the lower tiles (a rank's share: gpn_refine_resid_part)
};

template <int KIND, int NRHS>
__global__ __launch_bounds__(256) void refine_resid_sym_kernel(RefineSymArgs p) {
  __shared__ __attribute__((aligned(16))) double xs[RDC][RT];
  __shared__ __attribute__((aligned(16))) double ys[RDC][RT];
  __shared__ double inv_ell[RDC];
  __shared__ double arow[NRHS][RT], acol[NRHS][RT];
  __shared__ double red[2][RT][17];
  const int q = blockIdx.x;                  // slot of the partials
  const int qg = q + p.q_off;                // the tile
  int ti = (int)((sqrt(8.0 * (double)qg + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > qg) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= qg) ++ti;
  const int tj = qg - ti * (ti + 1) / 2;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = ti * RT, j0 = tj * RT;
  const double var = p.variance[0], noise = p.noise[0];
  const bool offdiag = ti != tj;

  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int d0 = 0; d0 < p.d; d0 += RDC) {
    if (tid < RDC) {
      const int dd_ = d0 + tid;
      inv_ell[tid] = dd_ < p.d ? 1.0 / p.ls[p.nls == 1 ? 0 : dd_] : 0.0;
    }
    __syncthreads();
    {
      const int pt = tid >> 2, c4 = (tid & 3) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd_ = d0 + c4 + c;
        double vx = 0.0, vy = 0.0;
        if (dd_ < p.d) {
          const double ie = inv_ell[c4 + c];
          if (i0 + pt < p.n) vx = p.X[(int64_t)(i0 + pt) * p.d + dd_] * ie;
          if (j0 + pt < p.n) vy = p.X[(int64_t)(j0 + pt) * p.d + dd_] * ie;
        }
        xs[c4 + c][pt] = vx;
        ys[c4 + c][pt] = vy;
      }
    }
    __syncthreads();
    const int dmax = min(RDC, p.d - d0);
    for (int dd_ = 0; dd_ < dmax; ++dd_) {
      const d2 xa = *reinterpret_cast<const d2*>(&xs[dd_][ty * 4]);
      const d2 xb = *reinterpret_cast<const d2*>(&xs[dd_][ty * 4 + 2]);
      const d2 ya = *reinterpret_cast<const d2*>(&ys[dd_][tx * 2]);
      const d2 yb = *reinterpret_cast<const d2*>(&ys[dd_][32 + tx * 2]);
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
  // entries (the assembly's arithmetic: kernel_fn.h), noise on the diagonal, nothing outside the matrix
  double v[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = i0 + ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = j0 + (b >> 1) * 32 + tx * 2 + (b & 1);
      double e = kernel_of_r2<KIND>(acc[a][b], var);
      if (row == col) e += noise;
      v[a][b] = (row < p.n && col < p.n) ? e : 0.0;
    }
  }
  RefineTailArgs tp;
  tp.a = p.a; tp.prow = p.prow; tp.pcol = p.pcol; tp.lds = p.lds; tp.n = p.n; tp.dy = p.dy;
  refine_tile_tail<NRHS>(tp, v, offdiag, q, i0, j0, arow, acol, red);
}

// The same pass for a Kyy that exists as a dense matrix (the dense-K fall-back of GPR: kernels the expression builder cannot take):
// the tile's entries are READ (lower tiles of the symmetric matrix; diag_add = the jitter the ladder settled on) instead of computed.
struct DenseResidArgs {
  const double* K;
  int64_t ldk;
  double diag_add;
  RefineTailArgs tail;
  int n;
};

template <int NRHS>
__global__ __launch_bounds__(256) void refine_resid_dense_kernel(DenseResidArgs p) {
  __shared__ double arow[NRHS][RT], acol[NRHS][RT];
  __shared__ double red[2][RT][17];
  const int q = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > q) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
  const int tj = q - ti * (ti + 1) / 2;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int i0 = ti * RT, j0 = tj * RT;
  double v[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = i0 + ty * 4 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = j0 + (b >> 1) * 32 + tx * 2 + (b & 1);
      double e = 0.0;
      if (row < p.n && col < p.n) {
        e = p.K[(int64_t)row * p.ldk + col];
        if (row == col) e += p.diag_add;
      }
      v[a][b] = e;
    }
  }
  refine_tile_tail<NRHS>(p.tail, v, ti != tj, q, i0, j0, arow, acol, red);
}

// ka[c][i] = sum_{J <= T} prow[(T, J)][c][r] + sum_{I > T} pcol[(I, T)][c][r]   (T = i / 64, r = i % 64), double-double,
// fixed order; 4 lanes per row take every 4th partial and are combined through LDS
__global__ __launch_bounds__(256) void refine_gather_kernel(const double* prow, const double* pcol, int nt, int dy, int64_t lds,
                                                            int64_t n, double* ka, int64_t q0 = 0, int64_t q1 = (int64_t)1 << 62) {
  __shared__ double sh[2][4][RT];
  const int T = blockIdx.x, c = blockIdx.y;
  const int r = threadIdx.x & (RT - 1), lane4 = threadIdx.x >> 6;
  dd acc{0.0, 0.0};
  const int total = nt;                       // T + 1 row partials + (nt - 1 - T) column partials
  for (int k = lane4; k < total; k += 4) {
    const double* src;
    const int64_t qq = k <= T ? (int64_t)T * (T + 1) / 2 + k : (int64_t)k * (k + 1) / 2 + T;
    if (qq < q0 || qq >= q1) continue;       // not in this launch's share of the tiles
    if (k <= T) src = prow + (((qq - q0) * dy + c) * RT + r) * 2;
    else src = pcol + (((qq - q0) * dy + c) * RT + r) * 2;
    dd_add(acc, dd{src[0], src[1]});
  }
  sh[0][lane4][r] = acc.hi;
  sh[1][lane4][r] = acc.lo;
  __syncthreads();
  if (lane4 == 0) {
    dd sum{sh[0][0][r], sh[1][0][r]};
    for (int k = 1; k < 4; ++k) dd_add(sum, dd{sh[0][k][r], sh[1][k][r]});
    const int64_t i = (int64_t)T * RT + r;
    if (i < n) {
      double* out = ka + ((int64_t)c * lds + i) * 2;
      out[0] = sum.hi;
      out[1] = sum.lo;
    }
  }
}

// r = (y - m) - sum_seg partial;  quad = sum_c sum_i a_i ((y - m)_i + r_i);  out3[1] = quad, out3[2] = LML (gpr.py:63-67)
__global__ __launch_bounds__(1024) void refine_finish_kernel(const double* Y, const double* M, const double* a, const double* partial,
                                                             int64_t n, int dy, int nseg, int64_t lds, double* out3, double* resid_norm) {
  constexpr int NT = 1024;
  __shared__ double rh[NT], rl[NT], rn[NT];
  const int tid = threadIdx.x;
  dd q{0.0, 0.0};
  double rmax = 0.0;
  for (int c = 0; c < dy; ++c)
    for (int64_t i = tid; i < n; i += NT) {
      dd ka{0.0, 0.0};
      for (int sgm = 0; sgm < nseg; ++sgm) {
        const double* pp = partial + (((int64_t)sgm * dy + c) * lds + i) * 2;
        dd_add(ka, dd{pp[0], pp[1]});
      }
      double y = Y[i * dy + c];
      if (M) y -= M[i * dy + c];                       // (one rounding, as in the extra rows the factorisation consumed)
      // t = y + r = 2 y - Kyy a, as a double-double
      dd tt{0.0, 0.0};
      dd_add(tt, dd{-ka.hi, -ka.lo});
      dd_add(tt, dd{y, 0.0});
      rmax = fmax(rmax, fabs(tt.hi));                  // |r_i| (diagnostic)
      dd_add(tt, dd{y, 0.0});
      const double ai = a[(int64_t)c * lds + i];
      dd_fma(q, ai, tt.hi);
      q.lo = fma(ai, tt.lo, q.lo);
    }
  rh[tid] = q.hi; rl[tid] = q.lo; rn[tid] = rmax;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (tid < s) {
      dd x{rh[tid], rl[tid]};
      dd_add(x, dd{rh[tid + s], rl[tid + s]});
      rh[tid] = x.hi; rl[tid] = x.lo;
      rn[tid] = fmax(rn[tid], rn[tid + s]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double quad = rh[0] + rl[0];
    const double logdet = out3[0];
    out3[1] = quad;
    out3[2] = -0.5 * quad - (double)dy * logdet - 0.5 * (double)dy * (double)n * 1.8378770664093454836;
    if (resid_norm) resid_norm[0] = rn[0];
  }
}

// ---- pieces for a factor that is spread over several GPUs (gptorch_amd/dist.py BlockCyclicGP._refine) ----------------------
// c[cc][col] += sum_r L[r][col] a[cc][r] for a rows x cols block of a row-major matrix (a tile row of the local factor times
// the back-substituted block it belongs to): one thread per column, the rows dealt over blockIdx.y chunks whose partial sums
// a second kernel adds in a fixed order (no atomics: the result does not depend on the schedule).
template <int NRHS>
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(const double* __restrict__ L, int64_t ld, int rows, int cols,
                                                             const double* __restrict__ a, int64_t lda, int nc, int rows_per_chunk,
                                                             double* __restrict__ part) {
  __shared__ double as[NRHS][256];
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  double acc[NRHS];
#pragma unroll
  for (int c = 0; c < NRHS; ++c) acc[c] = 0.0;
  for (int rb = r0; rb < r1; rb += 256) {
    __syncthreads();
    for (int c = 0; c < nc; ++c) as[c][threadIdx.x] = rb + (int)threadIdx.x < r1 ? a[(int64_t)c * lda + rb + threadIdx.x] : 0.0;
    __syncthreads();
    if (col < cols) {
      const int rn = min(256, r1 - rb);
      const double* Lp = L + (int64_t)rb * ld + col;
      for (int r = 0; r < rn; ++r) {
        const double l = Lp[(int64_t)r * ld];
#pragma unroll
        for (int c = 0; c < NRHS; ++c) acc[c] = fma(l, as[c][r], acc[c]);
      }
    }
  }
  if (col < cols)
    for (int c = 0; c < nc; ++c) part[((int64_t)blockIdx.y * nc + c) * cols + col] = acc[c];
}

__global__ __launch_bounds__(256) void gemv_t_reduce_kernel(const double* part, int nchunk, int nc, int cols, double* c, int64_t ldc) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  for (int cc = 0; cc < nc; ++cc) {
    double sum = 0.0;
    for (int k = 0; k < nchunk; ++k) sum += part[((int64_t)k * nc + cc) * cols + col];
    c[(int64_t)cc * ldc + col] += sum;
  }
}

static int gemv_t_chunks(int64_t rows, int64_t cols) {
  const int64_t col_wgs = (cols + 255) / 256;
  int64_t nchunk = (1024 + col_wgs - 1) / col_wgs;
  nchunk = std::max<int64_t>(1, std::min<int64_t>(nchunk, (rows + 31) / 32));
  return (int)nchunk;
}

GPN_SWITCH int g_backsub_persistent = 1;   // 0 = one launch per 128-column block (backsub_step_kernel; A/B and bit-identity test, tools' build)

struct RefineLayout { int64_t lds, s, a, partial, norm, prow, pcol, total; int nseg, tiles_per_seg; };
static RefineLayout refine_layout(int64_t n, int dy) {
  RefineLayout L;
  L.lds = round_up(n, LEAF);
  const int64_t ntile = (n + RT - 1) / RT;
  L.nseg = 1;                 // Kyy a_hat per row (double-double), written by refine_gather_kernel
  L.tiles_per_seg = (int)ntile;
  L.s = 0;
  L.a = L.s + (int64_t)dy * L.lds;
  L.partial = L.a + (int64_t)dy * L.lds;
  L.norm = L.partial + (int64_t)L.nseg * dy * L.lds * 2;
  // symmetric pass: one row partial and one column partial per lower tile
  const int64_t ntri = ntile * (ntile + 1) / 2;
  L.prow = L.norm + 8;
  L.pcol = L.prow + ntri * dy * RT * 2;
  L.total = L.pcol + ntri * dy * RT * 2;
  return L;
}

}  // namespace gpn

using namespace gpn;

GPN_DEBUG_ONLY(extern "C" int gpn_debug_set_backsub_persistent(int on) { gpn::g_backsub_persistent = on; return GPN_OK; })

extern "C" int64_t gpn_lml_refine_work_bytes(int64_t n, int dy) {
  if (n < 0 || dy <= 0) return 0;
  return refine_layout(n, dy).total * (int64_t)sizeof(double);
}

// a_hat = L^-T alpha into work (s, a), from the factor's extra rows
static int refine_backsub(hipStream_t s, const double* A, int64_t lda, const double* winv, int64_t n, int dy, double* work, const RefineLayout& L) {
  double* sv = work + L.s;
  double* av = work + L.a;
  const int nb_ = (int)((n + LEAF - 1) / LEAF);
  const int passes = dy == 1 ? 1 : (dy + RDY - 1) / RDY;
  if (g_backsub_persistent) {
    // one launch per group of right-hand sides; a_hat starts as the sentinel its readers spin on
    const int64_t cnt = (int64_t)dy * L.lds;
    int* tickets = reinterpret_cast<int*>(sv);               // (this path holds s_j in LDS: the s buffer carries one ticket counter per pass)
    hipLaunchKernelGGL(backsub_sentinel_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, reinterpret_cast<unsigned long long*>(av), cnt,
                       tickets, passes);
    int pass = 0;
    for (int c0 = 0; c0 < dy; c0 += (dy == 1 ? 1 : RDY), ++pass) {
      if (dy == 1) hipLaunchKernelGGL(backsub_persistent_kernel<1>, dim3((unsigned)nb_), dim3(256), 0, s, A, lda, winv, nb_, n, dy, c0, av, L.lds, tickets + pass);
      else hipLaunchKernelGGL(backsub_persistent_kernel<RDY>, dim3((unsigned)nb_), dim3(256), 0, s, A, lda, winv, nb_, n, dy, c0, av, L.lds, tickets + pass);
    }
    GPN_LAUNCH_CHECK();
    return GPN_OK;
  }
  hipLaunchKernelGGL(refine_init_kernel, dim3((unsigned)((L.lds + 255) / 256)), dim3(256), 0, s, A, lda, n, dy, sv, L.lds);
  GPN_LAUNCH_CHECK();
  const int nb = (int)((n + LEAF - 1) / LEAF);
  for (int c0 = 0; c0 < dy; c0 += (dy == 1 ? 1 : RDY))
    for (int k = nb; k >= 1; --k) {                 // launch k: update blocks < k by a_k (k < nb), then a_{k-1}
      const unsigned wgs = (unsigned)(k == nb ? 1 : k);             // the first launch only forms a_{nb-1}: one workgroup
      if (dy == 1) hipLaunchKernelGGL(backsub_step_kernel<1>, dim3(wgs), dim3(256), 0, s, A, lda, winv, k, nb, n, dy, c0, sv, av, L.lds);
      else hipLaunchKernelGGL(backsub_step_kernel<RDY>, dim3(wgs), dim3(256), 0, s, A, lda, winv, k, nb, n, dy, c0, sv, av, L.lds);
    }
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// Kyy a per row from the tile partials (double-double), then r, the dot products and the LML
static int refine_gather_finish(hipStream_t s, const double* Y, const double* M, int64_t n, int dy, double* work, const RefineLayout& L, double* out3) {
  const int64_t ntile = (n + RT - 1) / RT;
  hipLaunchKernelGGL(refine_gather_kernel, dim3((unsigned)ntile, (unsigned)dy), dim3(256), 0, s, work + L.prow, work + L.pcol, (int)ntile, dy,
                     L.lds, n, work + L.partial);
  GPN_LAUNCH_CHECK();
  hipLaunchKernelGGL(refine_finish_kernel, dim3(1), dim3(1024), 0, s, Y, M, work + L.a, work + L.partial, n, dy, 1, L.lds, out3, work + L.norm);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_lml_refine(void* stream, int kind, const double* X, int64_t n, int d,
                              const double* Y, const double* M, int dy,
                              const double* variance, const double* length_scales, int nls, const double* noise,
                              const double* A, int64_t lda, const double* winv, double* work, double* out3) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (!X) return -3;
  if (n < 0) return -4;
  if (d <= 0) return -5;
  if (!Y) return -6;
  if (dy <= 0) return -8;
  if (!variance || !length_scales) return -9;
  if (nls != 1 && nls != d) return -11;
  if (!noise) return -12;
  if (!A) return -13;
  if (lda != gpn_factor_ld(n, dy)) return -14;
  if (!winv) return -15;
  if (!work) return -16;
  if (!out3) return -17;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RefineLayout L = refine_layout(n, dy);
  int rc = refine_backsub(s, A, lda, winv, n, dy, work, L);
  if (rc != GPN_OK) return rc;
  const int64_t ntile = (n + RT - 1) / RT;
  RefineSymArgs p;
  p.X = X; p.variance = variance; p.ls = length_scales; p.noise = noise; p.a = work + L.a;
  p.prow = work + L.prow; p.pcol = work + L.pcol; p.lds = L.lds;
  p.n = (int)n; p.d = d; p.nls = nls; p.dy = dy; p.q_off = 0;
  const dim3 grid((unsigned)(ntile * (ntile + 1) / 2));
#define GPN_RESID(KIND)                                                                                 \
  do {                                                                                                  \
    if (dy == 1) hipLaunchKernelGGL((refine_resid_sym_kernel<KIND, 1>), grid, dim3(256), 0, s, p);      \
    else hipLaunchKernelGGL((refine_resid_sym_kernel<KIND, RDY>), grid, dim3(256), 0, s, p);            \
  } while (0)
  switch (kind) {
    case GPN_RBF: GPN_RESID(GPN_RBF); break;
    case GPN_MATERN52: GPN_RESID(GPN_MATERN52); break;
    case GPN_MATERN32: GPN_RESID(GPN_MATERN32); break;
    case GPN_EXP: GPN_RESID(GPN_EXP); break;
    default: GPN_RESID(GPN_PERIODIC); break;
  }
#undef GPN_RESID
  GPN_LAUNCH_CHECK();
  return refine_gather_finish(s, Y, M, n, dy, work, L, out3);
}

// the same step for a covariance EXPRESSION (kexpr.hip: sums of products of native leaves -- the composite kernels of
// kernels.py:286-306): the residual pass evaluates the expression per tile exactly as the fused assembly did
extern "C" int gpn_lml_refine_expr(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                   const double* theta, const double* X, int64_t n, int d, const double* Y, const double* M, int dy,
                                   const double* noise, const double* A, int64_t lda, const double* winv, double* work, double* out3) {
  if (!terms || !group_start) return -2;
  if (!theta) return -6;
  if (!X) return -7;
  if (n < 0) return -8;
  if (d <= 0) return -9;
  if (!Y) return -10;
  if (dy <= 0) return -12;
  if (!noise) return -13;
  if (!A) return -14;
  if (lda != gpn_factor_ld(n, dy)) return -15;
  if (!winv) return -16;
  if (!work) return -17;
  if (!out3) return -18;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RefineLayout L = refine_layout(n, dy);
  int rc = refine_backsub(s, A, lda, winv, n, dy, work, L);
  if (rc != GPN_OK) return rc;
  const int64_t ntile = (n + RT - 1) / RT;
  rc = expr_resid(s, terms, nterms, group_start, ngroups, theta, X, n, d, noise, work + L.a, dy, L.lds, 0, ntile * (ntile + 1) / 2,
                  work + L.prow, work + L.pcol);
  if (rc != GPN_OK) return rc;
  return refine_gather_finish(s, Y, M, n, dy, work, L, out3);
}

// the same step for a Kyy given as a dense symmetric matrix (K [n, n], leading dimension ldk, as it was factorised up to
// diag_add on the diagonal): gptorch_amd/_ops.py DenseLogLik
extern "C" int gpn_lml_refine_dense(void* stream, const double* K, int64_t ldk, double diag_add, int64_t n, const double* Y, const double* M, int dy,
                                    const double* A, int64_t lda, const double* winv, double* work, double* out3) {
  if (!K) return -2;
  if (ldk < n) return -3;
  if (n < 0) return -5;
  if (!Y) return -6;
  if (dy <= 0) return -8;
  if (!A) return -9;
  if (lda != gpn_factor_ld(n, dy)) return -10;
  if (!winv) return -11;
  if (!work) return -12;
  if (!out3) return -13;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RefineLayout L = refine_layout(n, dy);
  int rc = refine_backsub(s, A, lda, winv, n, dy, work, L);
  if (rc != GPN_OK) return rc;
  const int64_t ntile = (n + RT - 1) / RT;
  DenseResidArgs p;
  p.K = K; p.ldk = ldk; p.diag_add = diag_add; p.n = (int)n;
  p.tail.a = work + L.a; p.tail.prow = work + L.prow; p.tail.pcol = work + L.pcol; p.tail.lds = L.lds; p.tail.n = (int)n; p.tail.dy = dy;
  const dim3 grid((unsigned)(ntile * (ntile + 1) / 2));
  if (dy == 1) hipLaunchKernelGGL(refine_resid_dense_kernel<1>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(refine_resid_dense_kernel<RDY>, grid, dim3(256), 0, s, p);
  GPN_LAUNCH_CHECK();
  return refine_gather_finish(s, Y, M, n, dy, work, L, out3);
}

// ---- the same step in pieces, for a factor spread over several GPUs (include/gpnative.h) -----------------------------------
extern "C" int64_t gpn_gemv_t_work_bytes(int64_t rows, int64_t cols, int dy) {
  if (rows <= 0 || cols <= 0 || dy <= 0) return 0;
  // MONOTONE in cols: a caller may size the scratch once for its widest call and reuse it for narrower ones
  // (gpn_dist_lml_refine does).  chunks(c) * c itself is not monotone (chunks = ceil(1024 / ceil(c / 256))); both bounds
  // below are, and each dominates chunks(c') * c' for every c' <= cols.
  const int64_t by_width = 262144 + round_up(cols, 256);            // (1024 / wgs + 1) * 256 wgs
  const int64_t by_rows = ((rows + 31) / 32) * cols;                // chunks <= ceil(rows / 32)
  return std::min(by_width, by_rows) * std::min(dy, RDY) * (int64_t)sizeof(double);
}

extern "C" int gpn_gemv_t_acc(void* stream, const double* L, int64_t ld, int64_t rows, int64_t cols, const double* a, int64_t lda,
                              int dy, double* c, int64_t ldc, double* work) {
  if (!L) return -2;
  if (ld < cols) return -3;
  if (rows < 0) return -4;
  if (cols < 0) return -5;
  if (!a) return -6;
  if (lda < rows) return -7;
  if (dy <= 0) return -8;
  if (!c) return -9;
  if (ldc < cols) return -10;
  if (!work) return -11;
  if (rows == 0 || cols == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nchunk = gemv_t_chunks(rows, cols);
  const int rpc = (int)((rows + nchunk - 1) / nchunk);
  const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)nchunk);
  for (int c0 = 0; c0 < dy; c0 += RDY) {
    const int nc = std::min(RDY, dy - c0);
    if (nc == 1) hipLaunchKernelGGL(gemv_t_partial_kernel<1>, grid, dim3(256), 0, s, L, ld, (int)rows, (int)cols, a + (int64_t)c0 * lda, lda, nc, rpc, work);
    else hipLaunchKernelGGL(gemv_t_partial_kernel<RDY>, grid, dim3(256), 0, s, L, ld, (int)rows, (int)cols, a + (int64_t)c0 * lda, lda, nc, rpc, work);
    hipLaunchKernelGGL(gemv_t_reduce_kernel, dim3(grid.x), dim3(256), 0, s, work, nchunk, nc, (int)cols, c + (int64_t)c0 * ldc, ldc);
  }
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int64_t gpn_refine_resid_part_work_bytes(int dy, int64_t ntiles) {
  if (dy <= 0 || ntiles <= 0) return 0;
  return 2 * ntiles * dy * RT * 2 * (int64_t)sizeof(double);
}

extern "C" int64_t gpn_refine_tile_count(int64_t n) {
  const int64_t ntile = (n + RT - 1) / RT;
  return ntile * (ntile + 1) / 2;
}

extern "C" int gpn_refine_resid_part(void* stream, int kind, const double* X, int64_t n, int d,
                                     const double* variance, const double* length_scales, int nls, const double* noise,
                                     const double* a, int dy, int64_t q0, int64_t q1, double* work, double* ka) {
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -2;
  if (!X) return -3;
  if (n <= 0) return -4;
  if (d <= 0) return -5;
  if (!variance || !length_scales) return -6;
  if (nls != 1 && nls != d) return -8;
  if (!noise) return -9;
  if (!a) return -10;
  if (dy <= 0) return -11;
  const int64_t ntile = (n + RT - 1) / RT, ntri = ntile * (ntile + 1) / 2;
  if (q0 < 0 || q1 < q0 || q1 > ntri) return -12;
  if (!work) return -14;
  if (!ka) return -15;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t lds = round_up(n, LEAF);
  const int64_t cnt = q1 - q0;
  RefineSymArgs p;
  p.X = X; p.variance = variance; p.ls = length_scales; p.noise = noise; p.a = a;
  p.prow = work; p.pcol = work + cnt * dy * RT * 2; p.lds = lds;
  p.n = (int)n; p.d = d; p.nls = nls; p.dy = dy; p.q_off = (int)q0;
  if (cnt > 0) {
    const dim3 grid((unsigned)cnt);
#define GPN_RESID(KIND)                                                                                 \
  do {                                                                                                  \
    if (dy == 1) hipLaunchKernelGGL((refine_resid_sym_kernel<KIND, 1>), grid, dim3(256), 0, s, p);      \
    else hipLaunchKernelGGL((refine_resid_sym_kernel<KIND, RDY>), grid, dim3(256), 0, s, p);            \
  } while (0)
    switch (kind) {
      case GPN_RBF: GPN_RESID(GPN_RBF); break;
      case GPN_MATERN52: GPN_RESID(GPN_MATERN52); break;
      case GPN_MATERN32: GPN_RESID(GPN_MATERN32); break;
      case GPN_EXP: GPN_RESID(GPN_EXP); break;
      default: GPN_RESID(GPN_PERIODIC); break;
    }
#undef GPN_RESID
    GPN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(refine_gather_kernel, dim3((unsigned)ntile, (unsigned)dy), dim3(256), 0, s, p.prow, p.pcol, (int)ntile, dy, lds, n, ka,
                     q0, q1);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// gpn_refine_resid_part for a covariance EXPRESSION (gpn_expr_term program, kexpr.hip): the share of the lower 64 x 64 tiles
// q0 <= q < q1 in Kyy a, Kyy re-computed from the points -- what a rank of the block-cyclic drivers contributes when the model's
// kernel is a Sum / Product tree (DistGPR over the reference's example model Linear + Rbf + Constant, examples/regression_1d.py:34-53).
// work: gpn_refine_resid_part_work_bytes(dy, q1 - q0); ka [dy][round_up(n, 128)][2].
extern "C" int gpn_refine_resid_part_expr(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                          const double* theta, const double* X, int64_t n, int d, const double* noise,
                                          const double* a, int dy, int64_t q0, int64_t q1, double* work, double* ka) {
  if (!terms || !group_start) return -2;
  if (!theta) return -6;
  if (!X) return -7;
  if (n <= 0) return -8;
  if (d <= 0) return -9;
  if (!noise) return -10;
  if (!a) return -11;
  if (dy <= 0) return -12;
  const int64_t ntile = (n + RT - 1) / RT, ntri = ntile * (ntile + 1) / 2;
  if (q0 < 0 || q1 < q0 || q1 > ntri) return -13;
  if (!work) return -15;
  if (!ka) return -16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t lds = round_up(n, LEAF);
  const int64_t cnt = q1 - q0;
  double* prow = work;
  double* pcol = work + cnt * dy * RT * 2;
  if (cnt > 0) {
    const int rc = expr_resid(s, terms, nterms, group_start, ngroups, theta, X, n, d, noise, a, dy, lds, q0, cnt, prow, pcol);
    if (rc != GPN_OK) return rc;
  }
  hipLaunchKernelGGL(refine_gather_kernel, dim3((unsigned)ntile, (unsigned)dy), dim3(256), 0, s, prow, pcol, (int)ntile, dy, lds, n, ka,
                     q0, q1);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_refine_finish(void* stream, const double* Y, const double* M, const double* a, const double* ka, int64_t n, int dy,
                                 double* out3) {
  if (!Y) return -2;
  if (!a) return -4;
  if (!ka) return -5;
  if (n < 0) return -6;
  if (dy <= 0) return -7;
  if (!out3) return -8;
  hipLaunchKernelGGL(refine_finish_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), Y, M, a, ka, n, dy, 1, round_up(n, LEAF), out3,
                     static_cast<double*>(nullptr));
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}
