// Blocked fp64 Cholesky for gfx950 (lower, in place, row-major) + the triangular
// solves built on it.  Replaces torch.cholesky / torch.triangular_solve under
// functions.cholesky / functions.trtrs (functions.py:46-47, 71-76).
//
// Two host drivers over the same kernels: the nested-panel driver with in-panel look-ahead on an auxiliary
// stream (potrf_lookahead, the default for the factorisation) and recursive blocking down to a
// 128x128 leaf (used for small n, the right-solves and as the cross-check in the tests):
//   potrf(A):  A11 = potrf(A11);  A21 <- A21 * L11^-T;  A22 -= A21 A21^T;  potrf(A22)
//   trsm(B,L): B1 <- B1 * L11^-T; B2 -= B1 * L21^T;     B2 <- B2 * L22^-T
// Every flop outside the 128x128 leaves is an "NT" fp64-MFMA contraction
// (gemm_f64.hip) whose K extent is as large as the recursion allows, so the
// N^2 matrix is streamed O(log N) times instead of N/nb times and the trailing
// updates stay MFMA-bound rather than HBM-bound.  The leaf kernel factors its
// block AND inverts it in one workgroup (blocked elimination on [A ; I], see
// below); panel solves are then products with the stored inverses, done
// IN PLACE: a LEAF-column-wide tile covers the whole K and N extent of its rows,
// and a workgroup only stores after its last load.
//
// "Extra rows" (gpnative.h): rows n..n+e-1 ride along in every panel solve and
// trailing update of the right spine of the recursion; on exit they hold
// (L^-1 R)^T -- alpha^T of gpr.py:62 -- for free.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <vector>
#include "gpn_common.h"

namespace gpn {

// ---------------------------------------------------------------------------------
// The first-generation 128 x 128 leaf (rounds 1-3; the factorisation's leaf is leaf16.hip since round 4).  What is left of it is its
// FACTOR = false form -- the inverse of a GIVEN lower-triangular block, gpn_trtri_diag -- on the same machinery: blocked
// elimination by 8 columns on the stacked matrix [L ; I] (256 x 128), whose identity rows come out as I L^-T = W^T; the trailing
// matrix lives in registers as 16 x 16 MFMA accumulator tiles (waves 0..7 own one tile row of each part, wave 8 inverts the
// 8 x 8 diagonal blocks); 3 barriers per 8 pivots, double-buffered panel.  (FACTOR = true -- Cholesky of the block on the same
// scheme, with its pipelined pivot wave and the stamped diagnostic build -- is in the history: rounds 1-3, LAB.md 8.)
// ---------------------------------------------------------------------------------
constexpr int XPS = 9;                 // padded row of the panel buffer (doubles)
constexpr int LEAF_THREADS = 576;

template <bool FACTOR>
__global__ __launch_bounds__(LEAF_THREADS) void potrf_leaf_kernel(double* A, int64_t lda, int kb_, int col0_,
                                                                  double* winv_, int32_t* info, int n_total) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  int kb = kb_, col0 = col0_;
  double* winv = winv_;
  if constexpr (!FACTOR) {
    col0 = blockIdx.x * LEAF;
    kb = min(LEAF, n_total - col0);
    A += (int64_t)col0 * lda + col0;
    winv += (int64_t)blockIdx.x * LEAF * LEAF;
  }
  __shared__ double Xp[2][256 * XPS];   // panel rows 0..127: A part, 128..255: identity (-> W^T) part
  // small blocks exchanged between the pivot wave and the tile waves, double-buffered by the
  // parity of the panel they belong to (the pivot wave runs ahead of the tile waves in PIPE mode)
  __shared__ double Dg2[2][64];         // L8 (row-major 8x8), for the output rows
  __shared__ double Ds2[2][64];         // L8 scaled by 1/diag (diagonal slot: 1/diag), for the row solves
  __shared__ int failflag;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7 tile waves, 8 pivot wave
  const bool tilewave = wave < 8;
  const int w = wave & 7;
  const int lr = lane >> 4, lc = lane & 15;      // D-layout: rows lr + 4r, column lc
  if (tid == 0) failflag = 0;
  if (!tilewave) __builtin_amdgcn_s_setprio(3);   // the pivot chain must not queue behind tile-wave VALU work

  // slot J (J = 0..7): A tile (w, J), used when J <= w; slot J+1: identity tile (8+w, J), used
  // when J >= w.  (slot w holds the A diagonal tile, slot w+1 the identity diagonal tile.)
  d4 acc[9];
  if (tilewave) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const bool isA = q <= w;
      const int J = isA ? q : q - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + lr + 4 * r, col = 16 * J + lc;   // row within its part
        double v;
        if (isA) {
          v = (row == col) ? 1.0 : 0.0;
          if (row < kb && col <= row) v = A[(int64_t)row * lda + col];
        } else {
          v = (row == col) ? 1.0 : 0.0;
        }
        acc[q][r] = v;
      }
    }
  }

  // raw 8-column panel (tile column Jp, half hp) -> LDS
  auto publish = [&](double* xp, int Jp, int hp) {
#pragma unroll
    for (int J = 0; J < 8; ++J) {
      if (J == Jp && (lc >> 3) == hp) {
        if (J <= w) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xp[(16 * w + lr + 4 * r) * XPS + (lc & 7)] = acc[J][r];
        }
        if (J >= w) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xp[(128 + 16 * w + lr + 4 * r) * XPS + (lc & 7)] = acc[J + 1][r];
        }
      }
    }
  };

  // rank-8 update with the solved panel in xp of the tile columns J in [jlo, jhi]
  auto update = [&](const double* xp, int jlo, int jhi, int J0) {
    const bool idlive = w <= J0;                 // identity rows 16w.. have met the pivots yet?
    const double* pa = xp + (16 * w + lc) * XPS + lr;
    double a0 = 0.0, a1 = 0.0, i0 = 0.0, i1 = 0.0;
    if (FACTOR) { a0 = -pa[0]; a1 = -pa[4]; }
    if (idlive) { i0 = -pa[128 * XPS]; i1 = -pa[128 * XPS + 4]; }
#pragma unroll
    for (int J = 0; J < 8; ++J) {
      if (J >= jlo && J <= jhi) {                // wave-uniform
        const double* pb = xp + (16 * J + lc) * XPS + lr;
        const double b0 = pb[0], b1 = pb[4];
        if (FACTOR && J <= w) {
          acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[J], 0, 0, 0);
          acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[J], 0, 0, 0);
        }
        if (idlive && J >= w) {
          acc[J + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(i0, b0, acc[J + 1], 0, 0, 0);
          acc[J + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, b1, acc[J + 1], 0, 0, 0);
        }
      }
    }
  };

  // P1 (pivot wave): ONE ELEMENT PER LANE -- lane 8i+c holds D[i][c] of the 8x8 diagonal block.
  // Per pivot j the serial chain is  readlane(d) -> v_rsq_f64 -> one cubic refinement ->
  // l = a*y -> DPP shift -> d' = a' - l*l -> readlane;  the rank-1 update of the other 63
  // elements is one masked FMA whose operands arrive through the LDS crossbar (ds_swizzle row
  // broadcast, ds_bpermute transpose-gather) and never sits on the chain.
  auto pivot_block = [&](double a, int c0, double& wt) {
    const int pi = lane >> 3, pc = lane & 7;
    (void)wt;
    double invd[8];
    int fail = 0;
    auto bcast = [](double v, int src) -> double {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    const int gather = ((pc << 3)) << 2;        // byte address of lane (pc, j) minus 4*j, for ds_bpermute
    double dn = a;                              // candidate next pivot (valid in the diagonal lanes)
    auto shr1 = [](double v) -> double {        // value of lane-1 (DPP row_shr:1; neighbours share a 16-lane row)
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    };
    double aleft = shr1(a);
    auto pivot = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      double d = bcast(dn, 9 * j);
      if (FACTOR) {
        if (!(d > 0.0)) {              // LAPACK dpotrf: ajj <= 0 or NaN
          if (!fail) fail = c0 + j + 1;
          d = 1.0;
        }
        // y = d^-1/2: v_rsq_f64 seed y0 + ONE cubically convergent step
        //   e = 1 - d y0^2;  y = y0 (1 + e p),  p = 1/2 + 3e/8.
        // Dependent fp64 ops cost ~38 cycles each on this chip, so the chain is kept to
        //   readlane -> rsq -> {d*y0, aleft*y0} -> e -> {p, (aleft*y0)*e} -> l' -> d' :
        // everything is expressed as x*y0*(1 + e p) so that no product waits for the refined y.
        const double y0 = __builtin_amdgcn_rsq(d);
        const double aly = aleft * y0;          // diagonal lane (j+1,j+1): L[j+1][j] before refinement
        const double ay0 = a * y0;
        const double e = fma(-d * y0, y0, 1.0);
        const double p = fma(e, 0.375, 0.5);
        const double ldiag = fma(aly * e, p, aly);
        dn = fma(-ldiag, ldiag, a);             // next pivot candidate (valid in lane 9(j+1))
        const double y = fma(y0 * e, p, y0);
        invd[j] = y;
        const double ay = fma(ay0 * e, p, ay0); // column j lanes: L[i][j]
        // rank-1 update of the trailing elements (c > j): a -= L[i][j] * L[c][j]
        constexpr int pat = (j << 5) | 0x18;    // ds_swizzle bit-mode: src = (lane & 0x18) | j  -> lane (i, j)
        const double li = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(ay), pat),
                                           __builtin_amdgcn_ds_swizzle(__double2loint(ay), pat));
        const double lcj = __hiloint2double(__builtin_amdgcn_ds_bpermute(gather + 4 * j, __double2hiint(ay)),
                                            __builtin_amdgcn_ds_bpermute(gather + 4 * j, __double2loint(ay)));
        double sq = d * y;                      // sqrt(d), off the critical chain
        sq = fma(fma(-sq, sq, d), 0.5 * y, sq);
        if (pc > j) a = fma(-li, lcj, a);
        else if (pc == j) a = (pi == j) ? sq : ay;
        aleft = shr1(a);                        // left neighbour's (updated) entry, for the next pivot
      } else {
        d = bcast(a, 9 * j);
        if (d == 0.0) {                // dtrtri: zero pivot
          if (!fail) fail = c0 + j + 1;
          d = 1.0;
        }
        invd[j] = 1.0 / d;
      }
    };
    pivot(std::integral_constant<int, 0>{}); pivot(std::integral_constant<int, 1>{});
    pivot(std::integral_constant<int, 2>{}); pivot(std::integral_constant<int, 3>{});
    pivot(std::integral_constant<int, 4>{}); pivot(std::integral_constant<int, 5>{});
    pivot(std::integral_constant<int, 6>{}); pivot(std::integral_constant<int, 7>{});
    // publish L8 (for the output rows) and, for the row solves, L8 scaled by its reciprocal
    // diagonal: x[c] = r[c]/L[c][c] - sum_k x[k] (L[c][k]/L[c][c]) has ONE dependent op per column
    double myinv = invd[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) myinv = (pi == j) ? invd[j] : myinv;
    if (pc <= pi) {
      Dg2[(c0 >> 3) & 1][pi * 8 + pc] = a;
      Ds2[(c0 >> 3) & 1][pi * 8 + pc] = (pc == pi) ? myinv : a * myinv;
    }
    if (lane == 0 && fail && failflag == 0) failflag = fail;     // keep the FIRST failing column
  };

  // P2 (threads 0..255 = waves 0..3): forward substitution of one panel row against L8;
  // the solved row goes back to LDS (the pivot wave streams it to global one phase later)
  auto solve_rows = [&](double* xp, int c0) {
    if (tid >= 256) return;
    const double* Ds = Ds2[(c0 >> 3) & 1];
    const double* Dg = Dg2[(c0 >> 3) & 1];
    const bool apart = tid < 128;
    const int rho = apart ? tid : tid - 128;
    const bool solve = apart ? (FACTOR && tid >= c0 + 8) : (rho <= c0 + 7);
    if (solve) {
      // right-looking order: after x[k] is final its contribution goes to ALL later columns at
      // once, so the dependent chain is 8 FMAs deep (one per column) with 7..1 independent FMAs
      // in between to fill the fp64 pipeline -- the row-by-row order chains 28 of them
      double x[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = xp[tid * XPS + c] * Ds[c * 8 + c];
#pragma unroll
      for (int k2 = 0; k2 < 7; ++k2) {
#pragma unroll
        for (int c = k2 + 1; c < 8; ++c) x[c] = fma(-x[k2], Ds[c * 8 + k2], x[c]);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) xp[tid * XPS + c] = x[c];
    } else if (!apart) {
#pragma unroll
      for (int c = 0; c < 8; ++c) xp[tid * XPS + c] = 0.0;     // W^T rows not reached yet: zeros
    } else if (FACTOR && tid >= c0 && tid < c0 + 8) {
#pragma unroll
      for (int c = 0; c < 8; ++c) xp[tid * XPS + c] = (c <= tid - c0) ? Dg[(tid - c0) * 8 + c] : 0.0;   // L8 itself
    }
  };

  // final values of panel (c0) from LDS to global, one row per thread, by waves 4..7 (idle
  // while waves 0..3 solve the next panel):  A rows t >= c0 -> A[t][c0..c0+7];
  // identity rows rho -> winv[c0+c][rho] (zeros beyond kb / above the diagonal)
  auto store_panel = [&](const double* xp, int c0) {
    const int t = tid - 256;                     // 0..255
    if (t < 0 || t >= 256) return;
    if (t < 128) {
      if (FACTOR && t >= c0 && t < kb) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c0 + c < kb && c0 + c <= t) A[(int64_t)t * lda + c0 + c] = xp[t * XPS + c];
      }
    } else {
      const int rho = t - 128;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bool in = (c0 + c < kb) && (rho < kb) && (rho <= c0 + c);
        winv[(int64_t)(c0 + c) * LEAF + rho] = in ? xp[t * XPS + c] : 0.0;
      }
    }
  };

  // ---- prologue: panel 0 ------------------------------------------------------------------
  double a_main = 0.0, wt = 0.0;              // pivot wave: its 8x8 block / carried L8^-T
  if (tilewave) publish(Xp[0], 0, 0);
  __syncthreads();
  if (!tilewave) {
    a_main = Xp[0][(lane >> 3) * XPS + (lane & 7)];
    pivot_block(a_main, 0, wt);
  }
  __syncthreads();
  if (!failflag) solve_rows(Xp[0], 0);
  __syncthreads();

  // ---- main loop --------------------------------------------------------------------------
  int done = 0;
  for (int kb8 = 0; kb8 < 16 && !failflag; ++kb8) {
    const int c0 = kb8 * 8;
    const int J0 = kb8 >> 1;
    const int jact = (c0 + 8) >> 4;          // tile column of the NEXT panel (8 when none)
    const int halfn = (kb8 + 1) & 1;
    double* cur = Xp[kb8 & 1];
    double* nxt = Xp[(kb8 + 1) & 1];
    // A
    if (tilewave) {
      if (kb8 < 15) {
        update(cur, jact, jact, J0);
        publish(nxt, jact, halfn);
      }
    }
    done = kb8 + 1;
    if (kb8 == 15) break;
    __syncthreads();
    // B
    if (tilewave) update(cur, jact + 1, 7, J0);
    else { a_main = nxt[(c0 + 8 + (lane >> 3)) * XPS + (lane & 7)]; pivot_block(a_main, c0 + 8, wt); }
    __syncthreads();
    if (failflag) break;                     // uniform
    // C
    solve_rows(nxt, c0 + 8);
    store_panel(cur, c0);
    __syncthreads();
  }
  if (!failflag) store_panel(Xp[1], 120);     // last panel (kb8 = 15 lives in buffer 1)
  __syncthreads();
  if (failflag) {
    if (tid == 0 && info) {
      if (failflag > LEAF) *info = GPN_INFO_INTERNAL;
      else if (*info == 0) *info = col0 + failflag;
    }
    // leave the rest of A untouched; publish a finite (zero) winv so later kernels stay finite
    for (int idx = tid; idx < LEAF * LEAF; idx += LEAF_THREADS) winv[idx] = 0.0;
  }
  (void)done;
}

// Switches (gpn_common.h: constants in the product library, per-thread variables behind the gpn_debug_set_* entry points at the
// end of this file in the tools' build) -- other PARAMETRISATIONS of the shipped driver
// (plain recursion, panel widths and nesting, left- / right-looking in-panel updates, the extra rows' kernel), which
// tests/test_gpu_parity.py::test_factorisation_drivers_agree holds against each other.  The schedules that were built, measured
// and dropped (look-ahead over panels in three forms incl. round 6's persistent bulk, look-ahead inside the outer panel, the
// fused chain step, both column passes in one launch, left-looking inner panels, split assembly) are in the history and in
// LAB.md 8 / 10 / 11 / 12 with their same-box logs under profiles/.
GPN_SWITCH int g_potrf_variant = 0;      // 0 = nested panels with in-panel look-ahead (default), 1 = plain recursion
GPN_SWITCH int g_panel_width = 0;        // inner panel width, 0 = by size
GPN_SWITCH int g_outer_width = 0;        // outer panel width: 0 = by size, -1 = one level
GPN_SWITCH int g_outer_width2 = 0;       // a third level
GPN_SWITCH int g_aux_left_looking = -1;  // in-panel updates beyond the next column block: -1 = by size, 0 right- / 1 left-looking
GPN_SWITCH int g_extra_rows_kernel = 1;  // 0 = the extra rows as one more tile row of the lower-tile launch

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

static void potrf_rec(Ctx& c, double* A, int64_t n, int64_t e, int64_t col0) {
  if (c.rc != GPN_OK || n <= 0) return;
  if (n <= LEAF) {
    c.rc = launch_leaf(c.s, A, c.lda, (int)n, (int)col0, c.winv + (col0 / LEAF) * (LEAF * LEAF), c.info, c.batch, c.sA, c.sW, 1);
    if (c.rc != GPN_OK) return;
    if (e > 0) trsm_rec(c, A + n * c.lda, e, c.lda, A, c.lda, n, col0, c.winv);
    return;
  }
  const int64_t h = split_point(n);
  potrf_rec(c, A, h, 0, col0);
  double* A21 = A + h * c.lda;
  const int64_t m = n - h + e;
  trsm_rec(c, A21, m, c.lda, A, c.lda, h, col0, c.winv);
  if (c.rc != GPN_OK) return;
  if (c.corner || e == 0) {
    c.rc = cgemm(c, c.s, m, m, h, -1.0, A21, c.lda, A21, c.lda, 1.0, A21 + h, c.lda, 1);
  } else {
    const int64_t ms = n - h;
    c.rc = cgemm(c, c.s, ms, ms, h, -1.0, A21, c.lda, A21, c.lda, 1.0, A21 + h, c.lda, 1);
    if (c.rc == GPN_OK)
      c.rc = cgemm(c, c.s, e, ms, h, -1.0, A21 + ms * c.lda, c.lda, A21, c.lda, 1.0, A21 + ms * c.lda + h, c.lda, 0);
  }
  potrf_rec(c, A21 + h, n - h, e, col0 + h);
}

// ---- flat right-looking driver with look-ahead -------------------------------------------
// The recursion above runs every launch of the factorisation back to back on one stream, and
// most of them are latency-bound (one leaf = one workgroup for ~19 us; panel solves and small
// updates of ~9 us each).  This driver shortens that serial chain.  Panels of PW columns; inside
// a panel, per LEAF-wide column block k:
//     main stream : leaf(k) -> solve ALL rows below against W_k (one in-place launch)
//                   -> update of the NEXT column block only (what leaf(k+1) waits for)
//     aux stream  : update of the remaining columns of the panel by block k (needs solve(k)),
//                   overlapped with leaf(k+1) on the main stream; for N >= 24576 the left-looking
//                   form instead (the column block after the next one by all solved panel columns)
// and one large K = PW contraction for everything right of the panel at its end.  The two
// streams are joined by events (fork/join, so the whole call can also be captured in a
// hipGraph); no data-dependent host logic.  The upper triangle inside the panel's diagonal
// square receives finite garbage from the rectangular updates: nothing reads it (the leaf
// masks j > i on load, every other consumer uses blocks strictly below the diagonal blocks
// or winv).
struct Aux {
  hipStream_t s1 = nullptr;                     // (one aux stream only: HIP multiplexes streams onto a few hardware queues)
  hipEvent_t solve[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t rest[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t extra_go = nullptr, extra_done = nullptr;     // the extra rows' share of an outer panel's trailing update (aux stream)
};
static std::mutex g_aux_mutex;
static std::unordered_map<hipStream_t, Aux> g_aux;

static Aux* aux_for(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_aux_mutex);
  auto it = g_aux.find(s);
  if (it != g_aux.end()) return &it->second;
  Aux a;
  int least = 0, greatest = 0;     // aux work is off the critical path: lowest priority
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
  if (hipStreamCreateWithPriority(&a.s1, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&a.extra_go, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&a.extra_done, hipEventDisableTiming) != hipSuccess) return nullptr;
  for (int i = 0; i < 4; ++i) {
    if (hipEventCreateWithFlags(&a.solve[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&a.rest[i], hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &g_aux.emplace(s, a).first->second;
}

// The extra rows' share of an outer panel's trailing update: R[e, ms] -= Pe[e, K] P[ms, K]^T, where P = the solved
// panel's rows below its diagonal square, Pe = P + ms * lda its e extra rows (right-hand sides carried through the
// factorisation) and R the extra rows right of the panel.  As tiles of the lower-tile launch those e rows are one more
// 128-row tile row of full-price MFMA work (1 % of C3's trailing updates, 4 % of C2's in a lock-step batch); here they
// are e dot products per matrix row, on the aux stream underneath that launch.  One wave per 4 matrix rows.
constexpr int XR_MAXE = 8;
constexpr int64_t XR_MIN_N = 20480;
template <int E>
__global__ __launch_bounds__(256) void extra_rows_update_kernel(const double* P, double* R, int64_t lda, int ms, int K, int e, int64_t sA) {
  P += (int64_t)blockIdx.y * sA;
  R += (int64_t)blockIdx.y * sA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = (blockIdx.x * 4 + wave) * 4;
  if (j0 >= ms) return;
  const double* Pe = P + (int64_t)ms * lda;
  const double* rows[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rows[r] = P + (int64_t)min(j0 + r, ms - 1) * lda;
  double acc[4][E];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < E; ++q) acc[r][q] = 0.0;
  for (int k = 2 * lane; k < K; k += 128) {
    double2 v[4], pe[E];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const double2*>(rows[r] + k);
#pragma unroll
    for (int q = 0; q < E; ++q) pe[q] = q < e ? *reinterpret_cast<const double2*>(Pe + (int64_t)q * lda + k) : double2{0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < E; ++q) acc[r][q] = fma(v[r].y, pe[q].y, fma(v[r].x, pe[q].x, acc[r][q]));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < E; ++q) {
      double t = acc[r][q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      acc[r][q] = t;
    }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < E; ++q)
        if (j0 + r < ms && q < e) R[(int64_t)q * lda + j0 + r] -= acc[r][q];
  }
}
static int extra_rows_update(hipStream_t s, const double* P, double* R, int64_t lda, int64_t ms, int64_t K, int64_t e, int batch, int64_t sA) {
  const dim3 grid((unsigned)((ms + 15) / 16), (unsigned)batch);
  if (e <= 1) hipLaunchKernelGGL(extra_rows_update_kernel<1>, grid, dim3(256), 0, s, P, R, lda, (int)ms, (int)K, (int)e, sA);
  else if (e <= 2) hipLaunchKernelGGL(extra_rows_update_kernel<2>, grid, dim3(256), 0, s, P, R, lda, (int)ms, (int)K, (int)e, sA);
  else if (e <= 4) hipLaunchKernelGGL(extra_rows_update_kernel<4>, grid, dim3(256), 0, s, P, R, lda, (int)ms, (int)K, (int)e, sA);
  else hipLaunchKernelGGL(extra_rows_update_kernel<XR_MAXE>, grid, dim3(256), 0, s, P, R, lda, (int)ms, (int)K, (int)e, sA);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

// Same-box sweeps (r1z, tools/potrf_ab.py): panel width 1024 / 1536 / 2048 -> C2 7.10 / 7.00 / 7.01 ms,
// N = 16384 31.9 (1536) vs 32.2 (2048), C3 201.5 / 200.0 / 201.1, C4 1492 / 1473 / 1472; with the
// left-looking aux update (below) the large sizes prefer 2048: C3 198.1, C4 1454 ms.
static inline bool large_problem(int64_t n) { return n >= 24576; }
// Nested panels (round 4).  The chain (leaf -> column solve -> K = 128 update of the next column block, plus the K = 128
// updates of the rest of the panel on the aux stream when it is wider than 256) runs inside INNER panels of w[0]
// columns; the update after an inner panel (K = w[0]) only reaches the end of the OUTER panel it sits in (a trapezoid:
// all rows below, lower-only in its top square), and everything right of an outer panel is updated once, at its end,
// with K = w[1] -- the lower-tile launch bench.py prices.  The inner width prices the chain's K = 128 work (HBM-bound
// column passes), the outer width the C-tile traffic of the big updates: with one level C2 wanted 256-column panels
// for the first and paid 11.5 GB of C traffic per evaluation for it (K = 256 updates of the whole trailing matrix).
// Same-box sweeps, ms per evaluation (round 4, Rbf D = 8, inner:outer):
//   N = 2048:  256 0.606 | 256:512 0.615 | 256:1024 0.621
//   N = 4096:  256 1.529 | 256:512 1.518 | 256:1024 1.535 | 256:2048 1.558      (x 8 in lock step: 4.89 | 4.68 | 4.65 | 4.69)
//   N = 8192:  256 5.52 | 256:512 5.33 | 256:1024 5.34 | 256:2048 5.43 | 512:1024 5.58 | 128:1024 5.40
//              (x 8 in lock step: 29.4 | 27.2 | 26.5 | 27.0 | 26.8 | 27.5 ms)
//   N = 12288: 512 14.37 | 512:1024 14.3 | 512:2048 14.27 | 256:1024 13.97 | 256:2048 14.04 | 1024:2048 14.5
//   N = 16384: 1024 29.7 | 1024:2048 29.97 | 512:1024 29.05 | 512:2048 29.3 | 256:1024 28.86 | 256:2048 29.05
//   N = 24576: 2048 83.99 | 1024 83.9 | 512:2048 82.2 | 256:2048 81.7 | 256:1024 82.7
//   N = 32768: 2048 184.6 | 512:2048 182.3 | 256:2048 182.7 | 256:1024 184.3 | 512:1024 184.4 | 1024:2048 184.2
//              (a third level -- 256:1024:2048, 256:512:2048, 256:1024:4096, 512:2048:8192 -- is within 0.3 ms of 512:2048)
struct PanelLevels { int n = 1; int64_t w[3] = {0, 0, 0}; };
static inline int64_t panel_width(int64_t n) {            // inner panels
  if (g_panel_width) return g_panel_width;
  return n < 20480 ? 256 : 512;
}
static inline PanelLevels panel_levels(int64_t n) {
  PanelLevels L;
  L.w[0] = panel_width(n);
  int64_t w1 = g_outer_width, w2 = g_outer_width2;
  if (w1 == 0) w1 = n <= 2048 ? 0 : n < 20480 ? 1024 : 2048;
  if (w1 == 2048 && w2 == 0 && n >= 49152) w2 = 4096;     // N = 65536: 512:2048 1344 | 512:2048:4096 1336 | one level of 2048: 1351 ms
  if (w1 > L.w[0]) { L.w[L.n] = (w1 / L.w[L.n - 1]) * L.w[L.n - 1]; ++L.n; }
  if (w2 > L.w[L.n - 1]) { L.w[L.n] = (w2 / L.w[L.n - 1]) * L.w[L.n - 1]; ++L.n; }
  return L;
}

static void potrf_lookahead(Ctx& c, double* A, int64_t n, int64_t e) {
  Aux* ax = aux_for(c.s);
  if (!ax) { c.rc = GPN_E_HIP; return; }
  const PanelLevels lev = panel_levels(n);
  const int64_t lda = c.lda, PW = lev.w[0];
  const bool left_looking = g_aux_left_looking < 0 ? large_problem(n) : g_aux_left_looking != 0;
  auto hip_ok = [&](hipError_t err) { if (err != hipSuccess && c.rc == GPN_OK) { set_hip_error(err, "potrf_lookahead"); c.rc = GPN_E_HIP; } };
  int step = 0, rest_idx = 0;
  bool rest_pending = false, extra_pending = false;
  for (int64_t p0 = 0; p0 < n && c.rc == GPN_OK; p0 += PW) {
    const int64_t pw = std::min(PW, n - p0), pend = p0 + pw;
    if (extra_pending) { hip_ok(hipStreamWaitEvent(c.s, ax->extra_done, 0)); extra_pending = false; }   // the extra rows of this panel's columns
    for (int64_t k0 = p0; k0 < pend && c.rc == GPN_OK; k0 += LEAF, ++step) {
      const int64_t kb = std::min<int64_t>(LEAF, n - k0);
      const int64_t c1 = k0 + kb;                 // first row/column after this block
      double* Akk = A + k0 * lda + k0;
      const double* Wk = c.winv + (k0 / LEAF) * (LEAF * LEAF);
      {
        const int rec = profile_on() ? profile_begin(c.s, c.batch * 2.0 * LEAF * LEAF * LEAF / 3.0, PROF_LEAF) : -1;
        const int lrc = launch_leaf(c.s, Akk, lda, (int)kb, (int)k0, const_cast<double*>(Wk), c.info, c.batch, c.sA, c.sW, 1);
        if (c.rc == GPN_OK) c.rc = lrc;
        if (rec >= 0) profile_end(c.s, rec);
      }
      const int64_t m = n + e - c1;                // rows below (incl. the extra rows)
      if (m <= 0 || c.rc != GPN_OK) continue;
      double* B = A + c1 * lda + k0;               // [m, kb] <- B W_k^T   (in place)
      c.rc = ccolpanel(c, c.s, 0, m, kb, B, lda, Wk, LEAF, B, lda);
      if (c.rc != GPN_OK || c1 >= pend) continue;  // last block of the panel: nothing left inside it
      const int64_t nb1 = std::min<int64_t>(LEAF, pend - c1);
      const int64_t c2 = c1 + nb1;
      if (rest_pending) {                          // column block c1 was last written on the aux stream
        hip_ok(hipStreamWaitEvent(c.s, ax->rest[rest_idx], 0));
        rest_pending = false;
      }
      const bool fork = c2 < pend;
      // next column block (rows c1.., columns c1..c2): what the next leaf and solve wait for.
      // The aux work is forked AFTER it: launched together, the 1000+ workgroups of the rest
      // update crowd this small launch out (16 us instead of 7); behind it they overlap with
      // the next leaf + solve instead.
      c.rc = cgemm(c, c.s, m, nb1, kb, -1.0, B, lda, B, lda, 1.0, A + c1 * lda + c1, lda, 0);
      if (fork) hip_ok(hipEventRecord(ax->solve[step & 3], c.s));
      if (fork && c.rc == GPN_OK) {                // the rest of the panel on the aux stream
        hip_ok(hipStreamWaitEvent(ax->s1, ax->solve[step & 3], 0));
        const int64_t m2 = n + e - c2;
        if (!left_looking) {
          // right-looking: all remaining columns of the panel by block k (K = LEAF)
          c.rc = cgemm(c, ax->s1, m2, pend - c2, kb, -1.0, A + c2 * lda + k0, lda, A + c2 * lda + k0, lda, 1.0,
                         A + c2 * lda + c2, lda, 0);
        } else {
          // left-looking: only the column block AFTER the next one, by every solved column of the
          // panel so far (K = c1 - p0) -- each block of the panel is read and written once here
          // instead of once per earlier block.  Pays from N ~ 32768 up, where the K = 128 rank
          // updates of 250+ tile rows outlast the leaf (C3 200.6 -> 198.9 ms, C4 1473 -> 1463 ms);
          // costs 2 % at C2, where the long-K launches of one tile column are latency-bound
          const int64_t nb2 = std::min<int64_t>(LEAF, pend - c2);
          c.rc = cgemm(c, ax->s1, m2, nb2, c1 - p0, -1.0, A + c2 * lda + p0, lda, A + c2 * lda + p0, lda, 1.0,
                         A + c2 * lda + c2, lda, 0);
        }
        rest_idx = step & 3;
        hip_ok(hipEventRecord(ax->rest[rest_idx], ax->s1));
        rest_pending = true;
      }
    }
    if (c.rc != GPN_OK) break;
    if (rest_pending) {                            // join before the large update reads the panel
      hip_ok(hipStreamWaitEvent(c.s, ax->rest[rest_idx], 0));
      rest_pending = false;
    }
    if (pend >= n) break;
    const int64_t m = n + e - pend;
    int l = 0;                                     // the widest level that ends here
    while (l + 1 < lev.n && pend % lev.w[l + 1] == 0) ++l;
    const int64_t o0 = l == 0 ? p0 : pend - lev.w[l];
    double* P = A + pend * lda + o0;               // [m, pend - o0] the solved panel below the diagonal square
    const int64_t kp = round_up(pend - o0, 16);
    if (l + 1 < lev.n) {
      // below the top level: the columns up to the end of the panel one level up only (all rows below incl. the extra
      // ones; lower-only in the top square)
      const int64_t oend = std::min(n, (pend / lev.w[l + 1] + 1) * lev.w[l + 1]);
      c.rc = cgemm(c, c.s, m, oend - pend, kp, -1.0, P, lda, P, lda, 1.0, A + pend * lda + pend, lda, 2);
    } else if (e > 0 && e <= XR_MAXE && g_extra_rows_kernel && n >= XR_MIN_N && (lda & 1) == 0 && (o0 & 1) == 0) {
      // matrix rows: lower-tile square here; the few extra rows: dot products on the aux stream underneath it
      // (the fork / join is ~25 us per outer panel: C2 5.36 -> 5.53 ms with it, x 8 in lock step neutral, C3 182.4 -> 181.3;
      //  against the THIN tile row of gemm_f64.hip that the extra rows are otherwise: N = 16384 28.42 vs 28.14 ms, C3 178.9 vs
      //  180.2, C4 1329.7 vs 1333 -- on from XR_MIN_N rows)
      const int64_t ms = n - pend;
      hip_ok(hipEventRecord(ax->extra_go, c.s));
      hip_ok(hipStreamWaitEvent(ax->s1, ax->extra_go, 0));
      c.rc = cgemm(c, c.s, ms, ms, kp, -1.0, P, lda, P, lda, 1.0, A + pend * lda + pend, lda, 1);
      if (c.rc == GPN_OK) c.rc = extra_rows_update(ax->s1, P, A + n * lda + pend, lda, ms, kp, e, c.batch, c.sA);
      hip_ok(hipEventRecord(ax->extra_done, ax->s1));
      extra_pending = true;
    } else if (c.corner || e == 0) {
      c.rc = cgemm(c, c.s, m, m, kp, -1.0, P, lda, P, lda, 1.0, A + pend * lda + pend, lda, 1);
    } else {
      const int64_t ms = n - pend;                 // matrix rows / columns left; the e extra rows: rectangular
      c.rc = cgemm(c, c.s, ms, ms, kp, -1.0, P, lda, P, lda, 1.0, A + pend * lda + pend, lda, 1);
      if (c.rc == GPN_OK)
        c.rc = cgemm(c, c.s, e, ms, kp, -1.0, P + ms * lda, lda, P, lda, 1.0, A + n * lda + pend, lda, 0);
    }
  }
  if (extra_pending) hip_ok(hipStreamWaitEvent(c.s, ax->extra_done, 0));     // (error exits: nothing of this call stays in flight unordered)
}


// U_ii <- W_ii^T for every LEAF x LEAF diagonal block
__global__ __launch_bounds__(256) void diag_transpose_kernel(const double* winv, double* U, int64_t ldu, int n, int64_t sW = 0,
                                                             int64_t sU = 0) {
  // 32x32 sub-tiles through LDS: blockIdx.y enumerates the (LEAF/32)^2 sub-tiles of W; blockIdx.z = model of a lock-step batch
  __shared__ double t[32][33];
  winv += (int64_t)blockIdx.z * sW;
  U += (int64_t)blockIdx.z * sU;
  const int blk = blockIdx.x, tid = threadIdx.x;
  const int si = blockIdx.y / (LEAF / 32), sj = blockIdx.y % (LEAF / 32);
  const double* W = winv + (int64_t)blk * LEAF * LEAF;
  const int tx = tid & 31, ty = tid >> 5;
  for (int k = ty; k < 32; k += 8) t[k][tx] = W[(si * 32 + k) * LEAF + sj * 32 + tx];
  __syncthreads();
  const int kb = min(LEAF, n - blk * LEAF);
  double* Ub = U + ((int64_t)blk * LEAF) * ldu + blk * LEAF;
  for (int k = ty; k < 32; k += 8) {
    const int i = sj * 32 + k, c = si * 32 + tx;     // U[i][c] = W[c][i]
    if (i < kb && c < kb) Ub[(int64_t)i * ldu + c] = t[tx][k];
  }
}

// U (upper, row-major) <- L^-T by recursion on the block structure:
//   U12 = -U11 * L21^T * L22^-T : one NT contraction (A = U11 upper: K range clipped)
//   followed by the in-place right solve with L22 (trsm_rec) -- no new primitive.
static void trtri_rec(Ctx& c, const double* L, int64_t ldl, double* U, int64_t ldu, int64_t n, int64_t diag0) {
  if (c.rc != GPN_OK || n <= LEAF) return;
  const int64_t h = split_point(n);
  trtri_rec(c, L, ldl, U, ldu, h, diag0);
  trtri_rec(c, L + h * ldl + h, ldl, U + h * ldu + h, ldu, n - h, diag0 + h);
  if (c.rc != GPN_OK) return;
  c.rc = gemm_nt(c.s, h, n - h, h, -1.0, U, ldu, L + h * ldl, ldl, 0.0, U + h, ldu, 0, GPN_TRI_A_UPPER);
  trsm_rec(c, U + h, h, ldu, L + h * ldl + h, ldl, n - h, diag0 + h, c.winv);
}

// ---- reductions / utilities -------------------------------------------------
__global__ __launch_bounds__(1024) void lml_reduce_kernel(const double* A, int64_t n, int64_t e, int64_t lda,
                                                          double* out3, int64_t sA) {
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
    out3[2] = -0.5 * quad - (double)e * logdet - 0.5 * (double)e * (double)n * 1.8378770664093454836;
  }
}

__global__ void transpose_kernel(const double* src, int64_t rows, int64_t cols, int64_t lds,
                                 double* dst, int64_t ldd) {
  __shared__ double t[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int64_t r = r0 + k, cc = c0 + tx;
    t[k][tx] = (r < rows && cc < cols) ? src[r * lds + cc] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int64_t cc = c0 + k, r = r0 + tx;   // dst[cc, r]
    if (cc < cols && r < rows) dst[cc * ldd + r] = t[tx][k];
  }
}

// `gridDim.z` square blocks at constant strides: dst_z[c, r] = src_z[r, c]
// (inner > 0: block z = z1 + inner * z2 -- node z1 of lock-step model z2, models at strides ssrc2 / sdst2)
__global__ void transpose_batched_kernel(const double* src, int64_t n, int64_t lds, int64_t ssrc,
                                         double* dst, int64_t ldd, int64_t sdst, int inner = 0, int64_t ssrc2 = 0, int64_t sdst2 = 0) {
  __shared__ double t[32][33];
  if (inner > 0) {
    const int z2 = blockIdx.z / inner, z1 = blockIdx.z - z2 * inner;
    src += (int64_t)z1 * ssrc + (int64_t)z2 * ssrc2;
    dst += (int64_t)z1 * sdst + (int64_t)z2 * sdst2;
  } else {
    src += (int64_t)blockIdx.z * ssrc;
    dst += (int64_t)blockIdx.z * sdst;
  }
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int64_t r = r0 + k, cc = c0 + tx;
    t[k][tx] = (r < n && cc < n) ? src[r * lds + cc] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int64_t cc = c0 + k, r = r0 + tx;
    if (cc < n && r < n) dst[cc * ldd + r] = t[tx][k];
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

extern "C" int64_t gpn_factor_ld(int64_t n, int64_t e) { return round_up(n + e, LEAF); }
// +16 zero rows: a contraction whose operand starts at row n (the extra rows) reads whole 16-row groups
extern "C" int64_t gpn_factor_rows(int64_t n, int64_t e) { return round_up(n + e, LEAF) + 16; }
extern "C" int64_t gpn_winv_bytes(int64_t n) {
  return (round_up(n, LEAF) / LEAF) * LEAF * LEAF * (int64_t)sizeof(double);
}

extern "C" int gpn_potrf_lower(void* stream, double* A, int64_t n, int64_t e, int64_t lda,
                               double* winv, int32_t* info) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < round_up(n + e, LEAF) || (lda % LEAF) != 0) return -5;
  if (!winv) return -6;
  if (!info) return -7;
  if (reinterpret_cast<uintptr_t>(A) & 15) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), lda, winv, info, GPN_OK};
  if (g_potrf_variant == 1 || n <= 2 * LEAF) potrf_rec(c, A, n, e, 0);
  else potrf_lookahead(c, A, n, e);
  return c.rc;
}

// panel width of the look-ahead driver for an n x n factorisation (what bench.py needs to count the
// algorithmic flops of the SYRK trailing updates: one lower-tile K = width contraction per panel)
extern "C" int64_t gpn_potrf_panel_width(int64_t n) {
  if (g_potrf_variant == 1 || n <= 2 * LEAF) return 0;
  const PanelLevels L = panel_levels(n);
  return L.w[L.n - 1];
}
extern "C" int gpn_potrf_panel_levels(int64_t n, int64_t* widths3) {
  if (!widths3) return -2;
  widths3[0] = widths3[1] = widths3[2] = 0;
  if (g_potrf_variant == 1 || n <= 2 * LEAF) return 0;
  const PanelLevels L = panel_levels(n);
  for (int i = 0; i < L.n; ++i) widths3[i] = L.w[i];
  return L.n;
}

// The library keeps one low-priority helper stream + a few events per caller stream that has run a
// factorisation (created on first use).  A caller that destroys a stream releases them here, so a
// recycled hipStream_t handle never meets stale helpers.  stream == NULL releases all of them.
extern "C" int gpn_release_stream(void* stream) {
  dist_release(static_cast<hipStream_t>(stream));
  potrf_persistent_release(static_cast<hipStream_t>(stream));
  std::lock_guard<std::mutex> lock(g_aux_mutex);
  auto drop = [](Aux& a) {
    if (a.s1) { (void)hipStreamSynchronize(a.s1); (void)hipStreamDestroy(a.s1); }
    if (a.extra_go) (void)hipEventDestroy(a.extra_go);
    if (a.extra_done) (void)hipEventDestroy(a.extra_done);
    for (int i = 0; i < 4; ++i) {
      if (a.solve[i]) (void)hipEventDestroy(a.solve[i]);
      if (a.rest[i]) (void)hipEventDestroy(a.rest[i]);
    }
  };
  if (!stream) {
    for (auto& kv : g_aux) drop(kv.second);
    g_aux.clear();
    return GPN_OK;
  }
  auto it = g_aux.find(static_cast<hipStream_t>(stream));
  if (it != g_aux.end()) { drop(it->second); g_aux.erase(it); }
  return GPN_OK;
}

// A tile column of a larger matrix: the n x n block at A is factored in place and the e rows below it
// (any number: they are a panel of the enclosing matrix, not right-hand sides with room beside them) come
// out as R L^-T, with NOTHING right of column n read or written -- lda >= round_up(n, 128) suffices.
// Same drivers and in-panel look-ahead as gpn_potrf_lower, so the leaf chain of the tile runs underneath
// the updates of all e rows instead of alone on the chip.  (gptorch_amd/dist.py, csrc/dist.hip.)
extern "C" int gpn_potrf_lower_panel(void* stream, double* A, int64_t n, int64_t e, int64_t lda,
                                     double* winv, int32_t* info) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < round_up(n, LEAF) || (lda % LEAF) != 0) return -5;
  if (!winv) return -6;
  if (!info) return -7;
  if (reinterpret_cast<uintptr_t>(A) & 15) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), lda, winv, info, GPN_OK};
  c.corner = false;
  if (g_potrf_variant == 1 || n <= 2 * LEAF) potrf_rec(c, A, n, e, 0);
  else potrf_lookahead(c, A, n, e);
  return c.rc;
}

// (tools' build only: the calling thread's driver parametrisation -- see the switches at the top of the drivers)
GPN_DEBUG_ONLY(
extern "C" int gpn_debug_set_potrf_variant(int v) {
  g_potrf_variant = v & 1;           // bit 0: plain recursion; bits 8..: inner panel width / 128
  g_panel_width = ((v >> 8) & 0xff) * LEAF;
  g_aux_left_looking = ((v >> 3) & 1) ? 1 : (((v >> 5) & 1) ? 0 : -1);   // bit 3: force left-looking aux update, bit 5: force right-looking
  return GPN_OK;
}
extern "C" int gpn_debug_set_extra_rows(int on) { g_extra_rows_kernel = on; return GPN_OK; }
extern "C" int gpn_debug_set_outer_width(int w1, int w2) { g_outer_width = w1; g_outer_width2 = w2; return GPN_OK; })

extern "C" int gpn_trtri_diag(void* stream, const double* L, int64_t n, int64_t ldl, double* winv, int32_t* info) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < n) return -4;
  if (!winv) return -5;
  if (n == 0) return GPN_OK;
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL((potrf_leaf_kernel<false>), dim3(nb), dim3(LEAF_THREADS), 0, static_cast<hipStream_t>(stream),
                     const_cast<double*>(L), ldl, 0, 0, winv, info, (int)n);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

extern "C" int gpn_trsm_right_lt(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                                 double* B, int64_t m, int64_t ldb) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!winv) return -5;
  if (!B) return -6;
  if (m < 0) return -7;
  if (ldb < round_up(n, LEAF) || (ldb % LEAF) != 0) return -8;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  if (n == 0 || m == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), ldl, const_cast<double*>(winv), nullptr, GPN_OK};
  trsm_rec(c, B, m, ldb, L, ldl, n, 0, winv);
  return c.rc;
}

// gpn_trsm_right_lt for `batch` factors of one shape in lock step: problem b solves against L + b sL / winv + b sW in place on
// B + b sB.  The same recursion, every launch once over all problems: per problem bit-identical to gpn_trsm_right_lt.
extern "C" int gpn_trsm_right_lt_batched(void* stream, const double* L, int64_t n, int64_t ldl, int64_t sL, const double* winv, int64_t sW,
                                         double* B, int64_t m, int64_t ldb, int64_t sB, int batch) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (batch < 1) return -12;
  if (batch > 1 && (sL < n * ldl || (sL & 1))) return -5;
  if (!winv) return -6;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -7;
  if (!B) return -8;
  if (m < 0) return -9;
  if (ldb < round_up(n, LEAF) || (ldb % LEAF) != 0) return -10;
  if (batch > 1 && (sB < m * ldb || (sB & 1))) return -11;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return GPN_E_ALIGN;
  if (n == 0 || m == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), ldl, const_cast<double*>(winv), nullptr, GPN_OK};
  c.batch = batch; c.sA = sL; c.sW = sW; c.sRhs = sB;
  trsm_rec(c, B, m, ldb, L, ldl, n, 0, winv);
  return c.rc;
}

// ---- right-solves against BIG inverted diagonal blocks -----------------------------------------------------------------
// gpn_trsm_right_lt walks the recursion down to the 128-wide leaf inverses: at m = 1024 right-hand sides (GPR._predict,
// gpr.py:104-106) its <= 512-wide levels are ~190 latency-bound launches -- 2.3 ms at N = 8192 for 6.9e10 flops (33
// TFLOP/s).  With the inverses of the BIGB x BIGB diagonal blocks formed once per factor (n BIGB^2 / 3 flops), the same
// solve is n / BIGB steps of two large contractions:  X_k = B_k W_k^T  (K-clipped),  B_rest -= X_k L(rest, k)^T.
static constexpr int64_t BIGB = 1024;      // (the block of gpn_block_inverse: part of the documented layout)

extern "C" int64_t gpn_block_inverse_bytes(int64_t n) {
  if (n <= 0) return 0;
  const int64_t nb = (n + BIGB - 1) / BIGB;
  return (nb * BIGB * BIGB + (BIGB + 16) * BIGB) * (int64_t)sizeof(double);       // the blocks + one scratch U
}

// wb[b] (BIGB x BIGB, ld BIGB, row-major lower, zero above the diagonal and beyond a ragged last block) = L_bb^-1
extern "C" int gpn_block_inverse(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv, double* wb) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!winv) return -5;
  if (!wb) return -6;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(wb) & 15)) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t nb = (n + BIGB - 1) / BIGB;
  double* U = wb + nb * BIGB * BIGB;
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t nk = std::min(BIGB, n - b * BIGB);
    GPN_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)((BIGB + 16) * BIGB) * sizeof(double), s));
    GPN_HIP_CHECK(hipMemsetAsync(wb + b * BIGB * BIGB, 0, (size_t)(BIGB * BIGB) * sizeof(double), s));
    int rc = gpn_trtri_upper(stream, L + b * BIGB * (ldl + 1), nk, ldl, winv + b * (BIGB / LEAF) * LEAF * LEAF, U, BIGB);
    if (rc != GPN_OK) return rc;
    rc = gpn_transpose(stream, U, nk, nk, BIGB, wb + b * BIGB * BIGB, BIGB);       // W = U^T
    if (rc != GPN_OK) return rc;
  }
  return GPN_OK;
}

// X[m, n] = B L^-T with the block inverses of gpn_block_inverse; B [m, n] (ldb) is CONSUMED (it receives the updates);
// B and X padded like factor buffers (rows to a multiple of 16, zero K padding), X != B.
extern "C" int gpn_trsm_right_lt_blocked(void* stream, const double* L, int64_t n, int64_t ldl, const double* wb,
                                         double* B, int64_t m, int64_t ldb, double* X, int64_t ldx) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!wb) return -5;
  if (!B) return -6;
  if (m < 0) return -7;
  if (ldb < round_up(n, 16) || (ldb & 1)) return -8;
  if (!X || X == B) return -9;
  if (ldx < round_up(n, 16) || (ldx & 1)) return -10;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(X) & 15) ||
      (reinterpret_cast<uintptr_t>(wb) & 15)) return GPN_E_ALIGN;
  if (n == 0 || m == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t nb = (n + BIGB - 1) / BIGB;
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t c0 = b * BIGB, nk = std::min(BIGB, n - c0), kp = round_up(nk, 16);
    int rc = gemm_nt(s, m, nk, kp, 1.0, B + c0, ldb, wb + b * BIGB * BIGB, BIGB, 0.0, X + c0, ldx, 0, GPN_TRI_B_LOWER);
    if (rc != GPN_OK) return rc;
    const int64_t rest = n - (c0 + nk);
    if (rest > 0) {
      rc = gemm_nt(s, m, rest, kp, -1.0, X + c0, ldx, L + (c0 + nk) * ldl + c0, ldl, 1.0, B + c0 + nk, ldb, 0);
      if (rc != GPN_OK) return rc;
    }
  }
  return GPN_OK;
}

// Level-parallel variant with a scratch matrix S (same shape as U, zero-initialised):
//   U12 = -U11 * L21^T * U22  as two NT contractions  T = U11 L21^T  (into S12)  and
//   U12 = -T * (U22^T)^T  with U22^T written into S22 by an HBM-bound transpose --
// no right-solve chain, and all nodes of one depth of the recursion tree are independent:
// equal-shaped ones go out as one strided-batch launch per operation.  (Dealing them onto side
// streams as well was measured and dropped: 199 vs 196 ms at N = 32768, 3.96 vs 3.89 at 8192.)
struct TNode { int64_t off, n, h; int depth; };
static void trtri_collect(std::vector<TNode>& v, int64_t off, int64_t n, int depth) {
  if (n <= LEAF) return;
  const int64_t h = split_point(n);
  v.push_back({off, n, h, depth});
  trtri_collect(v, off, h, depth + 1);
  trtri_collect(v, off + h, n - h, depth + 1);
}

// batch > 1: `batch` lock-step models (L, U, S of model b at b * sLm / sUm / sSm): every launch of the single-model schedule
// becomes ONE launch over all models (equal nodes of a level x models: two-level strided batch).  Same launches per model,
// same per-entry summation order: each model's U is bit-identical to its own gpn_trtri_upper_ws.
static int trtri_levels(hipStream_t s, const double* L, int64_t ldl, double* U, int64_t ldu, double* S, int64_t lds,
                        int64_t n, int batch = 1, int64_t sLm = 0, int64_t sUm = 0, int64_t sSm = 0) {
  std::vector<TNode> nodes;
  trtri_collect(nodes, 0, n, 0);
  int maxd = -1;
  for (const TNode& t : nodes) maxd = std::max(maxd, t.depth);
  for (int d = maxd; d >= 0; --d) {
    std::vector<const TNode*> lvl;
    for (const TNode& t : nodes) if (t.depth == d) lvl.push_back(&t);
    // Nodes of one depth with the same shape at a constant spacing (all of them when n is a
    // power-of-two multiple of the leaf) go out as ONE strided-batch launch per operation;
    // irregular and large nodes follow one by one.
    size_t i0 = 0;
    std::vector<const TNode*> single;
    while (i0 < lvl.size()) {
      size_t i1 = i0 + 1;
      if (i1 < lvl.size() && lvl[i1]->n == lvl[i0]->n && lvl[i1]->h == lvl[i0]->h) {
        const int64_t step = lvl[i1]->off - lvl[i0]->off;
        while (i1 < lvl.size() && lvl[i1]->n == lvl[i0]->n && lvl[i1]->h == lvl[i0]->h &&
               lvl[i1]->off - lvl[i1 - 1]->off == step) ++i1;
      }
      const int cnt = (int)(i1 - i0);
      if (cnt >= 2 && lvl[i0]->n <= 2048) {
        const TNode& t = *lvl[i0];
        const int64_t h = t.h, m2 = t.n - t.h, o = t.off, step = lvl[i0 + 1]->off - o;
        const int64_t sU = step * (ldu + 1), sL = step * (ldl + 1), sS = step * (lds + 1);
        dim3 grid((unsigned)((m2 + 31) / 32), (unsigned)((m2 + 31) / 32), (unsigned)(cnt * batch));
        hipLaunchKernelGGL(transpose_batched_kernel, grid, dim3(256), 0, s, U + (o + h) * ldu + o + h, m2, ldu, sU,
                           S + (o + h) * lds + o + h, lds, sS, batch > 1 ? cnt : 0, sUm, sSm);
        GPN_LAUNCH_CHECK();
        int rc = gemm_nt_strided2(s, h, m2, h, 1.0, U + o * ldu + o, ldu, L + (o + h) * ldl + o, ldl, 0.0,
                                  S + o * lds + o + h, lds, 0, GPN_TRI_A_UPPER, cnt, sU, sL, sS, batch, sUm, sLm, sSm);
        if (rc != GPN_OK) return rc;
        rc = gemm_nt_strided2(s, h, m2, round_up(m2, 16), -1.0, S + o * lds + o + h, lds, S + (o + h) * lds + o + h, lds,
                              0.0, U + o * ldu + o + h, ldu, 0, GPN_TRI_B_LOWER, cnt, sS, sS, sU, batch, sSm, sSm, sUm);
        if (rc != GPN_OK) return rc;
      } else {
        for (size_t i = i0; i < i1; ++i) single.push_back(lvl[i]);
      }
      i0 = i1;
    }
    for (const TNode* tp : single) {
      const TNode& t = *tp;
      const int64_t h = t.h, m2 = t.n - t.h, o = t.off;
      const double* U11 = U + o * ldu + o;
      const double* L21 = L + (o + h) * ldl + o;
      const double* U22 = U + (o + h) * ldu + o + h;
      double* S12 = S + o * lds + o + h;
      double* S22 = S + (o + h) * lds + o + h;
      if (batch > 1) {
        dim3 grid((unsigned)((m2 + 31) / 32), (unsigned)((m2 + 31) / 32), (unsigned)batch);
        hipLaunchKernelGGL(transpose_batched_kernel, grid, dim3(256), 0, s, U22, m2, ldu, sUm, S22, lds, sSm, 0, (int64_t)0, (int64_t)0);
      } else {
        dim3 grid((unsigned)((m2 + 31) / 32), (unsigned)((m2 + 31) / 32));
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, U22, m2, m2, ldu, S22, lds);
      }
      GPN_LAUNCH_CHECK();
      int rc = gemm_nt_strided2(s, h, m2, h, 1.0, U11, ldu, L21, ldl, 0.0, S12, lds, 0, GPN_TRI_A_UPPER, 1, 0, 0, 0, batch, sUm, sLm, sSm);
      if (rc != GPN_OK) return rc;
      rc = gemm_nt_strided2(s, h, m2, round_up(m2, 16), -1.0, S12, lds, S22, lds, 0.0, U + o * ldu + o + h, ldu, 0, GPN_TRI_B_LOWER,
                            1, 0, 0, 0, batch, sSm, sSm, sUm);
      if (rc != GPN_OK) return rc;
    }
  }
  return GPN_OK;
}

// U_b = L_b^-T for `batch` lock-step models (gpn_lml_backward_batched): model b's factor at L + b sL, leaf inverses at
// winv + b sW, U / S at + b sU / + b sS (zero-initialised by the caller)
int gpn::trtri_upper_ws_batched(hipStream_t s, const double* L, int64_t n, int64_t ldl, int64_t sL, const double* winv, int64_t sW,
                                double* U, int64_t ldu, int64_t sU, double* S, int64_t lds, int64_t sS, int batch) {
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL(diag_transpose_kernel, dim3(nb, (LEAF / 32) * (LEAF / 32), (unsigned)batch), dim3(256), 0, s, winv, U, ldu, (int)n,
                     sW, sU);
  GPN_LAUNCH_CHECK();
  return trtri_levels(s, L, ldl, U, ldu, S, lds, n, batch, sL, sU, sS);
}

extern "C" int gpn_trtri_upper_ws(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                                  double* U, int64_t ldu, double* S, int64_t lds) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!winv) return -5;
  if (!U) return -6;
  if (ldu < round_up(n, LEAF) || (ldu % LEAF) != 0) return -7;
  if (!S) return -8;
  if (lds < round_up(n, LEAF) || (lds % LEAF) != 0) return -9;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(U) & 15) ||
      (reinterpret_cast<uintptr_t>(S) & 15)) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL(diag_transpose_kernel, dim3(nb, (LEAF / 32) * (LEAF / 32)), dim3(256), 0, s, winv, U, ldu, (int)n, (int64_t)0, (int64_t)0);
  GPN_LAUNCH_CHECK();
  return trtri_levels(s, L, ldl, U, ldu, S, lds, n);
}

extern "C" int gpn_trtri_upper(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                               double* U, int64_t ldu) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (!winv) return -5;
  if (!U) return -6;
  if (ldu < round_up(n, LEAF) || (ldu % LEAF) != 0) return -7;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(U) & 15)) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned nb = (unsigned)((n + LEAF - 1) / LEAF);
  hipLaunchKernelGGL(diag_transpose_kernel, dim3(nb, (LEAF / 32) * (LEAF / 32)), dim3(256), 0, s, winv, U, ldu, (int)n, (int64_t)0, (int64_t)0);
  GPN_LAUNCH_CHECK();
  Ctx c{s, ldl, const_cast<double*>(winv), nullptr, GPN_OK};
  trtri_rec(c, L, ldl, U, ldu, n, 0);
  return c.rc;
}

// U_b = L_b^-T for `batch` factors of one shape: gpn_trtri_upper_ws's schedule with every launch once over all problems
// (n > 256; up to there gpn_trtri_upper problem by problem -- a handful of launches each).  U and S zero-initialised by the
// caller; per problem bit-identical to the single-problem entry points.
extern "C" int gpn_trtri_upper_batched(void* stream, const double* L, int64_t n, int64_t ldl, int64_t sL, const double* winv, int64_t sW,
                                       double* U, int64_t ldu, int64_t sU, double* S, int64_t lds, int64_t sS, int batch) {
  if (!L) return -2;
  if (n < 0) return -3;
  if (ldl < round_up(n, LEAF) || (ldl % LEAF) != 0) return -4;
  if (batch < 1) return -14;
  if (batch > 1 && (sL < n * ldl || (sL & 1))) return -5;
  if (!winv) return -6;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -7;
  if (!U) return -8;
  if (ldu < round_up(n, LEAF) || (ldu % LEAF) != 0) return -9;
  if (batch > 1 && (sU < n * ldu || (sU & 1))) return -10;
  if ((reinterpret_cast<uintptr_t>(L) & 15) || (reinterpret_cast<uintptr_t>(U) & 15)) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  if (n <= 2 * LEAF) {
    for (int z = 0; z < batch; ++z) {
      const int rc = gpn_trtri_upper(stream, L + z * sL, n, ldl, winv + z * sW, U + z * sU, ldu);
      if (rc != GPN_OK) return rc;
    }
    return GPN_OK;
  }
  if (!S) return -11;
  if (lds < round_up(n, LEAF) || (lds % LEAF) != 0) return -12;
  if (batch > 1 && (sS < n * lds || (sS & 1))) return -13;
  if (reinterpret_cast<uintptr_t>(S) & 15) return GPN_E_ALIGN;
  // (the transposes put equal nodes x models into gridDim.z: chunks of models that fit)
  const int64_t max_models = std::max<int64_t>(1, 65535 / std::max<int64_t>(1, n / 256 + 1));
  for (int z0 = 0; z0 < batch; z0 += (int)max_models) {
    const int nb = (int)std::min<int64_t>(max_models, batch - z0);
    const int rc = trtri_upper_ws_batched(static_cast<hipStream_t>(stream), L + z0 * sL, n, ldl, sL, winv + z0 * sW, sW, U + z0 * sU, ldu, sU,
                                          S + z0 * sS, lds, sS, nb);
    if (rc != GPN_OK) return rc;
  }
  return GPN_OK;
}

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

// ---- `batch` factorisations in lock step -----------------------------------------------------------------------------
// The reference evaluates one model per optimiser step (gptorch/models/base.py:260-269); a hyper-parameter search runs
// many independent models of one shape.  Below N ~ 10^4 one factorisation cannot fill the chip -- its chain of N / 128
// leaf steps runs on ONE compute unit for a third of the time -- so B of them share every launch: the leaf runs as a grid
// of B workgroups, the column passes and every contraction as one strided-batch launch.  Same drivers, same kernels,
// same per-entry summation order as gpn_potrf_lower: every factor is bit-identical to its sequential factorisation.
extern "C" int gpn_potrf_lower_batched(void* stream, double* A, int64_t n, int64_t e, int64_t lda, int64_t sA,
                                       double* winv, int64_t sW, int32_t* info, int batch) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < round_up(n + e, LEAF) || (lda % LEAF) != 0) return -5;
  if (batch < 1) return -10;
  if (batch > 1 && (sA < gpn_factor_rows(n, e) * lda || (sA & 1))) return -6;
  if (!winv) return -7;
  if (batch > 1 && sW < gpn_winv_bytes(n) / (int64_t)sizeof(double)) return -8;
  if (!info) return -9;
  if (reinterpret_cast<uintptr_t>(A) & 15) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  Ctx c{static_cast<hipStream_t>(stream), lda, winv, info, GPN_OK};
  c.batch = batch; c.sA = sA; c.sW = sW;
  if (g_potrf_variant == 1 || n <= 2 * LEAF) potrf_rec(c, A, n, e, 0);
  else potrf_lookahead(c, A, n, e);
  return c.rc;
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

extern "C" int gpn_transpose(void* stream, const double* src, int64_t rows, int64_t cols, int64_t lds,
                             double* dst, int64_t ldd) {
  if (!src) return -2;
  if (rows < 0) return -3;
  if (cols < 0) return -4;
  if (lds < cols) return -5;
  if (!dst) return -6;
  if (ldd < rows) return -7;
  if (rows == 0 || cols == 0) return GPN_OK;
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), src, rows, cols, lds, dst, ldd);
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
