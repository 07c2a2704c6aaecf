// Triangular solves and inversions on a factor of potrf.hip: the right-solve X L^T = B (functions.trtrs, functions.py:71-76) by
// recursion down to the inverted 128 x 128 leaf blocks or through inverted 1024 x 1024 blocks, U = L^-T (the backward of the LML,
// SURVEY.md 8(a) a9) by recursion or level-parallel over the recursion tree, their lock-step forms, and the inverse of a GIVEN
// triangular leaf block (gpn_trtri_diag: the first-generation leaf kernel's inverse-only form).
#include <algorithm>
#include <vector>
#include "potrf_ctx.h"

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

}  // namespace gpn

using namespace gpn;

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
