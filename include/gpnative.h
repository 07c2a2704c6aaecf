/*
 * gpnative.h -- C ABI of libgpnative.so: the MI355X (gfx950) exact-GP hot path.
 *
 * The reference (cics-nd/gptorch v0.3.2) is pure Python over PyTorch ops and has
 * NO FFI/plugin layer; this header is the boundary a maintainer would bind
 * (ctypes stub: INTEGRATION.md) underneath the reference's Python call surface.
 * Each entry point names the reference interface it replaces (file:line relative
 * to the reference tree).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to row-major fp64 unless noted `host`;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *    nothing synchronises the host (graph-capturable) unless stated;
 *  - return value: 0 = ok, <0 = -(index of the bad argument) or a GPN_E* code;
 *    never throws, never allocates device memory: workspaces are sized by the
 *    *_work_bytes / gpn_winv_bytes queries and owned by the caller;
 *  - LAPACK-style `info` lives on the device (int32): 0 = ok, j>0 = the leading
 *    minor of order j is not positive definite (first failing pivot, 1-based).
 *    The Python shell maps info>0 to the jitter ladder of functions.py:20-43.
 *    info == GPN_INFO_INTERNAL (< 0) is NOT about the matrix: a kernel-internal
 *    hand-over failed; the shell raises instead of adding jitter.
 *
 * "Factor buffers": dense factorisation kernels work on a caller-owned,
 * ZERO-INITIALISED buffer `A` of `gpn_factor_rows(n,e)` rows and leading
 * dimension `lda = gpn_factor_ld(n,e)` (a multiple of 128), holding the n x n
 * matrix in its top-left corner (lower triangle significant) and `e` optional
 * "extra rows" n..n+e-1 that are carried through the factorisation: on exit
 * they hold (L^-1 * R)^T for the right-hand sides R^T stored in them on entry
 * (forward substitution fused into the panel solves -- this is how
 * alpha = L^-1 (y - m) of gpr.py:62 is produced without a separate TRSV).
 */
#ifndef GPNATIVE_H
#define GPNATIVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPN_VERSION 1

/* kernel kinds: kernels.py:215-222 (Rbf), 204-212 (Matern52), 196-201 (Matern32), 182-193 (Exp/Matern12) */
enum { GPN_RBF = 0, GPN_MATERN52 = 1, GPN_MATERN32 = 2, GPN_EXP = 3,
       GPN_SQDIST = 4 /* util.squared_distance itself (util.py:73-88): K = r^2, variance ignored */,
       GPN_PERIODIC = 5 /* kernels.py:228-235: variance * cos(r) */ };
enum { GPN_FULL = 0, GPN_LOWER = 1 };

enum {
  GPN_OK = 0,
  GPN_E_HIP = -100,      /* a HIP runtime call failed: see gpn_last_hip_error() */
  GPN_E_ALIGN = -101,    /* pointer / leading dimension violates an alignment rule */
  GPN_E_UNSUPPORTED = -102
};
/* device-side `info` value for an internal failure of the leaf kernel (see conventions) */
#define GPN_INFO_INTERNAL (-1000)

int gpn_version(void);
/* name of the gfx target compiled into the library, e.g. "gfx950" */
const char* gpn_arch(void);
/* text of the last HIP error seen by this thread ("" if none) */
const char* gpn_last_hip_error(void);

/* ---- factor-buffer geometry --------------------------------------------- */
int64_t gpn_factor_ld(int64_t n, int64_t e);    /* round_up(n+e, 128) */
int64_t gpn_factor_rows(int64_t n, int64_t e);  /* round_up(n+e, 128) + 16 (zero apron) */
/* bytes of the `winv` workspace (inverses of the 128x128 diagonal leaf blocks) */
int64_t gpn_winv_bytes(int64_t n);

/* ---- K assembly ------------------------------------------------------------
 * Replaces util.squared_distance (util.py:73-88), Stationary.squared_dist/dist
 * (kernels.py:149-172), Rbf.K / Matern52.K / Matern32.K / Exp.K (kernels.py:
 * 182-222) and, with `noise` != NULL, GPR._compute_kyy (gpr.py:69-86).
 *   K[i,j] = variance * k(|| (X_i - X2_j) / ell ||)   (+ noise on i==j when X2==NULL)
 * X[n,d], X2[m,d] (NULL => X2 = X, m ignored) contiguous row-major;
 * variance[1], length_scales[nls] (nls = 1 or d), noise[1] or NULL: device scalars
 * in CONSTRAINED space.  uplo = GPN_LOWER (only with X2 == NULL) writes tiles on
 * or below the diagonal only.  Distances are direct differences (no Gram trick),
 * so r^2 >= 0 by construction (the clamp of util.py:88 is the identity). */
int gpn_kernel_matrix(void* stream, int kind,
                      const double* X, int64_t n, const double* X2, int64_t m, int d,
                      const double* variance, const double* length_scales, int nls,
                      const double* noise, int uplo, double* K, int64_t ldk);

/* extra rows of a factor buffer: E[c, i] = Y[i, c] - M[i, c] (M may be NULL),
 * Y, M [n,dy] row-major contiguous; E = A + n*lda.  (gpr.py:62 `y - mean(x)`) */
int gpn_pack_rhs(void* stream, const double* Y, const double* M, int64_t n, int dy,
                 double* E, int64_t lde);

/* ---- Cholesky --------------------------------------------------------------
 * Replaces functions.cholesky's torch.cholesky call (functions.py:46-47; the
 * jitter ladder of functions.py:20-43 stays in the Python shell and replays on
 * info>0).  In-place lower factorisation of the n x n top-left block of the
 * factor buffer A; the strict upper triangle is neither read nor written.
 * `winv` (gpn_winv_bytes(n)) receives the inverses of the 128x128 diagonal blocks
 * of L and must be kept with L for the solves below.  e extra rows are carried
 * (see header comment).  *info must be 0 on entry. */
int gpn_potrf_lower(void* stream, double* A, int64_t n, int64_t e, int64_t lda,
                    double* winv, int32_t* info);

/* The same factorisation for a tile column of a LARGER matrix: the e rows below the n x n block (any
 * number -- a panel of the enclosing matrix) come out as R L^-T and nothing right of column n is read or
 * written; lda >= round_up(n, 128).  What the block-cyclic drivers call on a diagonal tile whose panel
 * rows live on the same GPU (gptorch_amd/dist.py, gpn_dist_lml_forward). */
int gpn_potrf_lower_panel(void* stream, double* A, int64_t n, int64_t e, int64_t lda,
                          double* winv, int32_t* info);

/* gpn_potrf_lower as ONE PERSISTENT LAUNCH (csrc/ppotrf.hip, round 6): a dataflow over 128 x 128 tiles -- leaf, tile solve
 * and tile update tasks with the K grouping and per-entry summation order of gpn_potrf_lower's schedule, so the factor, the
 * leaf inverses and the extra rows are BIT-IDENTICAL to gpn_potrf_lower's -- run by one workgroup per compute unit, the first
 * few of which serve only the critical chain (leaf -> solve -> update of the next diagonal block) while the others stream the
 * trailing updates: the N/128 leaves no longer run alone on the chip.  For the latency regime of one model per optimiser step
 * (gptorch/models/base.py:260-269; functions.py:46-47): gpn_potrf_persistent_supported(n, e) says which sizes it takes
 * (n a multiple of 128 in [2560, 20480), e <= 16); for any other size, and on a stream under capture before the size's first
 * call outside capture, it returns GPN_E_UNSUPPORTED and nothing is enqueued.  The task table of a size is built on the host
 * and kept on the device by the library (a few hundred KB per size; a runtime area per caller stream and size:
 * gpn_release_stream frees those).  Same contract for A, winv, info as gpn_potrf_lower; an internal failure (a bounded spin
 * that ran out) is reported as info = GPN_INFO_INTERNAL. */
int gpn_potrf_lower_persistent(void* stream, double* A, int64_t n, int64_t e, int64_t lda,
                               double* winv, int32_t* info);
int gpn_potrf_persistent_supported(int64_t n, int64_t e);
/* The task graph of gpn_potrf_lower_persistent, for inspection (tests replay it on the host in random valid orders):
 * counts (3 + 64 entries) = {tasks, successor entries, queues in use, tasks of queue 0 (the critical chains), 1, ...: queue
 * 1 + p = tasks whose output tile row lies in outer panel p; the extra-rows tile last}; tasks12 = 12 ints per task:
 * type (0 leaf, 1 tile solve, 2 tile update, 3 chain step = solve of (c, c-1) + last update of (c, c) + leaf(c), 4 the
 * diagonal tile's partial sums, 5 = 64 rows of the last block of the update of tile (c+1, c)), queue, tile row i, tile column j, K range [k0, k1) in 128-column blocks, predecessor count,
 * flags (1: raw sums to scratch instead of the update, 2: continue from the scratch sums, 4: rows 64 .. 127), succ_begin, succ_mid, succ_end, 0 --
 * the successors succ[succ_begin .. succ_mid) are released when a step's solved tile is out, succ[succ_mid .. succ_end) at the
 * end of the task.  succ and tasks12 are filled when given (capacities in entries).  Tasks are listed in a valid sequential
 * order. */
int gpn_potrf_persistent_plan(int64_t n, int64_t e, int64_t* counts, int32_t* tasks12, int64_t cap_tasks, int32_t* succ,
                              int64_t cap_succ);

/* `batch` factorisations of identical shape in LOCK STEP: problem b at A + b*sA (a factor buffer each: sA >=
 * gpn_factor_rows(n,e)*lda, even), winv + b*sW (sW >= gpn_winv_bytes(n)/8), info[b].  Same drivers and kernels as
 * gpn_potrf_lower with every launch covering all problems -- the 128x128 leaf as a grid of `batch` workgroups, the
 * column passes and contractions as strided-batch launches -- and the same summation order per entry: factor b is
 * BIT-IDENTICAL to gpn_potrf_lower on problem b alone.  Replaces `batch` sequential torch.cholesky calls
 * (functions.py:46-47) of models evaluated one per optimiser step (gptorch/models/base.py:260-269): below
 * N ~ 10^4 one factorisation leaves most of the chip idle during its N/128 leaf steps. */
int gpn_potrf_lower_batched(void* stream, double* A, int64_t n, int64_t e, int64_t lda, int64_t sA,
                            double* winv, int64_t sW, int32_t* info, int batch);
/* gpn_lml_reduce for `batch` factor buffers at stride sA: out3[3*b .. 3*b+2]. */
int gpn_lml_reduce_batched(void* stream, const double* A, int64_t n, int64_t e, int64_t lda, int64_t sA,
                           double* out3, int batch);

/* (Outer) panel width the driver uses for an n x n factorisation (0 = the recursive driver): each such panel ends
 * with one lower-tile K = width contraction -- the SYRK trailing update priced by bench.py. */
int64_t gpn_potrf_panel_width(int64_t n);
/* All nesting levels of the driver's panels, innermost first (potrf.hip panel_levels): widths3[0] = the inner panels
 * the leaf chain runs in, widths3[count-1] = gpn_potrf_panel_width(n); returns the count (0 = the recursive driver). */
int gpn_potrf_panel_levels(int64_t n, int64_t* widths3);
/* Release the helper streams/events the library keeps for `stream` (created by the first
 * factorisation / distributed evaluation enqueued on it); call before destroying the stream.
 * NULL = release all. */
int gpn_release_stream(void* stream);

/* winv <- inverses of the 128x128 diagonal blocks of a GIVEN lower-triangular L
 * (n x n, row-major, ldl): what functions.trtrs (functions.py:71-76) needs when
 * its triangular argument did not come from gpn_potrf_lower.  info (may be NULL):
 * j>0 = zero pivot at column j. */
int gpn_trtri_diag(void* stream, const double* L, int64_t n, int64_t ldl, double* winv, int32_t* info);

/* X * L^T = B  in place on B[m,n] (row-major, ldb; rows/ld padded like a factor
 * buffer), i.e. X^T = L^-1 B^T: functions.trtrs(b, L) (functions.py:71-76) for
 * b = B^T.  L/winv from gpn_potrf_lower. */
int gpn_trsm_right_lt(void* stream, const double* L, int64_t n, int64_t ldl,
                      const double* winv, double* B, int64_t m, int64_t ldb);

/* out[0] = sum_i log L[i,i]  (functions.lt_log_determinant, functions.py:61-68)
 * out[1] = sum of squares of the e x n extra rows (= ||alpha||_F^2, gpr.py:66)
 * out[2] = LML of gpr.py:63-67 = -0.5*out[1] - e*out[0] - 0.5*e*n*log(2*pi) */
int gpn_lml_reduce(void* stream, const double* A, int64_t n, int64_t e, int64_t lda,
                   double* out3);

/* ---- dense contractions (exposed for tests and for the predict path) -------
 * C[M,N] = alpha * A[M,K] * B[N,K]^T + beta * C   (all row-major; "NT" form).
 * lower != 0: square case M == N, only tiles on/below the diagonal are computed
 * and entries with j > i are not written (SYRK-style trailing update).
 * Requirements: K % 16 == 0; lda, ldb % 2 == 0; A, B 16-byte aligned; rows of A
 * (B) readable up to round_up(M,16) (round_up(N,16)). */
/* tri: structure hints that clip the K range per tile (operand entries in the clipped
 * range MUST be zero): A_UPPER: A[i,k] = 0 for k < i; A_LOWER: A[i,k] = 0 for k > i
 * (to tile granularity: k >= 128*ceil((i+1)/128)); likewise for B with its row index j. */
enum { GPN_TRI_A_UPPER = 1, GPN_TRI_A_LOWER = 2, GPN_TRI_B_UPPER = 4, GPN_TRI_B_LOWER = 8 };
/* lower == 2 ("trapezoid", M >= N, tri == 0): the N x N top square is treated as with lower == 1 and the
 * (M - N) x N rectangle below it is computed whole -- one tile column of a block-cyclic trailing update including
 * its diagonal tile, in ONE launch. */
int gpn_gemm_nt(void* stream, int64_t M, int64_t N, int64_t K, double alpha,
                const double* A, int64_t lda, const double* B, int64_t ldb,
                double beta, double* C, int64_t ldc, int lower, int tri);

/* "Staircase": C is M x (nblocks * blk) and column block b (blk columns, blk % 128 == 0) only has the rows from
 * b * step on (step % 128 == 0); with diag != 0 the first blk x blk square of every block is lower-only (entries
 * above its diagonal are not written).  All the local tile columns of one block-cyclic trailing update -- each
 * starting Pc/Pr tile rows below its left neighbour -- in ONE launch (gptorch_amd/dist.py, csrc/dist.hip), instead
 * of one launch with its own partial last round per tile column.  B is [nblocks * blk, K].  With diag every block that
 * has rows at all must have at least blk of them (status -15 otherwise: a partial diagonal square is not supported). */
int gpn_gemm_nt_stair(void* stream, int64_t M, int64_t nblocks, int64_t blk, int64_t K, double alpha,
                      const double* A, int64_t lda, const double* B, int64_t ldb,
                      double beta, double* C, int64_t ldc, int64_t step, int diag);

/* `batch` problems of identical shape in ONE launch: problem z uses A + z*sA, B + z*sB, C + z*sC
 * (strides in elements; sA, sB even).  With sA = sB = K and one set of rows this is a split-K
 * contraction into `batch` partial results -- how the sparse model's M x M accumulation
 * (sparse_gpr.py:131 `AAT`) fills the GPU when M^2 alone has too few tiles. */
int gpn_gemm_nt_batched(void* stream, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                        double beta, double* C, int64_t ldc, int64_t sC, int lower, int tri, int batch);

/* gpn_gemm_nt_batched with one scale per problem, read from DEVICE memory: C_z = alphas[z] A_z B_z^T + beta C_z; per problem
 * bit-identical to gpn_gemm_nt(alpha = alphas[z]).  Lock-step sparse models scale by their own 1 / noise variance
 * (sparse_gpr.py:131-135 divides by sigma per model). */
int gpn_gemm_nt_batched_scaled(void* stream, int64_t M, int64_t N, int64_t K, const double* alphas,
                               const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB,
                               double beta, double* C, int64_t ldc, int64_t sC, int lower, int tri, int batch);

/* ---- lock-step forms of the single-purpose entry points (round 6) -------------
 * `batch` models of one shape, every launch once over all of them: model b at pointer + b * stride (strides in doubles;
 * sX = 0: shared points), hyper-parameters consecutive (variance[b], length_scales + b nls, noise[b]).  Each is per model
 * BIT-IDENTICAL to its single-model entry point.  They carry the sparse bound (sparse_gpr.py:108-153) of B restarts in lock
 * step -- K(Z_b), K(x, Z_b), the right-solve against chol K(Z_b), L_b^-T, the sweeps against dF/dKuu and dF/dKuf -- where the
 * reference runs one model per optimiser step (models/base.py:260-269). */
int gpn_kernel_matrix_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n,
                              const double* X2, int64_t sX2, int64_t m, int d,
                              const double* variance, const double* length_scales, int nls, const double* noise,
                              int uplo, double* K, int64_t ldk, int64_t sK);
int gpn_trsm_right_lt_batched(void* stream, const double* L, int64_t n, int64_t ldl, int64_t sL, const double* winv, int64_t sW,
                              double* B, int64_t m, int64_t ldb, int64_t sB, int batch);
/* U, S zero-initialised by the caller (S may be NULL up to n = 256) */
int gpn_trtri_upper_batched(void* stream, const double* L, int64_t n, int64_t ldl, int64_t sL, const double* winv, int64_t sW,
                            double* U, int64_t ldu, int64_t sU, double* S, int64_t lds, int64_t sS, int batch);
/* out [batch, 1 + nls]; work: batch * gpn_grad_work_bytes(n, m, nls, 0) */
int gpn_kernel_grad_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n,
                            const double* X2, int64_t sX2, int64_t m, int d,
                            const double* variance, const double* length_scales, int nls,
                            const double* G, int64_t ldg, int64_t sG, double* work, double* out);
/* out [batch, m, d]; work: batch * gpn_grad_x2_work_bytes(n, m, d) */
int gpn_kernel_grad_x2_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n,
                               const double* X2, int64_t sX2, int64_t m, int d,
                               const double* variance, const double* length_scales, int nls,
                               const double* G, int64_t ldg, int64_t sG, double scale, int accumulate,
                               double* work, double* out);

/* ---- backward of the LML (closed form) ---------------------
 * U <- L^-T (upper triangular, row-major) into a ZERO-INITIALISED buffer of
 * gpn_factor_rows(n,0) x ldu (ldu = gpn_factor_ld(n,0)); L/winv from gpn_potrf_lower.
 * Then Kyy^-1 = U U^T is one SYRK-style gpn_gemm_nt(lower=1, tri=A_UPPER|B_UPPER)
 * and a = Kyy^-1 (y-m) = U alpha one gpn_gemm_nt(tri=B_UPPER).  Together they replace
 * PyTorch's CholeskyBackward0 / TriangularSolveBackward0 under gpr.py:47-67. */
int gpn_trtri_upper(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                    double* U, int64_t ldu);
/* Same result, throughput-oriented: with a ZERO-INITIALISED scratch matrix S of the same shape
 * as U the off-diagonal blocks are U12 = -(U11 L21^T) (U22^T)^T -- two NT contractions and an
 * HBM-bound transpose per node of the recursion tree, no right-solve chain -- and the
 * equal-shaped nodes of one tree depth go out as one strided-batch launch per operation. */
int gpn_trtri_upper_ws(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv,
                       double* U, int64_t ldu, double* S, int64_t lds);

/* workspace bytes for the two gradient sweeps below (lml != 0: gpn_lml_grad) */
int64_t gpn_grad_work_bytes(int64_t n, int64_t m, int nls, int lml);

/* out[0] = dLML/d variance, out[1..nls] = dLML/d length_scales, out[1+nls] = dLML/d noise
 * (all w.r.t. the CONSTRAINED values) from G = 1/2 (a a^T - dy Kinv), Kinv lower [n,n],
 * at = a^T [dy, n]; K and dK/dtheta are recomputed from X on the fly (kernels.py:149-222). */
int gpn_lml_grad(void* stream, int kind, const double* X, int64_t n, int d,
                 const double* variance, const double* length_scales, int nls,
                 const double* Kinv, int64_t ldk, const double* at, int64_t ldat, int dy,
                 double* work, double* out);
/* gpn_lml_grad for `batch` lock-step models as one sweep launch + one reduction launch: model b reads X + b*sX (0: shared),
 * variance[b], length_scales[b*nls ..], Kinv + b*sK, at + b*sAt and writes out[b*(2+nls) ..]; work: batch *
 * gpn_grad_work_bytes(n, n, nls, 1).  Bit-identical per model to gpn_lml_grad. */
int gpn_lml_grad_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                         const double* variance, const double* length_scales, int nls,
                         const double* Kinv, int64_t ldk, int64_t sK, const double* at, int64_t ldat, int64_t sAt, int dy,
                         double* work, double* out);

/* autograd backward of gpn_kernel_matrix: out[0] = sum G*dK/dvariance,
 * out[1..nls] = sum G*dK/dlength_scales for a given dense G [n, m]. */
int gpn_kernel_grad(void* stream, int kind, const double* X, int64_t n, const double* X2, int64_t m, int d,
                    const double* variance, const double* length_scales, int nls,
                    const double* G, int64_t ldg, double* work, double* out);

/* autograd backward of gpn_kernel_matrix w.r.t. the POINTS of the second argument
 * (what the reference gets from autograd through util.py:73-88 / kernels.py:149-222, e.g.
 * for the inducing points Z in sparse_gpr.py:126-129):
 *     out[j, c] (+)= scale * sum_i G[i, j] * dK(x_i, z_j)/dz_jc      out [m, d] contiguous
 * accumulate != 0 adds to out.  For the gradient w.r.t. X pass G^T with the roles swapped;
 * for a symmetric K(Z, Z) with symmetric G pass X = X2 = Z and scale = 2.
 * work: gpn_grad_x2_work_bytes(n, m, d) bytes. */
int64_t gpn_grad_x2_work_bytes(int64_t n, int64_t m, int d);
int gpn_kernel_grad_x2(void* stream, int kind, const double* X, int64_t n, const double* X2, int64_t m, int d,
                       const double* variance, const double* length_scales, int nls,
                       const double* G, int64_t ldg, double scale, int accumulate,
                       double* work, double* out);

/* ---- composite covariance functions (kernels.py:286-306 Sum / Product over the leaves below) in one pass --------
 * The reference composes `k1 + k2`, `k1 * k2` with elementwise torch ops on dense N x M matrices, one assembly chain per
 * leaf -- e.g. its own example model Linear + Rbf + Constant (examples/regression_1d.py:34-53).  Here the expression is
 * expanded by the caller into a SUM OF PRODUCTS of leaf terms,  K = sum_g prod_{t in group g} term_t,  and evaluated per
 * tile: the matrix is written once (uplo / noise / ldk exactly as gpn_kernel_matrix, so a GPR over a composite kernel
 * assembles straight into the factor buffer).  Leaves: a stationary kernel (kernels.py:108-235; kind = GPN_RBF ...
 * GPN_PERIODIC, variance at theta[var_off], length-scale(s) at theta[ls_off .. +nls), nls in {1, d}); Linear
 * (kernels.py:238-265: sum_d x_d v_d x'_d, v at theta[var_off .. +nvar), nvar in {1, d}); Constant / Bias
 * (kernels.py:95-105: theta[var_off]); White (kernels.py:83-92: theta[var_off] on the diagonal of K(X), 0 for K(X, X2)).
 * theta: ONE device array holding every leaf's constrained parameter values. */
enum { GPN_TERM_STATIONARY = 0, GPN_TERM_LINEAR = 1, GPN_TERM_CONSTANT = 2, GPN_TERM_WHITE = 3 };
enum { GPN_EXPR_MAX_TERMS = 16, GPN_EXPR_MAX_GROUPS = 8 };
typedef struct gpn_expr_term {
  int type;      /* GPN_TERM_* */
  int kind;      /* stationary: GPN_RBF ... GPN_PERIODIC */
  int var_off;   /* offset of the variance(s) in theta */
  int ls_off;    /* stationary: offset of the length-scale(s) in theta */
  int nls;       /* stationary: 1 or d */
  int nvar;      /* linear: 1 or d (1 otherwise) */
} gpn_expr_term;
/* terms[nterms] (host), group_start[ngroups + 1] (host): group g = terms group_start[g] .. group_start[g+1]-1. */
int gpn_kernel_matrix_expr(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                           const double* theta, const double* X, int64_t n, const double* X2, int64_t m, int d,
                           const double* noise, int uplo, double* K, int64_t ldk);
/* The matching backward sweep for ONE leaf instance `target` (index into terms):
 *     out[p] = sum_ij W_ij * prod_{t' != target in its group} term_t'(x_i, x_j) * d term_target(x_i, x_j) / d theta_p
 * with p over the target's parameters in the order variance, length-scale(s) (stationary) / variance(s) (linear) /
 * variance (constant, white), everything re-computed from the points.  W is either a given dense [n, m] matrix G
 * (at == NULL: the autograd backward of Kernel.K) or, with at != NULL (X2 must be NULL), the closed-form dLML/dKyy =
 * 1/2 (a a^T - dy Kyy^-1) formed on the fly from G = Kyy^-1 (lower triangle read) and a^T [dy, n]; then want_trace != 0
 * appends trace(W) (= dLML/d noise) as one more output.  One launch reads W once; a kernel of T leaf instances costs T
 * launches.  Per-dimension parameters (ARD length-scales, Linear with one variance per input) need d <= 16
 * (GPN_E_UNSUPPORTED beyond).  work: gpn_kernel_expr_grad_work_bytes(n, m, d, lml) bytes. */
/* Lock-step forms for `batch` models whose expressions have ONE structure (the same term table, every model its own parameter
 * values theta + b sTheta) -- the reference's example model Linear + Rbf + Constant (examples/regression_1d.py:34-53) in a
 * multi-start search: Kyy_b (symmetric, lower tiles, noise[b] on the diagonal) into K + b sK in one launch, and the gradient sweep
 * of one leaf instance in LML mode (G + b sG = Kyy_b^-1, at + b sAt = a_b^T) over all models as one sweep + one reduction launch
 * (out [batch, nout]; work: batch * gpn_kernel_expr_grad_work_bytes(n, n, d, 1)).  Per model bit-identical to the single forms. */
int gpn_kernel_matrix_expr_batched(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                   int batch, const double* theta, int64_t sTheta, const double* X, int64_t sX, int64_t n, int d,
                                   const double* noise, double* K, int64_t ldk, int64_t sK);
int gpn_kernel_expr_grad_batched(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                                 int batch, const double* theta, int64_t sTheta, int target, const double* X, int64_t sX,
                                 int64_t n, int d, const double* G, int64_t ldg, int64_t sG, const double* at, int64_t ldat,
                                 int64_t sAt, int dy, int want_trace, double* work, double* out);
int64_t gpn_kernel_expr_grad_work_bytes(int64_t n, int64_t m, int d, int lml);
int gpn_kernel_expr_grad(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                         const double* theta, int target, const double* X, int64_t n, const double* X2, int64_t m, int d,
                         const double* G, int64_t ldg, const double* at, int64_t ldat, int dy, int want_trace,
                         double* work, double* out);

/* ---- whole-path entry points: one call per reference method -------------------
 * Fixed sequences of the entry points above on ONE stream (no host synchronisation, no
 * allocation) for callers that do not want to issue them one by one.
 *
 * gpn_lml_forward = GPR.log_likelihood (gpr.py:47-67): K(X)+noise*I assembled into the lower
 * triangle of the factor buffer A (gpn_factor_rows(n,dy) x lda, lda = gpn_factor_ld(n,dy),
 * zero-initialised ONCE by the caller; reusable across calls), (Y - M)^T packed into the extra
 * rows, factorisation with fused forward substitution, reductions.  out3 as gpn_lml_reduce
 * (out3[2] = LML).  *info is cleared and set here: info > 0 => replay with noise + 10^(-10+i)
 * (functions.py:20-43).  A/winv afterwards hold L, alpha^T and the leaf inverses for the two
 * calls below. */
int gpn_lml_forward(void* stream, int kind, const double* X, int64_t n, int d,
                    const double* Y, const double* M, int dy,
                    const double* variance, const double* length_scales, int nls,
                    const double* noise, double* A, int64_t lda, double* winv,
                    int32_t* info, double* out3);

/* gpn_lml_forward that ALSO keeps a pristine copy of the lower triangle of Kyy (noise on the diagonal) in Ksave [n, lda] -- the
 * factorisation overwrites it in A.  With it the refinement step reads the matrix back,
 *     gpn_lml_refine_dense(stream, Ksave, lda, 0.0, n, Y, M, dy, A, lda, winv, work, out3),
 * instead of re-computing every kernel entry for its residual pass as gpn_lml_refine does (same entries bit for bit, same
 * partial sums: the same refined value).  Costs n * lda * 8 bytes of memory and a second stream of writes in the assembly. */
int gpn_lml_forward_saving(void* stream, int kind, const double* X, int64_t n, int d,
                           const double* Y, const double* M, int dy,
                           const double* variance, const double* length_scales, int nls,
                           const double* noise, double* A, int64_t lda, double* winv,
                           int32_t* info, double* out3, double* Ksave);

/* gpn_lml_forward for `batch` models of one shape (n, d, dy, nls) in lock step -- hyper-parameter restarts: the reference
 * evaluates one model per optimiser step (gptorch/models/base.py:260-269).  Model b: points X + b*sX (sX = 0: shared),
 * targets Y + b*sY (sY = 0: shared), mean values M + b*sM (M may be NULL), variance[b], length_scales[b*nls ..],
 * noise[b]; factor buffer A + b*sA, leaf inverses winv + b*sW (see gpn_potrf_lower_batched), info[b], out3[3*b ..].
 * One assembly launch, one right-hand-side launch, the batched factorisation, one reduction launch; every model's
 * out3 / factor / info is bit-identical to gpn_lml_forward on that model alone.  info[b] > 0: replay THAT model through
 * gpn_lml_forward with noise + 10^(-10+i) (functions.py:20-43). */
int gpn_lml_forward_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                            const double* Y, int64_t sY, const double* M, int64_t sM, int dy,
                            const double* variance, const double* length_scales, int nls, const double* noise,
                            double* A, int64_t lda, int64_t sA, double* winv, int64_t sW, int32_t* info, double* out3);

/* gpn_lml_refine: one step of iterative refinement of the quadratic form of gpr.py:61-67, to be called after a
 * gpn_lml_forward that returned info == 0 (same arguments; A / winv read only).  The factor satisfies
 * L L^T = Kyy + E, so |alpha|^2 = y^T (Kyy + E)^-1 y carries -a^T E a -- a few 1e-9 ABSOLUTE at N = 32768, where
 * |LML| = 1.5e5 and north_star's 1e-8 is 7e-14 relative (the reference's own fp64 value is 3.4e-9 from the exact one
 * there).  This call computes a_hat = L^-T alpha (back-substitution), r = (y - m) - Kyy a_hat with Kyy re-computed
 * from the points and double-double accumulation, and replaces out3[1] by y^T a_hat + a_hat^T r (exact up to
 * r^T Kyy^-1 r = O(|E|^2)) and out3[2] by the LML with it; out3[0] (log-det) is kept.  Cost: one pass over L in
 * n/128 dependent launches + one pass of kernel evaluations (about 3 % of an evaluation at N = 32768); pointless
 * below N of about 10^4, where the plain value is already within 1e-9.  work: gpn_lml_refine_work_bytes(n, dy). */
/* gpn_lml_forward_batched / gpn_lml_backward_batched for models of DIFFERENT sizes (round 6: cross-validation folds of unequal
 * length, learning curves; the reference evaluates one model at a time, models/base.py:260-269).  Model b has n_of[b] <= n points
 * (n_of: DEVICE array of int32, every entry > 256), its points at X + b sX and right-hand sides at Y + b sY, both padded to n rows
 * (the padding is not read into any result).  It is evaluated as the n x n problem [Kyy_b 0; 0 I] with zero right-hand sides on
 * the identity rows, by the same launches as an equal-size batch: factor, alpha, out3 and gradients are BIT-IDENTICAL to
 * gpn_lml_forward / gpn_lml_backward on the model's own n_of[b] points, provided n and all n_of[b] select the same panel levels
 * (gpn_potrf_panel_levels; both sides of 2048 rows differ) and stay below the refinement threshold.  Zero mean only (M = NULL).
 * backward work: gpn_lml_backward_batched_work_bytes(n, dy, nls, batch). */
int gpn_lml_forward_ragged(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, const int32_t* n_of, int d,
                           const double* Y, int64_t sY, int dy,
                           const double* variance, const double* length_scales, int nls, const double* noise,
                           double* A, int64_t lda, int64_t sA, double* winv, int64_t sW, int32_t* info, double* out3);
int gpn_lml_backward_ragged(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, const int32_t* n_of, int d,
                            const double* variance, const double* length_scales, int nls,
                            const double* A, int64_t lda, int64_t sA, const double* winv, int64_t sW, int dy,
                            double* work, double* grads);

int64_t gpn_lml_refine_work_bytes(int64_t n, int dy);
int gpn_lml_refine(void* stream, int kind, const double* X, int64_t n, int d,
                   const double* Y, const double* M, int dy,
                   const double* variance, const double* length_scales, int nls, const double* noise,
                   const double* A, int64_t lda, const double* winv, double* work, double* out3);

/* gpn_lml_refine for a covariance EXPRESSION (gpn_kernel_matrix_expr's program: the composite kernels of kernels.py:286-306),
 * after a factorisation of that expression's Kyy that carried the residual as extra rows: same step, the residual pass evaluates
 * the expression per tile exactly as the fused assembly did.  out3[0] = sum log L_ii on entry; work: gpn_lml_refine_work_bytes. */
int gpn_lml_refine_expr(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                        const double* theta, const double* X, int64_t n, int d, const double* Y, const double* M, int dy,
                        const double* noise, const double* A, int64_t lda, const double* winv, double* work, double* out3);

/* ... and for a Kyy that exists as a dense symmetric matrix K [n, n] (factorised as K + diag_add I; the dense-K fall-back of GPR). */
int gpn_lml_refine_dense(void* stream, const double* K, int64_t ldk, double diag_add, int64_t n, const double* Y, const double* M, int dy,
                         const double* A, int64_t lda, const double* winv, double* work, double* out3);

/* gpn_refine_resid_part for a covariance expression (gpn_expr_term program): the share of the lower 64 x 64 tiles q0 <= q < q1 in
 * Kyy a with Kyy re-computed from the points in double-double -- a rank's contribution to the block-cyclic refinement step when
 * the model's kernel is a Sum / Product tree (DistGPR over the reference's example model, examples/regression_1d.py:34-53).
 * work: gpn_refine_resid_part_work_bytes(dy, q1 - q0) bytes; ka [dy][round_up(n, 128)][2] (hi, lo). */
int gpn_refine_resid_part_expr(void* stream, const gpn_expr_term* terms, int nterms, const int* group_start, int ngroups,
                               const double* theta, const double* X, int64_t n, int d, const double* noise,
                               const double* a, int dy, int64_t q0, int64_t q1, double* work, double* ka);

/* The same refinement step in pieces, for a factor that is spread over several GPUs (gptorch_amd/dist.py
 * BlockCyclicGP._refine: 2-D block-cyclic tiles; the exchange between the pieces is the caller's).  Vectors are
 * [dy][ld] row-major, one right-hand side per row.
 *   (a diagonal tile's own solve a_J = L_JJ^-T s_J is gpn_trtri_diag + gpn_trtri_upper once + one gpn_gemm_nt per use)
 *   gpn_gemv_t_acc     c[cc][col] += sum_r L[r][col] a[cc][r] for a rows x cols block of the local factor: what the tile
 *                      row of a back-substituted block owes the blocks left of it (fixed summation order);
 *                      work: gpn_gemv_t_work_bytes(rows, cols, dy)
 *   gpn_refine_resid_part   Kyy a_hat restricted to the lower 64 x 64 tiles q0 <= q < q1 of gpn_refine_tile_count(n)
 *                      (row-major enumeration of the lower triangle; each tile also feeds its mirror entries), Kyy
 *                      re-computed from the points, double-double: ka[dy][round_up(n,128)][2] = (hi, lo) per row.  The
 *                      shares of all ranks add up to Kyy a_hat; a rank's share does not depend on where the factor's
 *                      tiles live.  work: gpn_refine_resid_part_work_bytes(dy, q1 - q0)
 *   gpn_refine_finish  out3[1] = y^T a_hat + a_hat^T r, out3[2] = the LML with it, r = (y - m) - ka in double-double;
 *                      out3[0] = sum log L_ii on entry. */
int64_t gpn_gemv_t_work_bytes(int64_t rows, int64_t cols, int dy);
int gpn_gemv_t_acc(void* stream, const double* L, int64_t ld, int64_t rows, int64_t cols, const double* a, int64_t lda,
                   int dy, double* c, int64_t ldc, double* work);
int64_t gpn_refine_tile_count(int64_t n);
int64_t gpn_refine_resid_part_work_bytes(int dy, int64_t ntiles);
int gpn_refine_resid_part(void* stream, int kind, const double* X, int64_t n, int d,
                          const double* variance, const double* length_scales, int nls, const double* noise,
                          const double* a, int dy, int64_t q0, int64_t q1, double* work, double* ka);
int gpn_refine_finish(void* stream, const double* Y, const double* M, const double* a, const double* ka, int64_t n, int dy,
                      double* out3);

/* gpn_lml_backward = the autograd backward of gpr.py:47-67 in closed form (what PyTorch's
 * CholeskyBackward0 + TriangularSolveBackward0 + elementwise chain compute for the reference):
 * U = L^-T, Kyy^-1 = U U^T, a = U alpha, one sweep.  grads[0] = dLML/d variance,
 * grads[1..nls] = dLML/d length_scales, grads[1+nls] = dLML/d noise (CONSTRAINED values);
 * grad_resid (may be NULL) [n, dy] = dLML/d(y - m) = -a.  A/winv from gpn_lml_forward with
 * info == 0; work: gpn_lml_backward_work_bytes(n, dy, nls) bytes (two factor-sized matrices). */
int64_t gpn_lml_backward_work_bytes(int64_t n, int dy, int nls);
int gpn_lml_backward(void* stream, int kind, const double* X, int64_t n, int d,
                     const double* variance, const double* length_scales, int nls,
                     const double* A, int64_t lda, const double* winv, int dy,
                     double* work, double* grads, double* grad_resid);

/* gpn_lml_backward for the `batch` models of a gpn_lml_forward_batched call, in lock step: the reference's training loop is
 * loss(); backward(); step() one model at a time (gptorch/models/base.py:260-269; the backward itself: gpr.py:47-67 through
 * autograd).  Model b: points X + b*sX (sX = 0: shared), variance[b], length_scales[b*nls ..], factor A + b*sA and leaf
 * inverses winv + b*sW exactly as gpn_lml_forward_batched left them (info[b] == 0).  Every launch of gpn_lml_backward's
 * schedule (leaf transposes, level-parallel triangular inversion, Kyy^-1 = U U^T, a^T = alpha^T U^T, gradient sweep,
 * reduction) goes out once over all models; grads [batch, 2 + nls] and grad_resid [batch, n, dy] (may be NULL) are
 * BIT-IDENTICAL per model to gpn_lml_backward on that model alone.  work: gpn_lml_backward_batched_work_bytes bytes
 * (2 factor-sized matrices per model). */
int64_t gpn_lml_backward_batched_work_bytes(int64_t n, int dy, int nls, int batch);
int gpn_lml_backward_batched(void* stream, int kind, int batch, const double* X, int64_t sX, int64_t n, int d,
                             const double* variance, const double* length_scales, int nls,
                             const double* A, int64_t lda, int64_t sA, const double* winv, int64_t sW, int dy,
                             double* work, double* grads, double* grad_resid);

/* The kernel-INDEPENDENT half of gpn_lml_backward_batched as an entry point of its own: for each of the `batch` factors of a
 * lock-step factorisation (gpn_potrf_lower_batched layout: model b at A + b*sA, winv + b*sW, alpha^T in the extra rows)
 * Kyy^-1 = U U^T (lower triangle) and a^T = alpha^T U^T, every launch once over all models, per model bit-identical to the
 * sequence gpn_trtri_upper_ws + gpn_gemm_nt(lower, A_UPPER|B_UPPER) + gpn_gemm_nt(B_UPPER) on that factor alone.  For
 * callers whose dKyy/dtheta is not a native stationary kind: composite kernels (kernels.py:286-306) sweep their own
 * expression against these with gpn_kernel_expr_grad, model by model (gptorch_amd/_expr.py BatchedExprLogLik).
 * gpn_lml_kinv_layout: out4 = {leading dimension, offset of Kyy^-1 [n, ld], offset of a^T [dy, ld], doubles from one model's
 * block of `work` to the next}; work: gpn_lml_kinv_batched_work_bytes bytes. */
int gpn_lml_kinv_layout(int64_t n, int dy, int64_t* out4);
int64_t gpn_lml_kinv_batched_work_bytes(int64_t n, int dy, int batch);
int gpn_lml_kinv_batched(void* stream, int batch, int64_t n, const double* A, int64_t lda, int64_t sA,
                         const double* winv, int64_t sW, int dy, double* work);

/* gpn_predict = GPR._predict (gpr.py:88-117), given the factor of gpn_lml_forward (whose extra rows hold
 * V = L^-1 (Y - m(X)), i.e. the training-side mean function went in through gpn_lml_forward's M):
 * mean [ns, dy] = Ms + A^T V with A = L^-1 K(X, x*) and Ms [ns, dy] = the mean function at the test points
 * (gpr.py:107-108; NULL = zero mean, mean_functions.py:42-49);
 * var = [ns] K_diag - colsumsq(A) (full_cov == 0; the reference expands it to [ns, dy])
 * or [ns, ns] K(x*) - A^T A.  work: gpn_predict_work_bytes(n, ns, dy) bytes. */
int64_t gpn_predict_work_bytes(int64_t n, int64_t ns, int dy);
int gpn_predict(void* stream, int kind, const double* X, int64_t n, int d,
                const double* Xs, int64_t ns, const double* Ms,
                const double* variance, const double* length_scales, int nls,
                const double* A, int64_t lda, const double* winv, int dy, int full_cov,
                double* work, double* mean, double* var);

/* The right-solve of gpn_predict against BIG inverted diagonal blocks (GPR._predict, gpr.py:104-106, for a model that
 * predicts more than once with one factor).  gpn_block_inverse: wb (gpn_block_inverse_bytes(n) bytes) <- the inverses of
 * the 1024 x 1024 diagonal blocks of L (n * 1024^2 / 3 flops, once per factor).  gpn_trsm_right_lt_blocked: X = B L^-T as
 * n / 1024 steps of two large contractions (B [m, n] is CONSUMED; X != B; both padded like factor buffers) instead of the
 * ~2 n / 128 small launches of gpn_trsm_right_lt -- 1024 right-hand sides at n = 8192: 2.3 -> 1.3 ms.
 * gpn_predict_blocked = gpn_predict with that right-solve; work: 2 * gpn_predict_work_bytes(n, ns, dy). */
int64_t gpn_block_inverse_bytes(int64_t n);
int gpn_block_inverse(void* stream, const double* L, int64_t n, int64_t ldl, const double* winv, double* wb);
int gpn_trsm_right_lt_blocked(void* stream, const double* L, int64_t n, int64_t ldl, const double* wb,
                              double* B, int64_t m, int64_t ldb, double* X, int64_t ldx);
int gpn_predict_blocked(void* stream, int kind, const double* X, int64_t n, int d,
                        const double* Xs, int64_t ns, const double* Ms,
                        const double* variance, const double* length_scales, int nls,
                        const double* A, int64_t lda, const double* winv, const double* wb, int dy, int full_cov,
                        double* work, double* mean, double* var);

/* ---- several GPUs: 2-D block-cyclic log marginal likelihood --------------------
 * The reference is single-GPU (gptorch/models/base.py:33 "Assume single GPU"); this is
 * GPR.log_likelihood (gpr.py:47-67) for a Gram matrix that is partitioned over a Pr x Pc process
 * grid (Pr divides Pc: 1x1, 1x2, 2x2, 2x4), one process per GPU, tile (I,J) of `tile` x `tile`
 * on rank (I mod Pr) * Pc + (J mod Pc).  Same algorithm, layout and launch sequence as
 * gptorch_amd/dist.py (csrc/dist.hip).  Every rank calls it collectively with the same arguments
 * (X [n,d] and Y [n,dy] replicated on every GPU; each rank assembles its own tiles).
 *
 * Communication goes through a callback table so that libgpnative does not link a communication
 * runtime.  libgpnative_rccl.so (gpn_rccl_comm_create below) fills it from three RCCL
 * communicators; any other transport can be supplied the same way.  Both callbacks ENQUEUE on
 * the given HIP stream and return 0 on success:
 *   bcast(ctx, which, buf, count, root, stream): in-place broadcast of `count` doubles inside this
 *       rank's process ROW (which = 0; root = the source's process-column index 0..Pc-1) or process
 *       COLUMN (which = 1; root = the source's process-row index 0..Pr-1) sub-communicator;
 *   allreduce(ctx, buf, count, stream): in-place sum over all Pr*Pc ranks. */
enum { GPN_DIST_FORCE_COLLECTIVES = 1, /* issue them in single-member communicators too (tests) */
       GPN_DIST_MESH_EXCHANGE = 2      /* transports that know two routes (libgpnative_rccl.so): move a panel by grouped
                                          point-to-point transfers over the direct xGMI links (root -> every peer for
                                          small panels, pipelined scatter + all-gather for large ones: gpn_mesh_plan)
                                          instead of the library broadcast.  Same bytes, same buffers: bit-identical. */ };
typedef struct gpn_dist_comm {
  void* ctx;
  int (*bcast)(void* ctx, int which, double* buf, int64_t count, int root, void* stream);
  int (*allreduce)(void* ctx, double* buf, int64_t count, void* stream);
  int flags;
} gpn_dist_comm;
/* bytes of the caller-owned device workspace of rank `rank` (256-byte aligned; contents arbitrary on
 * entry: the call clears what it needs).  <0: bad grid / tile. */
int64_t gpn_dist_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile);
/* out4 (device): [0] = sum log L_ii, [1] = |alpha|^2, [2] = LML (gpr.py:63-67), [3] = LAPACK-style
 * info of the WHOLE matrix as a double (0 = ok, j > 0 = first failing pivot: replay with
 * noise + 10^(-10+i) as functions.py:20-43 does; GPN_INFO_INTERNAL = internal failure), identical on
 * every rank.  comm may be NULL for a 1 x 1 grid.  No host synchronisation.
 * out4[1] is the PLAIN |alpha|^2: from about 10^4 rows on the caller that wants north_star's 1e-8 absolute follows this call
 * with gpn_dist_lml_refine (below).
 * Errors on a multi-rank grid: a non-zero status (bad argument aside) means this rank stopped issuing the evaluation's
 * collectives part-way; its peers may be blocked inside the transport.  Nothing is drained (a collective whose peers never
 * arrive cannot complete): abort the communicators on every rank (ncclCommAbort) before reusing them. */
int gpn_dist_lml_forward(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                         const double* X, int64_t n, int d, const double* Y, int dy,
                         const double* variance, const double* length_scales, int nls, const double* noise,
                         int64_t tile, double* work, int64_t work_bytes, double* out4);
/* The same evaluation PLUS its closed-form backward on the same grid (what autograd gives the reference under
 * gpr.py:47-67; gptorch_amd/dist.py BlockCyclicGP.log_likelihood_and_grad): the factorisation also carries
 * identity blocks, so U = L^-T falls out of the same panel solves / updates; Kyy^-1 = U U^T is accumulated with the
 * tile columns of U travelling like factorisation panels; every rank contracts its own tiles of
 * G = 1/2 (a a^T - dy Kyy^-1) with dK/dtheta (gpn_kernel_grad) and 2 + nls scalars are all-reduced.
 * grads (device, [2 + nls]): dLML/d variance, dLML/d length_scales, dLML/d noise w.r.t. the CONSTRAINED values;
 * grad_resid (device, [n, dy], may be NULL) = dLML/d(y - m) = -a.  Identical on every rank.  Meaningful only when
 * out4[3] == 0.  Workspace: gpn_dist_grad_work_bytes (about 3x the forward's local matrix). */
int64_t gpn_dist_grad_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile);
int gpn_dist_lml_grad(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                      const double* X, int64_t n, int d, const double* Y, int dy,
                      const double* variance, const double* length_scales, int nls, const double* noise,
                      int64_t tile, double* work, int64_t work_bytes, double* out4, double* grads, double* grad_resid);
/* gpn_dist_lml_refine: the refinement step of gpn_lml_refine for the distributed factor, to be called after a
 * gpn_dist_lml_forward that reported info == 0, with the SAME arguments and that call's workspace untouched (it holds L and
 * alpha).  out4[0] (sum log L_ii) is read; out4[1] <- y^T Kyy^-1 y to second order in the factor's error, out4[2] <- the LML with
 * it; identical on every rank.  Sequence: alpha replicated (one all-reduce of n*dy), each rank inverts its diagonal tiles, a = L^-T
 * alpha tile row by tile row (two small all-reduces per tile row), each rank's share of Kyy a from the points in double-double
 * (one all-reduce of n*dy*2), finish.  At N = 65536 the plain value is 6e-8 from the CPU reference on a 1 x 2 grid and 9e-9 on
 * 2 x 4, the refined one 2-4e-9 on every grid.  rwork: gpn_dist_lml_refine_work_bytes(...) bytes, 256-byte aligned. */
int64_t gpn_dist_lml_refine_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile);
int gpn_dist_lml_refine(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                        const double* X, int64_t n, int d, const double* Y, int dy,
                        const double* variance, const double* length_scales, int nls, const double* noise,
                        int64_t tile, const double* work, double* rwork, int64_t rwork_bytes, double* out4);
/* The sub-buffers of a rank's workspace as (offset, reserved size) pairs in doubles, in declaration order: which = 0 the
 * workspace of gpn_dist_lml_forward (A, left[2], right[2], diag, winv, xrow, xcol, stats, info, sums), 1 of gpn_dist_lml_grad
 * (those + kinv, alphaT, aT, al, part, arow, acol, gwork, gout, acc), 2 of gpn_dist_lml_refine (alpha, a, owed, buf, sj, aj, ar,
 * ka, U, S, W, winv, gwork, rwork).  Pure host function, for layout checks (the sanitizer leg of the CPU suite sweeps it over
 * grids / sizes / tiles); returns the number of sub-buffers (min(that, cap) pairs written to out), < 0: bad arguments. */
int gpn_dist_layout(int which, int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile, int64_t* out, int cap);

/* GPR._predict (gpr.py:88-117) on the grid (gptorch_amd/dist.py BlockCyclicGP.predict): the ns test points ride through ONE
 * factorisation as further residual rows -- K(x*, X) below (y - m)^T comes out as A^T = (L^-1 K(X, x*))^T exactly like alpha^T
 * does, spread over the tile columns of the residual's process row -- and mean = Ms + A^T alpha [ns, dy], var = variance -
 * colsumsq(A) [ns] (full_cov == 0) or K(x*) - A^T A [ns, ns] are column-partial sums and ONE all-reduce.  Y is the residual
 * y - m(X) [n, dy], Ms the mean function at the test points [ns, dy] (gpr.py:107; NULL = zero); mean / var (device) come out
 * identical on every rank; out4 as gpn_dist_lml_forward (the reference re-factorises on every predict call as well,
 * gpr.py:104; a non-zero out4[3] is replayed with jitter by the caller).  work: gpn_dist_predict_work_bytes. */
int64_t gpn_dist_predict_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t ns, int64_t tile, int full_cov);
int gpn_dist_predict(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                     const double* X, int64_t n, int d, const double* Y, int dy, const double* Xs, int64_t ns, const double* Ms,
                     const double* variance, const double* length_scales, int nls, const double* noise,
                     int64_t tile, int full_cov, double* work, int64_t work_bytes, double* out4, double* mean, double* var);
/* libgpnative_rccl.so only: a callback table over RCCL communicators (ncclComm_t passed as void*):
 * `row` spans this rank's process row with rank-in-communicator = process-column index, `col` its
 * process column with rank-in-communicator = process-row index, `world` all ranks.  row / col may be
 * NULL where that communicator has a single member.  Destroy with gpn_rccl_comm_destroy (the
 * communicators themselves stay the caller's). */
gpn_dist_comm* gpn_rccl_comm_create(void* row, void* col, void* world);
void gpn_rccl_comm_destroy(gpn_dist_comm* comm);
/* libgpnative_rccl.so only: the point-to-point schedule of ONE mesh broadcast (GPN_DIST_MESH_EXCHANGE) of `count`
 * doubles from member `root` of a p-member communicator, as seen by member `me`: quintuples (stage, kind 0 = send /
 * 1 = recv, peer, offset, length) into ops[5 * cap]; one stage = one ncclGroupStart/End.  Returns the number of
 * quintuples (call again with a larger buffer if > cap), < 0: bad arguments.  Pure host function (no RCCL call) --
 * the same plan as gptorch_amd/dist.py mesh_plan, checked against it by tests/test_abi.py.  Defaults used by the
 * adapter: stages = 4, direct_below = 4 MiB / 8 (GPN_DIST_MESH_STAGES / GPN_DIST_MESH_DIRECT_BYTES override). */
int64_t gpn_mesh_plan(int p, int root, int me, int64_t count, int stages, int64_t direct_below, int64_t* ops, int64_t cap);

/* ---- small utilities -------------------------------------------------------- */
/* zero `bytes` bytes at dst on the stream (hipMemsetAsync): factor buffers, the backward's U / scratch matrices */
int gpn_fill_zero(void* stream, void* dst, int64_t bytes);
/* dst[r, c] = src[c, r] for src[rows, cols] */
int gpn_transpose(void* stream, const double* src, int64_t rows, int64_t cols, int64_t lds,
                  double* dst, int64_t ldd);
/* dst[rows, cols] (ldd) <- src (lds); if tril != 0 entries with c > r are written as 0
 * (torch.cholesky returns a zeroed upper triangle) */
int gpn_copy_matrix(void* stream, const double* src, int64_t rows, int64_t cols, int64_t lds,
                    double* dst, int64_t ldd, int tril);
/* out[z] = sum_{i < rows, j < cols} x_z[i ldx + j] * y_z[i ldy + j] (y NULL: the plain sum), problem z at x + z sx, y + z sy:
 * the scalar sums of the sparse bound (sparse_gpr.py:139-151: tr(A A^T), |err|^2 ...) for `batch` models in one launch.  A
 * fixed summation order per problem: out[z] does not depend on `batch`. */
int gpn_dot2d_batched(void* stream, const double* x, int64_t ldx, int64_t sx, const double* y, int64_t ldy, int64_t sy,
                      int64_t rows, int64_t cols, double* out, int batch);
/* out[r] = sum_c A[r,c]^2  (the (A*A).sum(0) of gpr.py:109-113 in transposed storage) */
int gpn_row_sumsq(void* stream, const double* A, int64_t rows, int64_t cols, int64_t lda,
                  double* out);


/* ---- optional launch profiler (bench.py's roofline legs) ---------------------- */
/* While enabled, every contraction / assembly / gradient-sweep / leaf launch of this library is bracketed by two HIP events
 * recorded ON THE LAUNCH STREAM (what SURVEY 8(d) asks the SYRK fraction to be measured with).  Off by default; the
 * events serialise nothing, but enabling it inside a timed region adds two event records per launch. */
int gpn_profile_enable(int on);
/* Synchronises every recorded event and clears the list.  out3_host[0] = launches, [1] = total ms, [2] = executed flops of all
 * contraction launches (2 M N K per launch; lower-tile launches count the tiles on or below the diagonal only). */
int gpn_profile_collect(double* out3_host);
/* ... per launch class c < nclasses (0 rectangular contraction, 1 lower-tile SYRK update, 2 in-place panel solve,
 * 3 K-clipped contraction, 4 K assembly, 5 gradient sweep, 6 leaf): out_host[3c] = launches, [3c+1] = ms, [3c+2] = work
 * (flops for 0-3, algorithmic bytes for 4-5). */
int gpn_profile_collect_classes(double* out_host, int nclasses);

#ifdef __cplusplus
}
#endif
#endif /* GPNATIVE_H */
