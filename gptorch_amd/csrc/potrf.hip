// Blocked fp64 Cholesky for gfx950 (lower, in place, row-major).  Replaces torch.cholesky under functions.cholesky
// (functions.py:46-47); the triangular solves and inversions built on the factor (functions.trtrs, functions.py:71-76) are in
// trisolve.hip, the recursion both share (trsm_rec) in potrf_ctx.h.
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
#include "potrf_ctx.h"

namespace gpn {

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
