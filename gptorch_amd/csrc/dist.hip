// gpn_dist_*: the 2-D block-cyclic log marginal likelihood behind the C ABI (SURVEY.md 8(b):
// "multi-GPU handle gpn_dist_* taking an RCCL communicator + process grid"; 8(e)).
//
// The reference has no multi-GPU path (gptorch/models/base.py:33 "Assume single GPU"); this is the
// distributed form of GPR.log_likelihood (gpr.py:47-67) for callers that are not Python.  It is the
// same algorithm, layout and launch sequence as gptorch_amd/dist.py (read that module's header for
// the layout): tile (I,J) of T x T on rank (I mod Pr) * Pc + (J mod Pc), Pr | Pc; one stacked
// row-major local matrix per rank whose rows taking part in panel k are contiguous; per panel
//   1. owner factors the diagonal tile, [L_kk | leaf inverses] packed down the process COLUMN;
//   2. every rank of that process column solves all its panel rows in one call;
//   3. the solved rows, packed, along each process ROW;
//   4. the tiles of my tile columns, packed, down each process column;
//   5. per local tile column one contraction over the stacked rows below it,
// with look-ahead: tile column k+1 is updated, factored and solved first and its two exchanges
// run on their own HIP streams underneath the two halves of the remaining update by panel k.
//
// Communication goes through a small callback table (gpn_dist_comm) so that the library itself
// does not link a communication runtime: libgpnative_rccl.so (csrc/rccl_adapter.cpp) provides the
// table over three ncclComm_t (process row, process column, world); the test-suite provides one over
// torch.distributed/gloo to run several ranks on a 1-GPU box.  No allocation, no host
// synchronisation: the caller owns one workspace of gpn_dist_work_bytes() bytes and reads the four
// result words back itself.
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "gpn_common.h"

namespace gpn {

struct DistGeom {
  int rank, pr, pc, my_r, my_c;
  int64_t n, T, nt;
  int dy;
  int64_t nrow_t, ncol_t, res_off, id_off, rows, ld;
  bool has_res, with_inv;
  int64_t rows_of(int64_t I) const { return I == nt ? dy : std::min<int64_t>(T, n - I * T); }
  int64_t rows_le(int64_t k, int r) const {          // how many of process row r's tile rows have index <= k
    const int64_t cnt = (nt - r + pr - 1) / pr;
    return k < r ? 0 : std::min<int64_t>((k - r) / pr + 1, std::max<int64_t>(cnt, 0));
  }
  int64_t cols_le(int64_t k) const { return k < my_c ? 0 : std::min<int64_t>((k - my_c) / pc + 1, ncol_t); }
  int64_t nreal_rows() const {                       // real (unpadded) rows of my matrix segment: a prefix
    if (nrow_t == 0) return 0;
    const int64_t last = my_r + (nrow_t - 1) * pr;
    return (nrow_t - 1) * T + rows_of(last);
  }
  int64_t nreal_cols() const {
    if (ncol_t == 0) return 0;
    const int64_t last = my_c + (ncol_t - 1) * pc;
    return (ncol_t - 1) * T + rows_of(last);
  }
};

static int make_geom(DistGeom& g, int rank, int pr, int pc, int64_t n, int dy, int64_t tile, bool with_inv = false) {
  if (pr < 1 || pc < 1 || pc % pr) return -4;
  if (rank < 0 || rank >= pr * pc) return -3;
  if (tile < LEAF || tile % LEAF) return -15;
  g.rank = rank; g.pr = pr; g.pc = pc; g.my_r = rank / pc; g.my_c = rank % pc;
  g.n = n; g.T = tile; g.dy = dy;
  g.nt = (n + tile - 1) / tile;
  g.nrow_t = g.nt > g.my_r ? (g.nt - g.my_r + pr - 1) / pr : 0;
  g.ncol_t = g.nt > g.my_c ? (g.nt - g.my_c + pc - 1) / pc : 0;
  g.has_res = (g.nt % pr) == g.my_r;
  g.res_off = g.nrow_t * tile;
  g.with_inv = with_inv;
  g.id_off = g.res_off + (g.has_res ? round_up(dy, LEAF) : 0);     // identity blocks (backward only): block i -> U_i,:
  g.rows = g.id_off + (with_inv ? g.nrow_t * tile : 0) + LEAF;
  g.ld = std::max<int64_t>(g.ncol_t, 1) * tile;
  return GPN_OK;
}

// workspace layout (doubles)
struct DistLayout {
  int64_t A, left[2], right[2], diag, winv, xrow, xcol, stats, info, sums, total;
  // backward only
  int64_t kinv, alphaT, aT, al, part, arow, acol, gwork, gout, acc;
};
// (rec: optional recorder of (offset, reserved doubles) per sub-buffer in declaration order -- gpn_dist_layout)
static DistLayout make_layout(const DistGeom& g, int d, std::vector<int64_t>* rec = nullptr) {
  DistLayout L;
  const int64_t T = g.T;
  const int64_t lrows = (g.rows + T - 1) / T * T + 16;
  const int64_t wn = gpn_winv_bytes(T) / 8;
  int64_t o = 0;
  auto take = [&](int64_t cnt) {                                                               // 256-byte granules
    const int64_t at = o;
    o += round_up(cnt, 32);
    if (rec) { rec->push_back(at); rec->push_back(o - at); }
    return at;
  };
  L.A = take(g.rows * g.ld);
  for (int i = 0; i < 2; ++i) L.left[i] = take(lrows * T);
  for (int i = 0; i < 2; ++i) L.right[i] = take((std::max<int64_t>(g.ncol_t, 1) * T + 16) * T);
  L.diag = take(T * T + wn);
  L.winv = take(wn);
  L.xrow = take(std::max<int64_t>(g.nrow_t, 1) * T * d);
  L.xcol = take(std::max<int64_t>(g.ncol_t, 1) * T * d);
  L.stats = take(3 * (g.ncol_t + 1) + g.dy + 8);      // lml_reduce triples per diagonal tile, residual row sums
  L.info = take(g.nt + 8);                            // int32 per tile column (stored in double-sized slots)
  L.sums = take(g.nt + 8);                            // all-reduced vector: log-det, |alpha|^2, info per tile
  if (g.with_inv) {
    const int64_t kp = round_up(g.dy, 16), rpad = std::max<int64_t>(g.nrow_t, 1) * T;
    L.kinv = take((rpad + LEAF) * g.ld);                // Kyy^-1 -> G, my tiles, same layout as the matrix segment
    L.alphaT = take((int64_t)g.dy * g.n);               // alpha^T, replicated
    L.aT = take((int64_t)g.dy * g.n);                   // a^T = (Kyy^-1 (y - m))^T, replicated
    L.al = take(kp * g.ld);                             // alpha^T of my tile columns (local column order)
    L.part = take(kp * rpad);                           // my partial of a^T (local row order)
    L.arow = take((rpad + 16) * kp);                    // a of my tile rows / columns, K-padded
    L.acol = take((std::max<int64_t>(g.ncol_t, 1) * T + 16) * kp);
    L.gwork = take(gpn_grad_work_bytes(rpad, T, d, 0) / 8 + 8);
    L.gout = take(d + 8);
    L.acc = take(d + 8);
  }
  L.total = o;
  return L;
}

// helper streams / events per caller stream (row and column exchanges run beside the compute stream)
struct DistAux {
  hipStream_t row_s = nullptr, col_s = nullptr;
  hipEvent_t ready = nullptr;                          // compute stream -> exchange streams
  hipEvent_t diag_done = nullptr;
  hipEvent_t row_done[2] = {nullptr, nullptr}, col_done[2] = {nullptr, nullptr};
};
static std::mutex g_dist_mutex;
static std::unordered_map<hipStream_t, DistAux> g_dist_aux;
static DistAux* dist_aux_for(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_dist_mutex);
  auto it = g_dist_aux.find(s);
  if (it != g_dist_aux.end()) return &it->second;
  DistAux a;
  bool good = hipStreamCreateWithFlags(&a.row_s, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&a.col_s, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&a.ready, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&a.diag_done, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < 2 && good; ++i)
    good = hipEventCreateWithFlags(&a.row_done[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&a.col_done[i], hipEventDisableTiming) == hipSuccess;
  if (!good) {                       // free what was created before the failing call
    if (a.row_s) (void)hipStreamDestroy(a.row_s);
    if (a.col_s) (void)hipStreamDestroy(a.col_s);
    if (a.ready) (void)hipEventDestroy(a.ready);
    if (a.diag_done) (void)hipEventDestroy(a.diag_done);
    for (int i = 0; i < 2; ++i) {
      if (a.row_done[i]) (void)hipEventDestroy(a.row_done[i]);
      if (a.col_done[i]) (void)hipEventDestroy(a.col_done[i]);
    }
    return nullptr;
  }
  return &g_dist_aux.emplace(s, a).first->second;
}
void dist_release(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_dist_mutex);
  auto drop = [](DistAux& a) {
    if (a.row_s) { (void)hipStreamSynchronize(a.row_s); (void)hipStreamDestroy(a.row_s); }
    if (a.col_s) { (void)hipStreamSynchronize(a.col_s); (void)hipStreamDestroy(a.col_s); }
    if (a.ready) (void)hipEventDestroy(a.ready);
    if (a.diag_done) (void)hipEventDestroy(a.diag_done);
    for (int i = 0; i < 2; ++i) {
      if (a.row_done[i]) (void)hipEventDestroy(a.row_done[i]);
      if (a.col_done[i]) (void)hipEventDestroy(a.col_done[i]);
    }
  };
  if (!s) { for (auto& kv : g_dist_aux) drop(kv.second); g_dist_aux.clear(); return; }
  auto it = g_dist_aux.find(s);
  if (it != g_dist_aux.end()) { drop(it->second); g_dist_aux.erase(it); }
}

// sums[0] = sum over my diagonal tiles of sum log L_ii, sums[1] = |alpha|^2 of my residual columns,
// sums[2 + k] = info word of tile column k (as a double; only its owner is non-zero)
__global__ void dist_local_sums_kernel(const double* stats, int ndiag, const double* rowsq, int dy, const int32_t* info, int nt,
                                       double* sums) {
  const int t = threadIdx.x;
  if (t == 0) {
    double s = 0.0;
    for (int i = 0; i < ndiag; ++i) s += stats[3 * i];
    sums[0] = s;
    double q = 0.0;
    for (int c = 0; c < dy; ++c) q += rowsq[c];
    sums[1] = q;
  }
  for (int k = t; k < nt; k += blockDim.x) sums[2 + k] = (double)info[k];
}

// out[0] = sum log L_ii, out[1] = |alpha|^2, out[2] = LML (gpr.py:63-67), out[3] = LAPACK-style info of the
// whole matrix (first failing pivot, 1-based; GPN_INFO_INTERNAL if any tile reported it; 0 = ok)
__global__ void dist_finish_kernel(const double* sums, int nt, int64_t T, int64_t n, int dy, double* out4) {
  if (threadIdx.x != 0) return;
  double info = 0.0;
  for (int k = 0; k < nt; ++k) {
    if (sums[2 + k] < 0.0) { info = (double)GPN_INFO_INTERNAL; break; }
    if (sums[2 + k] > 0.0 && info == 0.0) info = (double)k * (double)T + sums[2 + k];
  }
  out4[0] = sums[0];
  out4[1] = sums[1];
  out4[2] = -0.5 * sums[1] - (double)dy * sums[0] - 0.5 * (double)dy * (double)n * 1.8378770664093454836;   // log(2 pi)
  out4[3] = info;
}

// identity block: ones on the diagonal of an n x n tile
__global__ void dist_set_identity_kernel(double* A, int64_t ld, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i * ld + i] = 1.0;
}
// out[p, c] = aT[c, gidx(p)] for the rows of my tile rows (or tile columns): tile q of mine is global tile
// first + q * stride, so gidx(p) = (first + (p / T) * stride) * T + p % T;  out is [*, kpad] row-major, zero beyond dy
__global__ void dist_a_pad_kernel(const double* aT, int64_t n, int dy, int64_t T, int first, int stride, int64_t nreal,
                                  double* out, int kpad) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nreal) return;
  const int64_t gi = ((int64_t)first + (p / T) * stride) * T + p % T;
  for (int c = 0; c < dy; ++c) out[p * kpad + c] = aT[(int64_t)c * n + gi];
}
// acc[slot] += trace of the n x n block G
__global__ void dist_trace_add_kernel(const double* G, int64_t ld, int64_t n, double* acc, int slot) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += G[i * ld + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[slot] += red[0];
}
__global__ void dist_axpy_kernel(double* acc, const double* x, double scale, int cnt) {
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) acc[i] += scale * x[i];
}
// grad_resid[i, c] = -aT[c, i]   (dLML/d(y - m) = -a)
__global__ void dist_neg_transpose_kernel(const double* aT, int64_t n, int dy, double* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < dy; ++c) out[i * dy + c] = -aT[(int64_t)c * n + i];
}

struct DistRun {
  hipStream_t s;
  DistAux* ax;
  const gpn_dist_comm* comm;
  DistGeom g;
  DistLayout L;
  double* W;
  int rc = GPN_OK;
  bool force() const { return comm && (comm->flags & GPN_DIST_FORCE_COLLECTIVES); }
  bool xrow() const { return comm && (g.pc > 1 || force()); }
  bool xcol() const { return comm && (g.pr > 1 || force()); }
  void ok(int r) { if (r != GPN_OK && rc == GPN_OK) rc = r; }
  void hip(hipError_t e, const char* where) { if (e != hipSuccess && rc == GPN_OK) { set_hip_error(e, where); rc = GPN_E_HIP; } }

  void active(int64_t k, int64_t& lo, int64_t& hi) const {
    lo = g.rows_le(k, g.my_r) * g.T;
    hi = g.with_inv ? g.id_off + g.rows_le(k, g.my_r) * g.T : g.res_off + (g.has_res ? g.dy : 0);
    hi = std::max(lo, hi);
  }

  // steps 1-2
  void panel_phase(int64_t k) {
    const int64_t T = g.T, nk = g.rows_of(k), ck = k % g.pc;
    if (g.my_c != ck || rc != GPN_OK) return;
    const int64_t lk = (k - g.my_c) / g.pc;
    int64_t lo, hi;
    active(k, lo, hi);
    const int64_t m = hi - lo;
    double* colk = W + L.A + lk * T;
    const bool mine = (k % g.pr) == g.my_r;
    double* Lp = W + L.diag;
    double* Wp = Lp + T * T;
    int32_t* info = reinterpret_cast<int32_t*>(W + L.info) + k;
    if (mine) {
      const int64_t d0 = (g.rows_le(k, g.my_r) - 1) * T;
      double* Lt = colk + d0 * g.ld;
      if (!xcol()) {
        // nobody else needs L_kk (Pr = 1): my panel rows ride along as extra rows of the same call, so the tile's
        // leaf chain runs underneath their solves / updates instead of alone on the chip
        ok(gpn_potrf_lower_panel(s, Lt, nk, hi - (d0 + nk), g.ld, Wp, info));
        return;
      }
      ok(gpn_potrf_lower_panel(s, Lt, nk, 0, g.ld, Wp, info));
      ok(gpn_copy_matrix(s, Lt, nk, nk, g.ld, Lp, T, 0));
    }
    if (xcol() && rc == GPN_OK) {
      // short, and everything after it depends on it -- but it goes through the column stream like the
      // panel exchange of step 4, so that one communicator is only ever driven from one stream
      hip(hipEventRecord(ax->ready, s), "panel_phase");
      hip(hipStreamWaitEvent(ax->col_s, ax->ready, 0), "panel_phase");
      ok(comm->bcast(comm->ctx, 1, Lp, T * T + gpn_winv_bytes(T) / 8, (int)(k % g.pr), ax->col_s));
      hip(hipEventRecord(ax->diag_done, ax->col_s), "panel_phase");
      hip(hipStreamWaitEvent(s, ax->diag_done, 0), "panel_phase");
    }
    if (m > 0 && rc == GPN_OK) ok(gpn_trsm_right_lt(s, Lp, nk, T, Wp, colk + lo * g.ld, m, g.ld));
  }

  // step 3 (asynchronous on the row stream): -> left operand buffer
  double* start_rows(int64_t k, bool& pending) {
    int64_t lo, hi;
    active(k, lo, hi);
    return start_rows(k, lo, hi, pending);
  }
  // ... of an explicit local row range [lo, hi) of tile column k (the backward sends the identity blocks)
  double* start_rows(int64_t k, int64_t lo, int64_t hi, bool& pending) {
    const int64_t T = g.T, nk = g.rows_of(k), ck = k % g.pc;
    const int64_t m = hi - lo;
    double* buf = W + L.left[k & 1];
    pending = false;
    if (m == 0 || rc != GPN_OK) return buf;
    if (nk < T) hip(hipMemsetAsync(buf, 0, (size_t)m * T * sizeof(double), s), "start_rows memset");
    if (g.my_c == ck) ok(gpn_copy_matrix(s, W + L.A + lo * g.ld + ((k - g.my_c) / g.pc) * T, m, nk, g.ld, buf, T, 0));
    if (xrow() && rc == GPN_OK) {
      hip(hipEventRecord(ax->ready, s), "start_rows");
      hip(hipStreamWaitEvent(ax->row_s, ax->ready, 0), "start_rows");
      ok(comm->bcast(comm->ctx, 0, buf, m * T, (int)ck, ax->row_s));
      hip(hipEventRecord(ax->row_done[k & 1], ax->row_s), "start_rows");
      pending = true;
    }
    return buf;
  }
  void wait_rows(int64_t k, bool& pending) {
    if (pending) hip(hipStreamWaitEvent(s, ax->row_done[k & 1], 0), "wait_rows");
    pending = false;
  }

  // step 4 (asynchronous on the column stream, after step 3 has completed): -> right operand
  double* start_cols(int64_t k, double* left, bool& pending) {
    const int rs = g.my_c % g.pr;
    const int64_t lj0 = g.cols_le(k), count = g.ncol_t - lj0;
    const int64_t J0 = lj0 * g.pc + g.my_c;
    return start_cols(k, left, count > 0 ? (J0 - rs) / g.pr - g.rows_le(k, rs) : 0, count, pending);
  }
  // ... `count` tiles starting at slot `first` of process row (c mod Pr)'s left buffer, every (Pc/Pr)-th
  double* start_cols(int64_t k, double* left, int64_t first, int64_t count, bool& pending) {
    const int64_t T = g.T;
    pending = false;
    const int rs = g.my_c % g.pr;
    if (count <= 0 || rc != GPN_OK) return W + L.right[k & 1];
    const int64_t step = g.pc / g.pr;
    const bool src = g.my_r == rs;
    double* buf;
    if (src && step == 1) buf = left + first * T * T;                     // already contiguous
    else {
      buf = W + L.right[k & 1];
      if (src)   // every step-th tile of my left buffer -> consecutive tiles: ONE strided copy launch with a tile (T*T doubles) as
                 // a "row" (not hipMemcpy2DAsync: its pitch is limited to the device's memPitch, about 2 GiB -- a 1 x 8
                 // grid with tile 8192 would need 4.3 GB)
        ok(gpn_copy_matrix(s, left + first * T * T, count, T * T, step * T * T, buf, T * T, 0));
    }
    if (xcol() && rc == GPN_OK) {
      hip(hipEventRecord(ax->ready, s), "start_cols");
      hip(hipStreamWaitEvent(ax->col_s, ax->ready, 0), "start_cols");
      ok(comm->bcast(comm->ctx, 1, buf, count * T * T, rs, ax->col_s));
      hip(hipEventRecord(ax->col_done[k & 1], ax->col_s), "start_cols");
      pending = true;
    }
    return buf;
  }
  void wait_cols(int64_t k, bool& pending) {
    if (pending) hip(hipStreamWaitEvent(s, ax->col_done[k & 1], 0), "wait_cols");
    pending = false;
  }

  // step 5 on my local tile columns [from, to): ONE staircase launch -- every tile column starts Pc/Pr tile rows below
  // its left neighbour, and where the diagonal tiles are mine (c mod Pr = r) each column's first tile is lower-only
  void update(int64_t k, const double* left, const double* right, int64_t from, int64_t to) {
    const int64_t T = g.T, nk = round_up(g.rows_of(k), 16);
    int64_t lo, hi;
    active(k, lo, hi);
    const int64_t base = g.cols_le(k);
    const int64_t a = std::max(from, base), b = std::min(to, g.ncol_t);
    if (b <= a || rc != GPN_OK) return;
    const int64_t J = a * g.pc + g.my_c;
    const int64_t r0 = g.rows_le(J - 1, g.my_r) * T;     // >= lo: J > k
    if (r0 >= hi) return;
    const int diag = (g.my_c % g.pr) == g.my_r;
    ok(gemm_nt_stair(s, hi - r0, b - a, T, nk, -1.0, left + (r0 - lo) * T, T, right + (a - base) * T * T, T, 1.0,
                     W + L.A + r0 * g.ld + a * T, g.ld, (g.pc / g.pr) * T, diag));
  }
};

}  // namespace gpn

using namespace gpn;

// var <- variance - var (diag) / var <- Kss - var (full), mean += Ms: the last step of gpr.py:107-117 after the all-reduce
__global__ void dist_predict_finish_kernel(double* mean, const double* Ms, int64_t nmean, double* var, const double* variance,
                                           const double* kss, int64_t nvar) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (Ms && i < nmean) mean[i] += Ms[i];
  if (i < nvar) var[i] = (kss ? kss[i] : variance[0]) - var[i];
}

// shared body of the entry points: assembly + factorisation (+ the closed-form backward on the grid, or prediction).
// Error behaviour on a multi-rank grid: a non-zero status means this rank stopped issuing the evaluation's collectives part-way
// (nothing is waited for here: a broadcast whose peers never arrive would block a drain for ever).  The other ranks of its
// process row / column are then blocked inside the transport: the caller must abort the communicators (ncclCommAbort) on every
// rank before reusing them -- include/gpnative.h says the same.
static int dist_evaluate(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                         const double* X, int64_t n, int d, const double* Y, int dy,
                         const double* variance, const double* length_scales, int nls, const double* noise,
                         int64_t tile, double* work, int64_t work_bytes, double* out4, bool with_grad,
                         double* grads, double* grad_resid,
                         const double* Xs = nullptr, int64_t ns = 0, const double* Ms = nullptr, int full_cov = 0,
                         double* pmean = nullptr, double* pvar = nullptr) {
  DistRun R;
  // prediction: the ns test points ride through the factorisation as further residual rows (K(x*, X) below (y - m)^T comes
  // out as A^T = (L^-1 K(X, x*))^T exactly like alpha^T does); the LML terms still count the first dy rows only
  int rc = make_geom(R.g, rank, pr, pc, n, dy + (int)ns, tile, with_grad);
  if (rc != GPN_OK) return rc;
  if (ns > 0 && (with_grad || !Xs || !pmean || !pvar)) return -20;
  if (pr * pc > 1 && (!comm || !comm->bcast || !comm->allreduce)) return -2;
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -6;
  if (!X) return -7;
  if (n <= 0) return -8;
  if (d <= 0) return -9;
  if (!Y) return -10;
  if (dy <= 0) return -11;
  if (!variance || !length_scales || !noise) return -12;
  if (nls != 1 && nls != d) return -14;
  if (!work) return -16;
  if (!out4) return -18;
  if (with_grad && !grads) return -19;
  R.L = make_layout(R.g, d);
  const int64_t pred_words = ns > 0 ? round_up(ns * dy + (full_cov ? 2 * ns * ns : ns), 32) : 0;   // partials (+ K(x*) for the full covariance)
  if (work_bytes < (R.L.total + pred_words) * (int64_t)sizeof(double)) return -17;
  if (reinterpret_cast<uintptr_t>(work) & 255) return GPN_E_ALIGN;
  R.s = static_cast<hipStream_t>(stream);
  R.comm = (pr * pc > 1 || (comm && (comm->flags & GPN_DIST_FORCE_COLLECTIVES))) ? comm : nullptr;
  R.W = work;
  R.ax = dist_aux_for(R.s);
  if (!R.ax) return GPN_E_HIP;
  const DistGeom& g = R.g;
  const DistLayout& L = R.L;
  const int64_t T = g.T, nt = g.nt;
  hipStream_t s = R.s;

  // ---- assembly: X rows in my tile-row / tile-column order, then one rectangular K per local tile column
  // The workspace's contents are arbitrary on entry.  What has to be zero: everything but the REAL rows of the matrix
  // segment (ragged-tile padding rows, residual and identity segments, Kyy^-1 accumulator, panel buffers, statistics)
  // and, in those rows, the padding columns of a ragged last tile column.  The tiles of the real rows at or below the
  // diagonal are overwritten by the assembly; the tiles above it are never read.  (At 1 x 1 and N = 65536 the matrix
  // segment is 34 GB: clearing it every evaluation was 10 ms.)
  const int64_t nrr = g.nreal_rows(), ncr = g.nreal_cols();
  GPN_HIP_CHECK(hipMemsetAsync(work + L.A + nrr * g.ld, 0, (size_t)(R.L.total - (L.A + nrr * g.ld)) * sizeof(double), s));
  if (nrr > 0 && ncr < g.ld)
    GPN_HIP_CHECK(hipMemset2DAsync(work + L.A + ncr, (size_t)g.ld * sizeof(double), 0, (size_t)(g.ld - ncr) * sizeof(double), (size_t)nrr, s));
  double* Xrow = work + L.xrow;
  double* Xcol = work + L.xcol;
  for (int64_t li = 0; li < g.nrow_t; ++li) {          // tile rows are T-row blocks of X at stride Pr*T
    const int64_t I = g.my_r + li * g.pr;
    GPN_HIP_CHECK(hipMemcpyAsync(Xrow + li * T * d, X + I * T * d, (size_t)g.rows_of(I) * d * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  for (int64_t lj = 0; lj < g.ncol_t; ++lj) {
    const int64_t J = g.my_c + lj * g.pc;
    GPN_HIP_CHECK(hipMemcpyAsync(Xcol + lj * T * d, X + J * T * d, (size_t)g.rows_of(J) * d * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  double* A = work + L.A;
  for (int64_t lj = 0; lj < g.ncol_t; ++lj) {
    const int64_t J = g.my_c + lj * g.pc, nJ = g.rows_of(J);
    if (g.has_res) {                                     // residual rows: (y)^T of this tile column
      rc = gpn_pack_rhs(stream, Y + J * T * dy, nullptr, nJ, dy, A + g.res_off * g.ld + lj * T, g.ld);
      if (rc != GPN_OK) return rc;
      if (ns > 0) {                                      // ... and K(x*, X_J) below them (prediction)
        rc = gpn_kernel_matrix(stream, kind, Xs, ns, Xcol + lj * T * d, nJ, d, variance, length_scales, nls, nullptr, GPN_FULL,
                               A + (g.res_off + dy) * g.ld + lj * T, g.ld);
        if (rc != GPN_OK) return rc;
      }
    }
    if (g.with_inv && J % g.pr == g.my_r) {              // identity block J: I at tile column J (-> U_J,: from there on)
      const int64_t li = (J - g.my_r) / g.pr;
      hipLaunchKernelGGL(dist_set_identity_kernel, dim3((unsigned)((nJ + 255) / 256)), dim3(256), 0, s,
                         A + (g.id_off + li * T) * g.ld + lj * T, g.ld, nJ);
      GPN_LAUNCH_CHECK();
    }
    const int64_t li0 = g.rows_le(J - 1, g.my_r);
    int64_t r0 = li0 * T;
    if (r0 >= nrr) continue;                             // no matrix tile of mine at or below the diagonal here
    const double* xj = Xcol + lj * T * d;
    if (li0 * g.pr + g.my_r == J) {                     // the diagonal tile is mine: K(X_J) + noise I (gpr.py:83-86)
      rc = gpn_kernel_matrix(stream, kind, xj, nJ, nullptr, nJ, d, variance, length_scales, nls, noise, GPN_FULL,
                             A + r0 * g.ld + lj * T, g.ld);
      if (rc != GPN_OK) return rc;
      r0 += T;
    }
    if (r0 < nrr) {
      rc = gpn_kernel_matrix(stream, kind, Xrow + r0 * d, nrr - r0, xj, nJ, d, variance, length_scales, nls, nullptr, GPN_FULL,
                             A + r0 * g.ld + lj * T, g.ld);
      if (rc != GPN_OK) return rc;
    }
  }

  // ---- factorisation with look-ahead
  R.panel_phase(0);
  double *left = nullptr, *right = nullptr;
  bool rows_pending = false, cols_pending = false;
  if (nt > 1) {
    left = R.start_rows(0, rows_pending);
    R.wait_rows(0, rows_pending);
    right = R.start_cols(0, left, cols_pending);
  }
  for (int64_t k = 0; k + 1 < nt && R.rc == GPN_OK; ++k) {
    R.wait_cols(k, cols_pending);                        // panel k is everywhere it is needed
    if ((k + 1) % g.pc == g.my_c) {
      const int64_t nxt = g.cols_le(k + 1) - 1;
      R.update(k, left, right, nxt, nxt + 1);            // tile column k+1 first ...
    }
    R.panel_phase(k + 1);                                // ... so only ITS panel is on the critical path
    if (k + 2 < nt) {
      bool rp = false, cp = false;
      double* nleft = R.start_rows(k + 1, rp);           // in flight under the first half ...
      const int64_t a = g.cols_le(k + 1);
      const int64_t half = a + (g.ncol_t - a + 1) / 2;
      R.update(k, left, right, a, half);
      R.wait_rows(k + 1, rp);
      double* nright = R.start_cols(k + 1, nleft, cp);   // ... and under the second half
      R.update(k, left, right, half, g.ncol_t);
      left = nleft; right = nright;
      cols_pending = cp;
    }
  }
  if (R.rc != GPN_OK) return R.rc;

  // ---- log-det partials, |alpha|^2, info words: one small all-reduce
  double* stats = work + L.stats;
  int ndiag = 0;
  for (int64_t lj = 0; lj < g.ncol_t; ++lj) {
    const int64_t J = g.my_c + lj * g.pc;
    if (J % g.pr != g.my_r) continue;
    const int64_t li = (J - g.my_r) / g.pr;
    rc = gpn_lml_reduce(stream, A + li * T * g.ld + lj * T, g.rows_of(J), 0, g.ld, stats + 3 * ndiag);
    if (rc != GPN_OK) return rc;
    ++ndiag;
  }
  double* rowsq = stats + 3 * (g.ncol_t + 1);
  if (g.has_res && ncr > 0) {
    rc = gpn_row_sumsq(stream, A + g.res_off * g.ld, dy, ncr, g.ld, rowsq);
    if (rc != GPN_OK) return rc;
  }
  double* sums = work + L.sums;
  hipLaunchKernelGGL(dist_local_sums_kernel, dim3(1), dim3(256), 0, s, stats, ndiag, rowsq, (g.has_res && ncr > 0) ? dy : 0,
                     reinterpret_cast<const int32_t*>(work + L.info), (int)nt, sums);
  GPN_LAUNCH_CHECK();
  if (R.comm) {
    rc = R.comm->allreduce(R.comm->ctx, sums, nt + 2, s);
    if (rc != GPN_OK) return rc;
  }
  hipLaunchKernelGGL(dist_finish_kernel, dim3(1), dim3(64), 0, s, sums, (int)nt, T, n, dy, out4);
  GPN_LAUNCH_CHECK();
  if (ns > 0) {
    // ---- prediction (gpr.py:104-117; gptorch_amd/dist.py BlockCyclicGP.predict): mean = A^T alpha and colsumsq(A) / A^T A are
    // sums over the tile columns, which are spread over the ranks of the residual's process row: partials + ONE all-reduce
    double* pm = work + R.L.total;                       // [ns * dy | ns or ns * ns | (full: K(x*))]
    double* pv = pm + ns * dy;
    const int64_t nvar = full_cov ? ns * ns : ns;
    GPN_HIP_CHECK(hipMemsetAsync(pm, 0, (size_t)(ns * dy + nvar) * sizeof(double), s));
    if (g.has_res && ncr > 0) {
      const double* alphaT = A + g.res_off * g.ld;       // [dy, ld]  (columns past my real ones are zero)
      const double* AT = alphaT + (int64_t)dy * g.ld;    // [ns, ld]
      rc = gemm_nt(s, ns, dy, g.ld, 1.0, AT, g.ld, alphaT, g.ld, 0.0, pm, dy, 0);
      if (rc != GPN_OK) return rc;
      if (full_cov) rc = gemm_nt(s, ns, ns, g.ld, 1.0, AT, g.ld, AT, g.ld, 0.0, pv, ns, 0);
      else rc = gpn_row_sumsq(stream, AT, ns, ncr, g.ld, pv);
      if (rc != GPN_OK) return rc;
    }
    if (R.comm) { rc = R.comm->allreduce(R.comm->ctx, pm, ns * dy + nvar, s); if (rc != GPN_OK) return rc; }
    double* kss = nullptr;
    if (full_cov) {
      kss = pv + nvar;
      rc = gpn_kernel_matrix(stream, kind, Xs, ns, nullptr, ns, d, variance, length_scales, nls, nullptr, GPN_FULL, kss, ns);
      if (rc != GPN_OK) return rc;
    }
    GPN_HIP_CHECK(hipMemcpyAsync(pmean, pm, (size_t)ns * dy * sizeof(double), hipMemcpyDeviceToDevice, s));
    GPN_HIP_CHECK(hipMemcpyAsync(pvar, pv, (size_t)nvar * sizeof(double), hipMemcpyDeviceToDevice, s));
    const int64_t cnt = std::max<int64_t>(ns * dy, nvar);
    hipLaunchKernelGGL(dist_predict_finish_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, pmean, Ms, ns * dy, pvar,
                       variance, kss, nvar);
    GPN_LAUNCH_CHECK();
    return GPN_OK;
  }
  if (!with_grad) return GPN_OK;

  // ---- backward on the same grid (gptorch_amd/dist.py BlockCyclicGP.backward) ------------------
  // a = Kyy^-1 (y - m) = U alpha,  G = 1/2 (a a^T - dy Kyy^-1),  dLML/dtheta = sum G o dKyy/dtheta
  const int kp = (int)round_up(dy, 16);
  double *alphaT = work + L.alphaT, *aT = work + L.aT, *al = work + L.al, *part = work + L.part;
  const int64_t rpad = std::max<int64_t>(g.nrow_t, 1) * T;
  // (1) alpha^T replicated: every residual holder contributes its tile columns, one all-reduce
  if (g.has_res)
    for (int64_t lj = 0; lj < g.ncol_t; ++lj) {
      const int64_t J = g.my_c + lj * g.pc;
      GPN_HIP_CHECK(hipMemcpy2DAsync(alphaT + J * T, (size_t)n * 8, A + g.res_off * g.ld + lj * T, (size_t)g.ld * 8,
                                     (size_t)g.rows_of(J) * 8, (size_t)dy, hipMemcpyDeviceToDevice, s));
    }
  if (R.comm) { rc = R.comm->allreduce(R.comm->ctx, alphaT, (int64_t)dy * n, s); if (rc != GPN_OK) return rc; }
  // (2) a^T = alpha^T U^T: my identity rows x my tile columns give a partial sum
  for (int64_t lj = 0; lj < g.ncol_t; ++lj) {
    const int64_t J = g.my_c + lj * g.pc;
    GPN_HIP_CHECK(hipMemcpy2DAsync(al + lj * T, (size_t)g.ld * 8, alphaT + J * T, (size_t)n * 8, (size_t)g.rows_of(J) * 8, (size_t)dy,
                                   hipMemcpyDeviceToDevice, s));
  }
  if (nrr > 0 && ncr > 0) {
    rc = gemm_nt(s, dy, nrr, g.ld, 1.0, al, g.ld, A + g.id_off * g.ld, g.ld, 0.0, part, rpad, 0);
    if (rc != GPN_OK) return rc;
    for (int64_t li = 0; li < g.nrow_t; ++li) {
      const int64_t I = g.my_r + li * g.pr;
      GPN_HIP_CHECK(hipMemcpy2DAsync(aT + I * T, (size_t)n * 8, part + li * T, (size_t)rpad * 8, (size_t)g.rows_of(I) * 8, (size_t)dy,
                                     hipMemcpyDeviceToDevice, s));
    }
  }
  if (R.comm) { rc = R.comm->allreduce(R.comm->ctx, aT, (int64_t)dy * n, s); if (rc != GPN_OK) return rc; }
  // (3) Kyy^-1 = U U^T on the owners of the matrix tiles: tile column K of U travels like a factorisation panel
  double* Kinv = work + L.kinv;
  {
    const int rs = g.my_c % g.pr;
    const int64_t first = (g.my_c - rs) / g.pr;          // slot of tile J = c in process row rs's buffer
    bool rp = false, cp = false;
    double* lft = R.start_rows(0, g.id_off, g.id_off + g.rows_le(0, g.my_r) * T, rp);
    for (int64_t K = 0; K < nt && R.rc == GPN_OK; ++K) {
      const int64_t nK = round_up(g.rows_of(K), 16);
      R.wait_rows(K, rp);
      const int64_t ncol = g.cols_le(K);
      double* rgt = R.start_cols(K, lft, first, ncol, cp);
      R.wait_cols(K, cp);
      double* nlft = nullptr;
      bool nrp = false;
      if (K + 1 < nt) nlft = R.start_rows(K + 1, g.id_off, g.id_off + g.rows_le(K + 1, g.my_r) * T, nrp);
      const int64_t hi = g.rows_le(K, g.my_r) * T;
      const int64_t r0 = g.rows_le((int64_t)g.my_c - 1, g.my_r) * T;   // my first tile column starts here, the next Pc/Pr tiles lower
      if (ncol > 0 && r0 < hi)
        R.ok(gemm_nt_stair(s, hi - r0, ncol, T, nK, 1.0, lft + r0 * T, T, rgt, T, 1.0, Kinv + r0 * g.ld, g.ld, (g.pc / g.pr) * T, 0));
      lft = nlft;
      rp = nrp;
    }
    if (R.rc != GPN_OK) return R.rc;
  }
  // (4) G in place over the stacked rows of every local tile column, contracted with dK/dtheta by the native sweep
  double *arow = work + L.arow, *acol = work + L.acol, *gwork = work + L.gwork, *gout = work + L.gout, *acc = work + L.acc;
  if (nrr > 0) {
    hipLaunchKernelGGL(dist_a_pad_kernel, dim3((unsigned)((nrr + 255) / 256)), dim3(256), 0, s, aT, n, dy, T, g.my_r, g.pr, nrr, arow, kp);
    GPN_LAUNCH_CHECK();
  }
  if (ncr > 0) {
    hipLaunchKernelGGL(dist_a_pad_kernel, dim3((unsigned)((ncr + 255) / 256)), dim3(256), 0, s, aT, n, dy, T, g.my_c, g.pc, ncr, acol, kp);
    GPN_LAUNCH_CHECK();
  }
  for (int64_t lj = 0; lj < g.ncol_t; ++lj) {
    const int64_t J = g.my_c + lj * g.pc, nJ = g.rows_of(J);
    const int64_t li0 = g.rows_le(J - 1, g.my_r);
    int64_t r0 = li0 * T;
    if (r0 >= nrr) continue;
    const double* xj = Xcol + lj * T * d;
    rc = gemm_nt(s, nrr - r0, nJ, kp, 0.5, arow + r0 * kp, kp, acol + lj * T * kp, kp, -0.5 * dy, Kinv + r0 * g.ld + lj * T, g.ld, 0);
    if (rc != GPN_OK) return rc;
    if (li0 * g.pr + g.my_r == J) {                     // diagonal tile: counted once; d/d noise = tr G
      double* G = Kinv + r0 * g.ld + lj * T;
      hipLaunchKernelGGL(dist_trace_add_kernel, dim3(1), dim3(256), 0, s, G, g.ld, nJ, acc, 1 + nls);
      GPN_LAUNCH_CHECK();
      rc = gpn_kernel_grad(stream, kind, Xrow + r0 * d, nJ, xj, nJ, d, variance, length_scales, nls, G, g.ld, gwork, gout);
      if (rc != GPN_OK) return rc;
      hipLaunchKernelGGL(dist_axpy_kernel, dim3(1), dim3(256), 0, s, acc, gout, 1.0, 1 + nls);
      GPN_LAUNCH_CHECK();
      r0 += T;
    }
    if (r0 < nrr) {                                      // symmetric partners (J, I): x 2
      rc = gpn_kernel_grad(stream, kind, Xrow + r0 * d, nrr - r0, xj, nJ, d, variance, length_scales, nls,
                           Kinv + r0 * g.ld + lj * T, g.ld, gwork, gout);
      if (rc != GPN_OK) return rc;
      hipLaunchKernelGGL(dist_axpy_kernel, dim3(1), dim3(256), 0, s, acc, gout, 2.0, 1 + nls);
      GPN_LAUNCH_CHECK();
    }
  }
  if (R.comm) { rc = R.comm->allreduce(R.comm->ctx, acc, 2 + nls, s); if (rc != GPN_OK) return rc; }
  GPN_HIP_CHECK(hipMemcpyAsync(grads, acc, (size_t)(2 + nls) * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (grad_resid) {
    hipLaunchKernelGGL(dist_neg_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, aT, n, dy, grad_resid);
    GPN_LAUNCH_CHECK();
  }
  return GPN_OK;
}

extern "C" int64_t gpn_dist_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile) {
  DistGeom g;
  if (make_geom(g, rank, pr, pc, n, dy, tile) != GPN_OK || d <= 0 || dy <= 0) return -1;
  return make_layout(g, d).total * (int64_t)sizeof(double);
}
extern "C" int64_t gpn_dist_grad_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile) {
  DistGeom g;
  if (make_geom(g, rank, pr, pc, n, dy, tile, true) != GPN_OK || d <= 0 || dy <= 0) return -1;
  return make_layout(g, d).total * (int64_t)sizeof(double);
}

extern "C" int64_t gpn_dist_predict_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t ns, int64_t tile, int full_cov) {
  DistGeom g;
  if (ns < 0 || dy <= 0 || d <= 0 || make_geom(g, rank, pr, pc, n, dy + (int)ns, tile) != GPN_OK) return -1;
  return (make_layout(g, d).total + round_up(ns * dy + (full_cov ? 2 * ns * ns : ns), 32)) * (int64_t)sizeof(double);
}

extern "C" int gpn_dist_predict(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                                const double* X, int64_t n, int d, const double* Y, int dy, const double* Xs, int64_t ns, const double* Ms,
                                const double* variance, const double* length_scales, int nls, const double* noise,
                                int64_t tile, int full_cov, double* work, int64_t work_bytes, double* out4, double* mean, double* var) {
  if (ns <= 0) return -13;
  return dist_evaluate(stream, comm, rank, pr, pc, kind, X, n, d, Y, dy, variance, length_scales, nls, noise, tile, work, work_bytes,
                       out4, false, nullptr, nullptr, Xs, ns, Ms, full_cov, mean, var);
}

extern "C" int gpn_dist_lml_forward(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                                    const double* X, int64_t n, int d, const double* Y, int dy,
                                    const double* variance, const double* length_scales, int nls, const double* noise,
                                    int64_t tile, double* work, int64_t work_bytes, double* out4) {
  return dist_evaluate(stream, comm, rank, pr, pc, kind, X, n, d, Y, dy, variance, length_scales, nls, noise, tile, work, work_bytes,
                       out4, false, nullptr, nullptr);
}

extern "C" int gpn_dist_lml_grad(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                                 const double* X, int64_t n, int d, const double* Y, int dy,
                                 const double* variance, const double* length_scales, int nls, const double* noise,
                                 int64_t tile, double* work, int64_t work_bytes, double* out4, double* grads, double* grad_resid) {
  return dist_evaluate(stream, comm, rank, pr, pc, kind, X, n, d, Y, dy, variance, length_scales, nls, noise, tile, work, work_bytes,
                       out4, true, grads, grad_resid);
}

// ---- gpn_dist_lml_refine: the refinement step of the quadratic form (refine.hip) on the grid ------------------------
// The sequence of gptorch_amd/dist.py BlockCyclicGP._refine over the callback table (world all-reduces only: a sum in which all
// ranks but one contribute zeros is the broadcast).  `work` is the forward call's workspace, untouched since: it holds L and alpha.
struct DistRefineLayout { int64_t alpha, a, owed, buf, sj, aj, ar, ka, U, S, W, winv, gwork, rwork, total; int64_t lv, lds, q0, q1; int ndiag; };
static DistRefineLayout refine_layout_dist(const DistGeom& g, std::vector<int64_t>* rec = nullptr) {
  DistRefineLayout R;
  const int64_t T = g.T;
  R.lv = g.nt * T;
  R.lds = round_up(g.n, LEAF);
  R.ndiag = 0;
  for (int64_t J = g.my_c; J < g.nt; J += g.pc) if (J % g.pr == g.my_r) ++R.ndiag;
  const int64_t ntri = gpn_refine_tile_count(g.n);
  const int64_t world = (int64_t)g.pr * g.pc;
  R.q0 = ntri * g.rank / world;
  R.q1 = ntri * (g.rank + 1) / world;
  int64_t o = 0;
  auto take = [&](int64_t cnt) {
    const int64_t at = o;
    o += round_up(std::max<int64_t>(cnt, 1), 32);
    if (rec) { rec->push_back(at); rec->push_back(o - at); }
    return at;
  };
  R.alpha = take((int64_t)g.dy * R.lv);
  R.a = take((int64_t)g.dy * R.lv);
  R.owed = take((int64_t)g.dy * std::max<int64_t>(g.ncol_t, 1) * T);
  R.buf = take((int64_t)g.dy * T);
  R.sj = take((int64_t)g.dy * T);
  R.aj = take((int64_t)g.dy * T);
  R.ar = take((int64_t)g.dy * R.lds);
  R.ka = take((int64_t)g.dy * R.lds * 2);
  R.U = take(T * T);
  R.S = take(T * T);
  R.W = take((int64_t)std::max(R.ndiag, 1) * T * T);
  R.winv = take(gpn_winv_bytes(T) / 8);
  R.gwork = take(gpn_gemv_t_work_bytes(T, std::max<int64_t>(g.ncol_t, 1) * T, g.dy) / 8);
  R.rwork = take(gpn_refine_resid_part_work_bytes(g.dy, R.q1 - R.q0) / 8);
  R.total = o;
  return R;
}

// The sub-buffers of a rank's workspace as (offset, reserved size) pairs in doubles, in the order the layout declares them
// (which = 0: gpn_dist_lml_forward -- A, left[2], right[2], diag, winv, xrow, xcol, stats, info, sums; 1: gpn_dist_lml_grad --
// the same + kinv, alphaT, aT, al, part, arow, acol, gwork, gout, acc; 2: gpn_dist_lml_refine -- alpha, a, owed, buf, sj, aj,
// ar, ka, U, S, W, winv, gwork, rwork).  Pure host function: the layout sweep of the sanitizer leg (tests/test_host_sanitizer.py)
// checks every offset + size against gpn_dist_*_work_bytes and against the sizes the driver's calls need.  Returns the number of
// sub-buffers (pairs written: min(that, cap)), < 0: bad arguments.
extern "C" int gpn_dist_layout(int which, int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile, int64_t* out, int cap) {
  if (which < 0 || which > 2) return -1;
  if (d <= 0 || dy <= 0 || n <= 0) return -5;
  if (!out && cap > 0) return -9;
  DistGeom g;
  const int rc = make_geom(g, rank, pr, pc, n, dy, tile, which == 1);
  if (rc != GPN_OK) return rc;
  std::vector<int64_t> rec;
  if (which == 2) (void)refine_layout_dist(g, &rec);
  else (void)make_layout(g, d, &rec);
  const int cnt = (int)(rec.size() / 2);
  for (int i = 0; i < std::min(cnt, cap); ++i) { out[2 * i] = rec[2 * i]; out[2 * i + 1] = rec[2 * i + 1]; }
  return cnt;
}

extern "C" int64_t gpn_dist_lml_refine_work_bytes(int rank, int pr, int pc, int64_t n, int d, int dy, int64_t tile) {
  (void)d;
  DistGeom g;
  if (n <= 0 || dy <= 0 || make_geom(g, rank, pr, pc, n, dy, tile) != GPN_OK) return -1;
  return refine_layout_dist(g).total * (int64_t)sizeof(double);
}

extern "C" int gpn_dist_lml_refine(void* stream, const gpn_dist_comm* comm, int rank, int pr, int pc, int kind,
                                   const double* X, int64_t n, int d, const double* Y, int dy,
                                   const double* variance, const double* length_scales, int nls, const double* noise,
                                   int64_t tile, const double* work, double* rwork, int64_t rwork_bytes, double* out4) {
  DistGeom g;
  int rc = make_geom(g, rank, pr, pc, n, dy, tile);
  if (rc != GPN_OK) return rc;
  if (pr * pc > 1 && (!comm || !comm->allreduce)) return -2;
  if (kind < GPN_RBF || kind > GPN_PERIODIC) return -6;
  if (!X) return -7;
  if (n <= 0) return -8;
  if (d <= 0) return -9;
  if (!Y) return -10;
  if (dy <= 0) return -11;
  if (!variance || !length_scales || !noise) return -12;
  if (nls != 1 && nls != d) return -14;
  if (!work) return -16;
  if (!rwork) return -17;
  if (!out4) return -19;
  const DistLayout L = make_layout(g, d);
  const DistRefineLayout R = refine_layout_dist(g);
  if (rwork_bytes < R.total * (int64_t)sizeof(double)) return -18;
  if (reinterpret_cast<uintptr_t>(rwork) & 255) return GPN_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const gpn_dist_comm* cm = (pr * pc > 1 || (comm && (comm->flags & GPN_DIST_FORCE_COLLECTIVES))) ? comm : nullptr;
  const int64_t T = g.T, nt = g.nt, lv = R.lv;
  const double* A = work + L.A;
  double* alpha = rwork + R.alpha;
  double* a = rwork + R.a;
  double* owed = rwork + R.owed;
  double* buf = rwork + R.buf;
  double* sj = rwork + R.sj;
  double* aj = rwork + R.aj;
  const int64_t ldo = std::max<int64_t>(g.ncol_t, 1) * T;
  auto zero = [&](double* ptr, int64_t cnt) { return hipMemsetAsync(ptr, 0, (size_t)cnt * sizeof(double), s); };
#define GPN_RC(expr) do { rc = (expr); if (rc != GPN_OK) return rc; } while (0)
  // 1. alpha^T from the residual segment of the factor, replicated
  GPN_HIP_CHECK(zero(alpha, (int64_t)dy * lv));
  GPN_HIP_CHECK(zero(a, (int64_t)dy * lv));
  GPN_HIP_CHECK(zero(owed, (int64_t)dy * ldo));
  if (g.has_res)
    for (int64_t lj = 0, J = g.my_c; J < nt; J += g.pc, ++lj)
      GPN_RC(gpn_copy_matrix(stream, A + g.res_off * g.ld + lj * T, dy, g.rows_of(J), g.ld, alpha + J * T, lv, 0));
  if (cm) GPN_RC(cm->allreduce(cm->ctx, alpha, (int64_t)dy * lv, stream));
  // 2. my diagonal tiles' inverses (row-major L^-1), before the serial sweep
  {
    int slot = 0;
    for (int64_t lj = 0, J = g.my_c; J < nt; J += g.pc, ++lj) {
      if (J % g.pr != g.my_r) continue;
      const int64_t li = (J - g.my_r) / g.pr, nJ = g.rows_of(J);
      const double* Lt = A + li * T * g.ld + lj * T;
      double* U = rwork + R.U;
      double* S = rwork + R.S;
      double* W = rwork + R.W + (int64_t)slot * T * T;
      GPN_HIP_CHECK(zero(U, T * T));
      GPN_HIP_CHECK(zero(S, T * T));
      GPN_RC(gpn_trtri_diag(stream, Lt, nJ, g.ld, rwork + R.winv, nullptr));
      if (nJ > 2 * LEAF) GPN_RC(gpn_trtri_upper_ws(stream, Lt, nJ, g.ld, rwork + R.winv, U, T, S, T));
      else GPN_RC(gpn_trtri_upper(stream, Lt, nJ, g.ld, rwork + R.winv, U, T));
      GPN_RC(gpn_transpose(stream, U, T, T, T, W, T));
      ++slot;
    }
  }
  // 3. a = L^-T alpha, tile row by tile row from the bottom
  {
    int slot = R.ndiag;                                   // my diagonal tiles are met in descending order
    for (int64_t J = nt - 1; J >= 0; --J) {
      const int64_t nJ = g.rows_of(J);
      const int cj = (int)(J % g.pc), rj = (int)(J % g.pr);
      const bool owner = g.my_r == rj && g.my_c == cj;
      GPN_HIP_CHECK(zero(buf, (int64_t)dy * T));
      if (g.my_c == cj) {
        const int64_t lj = (J - g.my_c) / g.pc;
        GPN_RC(gpn_copy_matrix(stream, owed + lj * T, dy, nJ, ldo, buf, T, 0));
      }
      if (cm) GPN_RC(cm->allreduce(cm->ctx, buf, (int64_t)dy * T, stream));      // (only process column cj contributes)
      GPN_HIP_CHECK(zero(aj, (int64_t)dy * T));
      if (owner) {
        --slot;
        GPN_RC(gpn_copy_matrix(stream, alpha + J * T, dy, T, lv, sj, T, 0));
        hipLaunchKernelGGL(dist_axpy_kernel, dim3(1), dim3(1024), 0, s, sj, buf, -1.0, (int)(dy * T));
        GPN_LAUNCH_CHECK();
        GPN_RC(gpn_gemv_t_acc(stream, rwork + R.W + (int64_t)slot * T * T, T, nJ, nJ, sj, T, dy, aj, T, rwork + R.gwork));
      }
      if (cm) GPN_RC(cm->allreduce(cm->ctx, aj, (int64_t)dy * T, stream));        // = broadcast from the owner
      GPN_RC(gpn_copy_matrix(stream, aj, dy, nJ, T, a + J * T, lv, 0));
      if (g.my_r == rj && J > 0) {
        const int64_t ncl = g.cols_le(J - 1), li = (J - g.my_r) / g.pr;
        if (ncl > 0) GPN_RC(gpn_gemv_t_acc(stream, A + li * T * g.ld, g.ld, nJ, ncl * T, aj, T, dy, owed, ldo, rwork + R.gwork));
      }
    }
  }
  // 4. my share of Kyy a (double-double), summed over the ranks; 5. finish
  double* ar = rwork + R.ar;
  double* ka = rwork + R.ka;
  GPN_RC(gpn_copy_matrix(stream, a, dy, R.lds, lv, ar, R.lds, 0));
  GPN_HIP_CHECK(zero(ka, (int64_t)dy * R.lds * 2));
  GPN_RC(gpn_refine_resid_part(stream, kind, X, n, d, variance, length_scales, nls, noise, ar, dy, R.q0, R.q1, rwork + R.rwork, ka));
  if (cm) GPN_RC(cm->allreduce(cm->ctx, ka, (int64_t)dy * R.lds * 2, stream));
  GPN_RC(gpn_refine_finish(stream, Y, nullptr, ar, ka, n, dy, out4));
#undef GPN_RC
  return GPN_OK;
}
