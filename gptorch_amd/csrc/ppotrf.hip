// The blocked fp64 Cholesky of potrf.hip as ONE persistent launch: a dataflow over 128 x 128 tiles (round 6).
// Replaces torch.cholesky under functions.cholesky (functions.py:46-47) for the latency regime N = 4096 ... 16384, where the
// launch-per-step driver leaves the chip idle: its N / 128 leaves run one after the other on ONE compute unit (22 % of a C2
// evaluation) and nothing can be put beside them from another stream -- the leaf needs an EMPTY compute unit and priorities only
// order dispatch (LAB.md 8-1c, 10-10; round 6's persistent bulk beside the launch-based chain, profiles/r6_lookahead_*: the
// chain's kernels ran 4-5 x slower next to resident tile workgroups: on one SIMD the older wave wins the matrix pipe).
//
// Here one workgroup per compute unit stays resident for the whole factorisation and the schedule is a task graph:
//   LEAF(k)            the 128 x 128 diagonal block: Cholesky + inverse (leaf16_body.h, the shipped leaf)
//   TRSM(i, k)         X(i,k) = A(i,k) W_k^T, in place, one 128-row tile (the arithmetic of colpanel.hip, mode 0)
//   UPD(i, j, k0..k1)  A(i,j) -= X(i, k0..k1) X(j, k0..k1)^T: one 128 x 128 tile of ONE of the shipped driver's update launches
//                      (next-column K = 128, inner-panel trapezoid K = 256, outer-panel K = 1024; gemm_tile.h, the shipped tile)
// with exactly the K grouping and per-entry summation order of potrf_lookahead: THE FACTOR IS BIT-IDENTICAL to gpn_potrf_lower's.
// Every task has <= 3 predecessors (TRSMs of a tile row complete in column order, so the last column of a K group stands for
// all of them) and a successor list; a finished task decrements its successors' counters and pushes the ones that reach zero
// onto one of three queues.  Roles are fixed at start by ticket: the first R workgroups serve queue 0 only -- the tasks inside
// the current outer panel's diagonal triangle, i.e. the critical chain leaf -> solve -> update -> leaf -- and never hold a
// long tile when a leaf becomes ready; the others serve queue 1 (tiles whose row panel is at most one outer panel below their
// column panel: what the NEXT panel's chain waits for), then queue 2 (the bulk), then queue 0.
// A workgroup only ever COMMITS to a task whose predecessors are done, so there is nothing to deadlock on; all spins are
// bounded and end in info = GPN_INFO_INTERNAL.
//
// Visibility between workgroups (MI355X_MICROARCH.md, "Valid forms"): every store of a tile that another workgroup reads is an
// agent-scope write-through (`sc1`) store; every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier,
// THEN wave 0 touches the successors' counters; the consumer pops (agent-scope atomics), runs ONE agent-scope acquire
// (invalidates its compute unit's L1), waits for it, meets at a barrier and reads with plain loads / LDS-DMA.
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "gpn_common.h"
#include "gemm_tile.h"
#include "leaf16_body.h"

namespace gpn {

enum { PT_LEAF = 0, PT_TRSM = 1, PT_UPD = 2 };
struct PTask {                 // 32 bytes
  int16_t type, queue;
  int16_t i, j, k0, k1;        // tile row / column (TRSM, LEAF: j = the column block k); UPD: K range in column blocks [k0, k1)
  int32_t ndeps;
  int32_t succ_begin, succ_end;
  int32_t pad[2];
};
static_assert(sizeof(PTask) == 32, "PTask layout");

// runtime words (ints): control, then per queue a head and a tail on lines of their own
constexpr int RT_TICKET = 0, RT_COMPLETED = 1, RT_DONE = 2, RT_ABORT = 3, RT_Q0 = 32, RT_QSTRIDE = 64, RT_FIXED = RT_Q0 + 3 * RT_QSTRIDE;
constexpr int PP_NQ = 3;

struct PArgs {
  double* A;
  int64_t lda;
  double* winv;
  int32_t* info;
  int T, e;                    // matrix tile rows; extra rows (tile row T when e > 0)
  const PTask* tasks;
  const int* succ;
  int* rt;
  int ntasks;
  int off_dep;
  int off_slots[PP_NQ];
  int R;                       // chain workgroups
  int spin_limit;
  unsigned long long* trace;   // tools' build: 4 stamps (100 MHz clock) + ticket per task, or NULL
};

__device__ __forceinline__ int pp_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pp_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// thread 0: the next task of this workgroup's queues, -1 when the factorisation is over (or aborted)
__device__ __forceinline__ int pp_pop(const PArgs& a, const bool chain) {
  int* rt = a.rt;
  const int order[3] = {chain ? 0 : 1, chain ? -1 : 2, chain ? -1 : 0};
  for (int spins = 0;; ++spins) {
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const int q = order[o];
      if (q < 0) continue;
      int* head = rt + RT_Q0 + RT_QSTRIDE * q;
      int* tail = head + 32;
      int h = pp_ld(head);
      const int t = pp_ld(tail);
      while (h < t) {
        int expect = h;
        if (__hip_atomic_compare_exchange_strong(head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          const int* slot = rt + a.off_slots[q] + h;
          int task, s2 = 0;
          while ((task = pp_ld(slot)) < 0) {           // the producer bumped the tail and is about to fill the slot
            if (++s2 > a.spin_limit) { pp_st(rt + RT_ABORT, 3); return -1; }
            __builtin_amdgcn_s_sleep(1);
          }
          return task;
        }
        h = expect;                                    // somebody else took it
      }
    }
    if (pp_ld(rt + RT_DONE) | pp_ld(rt + RT_ABORT)) return -1;
    if (spins > a.spin_limit) { pp_st(rt + RT_ABORT, 2); return -1; }
    __builtin_amdgcn_s_sleep(4);
  }
}

// wave 0, after the workgroup's stores have drained: release the successors
__device__ __forceinline__ void pp_notify(const PArgs& a, const PTask& tk, const int lane) {
  int* rt = a.rt;
  for (int base = tk.succ_begin; base < tk.succ_end; base += 64) {
    const int idx = base + lane;
    int s = -1, q = -1;
    bool ready = false;
    if (idx < tk.succ_end) {
      s = a.succ[idx];
      const int old = __hip_atomic_fetch_sub(rt + a.off_dep + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ready = old == 1;
      if (ready) q = a.tasks[s].queue;
    }
#pragma unroll
    for (int qq = 0; qq < PP_NQ; ++qq) {
      const bool mine = ready && q == qq;
      const unsigned long long m = __ballot(mine);
      if (m == 0) continue;
      const int cnt = __popcll(m);
      const int leader = __ffsll((long long)m) - 1;
      int basei = 0;
      if (lane == leader) basei = __hip_atomic_fetch_add(rt + RT_Q0 + RT_QSTRIDE * qq + 32, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      basei = __shfl(basei, leader, 64);
      if (mine) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        pp_st(rt + a.off_slots[qq] + basei + rank, s);
      }
    }
  }
  if (lane == 0) {
    const int c = __hip_atomic_fetch_add(rt + RT_COMPLETED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c == a.ntasks - 1) pp_st(rt + RT_DONE, 1);
  }
}

// X = B W^T in place for one tile of `rows` <= 128 rows (B [rows, 128] at ld lda; W = the leaf's inverse, [128][128], lower
// triangular).  colpanel.hip mode 0's arithmetic entry for entry: per 16 x 16 output tile the 8-k groups j = 0 .. (last
// column of the tile) / 8 in order, the pair of MFMAs (k = 8j + {0,2,4,6}, then + {1,3,5,7}) per group, accumulators from zero.
// The whole tile is parked in LDS (row stride 1 KiB + 16 B) with all its loads in flight at once; the 8 matrix waves own one
// 16-column tile each ({w, 7 - w} on the two waves of a SIMD: equal work in the triangular product); 12 waves load.
constexpr int PP_LDS_ROW = LEAF * 8 + 16;
__device__ __forceinline__ void pp_trsm_tile(double* B, const int64_t lda, const int rows, const double* W, const int tid) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef double d4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char pp_lds[];
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int rt16 = (rows + 15) >> 4;                   // 16-row tiles
  const int total = rt16 * 16 * 64;                    // 16-byte segments of the padded tile
  constexpr int PER = (LEAF * 64 + L16_THREADS - 1) / L16_THREADS;      // 11
  d2 v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = tid + u * L16_THREADS;
    const int row = idx >> 6, seg = idx & 63;
    v[u] = (idx < total && row < rows) ? *reinterpret_cast<const d2*>(B + (int64_t)row * lda + seg * 2) : d2{0.0, 0.0};
  }
  const bool mm = wave < 8;
  const int ct = wave < 4 ? wave : 11 - wave;          // (waves w and w + 4 share a SIMD)
  d2 b[16];
  if (mm) {
    const double* wsrc = W + (int64_t)(ct * 16 + lr) * LEAF + 2 * lq;
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = (8 * j <= ct * 16 + 15) ? *reinterpret_cast<const d2*>(wsrc + 8 * j) : d2{0.0, 0.0};
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = tid + u * L16_THREADS;
    if (idx < total) *reinterpret_cast<d2*>(pp_lds + (idx >> 6) * PP_LDS_ROW + (idx & 63) * 16) = v[u];
  }
  __syncthreads();
  if (mm) {
#pragma unroll 1
    for (int i = 0; i < rt16; ++i) {
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (8 * j > ct * 16 + 15) continue;            // W[col][k] = 0 for k > col (wave-uniform)
        const d2 av = *reinterpret_cast<const d2*>(pp_lds + (i * 16 + lr) * PP_LDS_ROW + (4 * j + lq) * 16);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b[j].y, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i * 16 + lq + 4 * r;
        if (row < rows) __hip_atomic_store(B + (int64_t)row * lda + ct * 16 + lr, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

__global__ __launch_bounds__(L16_THREADS) void ppotrf_kernel(PArgs a) {
  __shared__ int s_task, s_role;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(a.rt + RT_TICKET, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_role = t < a.R ? 1 : 0;
  }
  __syncthreads();
  const bool chain = s_role != 0;
  unsigned long long t_pop0 = 0, t_pop1 = 0, t_run = 0;
  for (;;) {
    if (tid == 0) {
      if (a.trace) t_pop0 = __builtin_amdgcn_s_memrealtime();
      const int t = pp_pop(a, chain);
      s_task = t;
      if (a.trace) t_pop1 = __builtin_amdgcn_s_memrealtime();
      if (t >= 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (a.trace) t_run = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    const int task = __builtin_amdgcn_readfirstlane(s_task);
    if (task < 0) break;
    // the thread index, opaque per iteration: the task bodies' lane-derived addresses must not be hoisted out of this loop
    // (they were: 201 spilled registers)
    int tid_it = tid;
    asm volatile("" : "+v"(tid_it));
    const PTask tk = a.tasks[task];
    const int rows_i = tk.i < a.T ? LEAF : a.e;
    if (tk.type == PT_LEAF) {
      Leaf16Args la;
      la.A = a.A + (int64_t)tk.i * LEAF * (a.lda + 1);
      la.lda = a.lda;
      la.kb = LEAF;
      la.col0 = tk.i * LEAF;
      la.winv = a.winv + (int64_t)tk.i * LEAF * LEAF;
      la.info = a.info;
      la.sA = la.sW = la.sInfo = 0;
      leaf16_body<false, true>(la, nullptr, 0, tid_it);
    } else if (tk.type == PT_TRSM) {
      pp_trsm_tile(a.A + (int64_t)tk.i * LEAF * a.lda + (int64_t)tk.j * LEAF, a.lda, rows_i, a.winv + (int64_t)tk.j * LEAF * LEAF, tid_it);
    } else {
      GemmArgs g;
      g.A = a.A + (int64_t)tk.i * LEAF * a.lda + (int64_t)tk.k0 * LEAF;
      g.B = a.A + (int64_t)tk.j * LEAF * a.lda + (int64_t)tk.k0 * LEAF;
      g.C = a.A + (int64_t)tk.i * LEAF * a.lda + (int64_t)tk.j * LEAF;
      g.lda = g.ldb = g.ldc = a.lda;
      g.M = rows_i; g.N = LEAF; g.K = (tk.k1 - tk.k0) * LEAF;
      g.mt = g.nt = 1;
      g.lower = tk.i == tk.j ? 1 : 0;
      g.q_off = g.q_cnt = g.q_mt = 0;
      g.lds_pad_kb = 0; g.group_h = 8; g.thin = 1;
      g.st_blk = g.st_step = g.st_diag = 0;
      g.tri = 0;
      g.alpha = -1.0; g.beta = 1.0;
      g.batch = 1; g.sA = g.sB = g.sC = 0;
      g.inner = 0; g.sA2 = g.sB2 = g.sC2 = 0;
      gemm_nt_tile<128, 128, 32, 64, true, 2, false, true, true, true>(g, 0, 1, true, tid_it);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave's write-through stores have left
    __syncthreads();
    unsigned long long t_end = 0;
    if (a.trace && tid == 0) t_end = __builtin_amdgcn_s_memrealtime();
    if (wave == 0) pp_notify(a, tk, lane);
    if (a.trace && tid == 0) {
      unsigned long long* o = a.trace + 6 * (size_t)task;
      o[0] = t_pop0; o[1] = t_pop1; o[2] = t_run; o[3] = t_end; o[4] = __builtin_amdgcn_s_memrealtime();
      o[5] = ((unsigned long long)(chain ? 1 : 0) << 32) | (unsigned)blockIdx.x;
    }
  }
  if (tid == 0 && pp_ld(a.rt + RT_ABORT) != 0)
    __hip_atomic_store(a.info, (int32_t)GPN_INFO_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the plan (host) ---------------------------------------------------------------------------------------------------------
struct PPlan {
  int64_t n = 0, e = 0;
  int T = 0, TR = 0;
  std::vector<PTask> tasks;
  std::vector<int> succ;
  std::vector<int> dep0;
  std::vector<int> initial[PP_NQ];
  int qcount[PP_NQ] = {0, 0, 0};
  int off_dep = 0, off_slots[PP_NQ] = {0, 0, 0}, rt_ints = 0;
};

static bool pp_supported(int64_t n, int64_t e) {
  if (n % LEAF != 0 || n < 2560 || n >= 20480 || e < 0 || e > 16) return false;
  int64_t w[3];
  return gpn_potrf_panel_levels(n, w) == 2 && w[0] == 256 && w[1] % w[0] == 0;
}

// Walks gpn_potrf_lower's schedule for this size (potrf.hip potrf_lookahead: inner panels of w0 columns -- two blocks --, outer
// panels of w1) and emits one task per 128 x 128 tile of every launch; predecessors from the last writer of each tile.
static int pp_build(int64_t n, int64_t e, PPlan& P) {
  if (!pp_supported(n, e)) return GPN_E_UNSUPPORTED;
  int64_t w[3];
  gpn_potrf_panel_levels(n, w);
  const int PW = (int)(w[0] / LEAF), OW = (int)(w[1] / LEAF);
  const int T = (int)(n / LEAF), TR = T + (e > 0 ? 1 : 0);
  P.n = n; P.e = e; P.T = T; P.TR = TR;
  std::vector<int> lastw((size_t)TR * T, -1);            // last task that wrote tile (i, j)
  std::vector<int> trsm((size_t)TR * T, -1);
  std::vector<std::vector<int>> preds;
  auto panel = [&](int t) { return t < T ? t / OW : (1 << 20); };
  auto add = [&](int type, int i, int j, int k0, int k1, std::initializer_list<int> deps, bool top) {
    PTask t{};
    t.type = (int16_t)type; t.i = (int16_t)i; t.j = (int16_t)j; t.k0 = (int16_t)k0; t.k1 = (int16_t)k1;
    std::vector<int> d;
    for (int x : deps) if (x >= 0 && std::find(d.begin(), d.end(), x) == d.end()) d.push_back(x);
    t.ndeps = (int)d.size();
    const int pi = panel(i), pj = panel(j);
    t.queue = (int16_t)((!top && pi == pj) ? 0 : (pi - pj <= 1 ? 1 : 2));
    P.tasks.push_back(t);
    preds.push_back(d);
    return (int)P.tasks.size() - 1;
  };
  auto upd = [&](int i, int j, int k0, int k1, bool top) {
    const int kl = k1 - 1;
    const int id = add(PT_UPD, i, j, k0, k1, {trsm[(size_t)i * T + kl], trsm[(size_t)j * T + kl], lastw[(size_t)i * T + j]}, top);
    lastw[(size_t)i * T + j] = id;
  };
  for (int p0 = 0; p0 < T; p0 += PW) {
    const int pend = std::min(p0 + PW, T);
    for (int k = p0; k < pend; ++k) {
      const int leaf = add(PT_LEAF, k, k, k, k + 1, {lastw[(size_t)k * T + k]}, false);
      lastw[(size_t)k * T + k] = leaf;
      for (int i = k + 1; i < TR; ++i) {
        const int id = add(PT_TRSM, i, k, k, k + 1, {leaf, lastw[(size_t)i * T + k]}, false);
        trsm[(size_t)i * T + k] = id;
        lastw[(size_t)i * T + k] = id;
      }
      if (k + 1 < pend)                                  // next column block of the inner panel: K = 128
        for (int i = k + 1; i < TR; ++i) upd(i, k + 1, k, k + 1, false);
    }
    if (pend >= T) break;
    const bool top = pend % OW == 0;
    if (!top) {                                          // trapezoid: to the end of the enclosing outer panel, K = the inner panel
      const int oend = std::min(T, (pend / OW + 1) * OW);
      for (int j = pend; j < oend; ++j)
        for (int i = j; i < TR; ++i) upd(i, j, p0, pend, false);
    } else {                                             // everything right of the outer panel, K = its width
      for (int j = pend; j < T; ++j)
        for (int i = j; i < TR; ++i) upd(i, j, pend - OW, pend, true);
    }
  }
  const int nt = (int)P.tasks.size();
  std::vector<int> cnt(nt, 0);
  for (int t = 0; t < nt; ++t) for (int d : preds[t]) ++cnt[d];
  int off = 0;
  for (int t = 0; t < nt; ++t) { P.tasks[t].succ_begin = off; off += cnt[t]; P.tasks[t].succ_end = P.tasks[t].succ_begin; }
  P.succ.assign((size_t)off, -1);
  for (int t = 0; t < nt; ++t) for (int d : preds[t]) P.succ[(size_t)P.tasks[d].succ_end++] = t;
  P.dep0.resize(nt);
  for (int t = 0; t < nt; ++t) {
    P.dep0[t] = P.tasks[t].ndeps;
    ++P.qcount[P.tasks[t].queue];
    if (P.tasks[t].ndeps == 0) P.initial[P.tasks[t].queue].push_back(t);
  }
  P.off_dep = RT_FIXED;
  int o = P.off_dep + ((nt + 31) & ~31);
  for (int q = 0; q < PP_NQ; ++q) { P.off_slots[q] = o; o += (P.qcount[q] + 31) & ~31; }
  P.rt_ints = o;
  return GPN_OK;
}

static void pp_image(const PPlan& P, std::vector<int>& img) {
  img.assign((size_t)P.rt_ints, 0);
  for (size_t t = 0; t < P.dep0.size(); ++t) img[(size_t)P.off_dep + t] = P.dep0[t];
  for (int q = 0; q < PP_NQ; ++q) {
    for (int s = 0; s < ((P.qcount[q] + 31) & ~31); ++s) img[(size_t)P.off_slots[q] + s] = -1;
    for (size_t s = 0; s < P.initial[q].size(); ++s) img[(size_t)P.off_slots[q] + s] = P.initial[q][s];
    img[RT_Q0 + RT_QSTRIDE * q + 32] = (int)P.initial[q].size();     // tail
  }
}

struct PDevPlan {
  PPlan plan;
  PTask* d_tasks = nullptr;
  int* d_succ = nullptr;
  int* d_image = nullptr;
};
static std::mutex g_pp_mutex;
static std::map<std::tuple<int, int64_t, int64_t>, PDevPlan*> g_pp_plans;                 // (device, n, e)
static std::map<std::tuple<hipStream_t, int64_t, int64_t>, int*> g_pp_runtime;            // one runtime area per caller stream and shape

#ifdef GPN_DEBUG_SWITCHES
static thread_local int g_pp_chain_wgs = 0, g_pp_grid = 0;
static thread_local unsigned long long* g_pp_trace = nullptr;
#else
static constexpr int g_pp_chain_wgs = 0, g_pp_grid = 0;
static constexpr unsigned long long* g_pp_trace = nullptr;
#endif

// gpn_potrf_lower as one persistent launch.  GPN_E_UNSUPPORTED: this size (or a stream under capture that has no plan yet)
// stays with the launch-based driver.  The plan (task table + successor lists + the pristine runtime image) is built on the
// host and uploaded ONCE per (device, n, e); each call copies the image over the stream's runtime area and launches.
int potrf_persistent(hipStream_t s, double* A, int64_t n, int64_t e, int64_t lda, double* winv, int32_t* info) {
  if (!pp_supported(n, e)) return GPN_E_UNSUPPORTED;
  int dev = 0;
  GPN_HIP_CHECK(hipGetDevice(&dev));
  PDevPlan* dp = nullptr;
  int* rt = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_pp_mutex);
    auto it = g_pp_plans.find(std::make_tuple(dev, n, e));
    auto rit = g_pp_runtime.find(std::make_tuple(s, n, e));
    if (it == g_pp_plans.end() || rit == g_pp_runtime.end()) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return GPN_E_UNSUPPORTED;   // no allocation under capture
    }
    if (it == g_pp_plans.end()) {
      PDevPlan* p = new PDevPlan();
      int rc = pp_build(n, e, p->plan);
      if (rc != GPN_OK) { delete p; return rc; }
      std::vector<int> img;
      pp_image(p->plan, img);
      GPN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->d_tasks), p->plan.tasks.size() * sizeof(PTask)));
      GPN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->d_succ), std::max<size_t>(1, p->plan.succ.size()) * sizeof(int)));
      GPN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->d_image), img.size() * sizeof(int)));
      GPN_HIP_CHECK(hipMemcpy(p->d_tasks, p->plan.tasks.data(), p->plan.tasks.size() * sizeof(PTask), hipMemcpyHostToDevice));
      GPN_HIP_CHECK(hipMemcpy(p->d_succ, p->plan.succ.data(), p->plan.succ.size() * sizeof(int), hipMemcpyHostToDevice));
      GPN_HIP_CHECK(hipMemcpy(p->d_image, img.data(), img.size() * sizeof(int), hipMemcpyHostToDevice));
      it = g_pp_plans.emplace(std::make_tuple(dev, n, e), p).first;
    }
    dp = it->second;
    if (rit == g_pp_runtime.end()) {
      int* r = nullptr;
      GPN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&r), (size_t)dp->plan.rt_ints * sizeof(int)));
      rit = g_pp_runtime.emplace(std::make_tuple(s, n, e), r).first;
    }
    rt = rit->second;
  }
  const PPlan& P = dp->plan;
  GPN_HIP_CHECK(hipMemcpyAsync(rt, dp->d_image, (size_t)P.rt_ints * sizeof(int), hipMemcpyDeviceToDevice, s));
  PArgs a;
  a.A = A; a.lda = lda; a.winv = winv; a.info = info;
  a.T = P.T; a.e = (int)e;
  a.tasks = dp->d_tasks; a.succ = dp->d_succ; a.rt = rt;
  a.ntasks = (int)P.tasks.size();
  a.off_dep = P.off_dep;
  for (int q = 0; q < PP_NQ; ++q) a.off_slots[q] = P.off_slots[q];
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 2) cus = 256;
  const int grid = g_pp_grid > 0 ? g_pp_grid : cus;
  a.R = std::min(grid - 1, g_pp_chain_wgs > 0 ? g_pp_chain_wgs : 12);
  a.spin_limit = 1 << 22;
  a.trace = g_pp_trace;
  static std::atomic<int> attr_done{0};
  if (!attr_done.load(std::memory_order_acquire)) {
    GPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ppotrf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L16_LDS_BYTES));
    attr_done.store(1, std::memory_order_release);
  }
  const int rec = profile_on() ? profile_begin(s, (double)n * n * n / 3.0, PROF_GEMM) : -1;
  hipLaunchKernelGGL(ppotrf_kernel, dim3((unsigned)grid), dim3(L16_THREADS), L16_LDS_BYTES, s, a);
  if (rec >= 0) profile_end(s, rec);
  GPN_LAUNCH_CHECK();
  return GPN_OK;
}

void potrf_persistent_release(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_pp_mutex);
  for (auto it = g_pp_runtime.begin(); it != g_pp_runtime.end();) {
    if (!s || std::get<0>(it->first) == s) { (void)hipFree(it->second); it = g_pp_runtime.erase(it); }
    else ++it;
  }
}

}  // namespace gpn

using namespace gpn;

extern "C" int gpn_potrf_lower_persistent(void* stream, double* A, int64_t n, int64_t e, int64_t lda, double* winv, int32_t* info) {
  if (!A) return -2;
  if (n < 0) return -3;
  if (e < 0) return -4;
  if (lda < round_up(n + e, LEAF) || (lda % LEAF) != 0) return -5;
  if (!winv) return -6;
  if (!info) return -7;
  if (reinterpret_cast<uintptr_t>(A) & 15) return GPN_E_ALIGN;
  if (n == 0) return GPN_OK;
  return potrf_persistent(static_cast<hipStream_t>(stream), A, n, e, lda, winv, info);
}

extern "C" int gpn_potrf_persistent_supported(int64_t n, int64_t e) { return pp_supported(n, e) ? 1 : 0; }

// The task graph of gpn_potrf_lower_persistent for an n x n factorisation with e extra rows, for inspection (tests replay it on
// the host): counts5 = {tasks, successor entries, tasks of queue 0, 1, 2}; tasks8 (8 ints per task: type, queue, i, j, k0, k1,
// predecessors, first successor) and succ are filled when given (capacities in entries).  Returns 0, GPN_E_UNSUPPORTED for a
// size the persistent driver does not take, -2 / -3 for a buffer that is too small.
extern "C" int gpn_potrf_persistent_plan(int64_t n, int64_t e, int64_t* counts5, int32_t* tasks8, int64_t cap_tasks, int32_t* succ,
                                         int64_t cap_succ) {
  PPlan P;
  const int rc = pp_build(n, e, P);
  if (rc != GPN_OK) return rc;
  if (counts5) {
    counts5[0] = (int64_t)P.tasks.size(); counts5[1] = (int64_t)P.succ.size();
    for (int q = 0; q < PP_NQ; ++q) counts5[2 + q] = P.qcount[q];
  }
  if (tasks8) {
    if (cap_tasks < (int64_t)P.tasks.size()) return -2;
    for (size_t t = 0; t < P.tasks.size(); ++t) {
      const PTask& k = P.tasks[t];
      int32_t* o = tasks8 + 8 * t;
      o[0] = k.type; o[1] = k.queue; o[2] = k.i; o[3] = k.j; o[4] = k.k0; o[5] = k.k1; o[6] = k.ndeps; o[7] = k.succ_begin;
    }
  }
  if (succ) {
    if (cap_succ < (int64_t)P.succ.size()) return -3;
    std::copy(P.succ.begin(), P.succ.end(), succ);
  }
  return GPN_OK;
}

#ifdef GPN_DEBUG_SWITCHES
extern "C" int gpn_debug_set_persistent(int chain_wgs, int grid) { g_pp_chain_wgs = chain_wgs; g_pp_grid = grid; return GPN_OK; }
// device buffer of 6 x 8 bytes per task (gpn_potrf_persistent_plan's count): per task the 100 MHz stamps {pop begins, task popped,
// acquire done, stores drained, successors released} and (chain role << 32 | workgroup); NULL switches the trace off
extern "C" int gpn_debug_persistent_trace(unsigned long long* buf) { g_pp_trace = buf; return GPN_OK; }
#endif
