// The blocked fp64 Cholesky of potrf.hip as ONE persistent launch: a dataflow over 128 x 128 tiles (round 6).
// Replaces torch.cholesky under functions.cholesky (functions.py:46-47) for the latency regime N = 4096 ... 16384, where the
// launch-per-step driver leaves the chip idle: its N / 128 leaves run one after the other on ONE compute unit (22 % of a C2
// evaluation) and nothing can be put beside them from another stream -- the leaf needs an EMPTY compute unit and priorities only
// order dispatch (LAB.md 8-1c, 10-10; round 6's persistent bulk beside the launch-based chain, profiles/r6_lookahead_*: the
// chain's kernels ran 4-5 x slower next to resident tile workgroups: on one SIMD the older wave wins the matrix pipe).
//
// Here one workgroup per compute unit stays resident for the whole factorisation and the schedule is a task graph:
//   LEAF(0)            the first 128 x 128 diagonal block: Cholesky + inverse (leaf16_body.h, the shipped leaf)
//   STEP(c), c >= 1    the critical chain's step as ONE task of ONE workgroup, no hand-off inside:
//                        X(c,c-1) = A(c,c-1) W_{c-1}^T  (kept in LDS; stored; its readers are released at once: "phase 1")
//                        A(c,c) -= X(c, G) X(c, G)^T    (lower 16 x 16 blocks only; G = the K group that ends at column c-1)
//                        leaf(c)
//   TRSM(i, k)         X(i,k) = A(i,k) W_k^T, in place, one 128-row tile (the arithmetic of colpanel.hip, mode 0)
//   UPD(i, j, k0..k1)  A(i,j) -= X(i, k0..k1) X(j, k0..k1)^T: one 128 x 128 tile of ONE of the shipped driver's update launches
//                      (next-column K = 128, inner-panel trapezoid K = 256, outer-panel K = 1024; gemm_tile.h, the shipped tile)
//   PRE / FIN          the two tiles every step waits for -- (c,c) and (c+1,c) -- get the last update of their column in two
//                      parts when its K group has more than one block: all blocks but the last as soon as THEY are solved
//                      (PRED(c), PRE2(c): raw sums to scratch), the last block when the step's solve is out (inside STEP(c);
//                      FIN2(c)).  The sums continue where they stopped: nothing is rounded in between.
// with exactly the K grouping and per-entry summation order of potrf_lookahead: THE FACTOR IS BIT-IDENTICAL to gpn_potrf_lower's.
// Every task has <= 4 predecessors (TRSMs of a tile row complete in column order, so the last column of a K group stands for
// all of them) and a successor list; a finished task decrements its successors' counters and pushes the ones that reach zero
// onto one of four queues.  Roles are fixed at start by ticket: the first R workgroups serve queue 0 (the two interleaved
// critical chains: STEP, the solve and the FIN of the tile below the next diagonal block, the PREs) and queue 1 (the rest of
// the current outer panel's diagonal triangle) only, and never hold a long tile when a step becomes ready; the others serve
// queue 2 (tiles whose row panel is at most one outer panel below their column panel: what the NEXT panel's chain waits for),
// then queue 3 (the bulk), then 1 and 0.
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

enum { PT_LEAF = 0, PT_TRSM = 1, PT_UPD = 2, PT_STEP = 3, PT_PRED = 4, PT_SUB = 5 };
// UPD: PF_ACC_OUT = raw sums to the column's scratch tile (PRE2); SUB / STEP: PF_ACC_IN = continue from the scratch sums;
// SUB / TRSM: PF_HALF0 / PF_HALF1 = rows 0 .. 63 / 64 .. 127 of the tile only (neither: the whole tile -- the extra-rows tile)
enum { PF_ACC_OUT = 1, PF_ACC_IN = 2, PF_HALF1 = 4, PF_HALF0 = 8 };
struct PTask {                 // 32 bytes
  int16_t type, queue;
  int16_t i, j, k0, k1;        // tile row / column (TRSM, LEAF, STEP: j = the column block); K range in column blocks [k0, k1)
  int32_t ndeps;
  int32_t succ_begin, succ_end;
  int32_t succ_mid;            // successors [succ_begin, succ_mid) are released at the task's phase 1 (STEP: the solve is out)
  int32_t flags;
};
static_assert(sizeof(PTask) == 32, "PTask layout");

// runtime words (ints): control, then per queue a head and a tail on lines of their own
constexpr int PP_MAXQ = 64;    // queue 0: the chains' tasks; 1 + 2 p, 2 + 2 p: tasks whose OUTPUT column lies in outer panel p (near / far rows)
constexpr int PP_BAND = 3;     // solves / short updates of tile rows at most this far below their column go to the chain queue
// runtime words (ints): control, then per queue a head and a tail on lines of their own, then (read-only) per queue its task
// count and the offset of its slots
constexpr int RT_TICKET = 0, RT_COMPLETED = 1, RT_DONE = 2, RT_ABORT = 3, RT_EPOCH = 64, RT_Q0 = 128, RT_QSTRIDE = 64;      // (RT_EPOCH on a line of its own)
constexpr int RT_QINFO = RT_Q0 + PP_MAXQ * RT_QSTRIDE, RT_FIXED = RT_QINFO + 2 * PP_MAXQ;
// scratch per column block: the diagonal tile's 36 lower 16 x 16 blocks (raw sums, [block][r][lane]) + one 128 x 128 tile
constexpr int64_t PP_SCR_DIAG = 36 * 256, PP_SCR_SUB = LEAF * LEAF, PP_SCR_SLOT = PP_SCR_DIAG + PP_SCR_SUB;

struct PArgs {
  double* A;
  int64_t lda;
  double* winv;
  int32_t* info;
  int T, e;                    // matrix tile rows; extra rows (tile row T when e > 0)
  const PTask* tasks;
  const int* succ;
  int* rt;
  double* scratch;             // PP_SCR_SLOT doubles per column block
  int ntasks;
  int off_dep;
  int nq;                      // queues in use
  int R;                       // chain workgroups
  int spin_limit;
  unsigned long long* trace;   // tools' build: 4 stamps (100 MHz clock) + ticket per task, or NULL
};

__device__ __forceinline__ int pp_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pp_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// thread 0: one task of queue q if it has any (-1: empty; -2: aborted)
__device__ __forceinline__ int pp_take(const PArgs& a, const int q, int h, const int t) {
  int* rt = a.rt;
  int* head = rt + RT_Q0 + RT_QSTRIDE * q;
  while (h < t) {
    int expect = h;
    if (__hip_atomic_compare_exchange_strong(head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      const int* slot = rt + rt[RT_QINFO + 2 * q + 1] + h;
      int task, s2 = 0;
      while ((task = pp_ld(slot)) < 0) {               // the producer bumped the tail and is about to fill the slot
        if (++s2 > a.spin_limit) { pp_st(rt + RT_ABORT, 3); return -2; }
        __builtin_amdgcn_s_sleep(1);
      }
      return task;
    }
    h = expect;                                        // somebody else took it
  }
  return -1;
}

// thread 0: the next task for this workgroup, -1 when the factorisation is over (or aborted).
// EARLIEST NEED FIRST.  Queue 0 holds the chains' tasks (steps, the tiles right below the diagonal): chain workgroups look there
// first, the others last.  Every other task sits in the queue of the outer panel its OUTPUT TILE ROW lies in: the chain needs row
// i complete when it reaches column i, and a task on row i reads nothing of the rows below it -- so rows are served top down,
// `qmin` = the first queue that still has tasks to hand out (its head has not reached its task count yet).  With queues by task
// KIND, or by the output COLUMN's panel, long outer-panel tiles of far rows (or the short tasks of far rows) were served before
// what the next steps were waiting for, and every outer panel ended with the chain waiting for milliseconds
// (profiles/r6_persistent_trace_*.txt).
__device__ __forceinline__ int pp_pop(const PArgs& a, const bool chain, int& qmin) {
  int* rt = a.rt;
  const int nq = a.nq;
  for (;;) {
    // the push epoch BEFORE the scan: an idle workgroup then watches this ONE word instead of sweeping every queue's head and
    // tail (170 idle workgroups sweeping 17 queues slowed every running task by 1.3-2 x: MI355X_MICROARCH.md, polling-cost)
    const int epoch = pp_ld(rt + RT_EPOCH);
    if (chain) {
      const int t = pp_take(a, 0, pp_ld(rt + RT_Q0), pp_ld(rt + RT_Q0 + 32));
      if (t != -1) return t < 0 ? -1 : t;
    }
    while (qmin < nq && pp_ld(rt + RT_Q0 + RT_QSTRIDE * qmin) >= rt[RT_QINFO + 2 * qmin]) ++qmin;
    for (int base = qmin; base < nq; base += 8) {
      int hh[8], tt[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) {                     // eight heads and tails in flight at once
        const int q = base + o;
        hh[o] = tt[o] = 0;
        if (q < nq) {
          hh[o] = pp_ld(rt + RT_Q0 + RT_QSTRIDE * q);
          tt[o] = pp_ld(rt + RT_Q0 + RT_QSTRIDE * q + 32);
        }
      }
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        if (hh[o] < tt[o]) {
          const int t = pp_take(a, base + o, hh[o], tt[o]);
          if (t != -1) return t < 0 ? -1 : t;
        }
      }
    }
    if (!chain) {
      const int t = pp_take(a, 0, pp_ld(rt + RT_Q0), pp_ld(rt + RT_Q0 + 32));
      if (t != -1) return t < 0 ? -1 : t;
    }
    // nothing anywhere: sleep until somebody pushes (or the factorisation ends)
    for (int idle = 0;; ++idle) {
      if (pp_ld(rt + RT_EPOCH) != epoch) break;
      if ((idle & 15) == 15 && (pp_ld(rt + RT_DONE) | pp_ld(rt + RT_ABORT))) return -1;
      if (idle > a.spin_limit) { pp_st(rt + RT_ABORT, 2); return -1; }
      if (idle < 8) __builtin_amdgcn_s_sleep(8);
      else __builtin_amdgcn_s_sleep(32);
    }
    if (pp_ld(rt + RT_DONE) | pp_ld(rt + RT_ABORT)) return -1;
  }
}

// one wave, after the workgroup's stores have drained: release the successors succ[sbegin .. send); last: the task is complete
__device__ __forceinline__ void pp_notify(const PArgs& a, const int sbegin, const int send, const bool last, const int lane) {
  int* rt = a.rt;
  bool pushed = false;
  for (int base = sbegin; base < send; base += 64) {
    const int idx = base + lane;
    int s = -1, q = -1;
    bool ready = false;
    if (idx < send) {
      s = a.succ[idx];
      const int old = __hip_atomic_fetch_sub(rt + a.off_dep + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ready = old == 1;
      if (ready) q = a.tasks[s].queue;
    }
    // the ready ones, queue by queue (usually one or two queues): one tail bump per queue, then every lane fills its slot
    unsigned long long todo = __ballot(ready);
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const int qq = __shfl(q, first, 64);
      const bool mine = ready && q == qq;
      const unsigned long long m = __ballot(mine);
      const int cnt = __popcll(m);
      int basei = 0;
      if (lane == first) basei = __hip_atomic_fetch_add(rt + RT_Q0 + RT_QSTRIDE * qq + 32, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      basei = __shfl(basei, first, 64);
      if (mine) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        pp_st(rt + rt[RT_QINFO + 2 * qq + 1] + basei + rank, s);
      }
      todo &= ~m;
      pushed = true;
    }
  }
  if (pushed && lane == 0) __hip_atomic_fetch_add(rt + RT_EPOCH, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (last && lane == 0) {
    const int c = __hip_atomic_fetch_add(rt + RT_COMPLETED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c == a.ntasks - 1) {
      pp_st(rt + RT_DONE, 1);
      __hip_atomic_fetch_add(rt + RT_EPOCH, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // wake the sleepers
    }
  }
}

constexpr int PP_LDS_ROW = LEAF * 8 + 16;
constexpr int PP_TILE_PER = (LEAF * 64 + L16_THREADS - 1) / L16_THREADS;      // 16-byte segments per thread of a 128 x 128 tile: 11
typedef double pd2 __attribute__((ext_vector_type(2)));
typedef double pd4 __attribute__((ext_vector_type(4)));

// a [rows, 128] tile (ld lda) into LDS, row stride PP_LDS_ROW, rows padded with zeros to a multiple of 16: all loads of the
// workgroup in flight at once; the caller puts a barrier behind it
__device__ __forceinline__ void pp_tile_load(pd2 (&v)[PP_TILE_PER], const double* B, const int64_t lda, const int rows, const int tid) {
  const int total = ((rows + 15) >> 4) * 16 * 64;
#pragma unroll
  for (int u = 0; u < PP_TILE_PER; ++u) {
    const int idx = tid + u * L16_THREADS;
    const int row = idx >> 6, seg = idx & 63;
    v[u] = (idx < total && row < rows) ? *reinterpret_cast<const pd2*>(B + (int64_t)row * lda + seg * 2) : pd2{0.0, 0.0};
  }
}
__device__ __forceinline__ void pp_tile_park(const pd2 (&v)[PP_TILE_PER], char* lds, const int rows, const int tid) {
  const int total = ((rows + 15) >> 4) * 16 * 64;
#pragma unroll
  for (int u = 0; u < PP_TILE_PER; ++u) {
    const int idx = tid + u * L16_THREADS;
    if (idx < total) *reinterpret_cast<pd2*>(lds + (idx >> 6) * PP_LDS_ROW + (idx & 63) * 16) = v[u];
  }
}

// X = B W^T in place for one tile of `rows` <= 128 rows (B [rows, 128] at ld lda; W = the leaf's inverse, [128][128], lower
// triangular).  colpanel.hip mode 0's arithmetic entry for entry: per 16 x 16 output tile the 8-k groups j = 0 .. (last
// column of the tile) / 8 in order, the pair of MFMAs (k = 8j + {0,2,4,6}, then + {1,3,5,7}) per group, accumulators from zero.
// The whole tile is parked in LDS (row stride 1 KiB + 16 B) with all its loads in flight at once; the 8 matrix waves own one
// 16-column tile each ({w, 7 - w} on the two waves of a SIMD: equal work in the triangular product); 12 waves load.
// KEEP: afterwards the LDS tile holds X (row-major, same stride) instead of B -- rows are replaced half by half, each half once
// every wave is done reading it (a row of X needs its own row of B only) -- and the workgroup has met at a barrier.
template <bool KEEP>
__device__ __forceinline__ void pp_trsm_tile(double* B, const int64_t lda, const int rows, const double* W, const int tid) {
  extern __shared__ __attribute__((aligned(16))) char pp_lds[];
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int rt16 = (rows + 15) >> 4;                   // 16-row tiles
  pd2 v[PP_TILE_PER];
  pp_tile_load(v, B, lda, rows, tid);
  const bool mm = wave < 8;
  const int ct = wave < 4 ? wave : 11 - wave;          // (waves w and w + 4 share a SIMD)
  pd2 b[16];
  if (mm) {
    const double* wsrc = W + (int64_t)(ct * 16 + lr) * LEAF + 2 * lq;
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = (8 * j <= ct * 16 + 15) ? *reinterpret_cast<const pd2*>(wsrc + 8 * j) : pd2{0.0, 0.0};
  }
  pp_tile_park(v, pp_lds, rows, tid);
  __syncthreads();
  auto solve_row_tile = [&](int i) -> pd4 {
    pd4 acc = pd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (8 * j > ct * 16 + 15) continue;              // W[col][k] = 0 for k > col (wave-uniform)
      const pd2 av = *reinterpret_cast<const pd2*>(pp_lds + (i * 16 + lr) * PP_LDS_ROW + (4 * j + lq) * 16);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b[j].y, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i * 16 + lq + 4 * r;
      if (row < rows) __hip_atomic_store(B + (int64_t)row * lda + ct * 16 + lr, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return acc;
  };
  if constexpr (!KEEP) {
    if (mm) {
#pragma unroll 1
      for (int i = 0; i < rt16; ++i) (void)solve_row_tile(i);
    }
  } else {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      pd4 x[4];
      if (mm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = solve_row_tile(4 * half + q);
      }
      __syncthreads();                                 // every wave is done with this half's rows of B
      if (mm) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<double*>(pp_lds + ((4 * half + q) * 16 + lq + 4 * r) * PP_LDS_ROW + (ct * 16 + lr) * 8) = x[q][r];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // X's write-through stores have left (the caller releases its readers)
    __syncthreads();
  }
}

// The lower 16 x 16 blocks of a diagonal tile, dealt to the 8 matrix waves: block idx = bi (bi + 1) / 2 + bj (bj <= bi) goes
// to wave idx % 8 (9 blocks per SIMD).  One 128-column block of the update  D -= X X^T  from the X tile parked in LDS: per
// block the 8-k groups in order, two MFMAs each -- gemm_tile.h's order for the same entries.
constexpr int PP_DQ = 5;
__device__ __forceinline__ void pp_diag_blocks(const int wave, int (&bi)[PP_DQ], int (&bj)[PP_DQ], int& nb) {
  nb = 0;
#pragma unroll
  for (int q = 0; q < PP_DQ; ++q) {
    const int idx = wave + 8 * q;
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= idx) ++i;
    bi[q] = i; bj[q] = idx - i * (i + 1) / 2;
    if (idx < 36) nb = q + 1;
  }
}
__device__ __forceinline__ void pp_diag_mfma(const char* lds, pd4 (&dacc)[PP_DQ], const int (&bi)[PP_DQ], const int (&bj)[PP_DQ], const int nb,
                                             const int lr, const int lq) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
#pragma unroll
    for (int q = 0; q < PP_DQ; ++q) {
      if (q >= nb) continue;                           // (wave-uniform)
      const pd2 av = *reinterpret_cast<const pd2*>(lds + (bi[q] * 16 + lr) * PP_LDS_ROW + (4 * j + lq) * 16);
      const pd2 bv = *reinterpret_cast<const pd2*>(lds + (bj[q] * 16 + lr) * PP_LDS_ROW + (4 * j + lq) * 16);
      dacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, dacc[q], 0, 0, 0);
      dacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.y, dacc[q], 0, 0, 0);
    }
  }
}
// the blocks [k0, k1) of row panel c's solved tiles, one after the other through the LDS tile
__device__ __forceinline__ void pp_diag_accumulate(const PArgs& a, const int c, const int k0, const int k1, pd4 (&dacc)[PP_DQ],
                                                   const int (&bi)[PP_DQ], const int (&bj)[PP_DQ], const int nb, const int tid) {
  extern __shared__ __attribute__((aligned(16))) char pp_lds[];
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const bool mm = __builtin_amdgcn_readfirstlane(tid >> 6) < 8;
#pragma unroll 1
  for (int kb = k0; kb < k1; ++kb) {
    pd2 v[PP_TILE_PER];
    pp_tile_load(v, a.A + (int64_t)c * LEAF * a.lda + (int64_t)kb * LEAF, a.lda, LEAF, tid);
    pp_tile_park(v, pp_lds, LEAF, tid);
    __syncthreads();
    if (mm) pp_diag_mfma(pp_lds, dacc, bi, bj, nb, lr, lq);
    __syncthreads();                                   // before the next block (or the solve) overwrites the tile
  }
}

// PRED(c): the diagonal tile's raw sums over the blocks [k0, k1) -> scratch
__device__ __forceinline__ void pp_pred(const PArgs& a, const PTask& tk, const int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bi[PP_DQ], bj[PP_DQ], nb;
  pp_diag_blocks(wave & 7, bi, bj, nb);
  pd4 dacc[PP_DQ];
#pragma unroll
  for (int q = 0; q < PP_DQ; ++q) dacc[q] = pd4{0.0, 0.0, 0.0, 0.0};
  pp_diag_accumulate(a, tk.i, tk.k0, tk.k1, dacc, bi, bj, nb, tid);
  if (wave < 8) {
    double* scr = a.scratch + (int64_t)tk.i * PP_SCR_SLOT;
#pragma unroll
    for (int q = 0; q < PP_DQ; ++q)
      if (q < nb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_store(scr + ((int64_t)(wave + 8 * q) * 4 + r) * 64 + lane, dacc[q][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// SUB: the update of 64 rows of tile (i, j) by the K group [k0, k1) of one or two 128-column blocks -- every update that is not
// an outer panel's: next-column (K = 128), trapezoid (K = 256), and the last block of the column's last update for the tile
// below the next diagonal block (PF_ACC_IN: the earlier blocks' sums come from scratch).  Half tiles on two compute units:
// these updates and the solves between them form a chain PER TILE ROW through the columns of an outer panel (solve (i,k) ->
// update (i,k+1) -> solve (i,k+1) -> ...), which as whole-tile tasks through the generic contraction took ~65 us per column
// against the diagonal chain's ~55 -- every outer panel ended with the steps waiting for the rows (profiles/r6_persistent_*).
// K is the whole problem (colpanel.hip's scheme): the left operand's 64 rows are parked in LDS with all loads in flight at once,
// a wave keeps its 16 columns of the right operand in registers; sums, K order and epilogue are gemm_tile.h's for the same
// entries (accumulators from zero or from the scratch sums, 8-k groups in order, C = fma(1, C, -sums)).
__device__ __forceinline__ void pp_sub(const PArgs& a, const PTask& tk, const int tid) {
  extern __shared__ __attribute__((aligned(16))) char pp_lds[];
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = (tk.flags & PF_HALF1) ? 1 : 0;
  const int rows = tk.i < a.T ? 64 : a.e;                  // (the extra-rows tile: one task, its e rows)
  const bool diag = tk.i == tk.j;
  const double* Ai = a.A + ((int64_t)tk.i * LEAF + 64 * h) * a.lda;                          // row panel of the output rows
  const double* Bj = a.A + (int64_t)tk.j * LEAF * a.lda;                                     // row panel of the output columns
  double* C = a.A + ((int64_t)tk.i * LEAF + 64 * h) * a.lda + (int64_t)tk.j * LEAF;
  constexpr int PER = (64 * 64 + L16_THREADS - 1) / L16_THREADS;                              // 6
  const bool mm = wave < 8;
  const int ct = wave & 7;
  pd4 acc[4];
  double cold[4][4];
  if (mm) {
    if (tk.flags & PF_ACC_IN) {
      // gemm_tile.h's dump of the 128 x 128 tile (8 waves of 32 x 64): row tile I, column tile ct -> wave (I / 2) * 2 + ct / 4
      const double* scr = a.scratch + (int64_t)tk.j * PP_SCR_SLOT + PP_SCR_DIAG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int I = 4 * h + i;
        const int64_t base = ((((int64_t)((I >> 1) * 2 + (ct >> 2)) * 2 + (I & 1)) * 4 + (ct & 3)) * 4) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = scr[base + r * 64];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = pd4{0.0, 0.0, 0.0, 0.0};
    }
  }
  // one 128-column block; last_c: the group's last block -- the old tile is requested behind its operands (the staging registers
  // are free again by then) and needed only after its MFMAs
  auto block = [&](const int kb, auto last_c) {
    constexpr bool last = decltype(last_c)::value;
    pd2 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * L16_THREADS;
      const int row = idx >> 6;
      v[u] = (idx < 64 * 64 && row < rows) ? *reinterpret_cast<const pd2*>(Ai + (int64_t)row * a.lda + (int64_t)kb * LEAF + (idx & 63) * 2) : pd2{0.0, 0.0};
    }
    pd2 b[16];
    if (mm) {
      const double* bsrc = Bj + (int64_t)(ct * 16 + lr) * a.lda + (int64_t)kb * LEAF + 2 * lq;
#pragma unroll
      for (int j = 0; j < 16; ++j) b[j] = *reinterpret_cast<const pd2*>(bsrc + 8 * j);
    }
    if (kb > tk.k0) __syncthreads();                       // every wave is done with the previous block's rows
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * L16_THREADS;
      if (idx < 64 * 64) *reinterpret_cast<pd2*>(pp_lds + (idx >> 6) * PP_LDS_ROW + (idx & 63) * 16) = v[u];
    }
    __syncthreads();
    if (mm) {
      if constexpr (last) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + lq + 4 * r;
            cold[i][r] = row < rows ? C[(int64_t)row * a.lda + ct * 16 + lr] : 0.0;
          }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const pd2 av = *reinterpret_cast<const pd2*>(pp_lds + (i * 16 + lr) * PP_LDS_ROW + (4 * j + lq) * 16);
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b[j].x, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b[j].y, acc[i], 0, 0, 0);
        }
      }
    }
  };
#pragma unroll 1
  for (int kb = tk.k0; kb < tk.k1 - 1; ++kb) block(kb, std::false_type{});
  block(tk.k1 - 1, std::true_type{});
  if (mm) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i * 16 + lq + 4 * r, col = ct * 16 + lr;
        if (row < rows && (!diag || col <= 64 * h + row))
          __hip_atomic_store(C + (int64_t)row * a.lda + col, fma(1.0, cold[i][r], -1.0 * acc[i][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
}

// STEP(c) up to the leaf: solve of tile (c, c-1), release of its readers, the diagonal tile's last update
__device__ __forceinline__ void pp_step(const PArgs& a, const PTask& tk, const int task_index, const int tid) {
  extern __shared__ __attribute__((aligned(16))) char pp_lds[];
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = tk.i;
  int bi[PP_DQ], bj[PP_DQ], nb;
  pp_diag_blocks(wave & 7, bi, bj, nb);
  double* Ccc = a.A + (int64_t)c * LEAF * (a.lda + 1);
  pp_trsm_tile<true>(a.A + (int64_t)c * LEAF * a.lda + (int64_t)(c - 1) * LEAF, a.lda, LEAF, a.winv + (int64_t)(c - 1) * LEAF * LEAF, tid);
  // X(c, c-1) is out (stores drained, barrier passed): an idle wave releases its readers while the matrix waves go on
  if (wave == 8) {
    pp_notify(a, tk.succ_begin, tk.succ_mid, false, lane);
    if (a.trace && lane == 0) a.trace[8 * (size_t)task_index + 6] = __builtin_amdgcn_s_memrealtime();      // phase 1 released
  }
  // the sums so far: PRED(c)'s over the earlier blocks of the group (the builder makes one whenever the group has any), or zero
  pd4 dacc[PP_DQ];
  if ((tk.flags & PF_ACC_IN) && wave < 8) {
    const double* scr = a.scratch + (int64_t)c * PP_SCR_SLOT;
#pragma unroll
    for (int q = 0; q < PP_DQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) dacc[q][r] = q < nb ? scr[((int64_t)(wave + 8 * q) * 4 + r) * 64 + lane] : 0.0;
  } else {
#pragma unroll
    for (int q = 0; q < PP_DQ; ++q) dacc[q] = pd4{0.0, 0.0, 0.0, 0.0};
  }
  // the old diagonal tile, requested now and needed after the MFMAs; `sc1` loads: they bypass this compute unit's L1, which must
  // not keep lines of a tile the leaf below re-reads after this workgroup has rewritten it
  double cold[PP_DQ][4];
  if (wave < 8) {
#pragma unroll
    for (int q = 0; q < PP_DQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bi[q] * 16 + lq + 4 * r, col = bj[q] * 16 + lr;
        cold[q][r] = (q < nb && col <= row) ? __hip_atomic_load(Ccc + (int64_t)row * a.lda + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
  }
  if (wave < 8) {
    pp_diag_mfma(pp_lds, dacc, bi, bj, nb, lr, lq);
#pragma unroll
    for (int q = 0; q < PP_DQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bi[q] * 16 + lq + 4 * r, col = bj[q] * 16 + lr;
        if (q < nb && col <= row)
          __hip_atomic_store(Ccc + (int64_t)row * a.lda + col, fma(1.0, cold[q][r], -1.0 * dacc[q][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the diagonal tile is out before the leaf reads it back
  __syncthreads();
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
  int qmin = 1;
  for (;;) {
    if (tid == 0) {
      if (a.trace) t_pop0 = __builtin_amdgcn_s_memrealtime();
      const int t = pp_pop(a, chain, qmin);
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
    const int half = (tk.flags & PF_HALF1) ? 1 : 0;
    const int rows_i = tk.i < a.T ? ((tk.flags & (PF_HALF0 | PF_HALF1)) ? 64 : LEAF) : a.e;
    if (tk.type == PT_LEAF || tk.type == PT_STEP) {
      if (tk.type == PT_STEP) pp_step(a, tk, task, tid_it);
      Leaf16Args la;
      la.A = a.A + (int64_t)tk.i * LEAF * (a.lda + 1);
      la.lda = a.lda;
      la.kb = LEAF;
      la.col0 = tk.i * LEAF;
      la.winv = a.winv + (int64_t)tk.i * LEAF * LEAF;
      la.info = a.info;
      la.sA = la.sW = la.sInfo = 0;
      leaf16_body<true>(la, 0, tid_it);
    } else if (tk.type == PT_TRSM) {
      pp_trsm_tile<false>(a.A + ((int64_t)tk.i * LEAF + 64 * half) * a.lda + (int64_t)tk.j * LEAF, a.lda, rows_i,
                          a.winv + (int64_t)tk.j * LEAF * LEAF, tid_it);
    } else if (tk.type == PT_PRED) {
      pp_pred(a, tk, tid_it);
    } else if (tk.type == PT_SUB) {
      pp_sub(a, tk, tid_it);
    } else {
      GemmArgs g;
      g.A = a.A + (int64_t)tk.i * LEAF * a.lda + (int64_t)tk.k0 * LEAF;
      g.B = a.A + (int64_t)tk.j * LEAF * a.lda + (int64_t)tk.k0 * LEAF;
      g.C = a.A + (int64_t)tk.i * LEAF * a.lda + (int64_t)tk.j * LEAF;
      g.lda = g.ldb = g.ldc = a.lda;
      g.M = rows_i; g.N = LEAF; g.K = (tk.k1 - tk.k0) * LEAF;
      g.mt = g.nt = 1;
      g.lower = tk.i == tk.j ? 1 : 0;
      g.group_h = 8; g.thin = 1;
      g.st_blk = g.st_step = g.st_diag = 0;
      g.tri = 0;
      g.alpha = -1.0; g.beta = 1.0;
      g.batch = 1; g.sA = g.sB = g.sC = 0;
      g.inner = 0; g.sA2 = g.sB2 = g.sC2 = 0;
      double* sub = a.scratch + (int64_t)tk.j * PP_SCR_SLOT + PP_SCR_DIAG;
      g.acc_in = (tk.flags & PF_ACC_IN) ? sub : nullptr;
      g.acc_out = (tk.flags & PF_ACC_OUT) ? sub : nullptr;
      gemm_nt_tile<128, 128, 32, 64, true, 2, false, true, true, true>(g, 0, 1, true, tid_it);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave's write-through stores have left
    __syncthreads();
    unsigned long long t_end = 0;
    if (a.trace && tid == 0) t_end = __builtin_amdgcn_s_memrealtime();
    if (wave == 0) pp_notify(a, tk.succ_mid, tk.succ_end, true, lane);
    if (a.trace && tid == 0) {
      unsigned long long* o = a.trace + 8 * (size_t)task;
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
  int nq = 0;
  std::vector<int> initial[PP_MAXQ];
  int qcount[PP_MAXQ] = {};
  int off_dep = 0, off_slots[PP_MAXQ] = {}, rt_ints = 0;
};

static bool pp_supported(int64_t n, int64_t e) {
  if (n % LEAF != 0 || n < 2560 || n >= 20480 || e < 0 || e > 16) return false;
  int64_t w[3];
  return gpn_potrf_panel_levels(n, w) == 2 && w[0] == 256 && w[1] % w[0] == 0;
}

// Walks gpn_potrf_lower's schedule for this size (potrf.hip potrf_lookahead: inner panels of w0 columns -- two blocks --, outer
// panels of w1) and emits one task per 128 x 128 tile of every launch -- with the chain's pieces fused / split as described at
// the top of this file; predecessors from the last writer of each tile.  A predecessor is (task, phase): phase 1 of a STEP = its
// solved tile (c, c-1), phase 2 = everything.  Tasks are created after their predecessors: the list is a sequential order.
static int pp_build(int64_t n, int64_t e, PPlan& P) {
  if (!pp_supported(n, e)) return GPN_E_UNSUPPORTED;
  int64_t w[3];
  gpn_potrf_panel_levels(n, w);
  const int PW = (int)(w[0] / LEAF), OW = (int)(w[1] / LEAF);
  const int T = (int)(n / LEAF), TR = T + (e > 0 ? 1 : 0);
  P.n = n; P.e = e; P.T = T; P.TR = TR;
  struct Dep { int id, phase; };
  typedef std::vector<Dep> Deps;
  std::vector<Deps> lastw((size_t)TR * T);               // last writer(s) of tile (i, j)
  std::vector<Deps> trsm((size_t)TR * T);                // who solved tile (i, k)
  std::vector<Deps> leafof((size_t)T);
  std::vector<Deps> preds;
  auto add = [&](int type, int i, int j, int k0, int k1, int flags, int queue, std::initializer_list<Deps> deps) {
    PTask t{};
    t.type = (int16_t)type; t.i = (int16_t)i; t.j = (int16_t)j; t.k0 = (int16_t)k0; t.k1 = (int16_t)k1;
    t.flags = flags;
    Deps d;
    for (const Deps& xs : deps)
      for (const Dep& x : xs) {
        bool seen = false;
        for (Dep& y : d) if (y.id == x.id) { y.phase = std::max(y.phase, x.phase); seen = true; }
        if (!seen) d.push_back(x);
      }
    t.ndeps = (int)d.size();
    t.queue = (int16_t)queue;
    P.tasks.push_back(t);
    preds.push_back(d);
    return (int)P.tasks.size() - 1;
  };
  // queue of an ordinary task: by the outer panel its output tile ROW lies in (pp_pop: earliest need first) -- the chain needs
  // tile row i complete when it reaches column i, and what a task on row i reads from other rows are rows ABOVE i --; the
  // extra-rows tile last.  The band of PP_BAND tile rows below the diagonal feeds the chains within a few steps: the chain queue.
  const int NP = (T + OW - 1) / OW;
  P.nq = 1 + NP + 1;
  if (P.nq > PP_MAXQ) return GPN_E_UNSUPPORTED;
  auto queue_of = [&](int i, int j, int kl, bool top) {
    (void)kl;
    if (!top && i - j <= PP_BAND && i < T) return 0;
    return 1 + (i < T ? i / OW : NP);
  };
  // start of the K group whose update is the LAST one of column c's tiles (it ends at column c - 1)
  auto group_start = [&](int c) { return c % PW != 0 ? c - 1 : (c % OW != 0 ? c - PW : c - OW); };
  auto at = [&](int i, int j) { return (size_t)i * T + j; };
  // update of tile (i, j) by the K group [k0, k1); top: an outer panel's (K = outer width)
  auto upd = [&](int i, int j, int k0, int k1, bool top) {
    const int kl = k1 - 1;
    const bool last_of_column = k1 == j;                 // the group that ends right left of the tile's column
    if (last_of_column && i == j) return;                // fused into STEP(j)
    if (last_of_column && i == j + 1 && i < T) {         // the tile below the next diagonal block: on the second critical chain
      Deps pre;
      if (k1 - k0 >= 2) pre = Deps{Dep{add(PT_UPD, i, j, k0, kl, PF_ACC_OUT, 0, {trsm[at(i, kl - 1)], trsm[at(j, kl - 1)]}), 2}};
      const int fl = pre.empty() ? 0 : PF_ACC_IN;
      const int h0 = add(PT_SUB, i, j, kl, k1, fl | PF_HALF0, 0, {trsm[at(i, kl)], trsm[at(j, kl)], lastw[at(i, j)], pre});
      const int h1 = add(PT_SUB, i, j, kl, k1, fl | PF_HALF1, 0, {trsm[at(i, kl)], trsm[at(j, kl)], lastw[at(i, j)], pre});
      lastw[at(i, j)] = Deps{Dep{h0, 2}, Dep{h1, 2}};
      return;
    }
    if (top) {                                           // an outer panel's update: the generic 128 x 128 tile task
      const int id = add(PT_UPD, i, j, k0, k1, 0, queue_of(i, j, kl, top), {trsm[at(i, kl)], trsm[at(j, kl)], lastw[at(i, j)]});
      lastw[at(i, j)] = Deps{Dep{id, 2}};
      return;
    }
    // next-column update / trapezoid: two half-tile tasks (one for the extra-rows tile)
    const int q = queue_of(i, j, kl, false);
    if (i < T) {
      const int h0 = add(PT_SUB, i, j, k0, k1, PF_HALF0, q, {trsm[at(i, kl)], trsm[at(j, kl)], lastw[at(i, j)]});
      const int h1 = add(PT_SUB, i, j, k0, k1, PF_HALF1, q, {trsm[at(i, kl)], trsm[at(j, kl)], lastw[at(i, j)]});
      lastw[at(i, j)] = Deps{Dep{h0, 2}, Dep{h1, 2}};
    } else {
      const int id = add(PT_SUB, i, j, k0, k1, 0, q, {trsm[at(i, kl)], trsm[at(j, kl)], lastw[at(i, j)]});
      lastw[at(i, j)] = Deps{Dep{id, 2}};
    }
  };
  for (int p0 = 0; p0 < T; p0 += PW) {
    const int pend = std::min(p0 + PW, T);
    for (int k = p0; k < pend; ++k) {
      if (k == 0) {
        const int leaf = add(PT_LEAF, 0, 0, 0, 1, 0, 0, {});
        leafof[0] = lastw[at(0, 0)] = Deps{Dep{leaf, 2}};
      }
      // (for k >= 1 the leaf of column k is the end of STEP(k), created with column k - 1's solves below)
      for (int i = k + 1; i < TR; ++i) {
        if (i == k + 1 && i < T) {
          // STEP(c), c = k + 1: solve of (c, k) + the diagonal tile's last update (group [g0, c)) + leaf(c)
          const int c = i, g0 = group_start(c);
          Deps pre;
          if (c - g0 >= 2) pre = Deps{Dep{add(PT_PRED, c, c, g0, c - 1, 0, 0, {trsm[at(c, c - 2)]}), 2}};
          const int id = add(PT_STEP, c, c, g0, c, pre.empty() ? 0 : PF_ACC_IN, 0, {leafof[k], lastw[at(c, k)], lastw[at(c, c)], pre});
          trsm[at(c, k)] = lastw[at(c, k)] = Deps{Dep{id, 1}};
          leafof[c] = lastw[at(c, c)] = Deps{Dep{id, 2}};
        } else {
          const int q = queue_of(i, k, k, false);
          if (i < T) {                                   // two half-tile solves (rows are independent)
            const int h0 = add(PT_TRSM, i, k, k, k + 1, PF_HALF0, q, {leafof[k], lastw[at(i, k)]});
            const int h1 = add(PT_TRSM, i, k, k, k + 1, PF_HALF1, q, {leafof[k], lastw[at(i, k)]});
            trsm[at(i, k)] = lastw[at(i, k)] = Deps{Dep{h0, 2}, Dep{h1, 2}};
          } else {
            const int id = add(PT_TRSM, i, k, k, k + 1, 0, q, {leafof[k], lastw[at(i, k)]});
            trsm[at(i, k)] = lastw[at(i, k)] = Deps{Dep{id, 2}};
          }
        }
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
  // successor lists: phase-1 successors first
  std::vector<int> c1(nt, 0), c2(nt, 0);
  for (int t = 0; t < nt; ++t) for (const Dep& d : preds[t]) ++(d.phase == 1 ? c1 : c2)[d.id];
  int off = 0;
  std::vector<int> f1(nt), f2(nt);
  for (int t = 0; t < nt; ++t) {
    P.tasks[t].succ_begin = off; f1[t] = off; off += c1[t];
    P.tasks[t].succ_mid = off; f2[t] = off; off += c2[t];
    P.tasks[t].succ_end = off;
  }
  P.succ.assign((size_t)off, -1);
  for (int t = 0; t < nt; ++t) for (const Dep& d : preds[t]) P.succ[(size_t)(d.phase == 1 ? f1 : f2)[d.id]++] = t;
  P.dep0.resize(nt);
  for (int t = 0; t < nt; ++t) {
    P.dep0[t] = P.tasks[t].ndeps;
    ++P.qcount[P.tasks[t].queue];
    if (P.tasks[t].ndeps == 0) P.initial[P.tasks[t].queue].push_back(t);
  }
  P.off_dep = RT_FIXED;
  int o = P.off_dep + ((nt + 31) & ~31);
  for (int q = 0; q < P.nq; ++q) { P.off_slots[q] = o; o += (P.qcount[q] + 31) & ~31; }
  P.rt_ints = o;
  return GPN_OK;
}

static void pp_image(const PPlan& P, std::vector<int>& img) {
  img.assign((size_t)P.rt_ints, 0);
  for (size_t t = 0; t < P.dep0.size(); ++t) img[(size_t)P.off_dep + t] = P.dep0[t];
  for (int q = 0; q < P.nq; ++q) {
    for (int s = 0; s < ((P.qcount[q] + 31) & ~31); ++s) img[(size_t)P.off_slots[q] + s] = -1;
    for (size_t s = 0; s < P.initial[q].size(); ++s) img[(size_t)P.off_slots[q] + s] = P.initial[q][s];
    img[RT_Q0 + RT_QSTRIDE * q + 32] = (int)P.initial[q].size();     // tail
    img[RT_QINFO + 2 * q] = P.qcount[q];
    img[RT_QINFO + 2 * q + 1] = P.off_slots[q];
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
struct PRuntime { int* rt = nullptr; double* scratch = nullptr; };
static std::map<std::tuple<hipStream_t, int64_t, int64_t>, PRuntime> g_pp_runtime;      // one runtime area (+ scratch tiles) per caller stream and shape

GPN_SWITCH int g_pp_chain_wgs = 0;                       // chain workgroups / grid (0: the defaults); the per-task trace buffer
GPN_SWITCH int g_pp_grid = 0;
GPN_SWITCH unsigned long long* g_pp_trace = nullptr;

// gpn_potrf_lower as one persistent launch.  GPN_E_UNSUPPORTED: this size (or a stream under capture that has no plan yet)
// stays with the launch-based driver.  The plan (task table + successor lists + the pristine runtime image) is built on the
// host and uploaded ONCE per (device, n, e); each call copies the image over the stream's runtime area and launches.
int potrf_persistent(hipStream_t s, double* A, int64_t n, int64_t e, int64_t lda, double* winv, int32_t* info) {
  if (!pp_supported(n, e)) return GPN_E_UNSUPPORTED;
  int dev = 0;
  GPN_HIP_CHECK(hipGetDevice(&dev));
  PDevPlan* dp = nullptr;
  PRuntime rt;
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
      PRuntime r;
      GPN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&r.rt), (size_t)dp->plan.rt_ints * sizeof(int)));
      GPN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&r.scratch), (size_t)dp->plan.T * PP_SCR_SLOT * sizeof(double)));
      rit = g_pp_runtime.emplace(std::make_tuple(s, n, e), r).first;
    }
    rt = rit->second;
  }
  const PPlan& P = dp->plan;
  GPN_HIP_CHECK(hipMemcpyAsync(rt.rt, dp->d_image, (size_t)P.rt_ints * sizeof(int), hipMemcpyDeviceToDevice, s));
  PArgs a;
  a.A = A; a.lda = lda; a.winv = winv; a.info = info;
  a.T = P.T; a.e = (int)e;
  a.tasks = dp->d_tasks; a.succ = dp->d_succ; a.rt = rt.rt; a.scratch = rt.scratch;
  a.ntasks = (int)P.tasks.size();
  a.off_dep = P.off_dep;
  a.nq = P.nq;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 2) cus = 256;
  const int grid = g_pp_grid > 0 ? g_pp_grid : cus;
  a.R = std::min(grid - 1, g_pp_chain_wgs > 0 ? g_pp_chain_wgs : 16);
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
    if (!s || std::get<0>(it->first) == s) { (void)hipFree(it->second.rt); (void)hipFree(it->second.scratch); it = g_pp_runtime.erase(it); }
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
// the host): counts (3 + 64 entries) = {tasks, successor entries, queues in use, tasks of queue 0, 1, ...}; tasks12 (12 ints per
// task: type, queue, i, j, k0, k1, predecessors, flags, succ_begin, succ_mid, succ_end, 0) and succ are filled when given
// (capacities in entries).  Returns 0, GPN_E_UNSUPPORTED for a size the persistent driver does not take, -2 / -3 for a buffer
// that is too small.
extern "C" int gpn_potrf_persistent_plan(int64_t n, int64_t e, int64_t* counts, int32_t* tasks12, int64_t cap_tasks, int32_t* succ,
                                         int64_t cap_succ) {
  PPlan P;
  const int rc = pp_build(n, e, P);
  if (rc != GPN_OK) return rc;
  if (counts) {
    counts[0] = (int64_t)P.tasks.size(); counts[1] = (int64_t)P.succ.size(); counts[2] = P.nq;
    for (int q = 0; q < PP_MAXQ; ++q) counts[3 + q] = q < P.nq ? P.qcount[q] : 0;
  }
  if (tasks12) {
    if (cap_tasks < (int64_t)P.tasks.size()) return -2;
    for (size_t t = 0; t < P.tasks.size(); ++t) {
      const PTask& k = P.tasks[t];
      int32_t* o = tasks12 + 12 * t;
      o[0] = k.type; o[1] = k.queue; o[2] = k.i; o[3] = k.j; o[4] = k.k0; o[5] = k.k1; o[6] = k.ndeps; o[7] = k.flags;
      o[8] = k.succ_begin; o[9] = k.succ_mid; o[10] = k.succ_end; o[11] = 0;
    }
  }
  if (succ) {
    if (cap_succ < (int64_t)P.succ.size()) return -3;
    std::copy(P.succ.begin(), P.succ.end(), succ);
  }
  return GPN_OK;
}

GPN_DEBUG_ONLY(
extern "C" int gpn_debug_set_persistent(int chain_wgs, int grid) { g_pp_chain_wgs = chain_wgs; g_pp_grid = grid; return GPN_OK; }
// device buffer of 8 x 8 bytes per task (gpn_potrf_persistent_plan's count): per task the 100 MHz stamps {pop begins, task popped,
// acquire done, stores drained, successors released} and (chain role << 32 | workgroup); NULL switches the trace off
extern "C" int gpn_debug_persistent_trace(unsigned long long* buf) { g_pp_trace = buf; return GPN_OK; })
