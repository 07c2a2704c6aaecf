// libgpnative_rccl.so: the gpn_dist_comm callback table (include/gpnative.h) over RCCL communicators.
// Kept out of libgpnative.so so that the kernel library does not depend on a communication runtime.
// RCCL collectives are enqueued on the HIP stream they are given (ncclBroadcast / ncclAllReduce),
// which is exactly the contract of the callbacks.  Two routes for a panel: ncclBroadcast (RCCL picks a ring / tree)
// or, with GPN_DIST_MESH_EXCHANGE in the table's flags, grouped ncclSend / ncclRecv over the direct xGMI links.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <new>
#include "../../include/gpnative.h"

#include <cstdlib>
#include <vector>

// ---- mesh broadcast: the schedule of gptorch_amd/dist.py `mesh_plan`, restated ---------------------------------
// One broadcast of `count` doubles from member `root` of a p-member communicator as grouped point-to-point
// transfers over the direct xGMI links ("do not route these through a ring"):
//   * q = p - 1 peers (members != root, ascending).  q == 1 or count < direct_below: the root sends the whole buffer
//     to every peer (one hop, q links in parallel);
//   * otherwise scatter + all-gather, pipelined: S = max(1, min(stages, count / q)) slices of q pieces; in stage t
//     (0..S) the root sends piece (t, i) to peer i while peer i forwards piece (t-1, i) to the q - 1 other peers.
//     Piece idx = t*q + i covers [idx*base + min(idx, extra), +base + (idx < extra)), base = count / (S q).
// Each stage is ONE ncclGroupStart/End; stage t+1 forwards what stage t received -- ordered by the stream.
// ops: quintuples (stage, kind 0 = send / 1 = recv, peer, offset, length), empty pieces dropped.  Returns the number
// of quintuples (also when it exceeds cap: call again with a larger buffer), < 0 on bad arguments.
extern "C" int64_t gpn_mesh_plan(int p, int root, int me, int64_t count, int stages, int64_t direct_below, int64_t* ops, int64_t cap) {
  if (p < 1 || root < 0 || root >= p || me < 0 || me >= p || count < 0) return -1;
  const int q = p - 1;
  int64_t n = 0;
  auto emit = [&](int64_t stage, int64_t kind, int64_t peer, int64_t off, int64_t len) {
    if (len <= 0) return;
    if (n < cap) { int64_t* o = ops + 5 * n; o[0] = stage; o[1] = kind; o[2] = peer; o[3] = off; o[4] = len; }
    ++n;
  };
  if (q == 0 || count == 0) return 0;
  auto peer_rank = [&](int i) { return i < root ? i : i + 1; };          // i-th member that is not the root
  if (q == 1 || count < direct_below) {
    if (me == root) for (int i = 0; i < q; ++i) emit(0, 0, peer_rank(i), 0, count);
    else emit(0, 1, root, 0, count);
    return n;
  }
  int64_t S = count / q < stages ? count / q : stages;
  if (S < 1) S = 1;
  const int64_t np = S * q, base = count / np, extra = count % np;
  auto off_of = [&](int64_t t, int64_t i) { int64_t idx = t * q + i; return idx * base + (idx < extra ? idx : extra); };
  auto len_of = [&](int64_t t, int64_t i) { int64_t idx = t * q + i; return base + (idx < extra ? 1 : 0); };
  for (int64_t t = 0; t <= S; ++t) {
    if (me == root) {
      if (t < S) for (int i = 0; i < q; ++i) emit(t, 0, peer_rank(i), off_of(t, i), len_of(t, i));
    } else {
      const int i = me < root ? me : me - 1;
      if (t < S) emit(t, 1, root, off_of(t, i), len_of(t, i));
      if (t >= 1) {
        for (int j = 0; j < q; ++j) if (j != i) emit(t, 0, peer_rank(j), off_of(t - 1, i), len_of(t - 1, i));
        for (int j = 0; j < q; ++j) if (j != i) emit(t, 1, peer_rank(j), off_of(t - 1, j), len_of(t - 1, j));
      }
    }
  }
  return n;
}

namespace {
struct Ctx { ncclComm_t row, col, world; const gpn_dist_comm* table; int stages; int64_t direct_below; };

int mesh_bcast(Ctx* x, ncclComm_t comm, double* buf, int64_t count, int root, hipStream_t stream) {
  int p = 0, me = 0;
  if (ncclCommCount(comm, &p) != ncclSuccess || ncclCommUserRank(comm, &me) != ncclSuccess) return -100;
  std::vector<int64_t> ops(5 * 64);
  int64_t n = gpn_mesh_plan(p, root, me, count, x->stages, x->direct_below, ops.data(), (int64_t)ops.size() / 5);
  if (n < 0) return -1;
  if (n > (int64_t)ops.size() / 5) {
    ops.resize(5 * n);
    n = gpn_mesh_plan(p, root, me, count, x->stages, x->direct_below, ops.data(), n);
  }
  int64_t i = 0;
  while (i < n) {                                   // one grouped call per stage
    const int64_t stage = ops[5 * i];
    if (ncclGroupStart() != ncclSuccess) return -100;
    bool ok = true;
    for (; i < n && ops[5 * i] == stage; ++i) {
      const int64_t* o = &ops[5 * i];
      ncclResult_t r = o[1] == 0 ? ncclSend(buf + o[3], (size_t)o[4], ncclDouble, (int)o[2], comm, stream)
                                 : ncclRecv(buf + o[3], (size_t)o[4], ncclDouble, (int)o[2], comm, stream);
      ok = ok && r == ncclSuccess;
    }
    if (ncclGroupEnd() != ncclSuccess || !ok) return -100;
  }
  return 0;
}

int bcast(void* c, int which, double* buf, int64_t count, int root, void* stream) {
  Ctx* x = static_cast<Ctx*>(c);
  ncclComm_t comm = which == 0 ? x->row : x->col;
  if (!comm) return count == 0 ? 0 : -1;
  if (x->table && (x->table->flags & GPN_DIST_MESH_EXCHANGE)) return mesh_bcast(x, comm, buf, count, root, static_cast<hipStream_t>(stream));
  return ncclBroadcast(buf, buf, (size_t)count, ncclDouble, root, comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : -100;
}
int allreduce(void* c, double* buf, int64_t count, void* stream) {
  Ctx* x = static_cast<Ctx*>(c);
  if (!x->world) return -1;
  return ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, x->world, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : -100;
}
}  // namespace

extern "C" gpn_dist_comm* gpn_rccl_comm_create(void* row, void* col, void* world) {
  gpn_dist_comm* t = new (std::nothrow) gpn_dist_comm;
  Ctx* x = new (std::nothrow) Ctx{static_cast<ncclComm_t>(row), static_cast<ncclComm_t>(col), static_cast<ncclComm_t>(world), t, 4, (4 << 20) / 8};
  if (!t || !x) { delete t; delete x; return nullptr; }
  if (const char* e = std::getenv("GPN_DIST_MESH_STAGES")) { int v = std::atoi(e); if (v >= 1) x->stages = v; }
  if (const char* e = std::getenv("GPN_DIST_MESH_DIRECT_BYTES")) { long long v = std::atoll(e); if (v >= 0) x->direct_below = v / 8; }
  t->ctx = x; t->bcast = bcast; t->allreduce = allreduce; t->flags = 0;
  if (const char* e = std::getenv("GPN_DIST_SCHEDULE")) if (e[0] == 'm') t->flags |= GPN_DIST_MESH_EXCHANGE;
  return t;
}
extern "C" void gpn_rccl_comm_destroy(gpn_dist_comm* t) {
  if (!t) return;
  delete static_cast<Ctx*>(t->ctx);
  delete t;
}
