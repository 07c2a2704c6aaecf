// libgpnative_rccl.so: the gpn_dist_comm callback table (include/gpnative.h) over RCCL communicators.
// Kept out of libgpnative.so so that the kernel library does not depend on a communication runtime.
// RCCL collectives are enqueued on the HIP stream they are given (ncclBroadcast / ncclAllReduce),
// which is exactly the contract of the callbacks; xGMI routing is RCCL's business.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <new>
#include "../../include/gpnative.h"

namespace {
struct Ctx { ncclComm_t row, col, world; };

int bcast(void* c, int which, double* buf, int64_t count, int root, void* stream) {
  Ctx* x = static_cast<Ctx*>(c);
  ncclComm_t comm = which == 0 ? x->row : x->col;
  if (!comm) return count == 0 ? 0 : -1;
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
  Ctx* x = new (std::nothrow) Ctx{static_cast<ncclComm_t>(row), static_cast<ncclComm_t>(col), static_cast<ncclComm_t>(world)};
  if (!t || !x) { delete t; delete x; return nullptr; }
  t->ctx = x; t->bcast = bcast; t->allreduce = allreduce; t->flags = 0;
  return t;
}
extern "C" void gpn_rccl_comm_destroy(gpn_dist_comm* t) {
  if (!t) return;
  delete static_cast<Ctx*>(t->ctx);
  delete t;
}
