// Data-parallel gradient exchange on the EXECUTOR'S OWN stream (one RCCL all-reduce(sum) per step over
// the flat [gradient sums | loss_sum | count] buffer, xGMI).  The collective is enqueued between the two native phases of
// the step (fwd+bwd -> all-reduce -> Adam) on the stream the kernels run on: no second stream, no event hand-offs, nothing
// the host has to wait for.  torch's ProcessGroupNCCL issues the same collective on a stream of its own and brackets it with
// two event waits; with a 0.1 ms step those hand-offs are a measurable share of the step (profiles/r02_*_collective.json).
//
// RCCL is resolved with dlopen at the first call, so the library loads (and every single-GPU entry point works) on a box
// without RCCL, and a process that already loaded RCCL (torch does) shares that one instance.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

using namespace hmp;

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  bool tried = false;
  char why[256] = {0};
};

RcclApi& api() {
  static RcclApi a;
  if (a.tried) return a;
  a.tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {  // an instance the process already holds (torch's) is shared
    a.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (a.handle) break;
  }
  for (int i = 0; i < 3 && !a.handle; ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!a.handle) {
    snprintf(a.why, sizeof(a.why), "librccl not found: %s", dlerror());
    return a;
  }
#define HMP_SYM(field, name)                                                     \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name));          \
  if (!a.field) { snprintf(a.why, sizeof(a.why), "librccl lacks %s", name); a.handle = nullptr; return a; }
  HMP_SYM(GetUniqueId, "ncclGetUniqueId")
  HMP_SYM(CommInitRank, "ncclCommInitRank")
  HMP_SYM(CommDestroy, "ncclCommDestroy")
  HMP_SYM(AllReduce, "ncclAllReduce")
  HMP_SYM(Broadcast, "ncclBroadcast")
  HMP_SYM(GetErrorString, "ncclGetErrorString")
  HMP_SYM(CommCount, "ncclCommCount")
  HMP_SYM(CommUserRank, "ncclCommUserRank")
#undef HMP_SYM
  return a;
}

#define HMP_RCCL(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess) HMP_FAIL(HMP_E_HIP, "%s failed: %s", #expr, api().GetErrorString(_r));              \
  } while (0)

}  // namespace

struct hmp_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" int hmp_comm_unique_id(void* id128) {
  HMP_CHECK_ARG(id128, "hmp_comm_unique_id: null argument");
  RcclApi& a = api();
  if (!a.handle) HMP_FAIL(HMP_E_UNSUPPORTED, "hmp_comm_unique_id: %s", a.why);
  static_assert(sizeof(ncclUniqueId) == HMP_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  HMP_RCCL(a.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return HMP_OK;
}

extern "C" int hmp_comm_create(const void* id128, int32_t rank, int32_t world, hmp_comm** out) {
  HMP_CHECK_ARG(id128 && out, "hmp_comm_create: null argument");
  HMP_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "hmp_comm_create: rank %d of %d", rank, world);
  RcclApi& a = api();
  if (!a.handle) HMP_FAIL(HMP_E_UNSUPPORTED, "hmp_comm_create: %s", a.why);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  hmp_comm* c = new hmp_comm;
  c->rank = rank; c->world = world;
  ncclResult_t r = a.CommInitRank(&c->comm, world, id, rank);  // communicator of the CURRENT device (one process per GPU)
  if (r != ncclSuccess) {
    delete c;
    HMP_FAIL(HMP_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, a.GetErrorString(r));
  }
  *out = c;
  return HMP_OK;
}

extern "C" void hmp_comm_destroy(hmp_comm* c) {
  if (!c) return;
  if (c->comm && api().handle) (void)api().CommDestroy(c->comm);
  delete c;
}

extern "C" int hmp_comm_query(hmp_comm* c, int32_t* n_ranks, int32_t* rank) {
  HMP_CHECK_ARG(c && c->comm && n_ranks && rank, "hmp_comm_query: bad argument");
  int cnt = 0, rk = 0;
  HMP_RCCL(api().CommCount(c->comm, &cnt));
  HMP_RCCL(api().CommUserRank(c->comm, &rk));
  *n_ranks = cnt;
  *rank = rk;
  return HMP_OK;
}

extern "C" int hmp_comm_allreduce_sum_f32(hmp_comm* c, float* d_buf, int64_t n, void* stream) {
  HMP_CHECK_ARG(c && c->comm && d_buf && n >= 0, "hmp_comm_allreduce_sum_f32: bad argument");
  if (n == 0) return HMP_OK;
  HMP_RCCL(api().AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
  return HMP_OK;
}

extern "C" int hmp_comm_broadcast_f32(hmp_comm* c, float* d_buf, int64_t n, int32_t root, void* stream) {
  HMP_CHECK_ARG(c && c->comm && d_buf && n >= 0 && root >= 0 && root < c->world, "hmp_comm_broadcast_f32: bad argument");
  if (n == 0) return HMP_OK;
  HMP_RCCL(api().Broadcast(d_buf, d_buf, (size_t)n, ncclFloat32, root, c->comm, (hipStream_t)stream));
  return HMP_OK;
}
