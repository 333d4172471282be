// C1 / C2 / C3: the collectives of the data-parallel exchange as a C-ABI over RCCL (`cfhip_comm_*`) —
// gradient all-reduce (replaces what torch-DDP's reducer would have done behind trainer.py:268-272), parameter
// broadcast (the DDP constructor's), embedding all-gather / reduce-scatter (CLIP contrastive).
//
// RCCL is reached through dlopen / dlsym, not linked: a PyTorch process already holds one copy of librccl.so
// (torch/lib) and a second, link-time copy from /opt/rocm would give two RCCL runtimes in one address space.  The
// library already loaded is preferred (RTLD_NOLOAD), then the usual search path.  The communicator is owned by the
// caller (opaque handle), every collective takes the HIP stream it runs on — the caller's own comm stream, so the
// exchange never lands on a stream this package did not check for a hardware queue of its own
// (functional.distinct_stream).  One communicator per process (one process per GPU), xGMI underneath.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_mu;

bool load_rccl() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.handle != nullptr) return true;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);  // the copy torch loaded, if any
  if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) {
    cfhip_set_error("comm: cannot load librccl.so: %s", dlerror());
    return false;
  }
#define CFHIP_SYM(field, name)                                              \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (g_rccl.field == nullptr) {                                            \
    cfhip_set_error("comm: librccl.so lacks %s", name);                     \
    return false;                                                           \
  }
  CFHIP_SYM(GetUniqueId, "ncclGetUniqueId")
  CFHIP_SYM(CommInitRank, "ncclCommInitRank")
  CFHIP_SYM(CommDestroy, "ncclCommDestroy")
  CFHIP_SYM(CommCount, "ncclCommCount")
  CFHIP_SYM(CommUserRank, "ncclCommUserRank")
  CFHIP_SYM(AllReduce, "ncclAllReduce")
  CFHIP_SYM(AllGather, "ncclAllGather")
  CFHIP_SYM(ReduceScatter, "ncclReduceScatter")
  CFHIP_SYM(Broadcast, "ncclBroadcast")
  CFHIP_SYM(GetErrorString, "ncclGetErrorString")
#undef CFHIP_SYM
  g_rccl.handle = h;
  return true;
}

inline bool dtype_of(int dtype, ncclDataType_t* out) {
  if (dtype == 0) *out = ncclFloat32;
  else if (dtype == 1) *out = ncclBfloat16;
  else return false;
  return true;
}

#define CFHIP_RCCL(call, what)                                                        \
  do {                                                                                \
    ncclResult_t r__ = (call);                                                        \
    if (r__ != ncclSuccess) {                                                         \
      cfhip_set_error("comm: %s failed: %s", what, g_rccl.GetErrorString(r__));      \
      return CFHIP_ERR_LAUNCH;                                                        \
    }                                                                                 \
  } while (0)

}  // namespace

extern "C" int cfhip_comm_unique_id(void* out128) {
  CFHIP_REQUIRE(out128 != nullptr, "comm_unique_id: null pointer");
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  CFHIP_RCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out128, &id, sizeof(id));
  return CFHIP_OK;
}

extern "C" int cfhip_comm_init(int rank, int world, const void* uid128, void** comm) {
  CFHIP_REQUIRE(uid128 != nullptr && comm != nullptr, "comm_init: null pointer");
  CFHIP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank %d of %d", rank, world);
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  ncclUniqueId id;
  memcpy(&id, uid128, sizeof(id));
  ncclComm_t c = nullptr;
  CFHIP_RCCL(g_rccl.CommInitRank(&c, world, id, rank), "ncclCommInitRank");  // the current HIP device is this rank's GPU
  *comm = c;
  return CFHIP_OK;
}

extern "C" int cfhip_comm_destroy(void* comm) {
  if (comm == nullptr) return CFHIP_OK;
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  CFHIP_RCCL(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(comm)), "ncclCommDestroy");
  return CFHIP_OK;
}

// What RCCL itself says about the communicator (not the caller's bookkeeping): ranks in it and this process's rank.
extern "C" int cfhip_comm_count(void* comm, int* world, int* rank) {
  CFHIP_REQUIRE(comm != nullptr && world != nullptr, "comm_count: null pointer");
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  CFHIP_RCCL(g_rccl.CommCount(reinterpret_cast<ncclComm_t>(comm), world), "ncclCommCount");
  if (rank != nullptr) CFHIP_RCCL(g_rccl.CommUserRank(reinterpret_cast<ncclComm_t>(comm), rank), "ncclCommUserRank");
  return CFHIP_OK;
}

extern "C" int cfhip_comm_allreduce(void* comm, void* buf, size_t count, int dtype, void* stream) {
  ncclDataType_t dt;
  CFHIP_REQUIRE(comm && buf, "comm_allreduce: null pointer");
  CFHIP_REQUIRE(dtype_of(dtype, &dt), "comm_allreduce: dtype %d (0 = f32, 1 = bf16)", dtype);
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  CFHIP_RCCL(g_rccl.AllReduce(buf, buf, count, dt, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                              reinterpret_cast<hipStream_t>(stream)), "ncclAllReduce");
  return CFHIP_OK;
}

extern "C" int cfhip_comm_allgather(void* comm, const void* send, void* recv, size_t count_per_rank, int dtype, void* stream) {
  ncclDataType_t dt;
  CFHIP_REQUIRE(comm && send && recv, "comm_allgather: null pointer");
  CFHIP_REQUIRE(dtype_of(dtype, &dt), "comm_allgather: dtype %d (0 = f32, 1 = bf16)", dtype);
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  CFHIP_RCCL(g_rccl.AllGather(send, recv, count_per_rank, dt, reinterpret_cast<ncclComm_t>(comm),
                              reinterpret_cast<hipStream_t>(stream)), "ncclAllGather");
  return CFHIP_OK;
}

extern "C" int cfhip_comm_reduce_scatter(void* comm, const void* send, void* recv, size_t count_per_rank, int dtype,
                                         void* stream) {
  ncclDataType_t dt;
  CFHIP_REQUIRE(comm && send && recv, "comm_reduce_scatter: null pointer");
  CFHIP_REQUIRE(dtype_of(dtype, &dt), "comm_reduce_scatter: dtype %d (0 = f32, 1 = bf16)", dtype);
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  CFHIP_RCCL(g_rccl.ReduceScatter(send, recv, count_per_rank, dt, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                                  reinterpret_cast<hipStream_t>(stream)), "ncclReduceScatter");
  return CFHIP_OK;
}

extern "C" int cfhip_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream) {
  ncclDataType_t dt;
  CFHIP_REQUIRE(comm && buf, "comm_broadcast: null pointer");
  CFHIP_REQUIRE(dtype_of(dtype, &dt), "comm_broadcast: dtype %d (0 = f32, 1 = bf16)", dtype);
  if (!load_rccl()) return CFHIP_ERR_LAUNCH;
  CFHIP_RCCL(g_rccl.Broadcast(buf, buf, count, dt, root, reinterpret_cast<ncclComm_t>(comm),
                              reinterpret_cast<hipStream_t>(stream)), "ncclBroadcast");
  return CFHIP_OK;
}
