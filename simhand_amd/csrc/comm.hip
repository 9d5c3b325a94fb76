// Thin RCCL wrappers behind an opaque communicator handle: what a maintainer who drives the step from
// outside torch.distributed binds for the exchange steps of the path -- all-gather of the projection / joint rows,
// all-reduce (MAX / MIN / SUM) of the distance statistics and of the parameter gradients.
//
// Replaces (reference): the scatter / gather / reduce-add of torch.nn.DataParallel under Lightning's strategy="dp"
// (src/experiments/main.py:152-163).
//
// librccl is resolved at RUN TIME (dlopen; the copy torch already mapped is reused when there is one), so the library has
// no link-time dependency on a particular RCCL build and loads on machines without it (the symbols then return an error).
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace sh {

// the slice of rccl.h this file needs (values are ABI constants of NCCL 2.x / RCCL)
typedef struct { char internal[128]; } nccl_unique_id;
typedef void* nccl_comm_t;
enum { kNcclInt64 = 4, kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclBfloat16 = 9 };
enum { kNcclSum = 0, kNcclMax = 2, kNcclMin = 3 };

struct RcclApi {
  int (*GetUniqueId)(nccl_unique_id*);
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int);
  int (*CommDestroy)(nccl_comm_t);
  int (*CommCount)(nccl_comm_t, int*);
  int (*CommUserRank)(nccl_comm_t, int*);
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
  bool ok;
};

static RcclApi g_rccl = {};
static std::once_flag g_rccl_once;

static void load_rccl() {
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);  // torch's copy, if this process already mapped it
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return;
#define SH_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name))
  SH_SYM(GetUniqueId, "ncclGetUniqueId");
  SH_SYM(CommInitRank, "ncclCommInitRank");
  SH_SYM(CommDestroy, "ncclCommDestroy");
  SH_SYM(CommCount, "ncclCommCount");
  SH_SYM(CommUserRank, "ncclCommUserRank");
  SH_SYM(AllGather, "ncclAllGather");
  SH_SYM(AllReduce, "ncclAllReduce");
  SH_SYM(GetErrorString, "ncclGetErrorString");
#undef SH_SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.CommCount && g_rccl.CommUserRank && g_rccl.AllGather &&
              g_rccl.AllReduce && g_rccl.GetErrorString;
}

static int rccl_ready() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) {
    set_error("librccl.so could not be loaded (dlopen / dlsym failed): RCCL collectives are unavailable in this process");
    return 1;
  }
  return 0;
}

static int rccl_check(int rc, const char* what) {
  if (rc != 0) {
    set_error("%s: %s", what, g_rccl.GetErrorString(rc));
    return 1;
  }
  return 0;
}

static int nccl_dtype(int dt) {
  switch (dt) {
    case SH_COMM_F32: return kNcclFloat32;
    case SH_COMM_F64: return kNcclFloat64;
    case SH_COMM_BF16: return kNcclBfloat16;
    case SH_COMM_I64: return kNcclInt64;
    default: return -1;
  }
}

}  // namespace sh

using namespace sh;

extern "C" {

int simhand_comm_unique_id(uint8_t* id) {
  SH_REQUIRE(id != nullptr, "comm_unique_id: NULL");
  if (rccl_ready()) return 1;
  nccl_unique_id u;
  if (rccl_check(g_rccl.GetUniqueId(&u), "ncclGetUniqueId")) return 1;
  memcpy(id, u.internal, SH_COMM_ID_BYTES);
  return 0;
}

int simhand_comm_init(const uint8_t* id, int world, int rank, void** comm) {
  SH_REQUIRE(id && comm && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
  if (rccl_ready()) return 1;
  nccl_unique_id u;
  memcpy(u.internal, id, SH_COMM_ID_BYTES);
  nccl_comm_t c = nullptr;
  if (rccl_check(g_rccl.CommInitRank(&c, world, u, rank), "ncclCommInitRank")) return 1;  // binds to the CURRENT device
  *comm = c;
  return 0;
}

int simhand_comm_destroy(void* comm) {
  SH_REQUIRE(comm != nullptr, "comm_destroy: NULL");
  if (rccl_ready()) return 1;
  return rccl_check(g_rccl.CommDestroy((nccl_comm_t)comm), "ncclCommDestroy");
}

int simhand_comm_world(void* comm, int* world, int* rank) {
  SH_REQUIRE(comm && world && rank, "comm_world: NULL");
  if (rccl_ready()) return 1;
  if (rccl_check(g_rccl.CommCount((nccl_comm_t)comm, world), "ncclCommCount")) return 1;
  return rccl_check(g_rccl.CommUserRank((nccl_comm_t)comm, rank), "ncclCommUserRank");
}

int simhand_comm_all_gather(void* comm, const void* send, void* recv, int64_t count, int dtype, sh_stream_t stream) {
  SH_REQUIRE(comm && send && recv && count >= 0, "comm_all_gather: bad arguments");
  SH_REQUIRE(nccl_dtype(dtype) >= 0, "comm_all_gather: dtype %d", dtype);
  if (rccl_ready()) return 1;
  return rccl_check(g_rccl.AllGather(send, recv, (size_t)count, nccl_dtype(dtype), (nccl_comm_t)comm, (hipStream_t)stream), "ncclAllGather");
}

int simhand_comm_all_reduce(void* comm, const void* send, void* recv, int64_t count, int dtype, int op, sh_stream_t stream) {
  SH_REQUIRE(comm && send && recv && count >= 0, "comm_all_reduce: bad arguments");
  SH_REQUIRE(nccl_dtype(dtype) >= 0, "comm_all_reduce: dtype %d", dtype);
  SH_REQUIRE(op == SH_COMM_SUM || op == SH_COMM_MAX || op == SH_COMM_MIN, "comm_all_reduce: op %d", op);
  if (rccl_ready()) return 1;
  const int nop = op == SH_COMM_SUM ? kNcclSum : (op == SH_COMM_MAX ? kNcclMax : kNcclMin);
  return rccl_check(g_rccl.AllReduce(send, recv, (size_t)count, nccl_dtype(dtype), nop, (nccl_comm_t)comm, (hipStream_t)stream), "ncclAllReduce");
}

}  // extern "C"
