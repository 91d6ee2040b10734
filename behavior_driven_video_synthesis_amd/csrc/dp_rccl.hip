// Data-parallel gradient exchange over RCCL behind the C ABI (dp_{init,allreduce_bucket,finalize}).
//
// Replaces the per-step parameter broadcast / input scatter / output gather of the reference's single-process
// nn.DataParallel (experiments/shape_and_pose_net.py:213-214,223-224,230-233): one process per GPU keeps a persistent
// replica, and the ONLY exchange of a training step is the sum of the flat gradient buckets -- one ncclAllReduce per bucket,
// in place, on the stream the caller names (the averager's communication stream), so that it overlaps with the rest of
// backward and can sit inside a captured hipGraph like any kernel launch.
//
// RCCL is bound at RUN time: dlopen("librccl.so.1") returns the instance PyTorch has already mapped (same soname) when there
// is one, the system's otherwise -- the library itself has no link-time dependency on RCCL, so it loads (and exports every
// symbol of include/vunet_hip.h) on machines and in tests that never call these three functions.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

// the part of rccl.h this file needs (ABI of NCCL 2.x / RCCL: stable opaque types and enum values)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
enum { ncclFloat32 = 7 };
enum { ncclSum = 0, ncclAvg = 4 };
typedef int (*get_unique_id_fn)(ncclUniqueId*);
typedef int (*comm_init_rank_fn)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*comm_destroy_fn)(ncclComm_t);
typedef const char* (*get_error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  all_reduce_fn all_reduce = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  ncclComm_t comm = nullptr;
  int world = 0, rank = -1;
};
Rccl g;

bool bind() {
  if (g.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g.handle) break;
  }
  if (!g.handle) return false;
  g.get_unique_id = (get_unique_id_fn)dlsym(g.handle, "ncclGetUniqueId");
  g.comm_init_rank = (comm_init_rank_fn)dlsym(g.handle, "ncclCommInitRank");
  g.all_reduce = (all_reduce_fn)dlsym(g.handle, "ncclAllReduce");
  g.comm_destroy = (comm_destroy_fn)dlsym(g.handle, "ncclCommDestroy");
  if (!g.get_unique_id || !g.comm_init_rank || !g.all_reduce || !g.comm_destroy) {
    dlclose(g.handle);
    g.handle = nullptr;
    return false;
  }
  return true;
}

}  // namespace

// 1 if RCCL can be bound in this process (librccl found, every entry point resolved), else 0.  Touches no GPU and no
// communicator: ranks exchange this flag BEFORE anyone enters the blocking ncclCommInitRank (a rank that cannot bind would
// otherwise leave the others waiting inside it forever).
extern "C" int vunet_dp_available(void) { return bind() ? 1 : 0; }

extern "C" int vunet_dp_unique_id(void* id128) {
  if (!id128) return VUNET_ERR_ARG;
  if (!bind()) return VUNET_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g.get_unique_id(&id) != 0) return VUNET_ERR_LAUNCH;
  memcpy(id128, id.internal, sizeof(id.internal));
  return VUNET_OK;
}

extern "C" int vunet_dp_init(int32_t world, int32_t rank, const void* id128) {
  if (world < 1 || rank < 0 || rank >= world || !id128) return VUNET_ERR_ARG;
  if (g.comm) return VUNET_ERR_ARG;   // one communicator per process (one process per GPU)
  if (!bind()) return VUNET_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  ncclComm_t c = nullptr;
  if (g.comm_init_rank(&c, world, id, rank) != 0 || !c) return VUNET_ERR_LAUNCH;
  g.comm = c;
  g.world = world;
  g.rank = rank;
  return VUNET_OK;
}

extern "C" int vunet_dp_world(void) { return g.comm ? g.world : 0; }

extern "C" int vunet_dp_allreduce_bucket(float* buf, int64_t n, int32_t average, void* stream) {
  if (!buf || n < 0) return VUNET_ERR_ARG;
  if (!g.comm) return VUNET_ERR_ARG;
  if (n == 0) return VUNET_OK;
  const int rc = g.all_reduce(buf, buf, (size_t)n, ncclFloat32, average ? ncclAvg : ncclSum, g.comm, (hipStream_t)stream);
  return rc == 0 ? VUNET_OK : VUNET_ERR_LAUNCH;
}

extern "C" int vunet_dp_finalize(void) {
  if (!g.comm) return VUNET_OK;
  const int rc = g.comm_destroy(g.comm);
  g.comm = nullptr;
  g.world = 0;
  g.rank = -1;
  return rc == 0 ? VUNET_OK : VUNET_ERR_LAUNCH;
}
