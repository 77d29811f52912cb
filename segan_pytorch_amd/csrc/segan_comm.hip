// segan_comm.hip — the data-parallel exchange of the GAN step behind the C ABI: thin wrappers
// over RCCL (one communicator per process = per GPU, ring / tree over xGMI chosen by RCCL).
//
// The reference has nothing here ("Multi-GPU is not supported yet", README.md:79); SURVEY.md
// section 8(b)/(e) specifies the seam: segan_comm_init / segan_allreduce / segan_comm_destroy, the
// only library-owned resources.  RCCL is bound at RUN time (dlopen by soname): a process that
// already carries an RCCL — PyTorch-ROCm ships its own librccl.so.1 — keeps exactly one copy,
// and a single-GPU user of the library needs no RCCL at all.
#include "segan_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) =
      nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

struct Comm {
  ncclComm_t comm;
  int world, rank, device;
};

int bind_rccl() {
  if (g_rccl.h) return SEGAN_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    segan_set_error("comm: cannot load RCCL (librccl.so.1): %s", dlerror());
    return SEGAN_ELAUNCH;
  }
  Rccl r;
  r.h = h;
#define BIND(field, sym)                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));       \
  if (!r.field) {                                                     \
    segan_set_error("comm: RCCL symbol %s missing", sym);             \
    return SEGAN_ELAUNCH;                                             \
  }
  BIND(GetUniqueId, "ncclGetUniqueId")
  BIND(CommInitRank, "ncclCommInitRank")
  BIND(CommDestroy, "ncclCommDestroy")
  BIND(AllReduce, "ncclAllReduce")
  BIND(Broadcast, "ncclBroadcast")
  BIND(AllGather, "ncclAllGather")
  BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
  g_rccl = r;
  return SEGAN_OK;
}

int rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return SEGAN_OK;
  segan_set_error("%s: RCCL error %d: %s", what, (int)r, g_rccl.GetErrorString(r));
  return SEGAN_ELAUNCH;
}

__global__ void comm_scale_kernel(float* __restrict__ p, float s, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] *= s;
}
}  // namespace

extern "C" int segan_comm_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

// rank 0 creates the rendezvous id (128 bytes) and hands it to the other ranks by whatever
// side channel the host has (a file, MPI, torch.distributed's store, a TCP socket)
extern "C" int segan_comm_unique_id(void* id_out) {
  SEGAN_REQUIRE(id_out, "comm_unique_id: NULL pointer");
  if (int e = bind_rccl()) return e;
  return rccl_check(g_rccl.GetUniqueId((ncclUniqueId*)id_out), "comm_unique_id");
}

// collective over all `world` ranks; the calling thread's current device becomes the
// communicator's device
extern "C" int segan_comm_init(void** comm_out, int world, int rank, const void* id) {
  SEGAN_REQUIRE(comm_out && id, "comm_init: NULL pointer");
  SEGAN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
  if (int e = bind_rccl()) return e;
  Comm* c = new Comm();
  c->world = world;
  c->rank = rank;
  (void)hipGetDevice(&c->device);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  if (int e = rccl_check(g_rccl.CommInitRank(&c->comm, world, uid, rank), "comm_init")) {
    delete c;
    return e;
  }
  *comm_out = c;
  return SEGAN_OK;
}

extern "C" int segan_comm_destroy(void* comm) {
  if (!comm) return SEGAN_OK;
  Comm* c = (Comm*)comm;
  const int e = rccl_check(g_rccl.CommDestroy(c->comm), "comm_destroy");
  delete c;
  return e;
}

// in-place sum over ranks of n floats, then * scale (1/world for the mean) — both enqueued on
// `stream`; asynchronous w.r.t. the host
extern "C" int segan_allreduce(void* comm, float* buf, size_t n, float scale, void* stream) {
  SEGAN_REQUIRE(comm && buf && n > 0, "allreduce: bad arguments");
  Comm* c = (Comm*)comm;
  hipStream_t st = (hipStream_t)stream;
  if (int e = rccl_check(g_rccl.AllReduce(buf, buf, n, ncclFloat32, ncclSum, c->comm, st),
                         "allreduce"))
    return e;
  if (scale != 1.0f) {
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(comm_scale_kernel, dim3(blocks), dim3(256), 0, st, buf, scale, n);
    return segan_check_launch("allreduce scale");
  }
  return SEGAN_OK;
}

extern "C" int segan_broadcast(void* comm, float* buf, size_t n, int root, void* stream) {
  SEGAN_REQUIRE(comm && buf && n > 0, "broadcast: bad arguments");
  Comm* c = (Comm*)comm;
  SEGAN_REQUIRE(root >= 0 && root < c->world, "broadcast: root %d of %d", root, c->world);
  return rccl_check(g_rccl.Broadcast(buf, buf, n, ncclFloat32, root, c->comm, (hipStream_t)stream),
                    "broadcast");
}

// recv[world][n] <- every rank's send[n] (synchronised BatchNorm partial statistics)
extern "C" int segan_allgather(void* comm, const float* send, float* recv, size_t n, void* stream) {
  SEGAN_REQUIRE(comm && send && recv && n > 0, "allgather: bad arguments");
  Comm* c = (Comm*)comm;
  return rccl_check(g_rccl.AllGather(send, recv, n, ncclFloat32, c->comm, (hipStream_t)stream),
                    "allgather");
}

extern "C" int segan_comm_rank(void* comm) { return comm ? ((Comm*)comm)->rank : -1; }
extern "C" int segan_comm_world(void* comm) { return comm ? ((Comm*)comm)->world : -1; }
