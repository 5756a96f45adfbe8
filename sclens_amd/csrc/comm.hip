// RCCL inside the library ("a single RCCL gather over xGMI"): one communicator per rank, created from a unique id
// that the host ships over whatever channel launched the ranks. Every collective runs on the library's own device buffers and
// on the owning context's stream -- no host-framework tensor, no staging copy, one HIP runtime.
//   * ensemble exchange (scLENS.jl:771-778 sharded as member t -> rank t mod G): ncclAllGather of the min_pc x N blocks
//   * spread first phase (scLENS.jl:704, :717-721): ncclBroadcast of Vr2 and of the seed block of the partial eigensolver
//   * row-sharded cells (scLENS.jl:332-361 as a sum over cell blocks): ncclAllReduce of the statistics and of the partial Gram
//     matrix, through sclens_hip_comm_allreduce_cb, which has the signature of sclens_hip_allreduce_fn
//   * the few host-side numbers of the control flow (six doubles per search evaluation, spectra): the *_host variants stage
//     through a small device scratch so that the host needs no second communication library.
// librccl is opened at run time (dlopen): processes that never create a communicator (single-GPU hosts, the CPU tests, the
// Julia shim on one GPU) do not load its ~0.5 GB image, and a process that has already loaded a copy (PyTorch-ROCm bundles one,
// same soname) shares that copy and its HIP runtime instead of mapping a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace scl {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string err;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    std::string why;
    for (const char* nm : names) {
      api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
      const char* e = dlerror();  // reading it clears it: read once per attempt
      if (e && why.empty()) why = e;
    }
    if (!api.handle) {
      api.err = "librccl not found: " + why;
      return;
    }
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(api.handle, n);
      if (!p && api.err.empty()) api.err = std::string("librccl lacks ") + n;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Reduce = reinterpret_cast<decltype(api.Reduce)>(sym("ncclReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
  });
  return &api;
}

struct Comm {
  Ctx* ctx = nullptr;        // owner: its device and stream carry every collective
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  void* scratch = nullptr;   // staging of the *_host variants
  size_t scratch_bytes = 0;
  long calls = 0;
  double bytes = 0.0;
};

#define SCL_NCCL(ctx, expr)                                                                                       \
  do {                                                                                                            \
    ncclResult_t r__ = (expr);                                                                                    \
    if (r__ != ncclSuccess)                                                                                       \
      return (ctx)->fail(SCLENS_ERR_HIP, std::string(#expr) + ": " + (rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r__) : "rccl error")); \
  } while (0)

static int comm_scratch(Comm* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return SCLENS_OK;
  // through the pool, like every other allocation of the library: a plain hipMalloc does not see the pool's idle blocks and can
  // fail while the library is sitting on tens of GB of cached memory
  if (c->scratch) pool_free(c->scratch, c->ctx->stream);
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  SCL_HIP(c->ctx, pool_malloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return SCLENS_OK;
}

int comm_unique_id(Ctx* ctx, uint8_t* id) {
  RcclApi* api = rccl_api();
  if (!api->handle || !api->err.empty()) return ctx->fail(SCLENS_ERR_NO_DEVICE, api->err);
  ncclUniqueId u;
  SCL_NCCL(ctx, api->GetUniqueId(&u));
  static_assert(sizeof(u) == SCLENS_HIP_COMM_ID_BYTES, "unique id size");
  memcpy(id, &u, sizeof(u));
  return SCLENS_OK;
}

int comm_create(Ctx* ctx, const uint8_t* id, int rank, int world, Comm** out) {
  RcclApi* api = rccl_api();
  if (!api->handle || !api->err.empty()) return ctx->fail(SCLENS_ERR_NO_DEVICE, api->err);
  if (world < 1 || rank < 0 || rank >= world) return ctx->fail(SCLENS_ERR_ARG, "comm_create: bad rank / world");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  Comm* c = new Comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  pool_trim(ctx->device);  // RCCL allocates its own buffers with hipMalloc: hand the idle blocks back first
  ncclResult_t r = api->CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return ctx->fail(SCLENS_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
  }
  *out = c;
  return SCLENS_OK;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  hipStreamSynchronize(c->ctx->stream);
  if (c->comm) rccl_api()->CommDestroy(c->comm);
  if (c->scratch) pool_free(c->scratch, nullptr);  // the stream has just been synchronised
  delete c;
}

int comm_info(Comm* c, int* world, int* rank, int* version) {
  RcclApi* api = rccl_api();
  int w = 0, r = 0, v = 0;
  SCL_NCCL(c->ctx, api->CommCount(c->comm, &w));      // what RCCL itself reports, not what the host passed
  SCL_NCCL(c->ctx, api->CommUserRank(c->comm, &r));
  SCL_NCCL(c->ctx, api->GetVersion(&v));
  if (world) *world = w;
  if (rank) *rank = r;
  if (version) *version = v;
  return SCLENS_OK;
}

// in-place sum over the ranks; dtype 0 = fp64, 1 = fp32. `dev` may belong to any context of this device: the collective is
// ordered after everything the caller has already synchronised (callers of ShardReduce::sum synchronise their stream first)
int comm_allreduce(Comm* c, void* dev, int64_t count, int dtype) {
  if (count <= 0) return SCLENS_OK;
  if (dtype != 0 && dtype != 1) return c->ctx->fail(SCLENS_ERR_ARG, "comm_allreduce: dtype must be 0 (fp64) or 1 (fp32)");
  SCL_NCCL(c->ctx, rccl_api()->AllReduce(dev, dev, (size_t)count, dtype == 0 ? ncclDouble : ncclFloat, ncclSum, c->comm, c->ctx->stream));
  SCL_HIP(c->ctx, hipStreamSynchronize(c->ctx->stream));
  c->calls += 1;
  c->bytes += (double)count * (dtype == 0 ? 8 : 4);
  return SCLENS_OK;
}

// in-place sum onto rank `root` (the other ranks' buffers are unchanged)
int comm_reduce(Comm* c, void* dev, int64_t count, int dtype, int root) {
  if (count <= 0) return SCLENS_OK;
  if (dtype != 0 && dtype != 1) return c->ctx->fail(SCLENS_ERR_ARG, "comm_reduce: dtype must be 0 (fp64) or 1 (fp32)");
  if (root < 0 || root >= c->world) return c->ctx->fail(SCLENS_ERR_ARG, "comm_reduce: bad root");
  SCL_NCCL(c->ctx, rccl_api()->Reduce(dev, dev, (size_t)count, dtype == 0 ? ncclDouble : ncclFloat, ncclSum, root, c->comm, c->ctx->stream));
  SCL_HIP(c->ctx, hipStreamSynchronize(c->ctx->stream));
  c->calls += 1;
  c->bytes += (double)count * (dtype == 0 ? 8 : 4);
  return SCLENS_OK;
}

int comm_broadcast(Comm* c, void* dev, int64_t nbytes, int root) {
  if (nbytes <= 0) return SCLENS_OK;
  if (root < 0 || root >= c->world) return c->ctx->fail(SCLENS_ERR_ARG, "comm_broadcast: bad root");
  SCL_NCCL(c->ctx, rccl_api()->Broadcast(dev, dev, (size_t)nbytes, ncclChar, root, c->comm, c->ctx->stream));
  SCL_HIP(c->ctx, hipStreamSynchronize(c->ctx->stream));
  c->calls += 1;
  c->bytes += (double)nbytes;
  return SCLENS_OK;
}

// recv[r * nbytes .. (r + 1) * nbytes) = rank r's send buffer
int comm_allgather(Comm* c, const void* send, void* recv, int64_t nbytes) {
  if (nbytes <= 0) return SCLENS_OK;
  SCL_NCCL(c->ctx, rccl_api()->AllGather(send, recv, (size_t)nbytes, ncclChar, c->comm, c->ctx->stream));
  SCL_HIP(c->ctx, hipStreamSynchronize(c->ctx->stream));
  c->calls += 1;
  c->bytes += (double)nbytes * c->world;
  return SCLENS_OK;
}

int comm_allgather_host(Comm* c, const void* send_h, void* recv_h, int64_t nbytes) {
  if (nbytes <= 0) return SCLENS_OK;
  const size_t nb = (size_t)nbytes;
  SCL_TRY(comm_scratch(c, nb * (size_t)(c->world + 1)));
  char* s = static_cast<char*>(c->scratch);
  hipStream_t st = c->ctx->stream;
  SCL_HIP(c->ctx, hipMemcpyAsync(s, send_h, nb, hipMemcpyHostToDevice, st));
  SCL_NCCL(c->ctx, rccl_api()->AllGather(s, s + nb, nb, ncclChar, c->comm, st));
  SCL_HIP(c->ctx, hipMemcpyAsync(recv_h, s + nb, nb * (size_t)c->world, hipMemcpyDeviceToHost, st));
  SCL_HIP(c->ctx, hipStreamSynchronize(st));
  c->calls += 1;
  c->bytes += (double)nb * c->world;
  return SCLENS_OK;
}

int comm_broadcast_host(Comm* c, void* buf_h, int64_t nbytes, int root) {
  if (nbytes <= 0) return SCLENS_OK;
  if (root < 0 || root >= c->world) return c->ctx->fail(SCLENS_ERR_ARG, "comm_broadcast_host: bad root");
  const size_t nb = (size_t)nbytes;
  SCL_TRY(comm_scratch(c, nb));
  hipStream_t st = c->ctx->stream;
  if (c->rank == root) SCL_HIP(c->ctx, hipMemcpyAsync(c->scratch, buf_h, nb, hipMemcpyHostToDevice, st));
  SCL_NCCL(c->ctx, rccl_api()->Broadcast(c->scratch, c->scratch, nb, ncclChar, root, c->comm, st));
  if (c->rank != root) SCL_HIP(c->ctx, hipMemcpyAsync(buf_h, c->scratch, nb, hipMemcpyDeviceToHost, st));
  SCL_HIP(c->ctx, hipStreamSynchronize(st));
  c->calls += 1;
  c->bytes += (double)nb;
  return SCLENS_OK;
}

}  // namespace scl

struct sclens_hip_comm { scl::Comm* c; };

extern "C" {

int sclens_hip_comm_unique_id(sclens_hip_ctx* ctx, uint8_t* id) {
  if (!ctx || !id) return SCLENS_ERR_ARG;
  return scl::comm_unique_id(&ctx->c, id);
}
int sclens_hip_comm_create(sclens_hip_ctx* ctx, const uint8_t* id, int rank, int world, sclens_hip_comm** out) {
  if (!ctx || !id || !out) return SCLENS_ERR_ARG;
  *out = nullptr;
  hipSetDevice(ctx->c.device);
  scl::Comm* c = nullptr;
  const int rc = scl::comm_create(&ctx->c, id, rank, world, &c);
  if (rc != SCLENS_OK) return rc;
  *out = new sclens_hip_comm{c};
  return SCLENS_OK;
}
void sclens_hip_comm_destroy(sclens_hip_comm* comm) {
  if (!comm) return;
  if (comm->c) {
    hipSetDevice(comm->c->ctx->device);
    scl::comm_destroy(comm->c);
  }
  delete comm;
}
#define COMM_GUARD(w) \
  if (!(w) || !(w)->c) return SCLENS_ERR_ARG; \
  hipSetDevice((w)->c->ctx->device)

int sclens_hip_comm_info(sclens_hip_comm* comm, int* world, int* rank, int* rccl_version) {
  COMM_GUARD(comm);
  return scl::comm_info(comm->c, world, rank, rccl_version);
}
int sclens_hip_comm_stats(sclens_hip_comm* comm, int64_t* calls, double* bytes) {
  COMM_GUARD(comm);
  if (calls) *calls = comm->c->calls;
  if (bytes) *bytes = comm->c->bytes;
  return SCLENS_OK;
}
int sclens_hip_comm_allreduce(sclens_hip_comm* comm, void* dev_ptr, int64_t count, int dtype) {
  COMM_GUARD(comm);
  return scl::comm_allreduce(comm->c, dev_ptr, count, dtype);
}
int sclens_hip_comm_broadcast(sclens_hip_comm* comm, void* dev_ptr, int64_t nbytes, int root) {
  COMM_GUARD(comm);
  return scl::comm_broadcast(comm->c, dev_ptr, nbytes, root);
}
int sclens_hip_comm_allgather(sclens_hip_comm* comm, const void* send_dev, void* recv_dev, int64_t nbytes) {
  COMM_GUARD(comm);
  return scl::comm_allgather(comm->c, send_dev, recv_dev, nbytes);
}
int sclens_hip_comm_allgather_host(sclens_hip_comm* comm, const void* send, void* recv, int64_t nbytes) {
  COMM_GUARD(comm);
  return scl::comm_allgather_host(comm->c, send, recv, nbytes);
}
int sclens_hip_comm_broadcast_host(sclens_hip_comm* comm, void* buf, int64_t nbytes, int root) {
  COMM_GUARD(comm);
  return scl::comm_broadcast_host(comm->c, buf, nbytes, root);
}
/* signature of sclens_hip_allreduce_fn with user = the sclens_hip_comm handle */
int sclens_hip_comm_allreduce_cb(void* user, void* dev_ptr, int64_t count, int dtype) {
  return sclens_hip_comm_allreduce(static_cast<sclens_hip_comm*>(user), dev_ptr, count, dtype);
}
int sclens_hip_comm_reduce_cb(void* user, void* dev_ptr, int64_t count, int dtype, int root) {
  sclens_hip_comm* comm = static_cast<sclens_hip_comm*>(user);
  COMM_GUARD(comm);
  return scl::comm_reduce(comm->c, dev_ptr, count, dtype, root);
}
const char* sclens_hip_comm_last_error(sclens_hip_comm* comm) { return (comm && comm->c) ? comm->c->ctx->err.c_str() : "null communicator"; }

}  // extern "C"
