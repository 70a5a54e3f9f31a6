// comm.hip — the in-library RCCL communicator (one process and one wc_ctx per GPU, RCCL over xGMI).
//
// librccl.so is opened at run time (dlopen): the library has no link-time dependency on it and a single-GPU user never loads
// it.  wc_comm_rccl_init() installs a wc_comm whose three collectives are ENQUEUED ON THE CTX'S STREAM - no host
// synchronisation, no Python, no callback into the caller's runtime:
//   allreduce_f64   ncclAllReduce(ncclDouble, ncclSum) in place: the packed {upper block triangle of H, g, cost} buffer of a
//                   linearisation and the candidate-cost scalar (csrc/window.hip)
//   alltoallv       grouped ncclSend / ncclRecv of bytes: the ONE exchange step of the sharded extraction (csrc/route.hip)
//   allgatherv      grouped ncclSend / ncclRecv: surfel lists (route.hip) and the gated neighbour lists of the query-sharded
//                   matcher (csrc/match.hip)
// The unique id is created by rank 0 (wc_comm_rccl_unique_id) and handed to the other ranks by whatever launched them
// (torch.distributed / MPI / a file): 128 opaque bytes.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>

#include "ctx.h"

// The few names of RCCL's C API this file uses, declared here so that the library BUILDS without the rccl-dev headers (it is
// only ever reached through dlopen): nccl.h of RCCL 2.x / ROCm 6-7 - ncclUniqueId is 128 opaque bytes, the enums below have
// been stable since NCCL 2.0.  wc_comm_rccl_init checks ncclGetVersion() >= 2.0 before trusting them.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
static_assert(sizeof(ncclUniqueId) == 128, "the 128-byte id of include/wildcat_hip.h");

namespace {

struct RcclApi {
  void *so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
};

RcclApi *rccl_api(std::string *err) {
  static RcclApi api;
  static bool tried = false, ok = false;
  if (!tried) {
    tried = true;
    // an RCCL that the process has loaded already (e.g. the one bundled with PyTorch) is reused: two RCCL instances in one
    // process would each bring up their own transport on the same GPUs
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      api.so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (api.so) break;
    }
    if (!api.so)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.so) break;
      }
    if (api.so) {
      ok = true;
#define WC_SYM(field, sym)                                   \
  api.field = (decltype(api.field))dlsym(api.so, sym);       \
  if (!api.field) ok = false;
      WC_SYM(GetUniqueId, "ncclGetUniqueId")
      WC_SYM(CommInitRank, "ncclCommInitRank")
      WC_SYM(CommDestroy, "ncclCommDestroy")
      WC_SYM(AllReduce, "ncclAllReduce")
      WC_SYM(Send, "ncclSend")
      WC_SYM(Recv, "ncclRecv")
      WC_SYM(GroupStart, "ncclGroupStart")
      WC_SYM(GroupEnd, "ncclGroupEnd")
      WC_SYM(GetErrorString, "ncclGetErrorString")
      WC_SYM(GetVersion, "ncclGetVersion")
#undef WC_SYM
      int ver = 0;
      if (ok && (api.GetVersion(&ver) != ncclSuccess || ver < 2000)) ok = false;  // (the local declarations above describe 2.x)
    }
  }
  if (!ok) {
    if (err) *err = api.so ? "librccl.so lacks an expected symbol or is older than 2.0" : (std::string("dlopen(librccl.so) failed: ") + (dlerror() ? dlerror() : "?"));
    return nullptr;
  }
  return &api;
}

struct RcclComm {
  RcclApi *api;
  ncclComm_t comm;
  wc_ctx *ctx;
  int rank, world;
};

int rc_allreduce(void *user, double *d_buf, uint64_t count) {
  RcclComm *c = (RcclComm *)user;
  return c->api->AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm, c->ctx->stream) == ncclSuccess ? 0 : 1;
}

int rc_alltoallv(void *user, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes) {
  RcclComm *c = (RcclComm *)user;
  const char *s = (const char *)d_send;
  char *r = (char *)d_recv;
  bool ok = c->api->GroupStart() == ncclSuccess;
  for (int p = 0; p < c->world && ok; ++p) {
    if (send_bytes[p]) ok = ok && c->api->Send(s, (size_t)send_bytes[p], ncclUint8, p, c->comm, c->ctx->stream) == ncclSuccess;
    if (recv_bytes[p]) ok = ok && c->api->Recv(r, (size_t)recv_bytes[p], ncclUint8, p, c->comm, c->ctx->stream) == ncclSuccess;
    s += send_bytes[p];
    r += recv_bytes[p];
  }
  ok = (c->api->GroupEnd() == ncclSuccess) && ok;
  return ok ? 0 : 1;
}

int rc_allgatherv(void *user, const void *d_send, uint64_t send_bytes, void *d_recv, const uint64_t *recv_bytes) {
  RcclComm *c = (RcclComm *)user;
  char *r = (char *)d_recv;
  bool ok = c->api->GroupStart() == ncclSuccess;
  for (int p = 0; p < c->world && ok; ++p) {
    if (send_bytes) ok = ok && c->api->Send(d_send, (size_t)send_bytes, ncclUint8, p, c->comm, c->ctx->stream) == ncclSuccess;
    if (recv_bytes[p]) ok = ok && c->api->Recv(r, (size_t)recv_bytes[p], ncclUint8, p, c->comm, c->ctx->stream) == ncclSuccess;
    r += recv_bytes[p];
  }
  ok = (c->api->GroupEnd() == ncclSuccess) && ok;
  return ok ? 0 : 1;
}

}  // namespace

// the in-library binding's all-reduce on a stream OTHER than the ctx stream (window.hip: the large collective of the two-collective form
// runs beside the bias elimination).  -1: the ctx's communicator is not this binding.  The caller orders the streams with events so
// that the communicator's collectives stay totally ordered.
int wc_rccl_allreduce_on(wc_ctx *ctx, double *d_buf, uint64_t count, hipStream_t st) {
  if (!ctx || !ctx->rccl || ctx->comm.user != ctx->rccl) return -1;
  RcclComm *c = (RcclComm *)ctx->rccl;
  return c->api->AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm, st) == ncclSuccess ? 0 : 1;
}

extern "C" int wc_comm_rccl_unique_id(char out128[128]) {
  if (!out128) return WC_ERR_ARG;
  RcclApi *api = rccl_api(nullptr);
  if (!api) return WC_ERR_HIP;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return WC_ERR_HIP;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(out128, &id, 128);
  return WC_OK;
}

extern "C" int wc_comm_rccl_init(wc_ctx *ctx, int rank, int world, const char id128[128]) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  std::string err;
  RcclApi *api = rccl_api(&err);
  if (!api) return wc_fail(ctx, WC_ERR_HIP, "%s", err.c_str());
  if (ctx->rccl) return wc_fail(ctx, WC_ERR_ARG, "wc_comm_rccl_init: the context already has an RCCL communicator");
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  RcclComm *c = new RcclComm{api, nullptr, ctx, rank, world};
  const ncclResult_t rc = api->CommInitRank(&c->comm, world, id, rank);
  if (rc != ncclSuccess) {
    delete c;
    return wc_fail(ctx, WC_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(rc));
  }
  ctx->rccl = c;
  wc_comm v;
  std::memset(&v, 0, sizeof(v));
  v.user = c, v.rank = rank, v.world = world;
  v.allreduce_f64 = rc_allreduce, v.alltoallv = rc_alltoallv, v.allgatherv = rc_allgatherv;
  v.stream_ordered = 1;
  return wc_ctx_set_comm(ctx, &v);
}

extern "C" int wc_comm_rccl_destroy(wc_ctx *ctx) {
  wc_dev_guard dg_(ctx);
  if (!ctx) return WC_ERR_ARG;
  if (!ctx->rccl) return WC_OK;
  RcclComm *c = (RcclComm *)ctx->rccl;
  (void)hipStreamSynchronize(ctx->stream);
  (void)wc_ctx_set_comm(ctx, nullptr);
  (void)c->api->CommDestroy(c->comm);
  delete c;
  ctx->rccl = nullptr;
  return WC_OK;
}

// Measurement helper (bench.py: multi_gpu_model): `reps` in-place all-reduces of `count` doubles through the ctx's communicator,
// enqueued back to back on the ctx stream between two HIP events -> microseconds per all-reduce, enqueue to completion.  With the
// in-library RCCL communicator of a world of one this is RCCL's floor for the payload (no wire); with N ranks every rank must call it.
extern "C" int wc_comm_allreduce_probe(wc_ctx *ctx, uint64_t count, int reps, double *h_us) {
  wc_dev_guard dg_(ctx);
  if (!ctx || !h_us || reps < 1 || count == 0) return wc_fail(ctx, WC_ERR_ARG, "%s: null or out-of-range argument", __func__);
  if (!ctx->have_comm || !ctx->comm.allreduce_f64) return wc_fail(ctx, WC_ERR_ARG, "%s: no communicator installed", __func__);
  WC_TRY(wc_ensure(ctx, ctx->b_route[0], (size_t)count * 8));
  WC_HIP(ctx, hipMemsetAsync(ctx->b_route[0].p, 0, (size_t)count * 8, ctx->stream));
  for (int warm = 0; warm < 3; ++warm)
    if (ctx->comm.allreduce_f64(ctx->comm.user, (double *)ctx->b_route[0].p, count) != 0) return wc_fail(ctx, WC_ERR_HIP, "all-reduce failed");
  WC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  WC_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int r = 0; r < reps; ++r)
    if (ctx->comm.allreduce_f64(ctx->comm.user, (double *)ctx->b_route[0].p, count) != 0) return wc_fail(ctx, WC_ERR_HIP, "all-reduce failed");
  WC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  WC_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  WC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *h_us = (double)ms * 1e3 / reps;
  return WC_OK;
}
