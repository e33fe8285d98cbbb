// RCCL transport of the row-sharded step (include/drx.h "drx_comm_*"; SURVEY.md §8b: "no global state except an optional
// drx_comm_init/destroy for RCCL").  One communicator per rank and a HIP stream of its own: every exchange of a step is an
// all-to-all(v) = one ncclGroup of ncclSend / ncclRecv pairs (point-to-point xGMI: all seven links of a GPU at once), enqueued from C —
// the r05 step spent 0.35 ms of Python per 0.45 ms device step in torch.distributed calls, and the chunked schedule of r06 issues 2 + 2 C
// exchanges per step where r05 issued 4.  Ordering with the training / run-ahead streams is by events: an exchange waits for what
// `after_stream` has queued, and a stream waits for an exchange by its TICKET.
//
// librccl is opened at run time (dlopen), so libdrx.so loads — and everything single-GPU works — on a box without it; the header is
// only needed for the types.  No reference equivalent: DRecPy is single-process (recommender_abc.py:16 is its only device line).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include "drx.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;               // the function table: resolved once, read-only afterwards
char g_last_error[256] = "";

int load_rccl() {
  if (g_rccl.lib) return DRX_OK;
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) { snprintf(g_last_error, sizeof g_last_error, "librccl not found: %s", dlerror()); return DRX_ECOMM; }
  Rccl r;
  r.lib = h;
#define SYM(field, name)                                                                   \
  *(void **)(&r.field) = dlsym(h, name);                                                   \
  if (!r.field) { snprintf(g_last_error, sizeof g_last_error, "librccl lacks %s", name); dlclose(h); return DRX_ECOMM; }
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl = r;
  return DRX_OK;
}

constexpr int kRing = 256;          // events in flight: far more than the exchanges of the few steps the host runs ahead

}  // namespace

struct DrxComm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done[kRing];           // done[t % kRing]: recorded on `stream` behind exchange t
  hipEvent_t before[kRing];         // recorded on the caller's stream in front of exchange t
  int64_t next = 0;
};

#define NCCL_TRY(call)                                                                                               \
  do {                                                                                                               \
    const ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess) {                                                                                         \
      snprintf(g_last_error, sizeof g_last_error, "%s: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
      return DRX_ECOMM;                                                                                              \
    }                                                                                                                \
  } while (0)
#define HIP_TRY(call)                                                                                                \
  do {                                                                                                               \
    const hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) { snprintf(g_last_error, sizeof g_last_error, "%s: %s", #call, hipGetErrorString(e_)); return -(int)e_; } \
  } while (0)

extern "C" {

const char *drx_comm_last_error(void) { return g_last_error; }

int drx_comm_unique_id(void *id128) {
  if (!id128) return DRX_EINVAL;
  static_assert(sizeof(ncclUniqueId) == DRX_COMM_ID_BYTES, "include/drx.h DRX_COMM_ID_BYTES");
  const int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return DRX_OK;
}

int drx_comm_create(const void *id128, int32_t world, int32_t rank, DrxComm **out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return DRX_EINVAL;
  const int rc = load_rccl();
  if (rc) return rc;
  DrxComm *c = new DrxComm();
  c->world = world; c->rank = rank;
  HIP_TRY(hipGetDevice(&c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  NCCL_TRY(g_rccl.CommInitRank(&c->comm, world, id, rank));
  int lo = 0, hi = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));                      // (hi = the numerically lowest = highest priority)
  HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
  for (int i = 0; i < kRing; ++i) {
    // `done` is waited for by kernels that read what a PEER's RCCL kernel wrote into this GPU's memory: a system-scope event;
    // `before` orders two streams of this device
    HIP_TRY(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->before[i], hipEventDisableTiming | hipEventDisableSystemFence));
  }
  *out = c;
  return DRX_OK;
}

int drx_comm_destroy(DrxComm *c) {
  if (!c) return DRX_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  for (int i = 0; i < kRing; ++i) {
    if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->before[i]) (void)hipEventDestroy(c->before[i]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return DRX_OK;
}

void *drx_comm_stream(DrxComm *c) { return c ? (void *)c->stream : nullptr; }

int64_t drx_comm_alltoallv(DrxComm *c, const void *send, const int64_t *send_off, const int64_t *send_bytes, void *recv,
                           const int64_t *recv_off, const int64_t *recv_bytes, void *after_stream) {
  if (!c || !send_off || !send_bytes || !recv_off || !recv_bytes) return DRX_EINVAL;
  const int64_t t = c->next;
  if (after_stream != (void *)c->stream) {
    HIP_TRY(hipEventRecord(c->before[t % kRing], (hipStream_t)after_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->before[t % kRing], 0));
  }
  bool any = false;
  for (int p = 0; p < c->world; ++p) any = any || send_bytes[p] > 0 || recv_bytes[p] > 0;
  if (any) {
    if (!send || !recv) return DRX_EINVAL;
    NCCL_TRY(g_rccl.GroupStart());
    for (int p = 0; p < c->world; ++p) {
      if (send_bytes[p] < 0 || recv_bytes[p] < 0) { (void)g_rccl.GroupEnd(); return DRX_EINVAL; }
      if (send_bytes[p] > 0)
        NCCL_TRY(g_rccl.Send((const char *)send + send_off[p], (size_t)send_bytes[p], ncclInt8, p, c->comm, c->stream));
      if (recv_bytes[p] > 0)
        NCCL_TRY(g_rccl.Recv((char *)recv + recv_off[p], (size_t)recv_bytes[p], ncclInt8, p, c->comm, c->stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
  }
  HIP_TRY(hipEventRecord(c->done[t % kRing], c->stream));
  c->next = t + 1;
  return t;
}

int drx_comm_wait(DrxComm *c, int64_t ticket, void *stream) {
  if (!c || ticket < 0 || ticket >= c->next) return DRX_EINVAL;
  // (a slot that a later exchange has re-recorded since: waiting for the later one covers the earlier — the stream is in order)
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->done[ticket % kRing], 0));
  return DRX_OK;
}

}  // extern "C"
