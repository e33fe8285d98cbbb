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
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
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

// One exchange as the caller posted it (DRX_COMM_THREAD: the issuing thread takes it from here)
struct Request {
  const void *send;
  void *recv;
  int64_t so[DRX_MAX_WORLD], sb[DRX_MAX_WORLD], ro[DRX_MAX_WORLD], rb[DRX_MAX_WORLD];
  bool ordered;                     // wait for before[t] first
};

struct DrxComm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done[kRing];           // done[t % kRing]: recorded on `stream` behind exchange t
  hipEvent_t before[kRing];         // recorded on the caller's stream in front of exchange t
  int64_t next = 0;
  // DRX_COMM_THREAD: the nccl* calls of an exchange (about 25 us of host time for one send / recv pair, more with 7 peers) are made
  // by a thread of the communicator's own, so that the caller's thread goes on queueing kernels meanwhile — the chunked schedule
  // issues 2 + 3 C exchanges per step and was bound by its ONE issuing thread (profiles/r06d_host_profile_*).  The caller posts a
  // request (and records before[t] itself: events are recorded in the caller's program order); drx_comm_wait spins until the thread
  // has recorded done[t] — in steady state it already has.
  bool threaded = false;
  Request *ring = nullptr;          // [kRing]
  std::atomic<int64_t> posted{0}, issued{0};
  std::atomic<int> failed{0};
  std::atomic<bool> stop{false};
  std::thread worker;
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

static int issue(DrxComm *c, const Request &q, int64_t t) {
  if (q.ordered) HIP_TRY(hipStreamWaitEvent(c->stream, c->before[t % kRing], 0));
  bool any = false;
  for (int p = 0; p < c->world; ++p) any = any || q.sb[p] > 0 || q.rb[p] > 0;
  if (any) {
    NCCL_TRY(g_rccl.GroupStart());
    for (int p = 0; p < c->world; ++p) {
      if (q.sb[p] > 0) NCCL_TRY(g_rccl.Send((const char *)q.send + q.so[p], (size_t)q.sb[p], ncclInt8, p, c->comm, c->stream));
      if (q.rb[p] > 0) NCCL_TRY(g_rccl.Recv((char *)q.recv + q.ro[p], (size_t)q.rb[p], ncclInt8, p, c->comm, c->stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
  }
  HIP_TRY(hipEventRecord(c->done[t % kRing], c->stream));
  return DRX_OK;
}

static void comm_thread(DrxComm *c) {
  (void)hipSetDevice(c->device);
  int idle = 0;
  while (!c->stop.load(std::memory_order_acquire)) {
    const int64_t t = c->issued.load(std::memory_order_relaxed);
    if (t < c->posted.load(std::memory_order_acquire)) {
      if (!c->failed.load(std::memory_order_relaxed)) {
        const int rc = issue(c, c->ring[t % kRing], t);
        if (rc) c->failed.store(rc, std::memory_order_release);
      }
      c->issued.store(t + 1, std::memory_order_release);
      idle = 0;
    } else if (++idle > 200000) {
      std::this_thread::sleep_for(std::chrono::microseconds(200));      // (nothing for a long while: evaluation, set-up — stop burning the core)
    } else if (idle > 4000) {
      std::this_thread::yield();                  // (between steps: give the core away; inside a step the next request is microseconds off)
    }
  }
}

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

int drx_comm_create(const void *id128, int32_t world, int32_t rank, uint32_t flags, DrxComm **out) {
  if (!id128 || !out || world < 1 || world > DRX_MAX_WORLD || rank < 0 || rank >= world) return DRX_EINVAL;
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
  if (flags & DRX_COMM_THREAD) {
    c->ring = new Request[kRing];
    c->threaded = true;
    c->worker = std::thread(comm_thread, c);
  }
  *out = c;
  return DRX_OK;
}

int drx_comm_destroy(DrxComm *c) {
  if (!c) return DRX_OK;
  if (c->threaded) {
    while (c->issued.load(std::memory_order_acquire) < c->posted.load(std::memory_order_acquire)) std::this_thread::yield();
    c->stop.store(true, std::memory_order_release);
    c->worker.join();
    delete[] c->ring;
  }
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
  for (int p = 0; p < c->world; ++p) {
    if (send_bytes[p] < 0 || recv_bytes[p] < 0) return DRX_EINVAL;
    if ((send_bytes[p] > 0 && !send) || (recv_bytes[p] > 0 && !recv)) return DRX_EINVAL;
  }
  const bool ordered = after_stream != (void *)c->stream;
  if (c->threaded) {
    if (c->failed.load(std::memory_order_acquire)) return c->failed.load();
    // (slot t % kRing — request and events — is free once exchange t - kRing has been issued; the ring never fills in practice)
    while (t - c->issued.load(std::memory_order_acquire) >= kRing - 1) std::this_thread::yield();
  }
  if (ordered) HIP_TRY(hipEventRecord(c->before[t % kRing], (hipStream_t)after_stream));
  Request local;
  Request &q = c->threaded ? c->ring[t % kRing] : local;
  q.send = send; q.recv = recv; q.ordered = ordered;
  for (int p = 0; p < c->world; ++p) { q.so[p] = send_off[p]; q.sb[p] = send_bytes[p]; q.ro[p] = recv_off[p]; q.rb[p] = recv_bytes[p]; }
  c->next = t + 1;
  if (c->threaded) {
    c->posted.store(t + 1, std::memory_order_release);
    return t;
  }
  const int rc = issue(c, q, t);
  return rc ? rc : t;
}

int drx_comm_wait(DrxComm *c, int64_t ticket, void *stream) {
  if (!c || ticket < 0 || ticket >= c->next) return DRX_EINVAL;
  if (c->threaded) {                 // done[ticket] must have been RECORDED before a stream can be told to wait for it
    while (c->issued.load(std::memory_order_acquire) <= ticket) __builtin_ia32_pause();      // (microseconds: the thread is mid-issue)
    if (c->failed.load(std::memory_order_acquire)) return c->failed.load();
  }
  // (a slot that a later exchange has re-recorded since: waiting for the later one covers the earlier — the stream is in order)
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->done[ticket % kRing], 0));
  return DRX_OK;
}

}  // extern "C"
