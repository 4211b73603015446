// exchange.hip -- the exchange step of the multi-device batch path behind include/sjpeg_hip.h:
// communicators and sjpeg_hip_gather_streams() on RCCL (xGMI inside a node).  The reference is
// single-threaded and single-device: no counterpart there (BASELINE.json config #4).
// RCCL is resolved at run time with dlopen / dlsym -- the library carries no link dependency on the
// 570 MB librccl, and a process that never gathers never loads it.
// A communicator carries the table of transport functions it was made with: RCCL's (one process per GPU), or
// the LOCAL transport further down -- the ranks are threads of ONE process, each with its own stream (and its
// own device, or several on one), the transfers are peer copies ordered by events.  The gather functions are
// the same code for both; the local transport is also how their N > 1 paths run on a one-GPU box
// (tests/test_gpu_parity.py::test_exchange_c_abi_with_several_ranks_on_one_device).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "sjpeg_hip.h"

extern "C" __attribute__((visibility("hidden"))) void sjpeg_hip_internal_set_last_error(const char* msg);

namespace {

thread_local std::string g_xerr;
int xfail(int code, const std::string& msg) {
  g_xerr = msg;
  sjpeg_hip_internal_set_last_error(g_xerr.c_str());   // (sjpeg_hip_last_error() lives in scan_engine.hip)
  return code;
}

struct Transport {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
};

typedef Transport Rccl;
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  Rccl& r = g_rccl;
  // the copy this process already has (PyTorch brings its own), else the loader's, else ROCm's
  const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  r.handle = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
  for (int i = 0; r.handle == nullptr && i < 3; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (r.handle == nullptr) { r.why = std::string("librccl.so.1 not found: ") + dlerror(); return; }
  bool ok = true;
#define sym(NAME, FN)                                                                     \
  do {                                                                                    \
    *(FN) = reinterpret_cast<std::remove_pointer_t<decltype(FN)>>(dlsym(r.handle, NAME)); \
    if (*(FN) == nullptr) { ok = false; r.why = std::string("librccl: no symbol ") + NAME; } \
  } while (0)
  sym("ncclGetUniqueId", &r.GetUniqueId);
  sym("ncclCommInitRank", &r.CommInitRank);
  sym("ncclCommDestroy", &r.CommDestroy);
  sym("ncclCommCount", &r.CommCount);
  sym("ncclCommUserRank", &r.CommUserRank);
  sym("ncclAllGather", &r.AllGather);
  sym("ncclSend", &r.Send);
  sym("ncclRecv", &r.Recv);
  sym("ncclGroupStart", &r.GroupStart);
  sym("ncclGroupEnd", &r.GroupEnd);
  sym("ncclGetErrorString", &r.GetErrorString);
#undef sym
  if (!ok) { r.handle = nullptr; }
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, load_rccl);
  return g_rccl.handle != nullptr ? &g_rccl : nullptr;
}

#define RCCL_TRY(r, expr)                                                                        \
  do {                                                                                           \
    const ncclResult_t e_ = (expr);                                                              \
    if (e_ != ncclSuccess) return xfail(SJPEG_HIP_ERUNTIME, std::string(#expr) + ": " + (r)->GetErrorString(e_)); \
  } while (0)
#define XHIP_TRY(expr)                                                                           \
  do {                                                                                           \
    const hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return xfail(SJPEG_HIP_ERUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)


// ------------------------------------------------------------------------------------------------------
// The LOCAL transport: the ranks of a communicator are threads of this process.  Same function table as
// RCCL's, so sjpeg_hip_gather_rows / _bytes / _streams above run unchanged on it.  A transfer is a
// hipMemcpyAsync on the RECEIVER's stream (peer copy when the ranks drive different devices), ordered behind
// the sender's stream by an event the sender records, and the sender's stream is ordered behind the copy by
// an event of the receiver -- the buffer is the sender's again when its stream gets there, as with ncclSend.
// What differs from RCCL: the calls meet on the HOST (mutex + condition variable), so a send returns once its
// receive has been enqueued, not before; a peer that never shows up is a timeout error, not a hang.
// Uses: a single-process server that drives the GPUs of a node from one thread each (no RCCL in the
// process), and the N > 1 tests of the gather functions on a box with one GPU.

constexpr int kLocalTimeoutS = 60;

struct LocalGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, joined = 0, left = 0;
  std::vector<uint8_t> taken;                    // ranks that have joined
  // barrier
  int bar_n = 0;
  uint64_t bar_gen = 0;
  // all-gather: what every rank contributes this round
  struct Slot { const void* send = nullptr; hipEvent_t ready = nullptr, done = nullptr; };
  std::vector<Slot> ag;
  // mailboxes [src * world + dst]: one message in flight per pair
  struct Box { int state = 0; const void* ptr = nullptr; size_t bytes = 0; hipEvent_t ready = nullptr, done = nullptr; };   // 0 empty, 1 posted, 2 taken
  std::vector<Box> box;
  bool broken = false;                           // a rank timed out or failed: everybody gives up
};

struct LocalRank {
  std::shared_ptr<LocalGroup> g;
  std::string key;
  int rank = 0;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;       // all-gather
  std::vector<hipEvent_t> ev_send, ev_recv;               // per peer
  int depth = 0;                                           // ncclGroupStart nesting
  struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t st; };
  std::vector<Op> pending;
};

std::mutex g_local_mu;
std::map<std::string, std::weak_ptr<LocalGroup>> g_local_groups;

ncclResult_t local_destroy(ncclComm_t comm);
LocalRank* local_join(const uint8_t* id, int rank, int world, std::string* why) {
  const std::string key(reinterpret_cast<const char*>(id), SJPEG_HIP_COMM_ID_BYTES);
  std::shared_ptr<LocalGroup> g;
  {
    std::lock_guard<std::mutex> lk(g_local_mu);
    g = g_local_groups[key].lock();
    if (!g) {
      g = std::make_shared<LocalGroup>();
      g->world = world;
      g->taken.assign(world, 0);
      g->ag.resize(world);
      g->box.resize(static_cast<size_t>(world) * world);
      g_local_groups[key] = g;
    }
  }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->world != world) { *why = "the group of this id has another world size"; return nullptr; }
    if (g->taken[rank]) { *why = "rank already taken in the group of this id"; return nullptr; }
    g->taken[rank] = 1;
    ++g->joined;
  }
  LocalRank* lr = new (std::nothrow) LocalRank;
  if (lr == nullptr) {                             // (the rank is free again: nothing of this call stays behind)
    std::lock_guard<std::mutex> lk(g->mu);
    g->taken[rank] = 0;
    --g->joined;
    *why = "host allocation failed";
    return nullptr;
  }
  lr->g = g; lr->key = key; lr->rank = rank;
  lr->ev_send.assign(world, nullptr); lr->ev_recv.assign(world, nullptr);
  bool ok = hipEventCreateWithFlags(&lr->ev_ready, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&lr->ev_done, hipEventDisableTiming) == hipSuccess;
  for (int k = 0; ok && k < world; ++k) {
    ok = hipEventCreateWithFlags(&lr->ev_send[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&lr->ev_recv[k], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {
    // a communicator without its events must not be handed out (every gather would fail with an opaque HIP error
    // while the peers wait out their 60 s -- ADVICE r05): the rank leaves the way it would through destroy
    *why = "hipEventCreate failed";
    (void)hipGetLastError();
    (void)local_destroy(reinterpret_cast<ncclComm_t>(lr));
    return nullptr;
  }
  return lr;
}

// every rank of the group has got here (false: timeout or a broken group)
bool local_barrier(LocalGroup* g, std::unique_lock<std::mutex>& lk) {
  if (g->broken) return false;
  const uint64_t gen = g->bar_gen;
  if (++g->bar_n == g->world) {
    g->bar_n = 0; ++g->bar_gen;
    g->cv.notify_all();
    return true;
  }
  const bool ok = g->cv.wait_for(lk, std::chrono::seconds(kLocalTimeoutS), [&] { return g->bar_gen != gen || g->broken; });
  if (!ok || g->broken) { g->broken = true; g->cv.notify_all(); return false; }
  return true;
}

size_t nccl_type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

ncclResult_t local_all_gather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t st) {
  LocalRank* me = reinterpret_cast<LocalRank*>(comm);
  LocalGroup* g = me->g.get();
  const size_t bytes = count * nccl_type_size(type);
  if (bytes == 0 || send == nullptr || recv == nullptr) return ncclInvalidArgument;
  if (hipEventRecord(me->ev_ready, st) != hipSuccess) return ncclUnhandledCudaError;
  std::unique_lock<std::mutex> lk(g->mu);
  g->ag[me->rank].send = send; g->ag[me->rank].ready = me->ev_ready;
  if (!local_barrier(g, lk)) return ncclInternalError;          // every contribution is posted
  ncclResult_t res = ncclSuccess;
  for (int k = 0; k < g->world; ++k) {
    uint8_t* const dst = static_cast<uint8_t*>(recv) + static_cast<size_t>(k) * bytes;
    if (k == me->rank) {
      if (dst != send && hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) res = ncclUnhandledCudaError;
      continue;
    }
    if (hipStreamWaitEvent(st, g->ag[k].ready, 0) != hipSuccess ||
        hipMemcpyAsync(dst, g->ag[k].send, bytes, hipMemcpyDefault, st) != hipSuccess) res = ncclUnhandledCudaError;
  }
  if (hipEventRecord(me->ev_done, st) != hipSuccess) res = ncclUnhandledCudaError;
  g->ag[me->rank].done = me->ev_done;
  if (!local_barrier(g, lk)) return ncclInternalError;          // every rank has read every contribution (in stream order)
  for (int k = 0; k < g->world; ++k) {
    if (k != me->rank && hipStreamWaitEvent(st, g->ag[k].done, 0) != hipSuccess) res = ncclUnhandledCudaError;
  }
  if (!local_barrier(g, lk)) return ncclInternalError;          // nobody re-records an event somebody still has to wait for
  return res;
}

// posts the sends, takes the receives, waits for the sends to be taken: the order that cannot deadlock for any
// pattern of matched pairs
ncclResult_t local_run(LocalRank* me, std::vector<LocalRank::Op>& ops) {
  LocalGroup* g = me->g.get();
  const auto limit = std::chrono::seconds(kLocalTimeoutS);
  ncclResult_t res = ncclSuccess;
  std::unique_lock<std::mutex> lk(g->mu);
  auto give_up = [&]() { g->broken = true; g->cv.notify_all(); return ncclInternalError; };
  for (auto& op : ops) {
    if (!op.send) continue;
    LocalGroup::Box& b = g->box[static_cast<size_t>(me->rank) * g->world + op.peer];
    if (!g->cv.wait_for(lk, limit, [&] { return b.state == 0 || g->broken; }) || g->broken) return give_up();
    if (hipEventRecord(me->ev_send[op.peer], op.st) != hipSuccess) res = ncclUnhandledCudaError;
    b.ptr = op.buf; b.bytes = op.bytes; b.ready = me->ev_send[op.peer]; b.state = 1;
    g->cv.notify_all();
  }
  for (auto& op : ops) {
    if (op.send) continue;
    LocalGroup::Box& b = g->box[static_cast<size_t>(op.peer) * g->world + me->rank];
    if (!g->cv.wait_for(lk, limit, [&] { return b.state == 1 || g->broken; }) || g->broken) return give_up();
    if (b.bytes != op.bytes) { res = ncclInvalidArgument; }       // (mismatched counts: RCCL would hang or corrupt)
    else if (hipStreamWaitEvent(op.st, b.ready, 0) != hipSuccess ||
             hipMemcpyAsync(op.buf, b.ptr, op.bytes, hipMemcpyDefault, op.st) != hipSuccess) res = ncclUnhandledCudaError;
    if (hipEventRecord(me->ev_recv[op.peer], op.st) != hipSuccess) res = ncclUnhandledCudaError;
    b.done = me->ev_recv[op.peer]; b.state = 2;
    g->cv.notify_all();
  }
  for (auto& op : ops) {
    if (!op.send) continue;
    LocalGroup::Box& b = g->box[static_cast<size_t>(me->rank) * g->world + op.peer];
    if (!g->cv.wait_for(lk, limit, [&] { return b.state == 2 || g->broken; }) || g->broken) return give_up();
    if (hipStreamWaitEvent(op.st, b.done, 0) != hipSuccess) res = ncclUnhandledCudaError;
    b.state = 0;
    g->cv.notify_all();
  }
  return res;
}

// ncclGroupStart / End carry no communicator: the open group belongs to the calling thread
struct LocalPending { LocalRank* rank; LocalRank::Op op; };
thread_local int t_local_depth = 0;
thread_local std::vector<LocalPending> t_local_pending;

ncclResult_t local_p2p(bool send, void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t st) {
  LocalRank* me = reinterpret_cast<LocalRank*>(comm);
  if (peer < 0 || peer >= me->g->world || peer == me->rank || buf == nullptr) return ncclInvalidArgument;
  const LocalRank::Op op{send, buf, count * nccl_type_size(type), peer, st};
  if (t_local_depth > 0) { t_local_pending.push_back({me, op}); return ncclSuccess; }
  std::vector<LocalRank::Op> one{op};
  return local_run(me, one);
}

ncclResult_t local_send(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t st) {
  return local_p2p(true, const_cast<void*>(buf), count, type, peer, comm, st);
}
ncclResult_t local_recv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t st) {
  return local_p2p(false, buf, count, type, peer, comm, st);
}
ncclResult_t local_group_start() { ++t_local_depth; return ncclSuccess; }
ncclResult_t local_group_end() {
  if (t_local_depth <= 0) return ncclInvalidUsage;
  if (--t_local_depth > 0) return ncclSuccess;
  ncclResult_t res = ncclSuccess;
  // (one communicator per group in this library; several would be run one after the other)
  while (!t_local_pending.empty()) {
    LocalRank* const who = t_local_pending.front().rank;
    std::vector<LocalRank::Op> ops;
    std::vector<LocalPending> rest;
    for (auto& p : t_local_pending) { if (p.rank == who) ops.push_back(p.op); else rest.push_back(p); }
    t_local_pending.swap(rest);
    const ncclResult_t e = local_run(who, ops);
    if (e != ncclSuccess && res == ncclSuccess) res = e;
  }
  return res;
}
ncclResult_t local_destroy(ncclComm_t comm) {
  LocalRank* me = reinterpret_cast<LocalRank*>(comm);
  if (me->ev_ready) (void)hipEventDestroy(me->ev_ready);
  if (me->ev_done) (void)hipEventDestroy(me->ev_done);
  for (hipEvent_t e : me->ev_send) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : me->ev_recv) if (e) (void)hipEventDestroy(e);
  bool last;
  {
    std::lock_guard<std::mutex> lk(me->g->mu);
    me->g->taken[me->rank] = 0;                  // (the rank may join again: a new communicator on the same id)
    last = --me->g->joined == 0;
    // (a group everybody has left is whole again: ranks that re-join its id before the last shared_ptr is gone
    // must not inherit the failure of the communicators that are destroyed)
    if (last) {
      me->g->broken = false; me->g->bar_n = 0;
      for (auto& b : me->g->box) b.state = 0;
    }
  }
  const std::string key = me->key;
  delete me;                                     // the last rank's shared_ptr frees the group; the registry holds a weak one
  {
    // the registry forgets ids whose group is gone (128 bytes of key per id otherwise, for the life of the process)
    std::lock_guard<std::mutex> lk(g_local_mu);
    auto it = g_local_groups.find(key);
    if (it != g_local_groups.end() && it->second.expired()) g_local_groups.erase(it);
  }
  return ncclSuccess;
}
ncclResult_t local_count(const ncclComm_t comm, int* n) { *n = reinterpret_cast<const LocalRank*>(comm)->g->world; return ncclSuccess; }
ncclResult_t local_user_rank(const ncclComm_t comm, int* r) { *r = reinterpret_cast<const LocalRank*>(comm)->rank; return ncclSuccess; }
const char* local_error_string(ncclResult_t e) {
  switch (e) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "local transport: a HIP call failed";
    case ncclInvalidArgument: return "local transport: invalid argument (NULL buffer, bad peer, or a receive whose length differs from the send's)";
    case ncclInvalidUsage: return "local transport: group end without start";
    default: return "local transport: a rank of the group did not arrive within 60 s (or failed): the group is broken";
  }
}

const Transport* local_transport() {
  static const Transport t = [] {
    Transport x;
    x.CommDestroy = local_destroy; x.CommCount = local_count; x.CommUserRank = local_user_rank;
    x.AllGather = local_all_gather; x.Send = local_send; x.Recv = local_recv;
    x.GroupStart = local_group_start; x.GroupEnd = local_group_end; x.GetErrorString = local_error_string;
    return x;
  }();
  return &t;
}

// row of this rank: {packed bytes, number of frames, size of frame 0, 1, ... (0 beyond nframes_local)}
__global__ void gather_row_kernel(const unsigned long long* offsets, const unsigned long long* sizes, int nframes_local,
                                  int per_max, unsigned long long* row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { row[0] = offsets[nframes_local]; row[1] = static_cast<unsigned long long>(nframes_local); }
  if (i < per_max) row[2 + i] = (i < nframes_local) ? sizes[i] : 0ull;
}

}  // namespace

struct sjpeg_hip_comm {
  const Transport* api = nullptr;     // RCCL's functions, or the local transport's
  ncclComm_t comm = nullptr;          // (local transport: a LocalRank*)
  int rank = 0, world = 1;
  bool owned = false;
  uint64_t* h_pinned = nullptr;       // the host read of the rows goes through pinned memory
  size_t h_cap = 0;
};

extern "C" {

int sjpeg_hip_comm_unique_id(uint8_t id[SJPEG_HIP_COMM_ID_BYTES]) {
  if (id == nullptr) return xfail(SJPEG_HIP_EINVAL, "id == NULL");
  const Rccl* r = rccl();
  if (r == nullptr) return xfail(SJPEG_HIP_ERUNTIME, g_rccl.why);
  static_assert(sizeof(ncclUniqueId) == SJPEG_HIP_COMM_ID_BYTES, "unique id size");
  ncclUniqueId u;
  RCCL_TRY(r, r->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

int sjpeg_hip_comm_create(const uint8_t id[SJPEG_HIP_COMM_ID_BYTES], int rank, int world, sjpeg_hip_comm** comm) {
  if (comm == nullptr) return xfail(SJPEG_HIP_EINVAL, "comm == NULL");
  *comm = nullptr;
  if (id == nullptr || world <= 0 || rank < 0 || rank >= world) return xfail(SJPEG_HIP_EINVAL, "bad id / rank / world");
  const Rccl* r = rccl();
  if (r == nullptr) return xfail(SJPEG_HIP_ERUNTIME, g_rccl.why);
  sjpeg_hip_comm* c = new (std::nothrow) sjpeg_hip_comm;
  if (c == nullptr) return xfail(SJPEG_HIP_ENOMEM, "host allocation failed");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  const ncclResult_t e = r->CommInitRank(&c->comm, world, u, rank);
  if (e != ncclSuccess) {
    delete c;
    return xfail(SJPEG_HIP_ERUNTIME, std::string("ncclCommInitRank: ") + r->GetErrorString(e));
  }
  c->api = r; c->rank = rank; c->world = world; c->owned = true;
  *comm = c;
  return 0;
}

int sjpeg_hip_comm_create_local(const uint8_t id[SJPEG_HIP_COMM_ID_BYTES], int rank, int world, sjpeg_hip_comm** comm) {
  if (comm == nullptr) return xfail(SJPEG_HIP_EINVAL, "comm == NULL");
  *comm = nullptr;
  if (id == nullptr || world <= 0 || rank < 0 || rank >= world) return xfail(SJPEG_HIP_EINVAL, "bad id / rank / world");
  sjpeg_hip_comm* c = new (std::nothrow) sjpeg_hip_comm;
  if (c == nullptr) return xfail(SJPEG_HIP_ENOMEM, "host allocation failed");
  std::string why;
  LocalRank* lr = local_join(id, rank, world, &why);
  if (lr == nullptr) { delete c; return xfail(SJPEG_HIP_ERUNTIME, "sjpeg_hip_comm_create_local: " + why); }
  c->api = local_transport();
  c->comm = reinterpret_cast<ncclComm_t>(lr);
  c->rank = rank; c->world = world; c->owned = true;
  *comm = c;
  return 0;
}

int sjpeg_hip_comm_adopt(void* nccl_comm, sjpeg_hip_comm** comm) {
  if (comm == nullptr) return xfail(SJPEG_HIP_EINVAL, "comm == NULL");
  *comm = nullptr;
  if (nccl_comm == nullptr) return xfail(SJPEG_HIP_EINVAL, "nccl_comm == NULL");
  const Rccl* r = rccl();
  if (r == nullptr) return xfail(SJPEG_HIP_ERUNTIME, g_rccl.why);
  sjpeg_hip_comm* c = new (std::nothrow) sjpeg_hip_comm;
  if (c == nullptr) return xfail(SJPEG_HIP_ENOMEM, "host allocation failed");
  c->api = r;
  c->comm = static_cast<ncclComm_t>(nccl_comm);
  if (r->CommCount(c->comm, &c->world) != ncclSuccess || r->CommUserRank(c->comm, &c->rank) != ncclSuccess) {
    delete c;
    return xfail(SJPEG_HIP_ERUNTIME, "ncclCommCount / ncclCommUserRank failed on the adopted communicator");
  }
  *comm = c;
  return 0;
}

void sjpeg_hip_comm_destroy(sjpeg_hip_comm* c) {
  if (c == nullptr) return;
  if (c->h_pinned != nullptr) (void)hipHostFree(c->h_pinned);
  if (c->owned && c->comm != nullptr && c->api != nullptr) (void)c->api->CommDestroy(c->comm);
  delete c;
}

int sjpeg_hip_comm_rank(const sjpeg_hip_comm* c) { return c ? c->rank : -1; }
int sjpeg_hip_comm_world(const sjpeg_hip_comm* c) { return c ? c->world : 0; }

int sjpeg_hip_gather_rows(sjpeg_hip_comm* c, const uint64_t* d_offsets, const uint64_t* d_sizes, int nframes_local,
                          int per_max, uint64_t* d_rows, uint64_t* h_rows, uint64_t* h_rank_offsets, void* stream) {
  if (c == nullptr || d_offsets == nullptr || d_rows == nullptr || h_rows == nullptr || h_rank_offsets == nullptr) {
    return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_rows: NULL argument");
  }
  if (nframes_local < 0 || per_max <= 0 || nframes_local > per_max || per_max > (1 << 20) ||
      (nframes_local > 0 && d_sizes == nullptr)) {
    return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_rows: bad frame counts");
  }
  const Transport* r = c->api;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t row = static_cast<size_t>(per_max) + 2;
  const size_t nrows = static_cast<size_t>(c->world) * row;
  if (c->h_cap < nrows) {
    if (c->h_pinned != nullptr) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr; c->h_cap = 0;
    XHIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), nrows * sizeof(uint64_t), hipHostMallocDefault));
    c->h_cap = nrows;
  }
  // this rank's row (in the scratch row behind the gathered ones), the all-gather, the one host read
  unsigned long long* const mine = reinterpret_cast<unsigned long long*>(d_rows) + nrows;
  hipLaunchKernelGGL(gather_row_kernel, dim3((per_max + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<const unsigned long long*>(d_offsets), reinterpret_cast<const unsigned long long*>(d_sizes),
                     nframes_local, per_max, mine);
  XHIP_TRY(hipGetLastError());
  RCCL_TRY(r, r->AllGather(mine, d_rows, row, ncclUint64, c->comm, st));
  XHIP_TRY(hipMemcpyAsync(c->h_pinned, d_rows, nrows * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  XHIP_TRY(hipStreamSynchronize(st));
  memcpy(h_rows, c->h_pinned, nrows * sizeof(uint64_t));
  uint64_t total = 0;
  bool lost = false;
  for (int k = 0; k < c->world; ++k) {
    const uint64_t* rk = h_rows + static_cast<size_t>(k) * row;
    h_rank_offsets[k] = total;
    total += rk[0];
    // a size of 0 among a rank's frames = a frame that did not fit its slot; a packed total that is not
    // the sum of its (16-aligned) frames = the packed buffer was too small for them
    // (bit 63 of a rank's byte count: sjpeg_hip_compact_streams could not fit the frames into its d_packed)
    if ((rk[0] >> 63) != 0 || rk[1] > static_cast<uint64_t>(per_max)) { lost = true; continue; }
    uint64_t sum16 = 0;
    for (uint64_t f = 0; f < rk[1]; ++f) {
      if (rk[2 + f] == 0) lost = true;
      sum16 += (rk[2 + f] + 15u) & ~uint64_t(15);
    }
    if (sum16 != rk[0]) lost = true;
  }
  h_rank_offsets[c->world] = total;
  if (lost) {
    // (the offsets are still usable as sizes: the flag is not part of them)
    for (int k = 0; k <= c->world; ++k) h_rank_offsets[k] &= ~(uint64_t(1) << 63);
    return xfail(SJPEG_HIP_ECAPACITY, "sjpeg_hip_gather_rows: a frame of size 0 (it did not fit its output slot) or a packed "
                                      "buffer that was too small on some rank -- nothing to send");
  }
  return 0;
}

int sjpeg_hip_gather_bytes(sjpeg_hip_comm* c, int root, const void* d_packed, int per_max, const uint64_t* h_rows,
                           const uint64_t* h_rank_offsets, void* d_gathered, size_t gathered_capacity, void* stream) {
  if (c == nullptr || h_rows == nullptr || h_rank_offsets == nullptr) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_bytes: NULL argument");
  if (root < 0 || root >= c->world || per_max <= 0) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_bytes: bad root / per_max");
  const Transport* r = c->api;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t row = static_cast<size_t>(per_max) + 2;
  const uint64_t my_bytes = h_rows[static_cast<size_t>(c->rank) * row];
  if (c->rank != root) {
    if (my_bytes > 0) {
      if (d_packed == nullptr) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_bytes: d_packed == NULL");
      RCCL_TRY(r, r->Send(d_packed, my_bytes, ncclUint8, root, c->comm, st));
    }
    return 0;
  }
  // (the root's buffer is the root's business: the other ranks have sent by now, so every byte is
  // received -- what lies behind the capacity is an argument error of the root alone, reported
  // BEFORE the exchange by sjpeg_hip_gather_streams, which knows the capacity on every rank)
  const uint64_t total = h_rank_offsets[c->world];
  if (total > 0 && d_gathered == nullptr) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_bytes: d_gathered == NULL on the root");
  if (total > gathered_capacity) {
    return xfail(SJPEG_HIP_ECAPACITY, "sjpeg_hip_gather_bytes: " + std::to_string(total) + " bytes to gather, capacity " +
                                          std::to_string(gathered_capacity) + " (size the buffer from h_rank_offsets[world])");
  }
  uint8_t* const dst = static_cast<uint8_t*>(d_gathered);
  if (my_bytes > 0) {
    if (d_packed == nullptr) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_bytes: d_packed == NULL");
    // (a root that coded its frames straight into their place -- packed output into d_gathered + its offset,
    // offset 0 for root 0 -- has nothing to copy)
    if (static_cast<const uint8_t*>(d_packed) != dst + h_rank_offsets[root]) {
      XHIP_TRY(hipMemcpyAsync(dst + h_rank_offsets[root], d_packed, my_bytes, hipMemcpyDeviceToDevice, st));
    }
  }
  RCCL_TRY(r, r->GroupStart());
  ncclResult_t first_bad = ncclSuccess;
  for (int k = 0; k < c->world; ++k) {
    const uint64_t n = h_rows[static_cast<size_t>(k) * row];
    if (k == root || n == 0) continue;
    const ncclResult_t e = r->Recv(dst + h_rank_offsets[k], n, ncclUint8, k, c->comm, st);
    if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
  }
  RCCL_TRY(r, r->GroupEnd());
  if (first_bad != ncclSuccess) return xfail(SJPEG_HIP_ERUNTIME, std::string("ncclRecv: ") + r->GetErrorString(first_bad));
  return 0;
}

int sjpeg_hip_gather_streams(sjpeg_hip_comm* c, int root, const void* d_packed, const uint64_t* d_offsets,
                             const uint64_t* d_sizes, int nframes_local, int per_max, uint64_t* d_rows,
                             void* d_gathered, size_t gathered_capacity, uint64_t* h_rows,
                             uint64_t* h_rank_offsets, void* stream) {
  if (c == nullptr) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_streams: comm == NULL");
  if (root < 0 || root >= c->world) return xfail(SJPEG_HIP_EINVAL, "sjpeg_hip_gather_streams: bad root");
  const int rc = sjpeg_hip_gather_rows(c, d_offsets, d_sizes, nframes_local, per_max, d_rows, h_rows, h_rank_offsets, stream);
  if (rc != 0) return rc;
  if (h_rank_offsets[c->world] > gathered_capacity) {        // the same verdict on every rank: nobody sends
    return xfail(SJPEG_HIP_ECAPACITY, "sjpeg_hip_gather_streams: " + std::to_string(h_rank_offsets[c->world]) +
                                          " bytes to gather, capacity " + std::to_string(gathered_capacity) + " -- nothing was sent");
  }
  return sjpeg_hip_gather_bytes(c, root, d_packed, per_max, h_rows, h_rank_offsets, d_gathered, gathered_capacity, stream);
}

}  // extern "C"
