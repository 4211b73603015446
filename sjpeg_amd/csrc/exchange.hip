// exchange.hip -- the exchange step of the multi-device batch path behind include/sjpeg_hip.h:
// communicators and sjpeg_hip_gather_streams() on RCCL (xGMI inside a node).  The reference is
// single-threaded and single-device: no counterpart there (BASELINE.json config #4).
// RCCL is resolved at run time with dlopen / dlsym -- the library carries no link dependency on the
// 570 MB librccl, and a process that never gathers never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

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

struct Rccl {
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

// row of this rank: {packed bytes, number of frames, size of frame 0, 1, ... (0 beyond nframes_local)}
__global__ void gather_row_kernel(const unsigned long long* offsets, const unsigned long long* sizes, int nframes_local,
                                  int per_max, unsigned long long* row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { row[0] = offsets[nframes_local]; row[1] = static_cast<unsigned long long>(nframes_local); }
  if (i < per_max) row[2 + i] = (i < nframes_local) ? sizes[i] : 0ull;
}

}  // namespace

struct sjpeg_hip_comm {
  ncclComm_t comm = nullptr;
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
  if (c->owned && c->comm != nullptr && g_rccl.handle != nullptr) (void)g_rccl.CommDestroy(c->comm);
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
  const Rccl* r = rccl();
  if (r == nullptr) return xfail(SJPEG_HIP_ERUNTIME, g_rccl.why);
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
  const Rccl* r = rccl();
  if (r == nullptr) return xfail(SJPEG_HIP_ERUNTIME, g_rccl.why);
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
