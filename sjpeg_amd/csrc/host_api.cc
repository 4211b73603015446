// host_api.cc -- the sjpeg-compatible public API (include/sjpeg.h) on top of the HIP scan
// engine (include/sjpeg_hip.h).
//
// Mirrors the reference's host-side sequencing (/root/reference/src/api.cc:32-304,
// src/enc.cc:391-448, src/encoders.cc:145-152,546-568): argument checks, parameter ->
// method mapping, quantizer finalisation, header emission, then ONE call into the device
// for everything the reference does per MCU.  There is no CPU implementation of the hot
// path in this library: without a gfx950 device every encode fails.
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <math.h>
#include <stdio.h>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include "jpeg_host.h"
#include "sjpeg.h"
#include "sjpeg_hip.h"

using sjpeg_host::HuffSpec;

namespace {

thread_local std::string g_api_error;

bool Fail(const std::string& msg) {
  g_api_error = msg;
  return false;
}
bool FailHip(const char* what) {
  g_api_error = std::string(what) + ": " + sjpeg_hip_last_error();
  return false;
}

// reference defaults, src/enc.cc:42-49
const float kDefaultQuality = 75.f;
const int kDefaultBias = 0x78;
const int kDefaultDeltaMaxLuma = 12;
const int kDefaultDeltaMaxChroma = 1;

struct DefaultMemory : public sjpeg::MemoryManager {
  void* Alloc(size_t size) override { return malloc(size); }
  void Free(void* const ptr) override { free(ptr); }
} g_default_memory;

// ---- sinks (reference: src/bit_writer.h:51-93, src/bit_writer.cc:30-87) ----------------

// new[]-backed sink for the uint8_t** flavours: ownership passes to the caller, who
// releases with delete[] / SjpegFreeBuffer (src/sjpeg.h:39-41).
class NewArraySink : public sjpeg::ByteSink {
 public:
  NewArraySink() : buf_(nullptr), pos_(0), cap_(0) {}
  ~NewArraySink() override { Reset(); }
  bool Commit(size_t used, size_t extra, uint8_t** data) override {
    pos_ += used;
    if (pos_ + extra > cap_) {
      size_t ncap = pos_ + extra + 256;
      if (ncap < 2 * cap_) ncap = 2 * cap_;
      uint8_t* nbuf = new (std::nothrow) uint8_t[ncap];
      if (nbuf == nullptr) return false;
      if (pos_ > 0) memcpy(nbuf, buf_, pos_);
      delete[] buf_;
      buf_ = nbuf;
      cap_ = ncap;
    }
    *data = buf_ + pos_;
    return true;
  }
  bool Finalize() override { return true; }
  void Reset() override { delete[] buf_; buf_ = nullptr; pos_ = cap_ = 0; }
  size_t Release(uint8_t** out) {
    *out = buf_;
    const size_t n = pos_;
    buf_ = nullptr; pos_ = cap_ = 0;
    return n;
  }
 private:
  uint8_t* buf_;
  size_t pos_, cap_;
};

template <class T> class ContainerSink : public sjpeg::ByteSink {
 public:
  explicit ContainerSink(T* c) : c_(c), pos_(0) {}
  bool Commit(size_t used, size_t extra, uint8_t** data) override {
    pos_ += used;
    c_->resize(pos_ + extra);
    if (c_->size() != pos_ + extra) return false;
    *data = extra ? reinterpret_cast<uint8_t*>(&(*c_)[pos_]) : nullptr;
    return true;
  }
  bool Finalize() override { c_->resize(pos_); return true; }
  void Reset() override { c_->clear(); pos_ = 0; }
 private:
  T* const c_;
  size_t pos_;
};

// ---- per-thread device context ------------------------------------------------------------

// The riskiness score table (reference data, supplied by the caller: see sjpeg_hip.h)
std::mutex g_risk_mutex;
std::vector<uint8_t> g_risk_table;
int g_risk_generation = 0;

bool ReadRiskTable(const char* path) {
  FILE* f = fopen(path, "rb");
  if (f == nullptr) return false;
  std::vector<uint8_t> t(SJPEG_HIP_RISKINESS_TABLE_SIZE + 1);
  const size_t n = fread(t.data(), 1, t.size(), f);
  fclose(f);
  if (n != SJPEG_HIP_RISKINESS_TABLE_SIZE) return false;
  t.resize(SJPEG_HIP_RISKINESS_TABLE_SIZE);
  g_risk_table.swap(t);
  ++g_risk_generation;
  return true;
}

// Where the table is looked for, in this order: what sjpeg_hip_set_riskiness_table() installed, the
// file named by SJPEG_HIP_RISKINESS_TABLE, `riskiness.bin` in the directory this shared library was
// loaded from (a deployment installs it once, next to the library: tools/extract_riskiness_table.py).
bool LoadRiskTableFromEnv() {                      // under g_risk_mutex
  if (!g_risk_table.empty()) return true;
  const char* path = getenv("SJPEG_HIP_RISKINESS_TABLE");
  if (path != nullptr && ReadRiskTable(path)) return true;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&ReadRiskTable), &info) != 0 && info.dli_fname != nullptr) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    dir = (slash == std::string::npos) ? std::string(".") : dir.substr(0, slash);
    if (ReadRiskTable((dir + "/riskiness.bin").c_str())) return true;
  }
  return false;
}

// SjpegRiskiness' arithmetic on the three sums (src/jpeg_tools.cc:212-236)
SjpegYUVMode RiskVerdict(uint64_t score_sum, uint64_t score_num, uint64_t gray_num, int width, int height,
                         float* risk) {
  const double count = static_cast<double>(score_num);
  double gray_count = static_cast<double>(gray_num);
  double total_score = (count > 0) ? score_sum / count : 0.;
  const double num_samples = (width - 1.) * (height - 1.);
  if (num_samples > 0.) gray_count /= num_samples;
  const double frac = 100. * count / (static_cast<double>(width) * height);
  if (frac < 1.) total_score = 0.;
  total_score = (total_score > 25.) ? 100. : total_score * 100. / 25.;
  if (risk != nullptr) *risk = static_cast<float>(total_score);
  return (gray_count > 0.995) ? SJPEG_YUV_400 : (total_score < 40.0) ? SJPEG_YUV_420
       : (total_score < 70.0) ? SJPEG_YUV_SHARP : SJPEG_YUV_444;
}

struct DeviceContext {
  sjpeg_hip_engine* engine = nullptr;
  void* d_risk = nullptr; int risk_generation = -1;   // device copy of the riskiness table
  uint64_t* d_sums = nullptr;
  int device = 0;
  void* d_in = nullptr;  size_t in_cap = 0;
  void* d_out = nullptr; size_t out_cap = 0;
  void* d_stats = nullptr; size_t stats_cap = 0;
  void* d_hist = nullptr; size_t hist_cap = 0;       // adaptive quantization: histogram + its sums
  void* d_planes = nullptr; size_t planes_cap = 0;   // SJPEG_YUV_SHARP: converted planes
  void* d_work = nullptr; size_t work_cap = 0;       //                  and the conversion's workspace
  uint64_t* d_size = nullptr;
  // All work of a host thread runs on the context's own non-blocking stream: concurrent encodes from
  // several threads do not serialise on the legacy NULL stream or wait for each other's kernels.
  hipStream_t stream = nullptr;
  // Pinned host memory the device writes directly (fine-grained, visible after a stream
  // synchronise): the coded size, and the whole stream of a small picture -- one wait instead of
  // two device -> host copies, which were a third of the fixed cost of a thumbnail encode.
  static constexpr size_t kMailData = 256 * 1024;     // streams up to this bound skip the copy
  static constexpr size_t kMailIn = 256 * 1024;       // pictures up to this many bytes are read in place
  uint8_t* h_mail = nullptr;                          // [0, 8): size word; [64, 64 + kMailData): stream; then pixels
  uint8_t* d_mail = nullptr;                          // the same memory as the device addresses it
  ~DeviceContext() {
    if (engine == nullptr) return;
    // (thread_local: this may run at process exit, after the HIP runtime has shut down -- every call
    // below then fails with hipErrorDeinitialized, which is ignored like any other error here)
    if (hipSetDevice(device) != hipSuccess) return;
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_stats) (void)hipFree(d_stats);
    if (d_risk) (void)hipFree(d_risk);
    if (d_sums) (void)hipFree(d_sums);
    if (d_hist) (void)hipFree(d_hist);
    if (d_planes) (void)hipFree(d_planes);
    if (d_work) (void)hipFree(d_work);
    if (d_size) (void)hipFree(d_size);
    if (h_mail) (void)hipHostFree(h_mail);
    sjpeg_hip_engine_destroy(engine);
  }
  bool ready = false;                                 // every resource below exists
  bool Init() {
    if (ready) return true;
    // (a failure leaves what was created to the destructor and is tried again by the next call:
    // nothing runs on a half-built context -- no NULL stream, no missing mailbox)
    if (engine == nullptr) {
      const char* env = getenv("SJPEG_HIP_DEVICE");
      device = env ? atoi(env) : 0;
      if (sjpeg_hip_engine_create(device, &engine) != 0) { engine = nullptr; return FailHip("sjpeg_hip_engine_create"); }
    }
    if (hipSetDevice(device) != hipSuccess) return Fail("hipSetDevice failed");
    if (stream == nullptr && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
      stream = nullptr;
      return Fail("hipStreamCreate failed");
    }
    if (d_size == nullptr && hipMalloc(reinterpret_cast<void**>(&d_size), sizeof(uint64_t)) != hipSuccess) {
      d_size = nullptr;
      return Fail("hipMalloc(size word) failed");
    }
    if (h_mail == nullptr && hipHostMalloc(reinterpret_cast<void**>(&h_mail), 64 + kMailData + kMailIn, hipHostMallocMapped) != hipSuccess) {
      h_mail = nullptr;
      return Fail("hipHostMalloc(mapped mailbox) failed");
    }
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d_mail), h_mail, 0) != hipSuccess) {
      return Fail("hipHostGetDevicePointer(mapped mailbox) failed");
    }
    ready = true;
    return true;
  }
  // stream-ordered copies (device -> host ones wait for the stream: the data is needed right away)
  bool ToHost(void* dst, const void* src, size_t n) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream) == hipSuccess &&
           hipStreamSynchronize(stream) == hipSuccess;
  }
  bool ToDevice(void* dst, const void* src, size_t n) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream) == hipSuccess;
  }
  bool ToDevice2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, stream) == hipSuccess;
  }
  // the caller's riskiness table on this device; false (with the reason) if none was supplied
  bool EnsureRiskTable() {
    std::lock_guard<std::mutex> lock(g_risk_mutex);
    if (!LoadRiskTableFromEnv()) {
      return Fail("the riskiness score table was not found: SJPEG_YUV_AUTO / SjpegCompress / SjpegRiskiness need "
                  "riskiness.bin (shipped next to libsjpeg_amd.so; a copy of the library installed elsewhere needs it "
                  "beside it, the file named by SJPEG_HIP_RISKINESS_TABLE, or sjpeg_hip_set_riskiness_table())");
    }
    if (risk_generation == g_risk_generation) return true;
    if (d_risk == nullptr && hipMalloc(&d_risk, SJPEG_HIP_RISKINESS_TABLE_SIZE) != hipSuccess) return Fail("hipMalloc failed");
    if (d_sums == nullptr && hipMalloc(reinterpret_cast<void**>(&d_sums), 3 * sizeof(uint64_t)) != hipSuccess) return Fail("hipMalloc failed");
    if (!ToDevice(d_risk, g_risk_table.data(), SJPEG_HIP_RISKINESS_TABLE_SIZE) || hipStreamSynchronize(stream) != hipSuccess) {
      return Fail("riskiness table upload failed");
    }
    risk_generation = g_risk_generation;
    return true;
  }
  // riskiness of a device-resident RGB / BGRA / RGBA picture
  bool Riskiness(const sjpeg_hip_source& dsrc, int W, int H, SjpegYUVMode* mode, float* risk) {
    if (!EnsureRiskTable()) return false;
    if (sjpeg_hip_riskiness_sums(&dsrc, W, H, 1, static_cast<const uint8_t*>(d_risk), d_sums, stream) != 0) {
      return Fail("sjpeg_hip_riskiness_sums failed");
    }
    uint64_t sums[3];
    if (!ToHost(sums, d_sums, sizeof(sums))) return Fail("riskiness read-back failed");
    *mode = RiskVerdict(sums[0], sums[1], sums[2], W, H, risk);
    return true;
  }
  // First capacity tried for a frame's stream: header + 1 byte per sample of the picture's planes
  // / 2 (0.75 bytes per pixel in 4:2:0, 1.5 in 4:4:4 -- three to six times what photographs code
  // to at q 75..90), never above the worst-case bound.  SJPEG_HIP_HOST_FIRST_CAPACITY=bound restores
  // the worst case from the first pass on.
  size_t FirstCapacity(int W, int H, int mode, size_t header, size_t bound) const {
    static const bool worst = [] {
      const char* v = getenv("SJPEG_HIP_HOST_FIRST_CAPACITY");
      return v != nullptr && strcmp(v, "bound") == 0;
    }();
    if (worst) return bound;
    const size_t px = static_cast<size_t>(W) * static_cast<size_t>(H);
    const size_t samples = mode == SJPEG_HIP_YUV444 ? 3 * px : mode == SJPEG_HIP_YUV400 ? px : px + px / 2;
    const size_t cap = header + 65536 + samples / 2;
    return cap < bound ? cap : bound;
  }
  // Gives the device memory this thread's context caches (pixels, stream, planes, engine scratch)
  // back; the next encode allocates what it needs again.
  void Trim() {
    if (engine == nullptr || hipSetDevice(device) != hipSuccess) return;
    if (stream) (void)hipStreamSynchronize(stream);
    void** const bufs[] = {&d_in, &d_out, &d_hist, &d_planes, &d_work};
    size_t* const caps[] = {&in_cap, &out_cap, &hist_cap, &planes_cap, &work_cap};
    for (int i = 0; i < 5; ++i) {
      if (*bufs[i]) (void)hipFree(*bufs[i]);
      *bufs[i] = nullptr; *caps[i] = 0;
    }
    (void)sjpeg_hip_engine_trim(engine);
  }
  // What the call in progress asked its buffers to hold (the sizes passed to Ensure, grown or not).
  size_t call_need = 0, call_out_need = 0;
  void BeginCall() { call_need = 0; call_out_need = 0; }
  // Over the cache limit (SJPEG_HIP_HOST_CACHE_BYTES, 1 GiB by default) the memory goes back -- but only
  // when the call that just finished needed less than half of what is held (a very large frame followed
  // by ordinary ones).  A steady stream of frames that need more than the limit themselves keeps its
  // buffers: trimming after every call would cost a device synchronisation + hipFree + hipMalloc per frame.
  void TrimIfOver() {
    static const size_t limit = [] {
      const char* v = getenv("SJPEG_HIP_HOST_CACHE_BYTES");
      return v != nullptr ? static_cast<size_t>(strtoull(v, nullptr, 0)) : (static_cast<size_t>(1) << 30);
    }();
    if (engine == nullptr) return;
    const size_t held = CachedBytes();
    if (held <= limit) return;
    // (the engine's scratch follows the output capacity: about 3.5 x of it)
    const size_t scratch = sjpeg_hip_engine_scratch_bytes(engine);
    const size_t needed = call_need + (scratch < 4 * call_out_need ? scratch : 4 * call_out_need);
    if (held / 2 > needed) Trim();
  }
  size_t CachedBytes() const {
    return in_cap + out_cap + hist_cap + planes_cap + work_cap + sjpeg_hip_engine_scratch_bytes(engine);
  }
  bool Ensure(void** p, size_t* cap, size_t need) {
    call_need += need;
    if (p == &d_out && need > call_out_need) call_out_need = need;
    if (need <= *cap) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    if (hipMalloc(p, need) != hipSuccess) return Fail("hipMalloc(" + std::to_string(need) + ") failed");
    *cap = need;
    return true;
  }
};
thread_local DeviceContext g_ctx;

}  // namespace

namespace sjpeg {

// The name and the friendship with EncoderParam come from the reference header; here it is
// a one-shot plan: resolved parameters -> tables + header -> device call -> sink.
struct HostSource {            // pixels in host memory, one of SJPEG_HIP_SRC_*
  int format;
  const uint8_t* plane[3];
  int stride[3];
};

struct Encoder {
  Encoder(const uint8_t* rgb, int W, int H, int stride, ByteSink* sink, MemoryManager* mem)
      : Encoder(HostSource{SJPEG_HIP_SRC_RGB, {rgb, nullptr, nullptr}, {stride, 0, 0}}, W, H, sink, mem) {}
  Encoder(const HostSource& src, int W, int H, ByteSink* sink, MemoryManager* mem)
      : src_(src), W_(W), H_(H), sink_(sink),
        mem_(mem ? mem : &g_default_memory), q_bias_(kDefaultBias), method_(4),
        yuv_mode_(SJPEG_YUV_420), passes_(1), qdelta_luma_(kDefaultDeltaMaxLuma),
        qdelta_chroma_(kDefaultDeltaMaxChroma) {
    memset(min_quant_, 1, sizeof(min_quant_));
    SetQuality(kDefaultQuality);
  }

  void SetQuality(float q) {                       // reference: src/enc.cc:100-104
    const float s = sjpeg_host::QualityToScale(q);
    sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[0], s, quant_[0]);
    sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[1], s, quant_[1]);
  }
  void SetMethod(int m) { method_ = m < 0 ? 0 : m > 8 ? 8 : m; }   // src/enc.cc:121-122
  void SetYuvMode(SjpegYUVMode m) { yuv_mode_ = m; }

  // reference: Encoder::InitFromParam, src/api.cc:145-181
  void InitFromParam(const EncoderParam& p) {
    sjpeg_host::ScaleMatrix(p.quant_[0], 100.f, quant_[0]);        // src/enc.cc:106-109
    sjpeg_host::ScaleMatrix(p.quant_[1], 100.f, quant_[1]);
    if (p.use_min_quant_) {
      sjpeg_host::MinMatrix(p.min_quant_[0], p.min_quant_tolerance_, min_quant_[0]);
      sjpeg_host::MinMatrix(p.min_quant_[1], p.min_quant_tolerance_, min_quant_[1]);
    } else {
      memset(min_quant_, 1, sizeof(min_quant_));
    }
    int method = p.Huffman_compress ? 1 : 0;
    if (p.adaptive_quantization) method += 3;
    if (p.use_trellis) method = (method == 4) ? 7 : (method == 6) ? 8 : method;
    SetMethod(method);
    q_bias_ = p.quantization_bias;
    qdelta_luma_ = p.qdelta_max_luma;
    qdelta_chroma_ = p.qdelta_max_chroma;
    meta_.iccp = p.iccp; meta_.exif = p.exif; meta_.app_markers = p.app_markers;
    meta_.xmp = p.xmp; meta_.xmp_split = p.xmp_split_point;
    passes_ = p.passes < 1 ? 1 : p.passes > 20 ? 20 : p.passes;
    if (passes_ > 1) {                                              // src/api.cc:170-176
      search_hook_ = (p.search_hook == nullptr) ? &default_hook_ : p.search_hook;
      search_ok_ = search_hook_->Setup(p);
    }
    yuv_mode_ = p.yuv_mode;
  }

  bool Run();

  bool RunImpl();

 private:
  HostSource src_;
  int W_, H_;
  ByteSink* sink_;
  MemoryManager* mem_;
  uint8_t quant_[2][64], min_quant_[2][64];
  int q_bias_, method_;
  SjpegYUVMode yuv_mode_;
  int passes_;
  int qdelta_luma_, qdelta_chroma_;
  sjpeg_host::Metadata meta_;
  SearchHook default_hook_;
  SearchHook* search_hook_ = nullptr;
  bool search_ok_ = true;
};

// The API promises "no exceptions" (include/sjpeg.h): a failed host allocation inside the pipeline
// (std::vector / std::string growth) ends the encode like any other failure.
bool Encoder::Run() {
  try {
    return RunImpl();
  } catch (const std::bad_alloc&) {
    sink_->Reset();
    return Fail("out of host memory");
  } catch (...) {
    sink_->Reset();
    return Fail("unexpected exception in the encoder");
  }
}

bool Encoder::RunImpl() {
  sink_->Reset();                                                   // src/enc.cc:90
  if (W_ > 65535 || H_ > 65535) return Fail("dimension > 65535");   // src/enc.cc:406
  if (src_.format == SJPEG_HIP_SRC_GRAY) yuv_mode_ = SJPEG_YUV_400;                 // src/encoders.cc:256-276
  else if (src_.format == SJPEG_HIP_SRC_YUV444) yuv_mode_ = SJPEG_YUV_444;          // :384-419
  else if (src_.format >= SJPEG_HIP_SRC_YUV420) yuv_mode_ = SJPEG_YUV_420;          // :281-344, :442-490
  if (yuv_mode_ != SJPEG_YUV_AUTO && yuv_mode_ != SJPEG_YUV_420 && yuv_mode_ != SJPEG_YUV_SHARP &&
      yuv_mode_ != SJPEG_YUV_444 && yuv_mode_ != SJPEG_YUV_400) {
    return Fail("unknown yuv_mode");                                 // src/encoders.cc:553-567
  }
  // method flags, reference: src/enc.cc:121-129
  const bool adaptive = method_ >= 3;
  const bool optimize = (method_ != 0) && (method_ != 3);
  const bool trellis = method_ >= 7;
  if (qdelta_luma_ < 0 || qdelta_luma_ > 12 || qdelta_chroma_ < 0 || qdelta_chroma_ > 12) {
    return Fail("qdelta_max_luma / qdelta_max_chroma must be in [0, 12]");
  }

  DeviceContext& ctx = g_ctx;
  if (!ctx.Init()) return false;
  // the caller's pixels are read by stream-ordered copies: whatever way this call ends, nothing of
  // it is still in flight when it returns
  // ... and a context that grew past its cache limit (one very large frame) gives the memory back
  struct SyncOnExit {
    DeviceContext* c;
    ~SyncOnExit() { (void)hipStreamSynchronize(c->stream); c->TrimIfOver(); }
  } sync_on_exit{&ctx};
  ctx.BeginCall();
  if (hipSetDevice(ctx.device) != hipSuccess) return Fail("hipSetDevice failed");

  // pixels -> device, plane by plane.  Rows keep a 16-byte aligned pitch; a bottom-up plane
  // (negative stride) is copied as the memory block it is and addressed with a negative stride.
  sjpeg_hip_source dsrc;
  memset(&dsrc, 0, sizeof(dsrc));
  dsrc.format = src_.format;
  {
    const size_t cw = (static_cast<size_t>(W_) + 1) / 2, ch = (static_cast<size_t>(H_) + 1) / 2;
    size_t row_bytes[3] = {0, 0, 0}, rows[3] = {0, 0, 0};
    int nplanes = 1;
    switch (src_.format) {
      case SJPEG_HIP_SRC_RGB: row_bytes[0] = 3 * static_cast<size_t>(W_); rows[0] = H_; break;
      case SJPEG_HIP_SRC_BGRA:
      case SJPEG_HIP_SRC_RGBA: row_bytes[0] = 4 * static_cast<size_t>(W_); rows[0] = H_; break;
      case SJPEG_HIP_SRC_GRAY: row_bytes[0] = W_; rows[0] = H_; break;
      case SJPEG_HIP_SRC_YUV444:
        nplanes = 3;
        for (int i = 0; i < 3; ++i) { row_bytes[i] = W_; rows[i] = H_; }
        break;
      case SJPEG_HIP_SRC_YUV420:
        nplanes = 3;
        row_bytes[0] = W_; rows[0] = H_;
        row_bytes[1] = row_bytes[2] = cw; rows[1] = rows[2] = ch;
        break;
      default:   // NV12 / NV21
        nplanes = 2;
        row_bytes[0] = W_; rows[0] = H_;
        row_bytes[1] = 2 * cw; rows[1] = ch;
        break;
    }
    size_t offset[3] = {0, 0, 0}, pitch[3] = {0, 0, 0}, total = 0;
    for (int i = 0; i < nplanes; ++i) {
      pitch[i] = (row_bytes[i] + 15) & ~static_cast<size_t>(15);
      offset[i] = total;
      total += pitch[i] * rows[i] + 64;
    }
    // a thumbnail is copied into pinned memory by the CPU and read by the kernels over the bus:
    // no runtime copy call at all (it cost more than the whole K1 launch of such a picture)
    const bool in_place = total <= DeviceContext::kMailIn;
    if (!in_place && !ctx.Ensure(&ctx.d_in, &ctx.in_cap, total)) return false;
    uint8_t* const h_in = ctx.h_mail + 64 + DeviceContext::kMailData;
    uint8_t* const d_in = in_place ? ctx.d_mail + 64 + DeviceContext::kMailData : static_cast<uint8_t*>(ctx.d_in);
    for (int i = 0; i < nplanes; ++i) {
      const long long st = src_.stride[i];
      const size_t host_pitch = static_cast<size_t>(st < 0 ? -st : st);
      const uint8_t* lowest = st < 0 ? src_.plane[i] + static_cast<long long>(rows[i] - 1) * st : src_.plane[i];
      uint8_t* d = d_in + offset[i];
      if (in_place) {
        for (size_t y = 0; y < rows[i]; ++y) memcpy(h_in + offset[i] + y * pitch[i], lowest + y * host_pitch, row_bytes[i]);
      } else if (!ctx.ToDevice2D(d, pitch[i], lowest, host_pitch, row_bytes[i], rows[i])) {
        return Fail("hipMemcpy2D(host -> device) failed");
      }
      dsrc.plane[i] = st < 0 ? d + pitch[i] * (rows[i] - 1) : d;
      dsrc.row_stride[i] = st < 0 ? -static_cast<long long>(pitch[i]) : static_cast<long long>(pitch[i]);
    }
  }
  if (yuv_mode_ == SJPEG_YUV_AUTO) {
    // EncoderFactory (src/encoders.cc:549-551): the picture decides
    if (!ctx.Riskiness(dsrc, W_, H_, &yuv_mode_, nullptr)) return false;
  }
  const bool sharp = (yuv_mode_ == SJPEG_YUV_SHARP);
  const int mode = (yuv_mode_ == SJPEG_YUV_444) ? SJPEG_HIP_YUV444
                 : (yuv_mode_ == SJPEG_YUV_400) ? SJPEG_HIP_YUV400 : SJPEG_HIP_YUV420;
  if (sharp) {
    // EncoderSharp420 (src/encoders.cc:512-541): the sharp conversion turns the RGB picture into
    // Y / U / V planes on the device; from here on this is the planar 4:2:0 encoder.
    const size_t cw = (static_cast<size_t>(W_) + 1) / 2, ch = (static_cast<size_t>(H_) + 1) / 2;
    const size_t ysz = static_cast<size_t>(W_) * H_, csz = cw * ch;
    const size_t wsz = sjpeg_hip_sharp_workspace(W_, H_, 1);
    if (!ctx.Ensure(&ctx.d_planes, &ctx.planes_cap, ysz + 2 * csz + 64)) return false;
    if (!ctx.Ensure(&ctx.d_work, &ctx.work_cap, wsz)) return false;
    uint8_t* const py = static_cast<uint8_t*>(ctx.d_planes);
    if (sjpeg_hip_sharp_yuv(&dsrc, W_, H_, 1, py, py + ysz, py + ysz + csz, 0, 0, ctx.d_work, wsz, ctx.stream) != 0) {
      return Fail("sjpeg_hip_sharp_yuv failed");
    }
    memset(&dsrc, 0, sizeof(dsrc));
    dsrc.format = SJPEG_HIP_SRC_YUV420;
    dsrc.plane[0] = py; dsrc.row_stride[0] = W_;
    dsrc.plane[1] = py + ysz; dsrc.row_stride[1] = static_cast<long long>(cw);
    dsrc.plane[2] = py + ysz + csz; dsrc.row_stride[2] = static_cast<long long>(cw);
  }
  const int nb_comps = (mode == SJPEG_HIP_YUV400) ? 1 : 3;

  // quantizers (src/enc.cc:394-397)
  sjpeg_hip_scan_tables tables;
  memset(&tables, 0, sizeof(tables));
  sjpeg_host::FinalizeQuantizer(quant_[0], min_quant_[0], q_bias_, 0, &tables);
  sjpeg_host::FinalizeQuantizer(quant_[1], min_quant_[1], q_bias_, 1, &tables);
  if (trellis) {
    // methods 7, 8: the trellis prices its rate with the standard AC tables (InitCodes(true) in
    // SinglePassScanOptimized, src/enc.cc:330-334), whatever tables the stream ends up with
    tables.flags |= SJPEG_HIP_QUANT_TRELLIS;
    for (int c = 0; c < 2; ++c) {
      uint32_t codes[256];
      memset(codes, 0, sizeof(codes));
      sjpeg_host::BuildCodes(sjpeg_host::DefaultHuff(1, c), codes);
      for (int i = 0; i < 256; ++i) tables.trellis_len[c][i] = static_cast<uint8_t>(codes[i] & 0xff);
    }
  }
  if (!ctx.Ensure(&ctx.d_stats, &ctx.stats_cap, 2 * 64 * 128 * sizeof(uint32_t))) return false;

  if (passes_ > 1 && !search_ok_) return Fail("SearchHook::Setup() failed");
  const int ntables = (nb_comps == 1) ? 1 : 2;
  constexpr size_t kHistBytes = 2 * 64 * 128 * sizeof(uint32_t);
  constexpr size_t kSumsBytes = 2 * 64 * sjpeg_host::kAdaptDeltas * 2 * sizeof(int64_t);
  constexpr size_t kTotLastBytes = 2 * 64 * 2 * sizeof(int32_t);
  if (adaptive) {
    // CollectHistograms on the GPU (src/enc.cc:425-429, src/dichotomy.cc:117-121); the histogram
    // stays on the device, only the sums AnalyseHisto makes of it come back
    if (!ctx.Ensure(&ctx.d_hist, &ctx.hist_cap, kHistBytes + kSumsBytes + kTotLastBytes + 128)) return false;
    if (sjpeg_hip_scan_histogram_src(ctx.engine, &dsrc, W_, H_, mode, 1,
                                 static_cast<uint32_t*>(ctx.d_hist), ctx.stream) != 0) {
      return FailHip("sjpeg_hip_scan_histogram");
    }
  }
  bool adapt_ok = true;
  auto adapt = [&]() {                              // AnalyseHisto: bin loops on the GPU, the rest here
    uint8_t* const base = static_cast<uint8_t*>(ctx.d_hist);
    int64_t* const d_sums = reinterpret_cast<int64_t*>(base + kHistBytes);
    int32_t* const d_totlast = reinterpret_cast<int32_t*>(base + kHistBytes + kSumsBytes);
    static thread_local int64_t sums[2][64][sjpeg_host::kAdaptDeltas][2];
    static thread_local int32_t totlast[2][64][2];
    // the float half on the device too (sjpeg_hip_adapt_decide: the host's result bit for bit; 128 bytes come back instead
    // of 52 KB) -- for the step limits the kernel's 25 candidates cover; others take the host's form below
    if (qdelta_luma_ >= -12 && qdelta_luma_ <= 12 && qdelta_chroma_ >= -12 && qdelta_chroma_ <= 12) {
      uint8_t* const d_q = base + kHistBytes + kSumsBytes + kTotLastBytes;
      uint8_t q_new[2][64];
      if (sjpeg_hip_adapt_sums(static_cast<const uint32_t*>(ctx.d_hist), 1, quant_, &min_quant_[0][0], d_sums,
                               d_totlast, ctx.stream) != 0 ||
          sjpeg_hip_adapt_decide(d_sums, d_totlast, 1, quant_, nb_comps == 1 ? SJPEG_HIP_YUV400 : SJPEG_HIP_YUV420,
                                 qdelta_luma_, qdelta_chroma_, d_q, ctx.stream) != 0 ||
          !ctx.ToHost(q_new, d_q, sizeof(q_new))) {
        adapt_ok = false;
        return;
      }
      memcpy(quant_, q_new, static_cast<size_t>(ntables) * 64);
      for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
        sjpeg_host::FinalizeQuantizer(quant_[idx], min_quant_[idx], q_bias_, idx, &tables);
      }
      return;
    }
    if (sjpeg_hip_adapt_sums(static_cast<const uint32_t*>(ctx.d_hist), 1, quant_, &min_quant_[0][0], d_sums,
                             d_totlast, ctx.stream) != 0 ||
        !ctx.ToHost(sums, d_sums, kSumsBytes) ||
        !ctx.ToHost(totlast, d_totlast, kTotLastBytes)) {
      adapt_ok = false;
      return;
    }
    sjpeg_host::AdaptDecide(sums, totlast, nb_comps, quant_, qdelta_luma_, qdelta_chroma_);
    for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
      sjpeg_host::FinalizeQuantizer(quant_[idx], min_quant_[idx], q_bias_, idx, &tables);
    }
  };
  auto symbol_stats = [&](uint32_t freq[2][272]) -> bool {
    if (sjpeg_hip_scan_symbol_stats_src(ctx.engine, &dsrc, W_, H_, mode, 1, &tables,
                                        static_cast<uint32_t*>(ctx.d_stats), ctx.stream) != 0) {
      return FailHip("sjpeg_hip_scan_symbol_stats");
    }
    if (!ctx.ToHost(freq, ctx.d_stats, 2 * 272 * sizeof(uint32_t))) {
      return Fail(std::string("statistics pass failed: ") + hipGetErrorString(hipGetLastError()));
    }
    return true;
  };

  // Trellis + search: the rate table is the encoder's live AC code array (Quantizer::codes_,
  // src/quantize.cc:151): InitCodes() writes the codes of the symbols a table HAS over it and leaves
  // the rest as they were, so it accumulates over the passes (standard tables first, then whatever
  // each size pass compiled).  tables.trellis_len plays that role here.
  bool final_tables_known = false;               // the last pass' run/levels are the stream
  HuffSpec pass_specs[4];
  uint8_t pass_rate[2][256];
  if (passes_ > 1) {
    // Encoder::LoopScan (src/dichotomy.cc:113-205): the search is a host control loop; each pass
    // costs one statistics / error / size pass on the GPU over the resident picture.
    SearchHook* const hook = search_hook_;
    uint8_t opt_quants[2][64];
    float best = 0.f, best_q = 0.f, best_result = 0.f;
    bool last_is_best = false;
    for (int p = 0; p < passes_; ++p) {
      hook->pass = p;
      for (int c = 0; c < 2; ++c) {
        hook->NextMatrix(c, quant_[c]);
        sjpeg_host::FinalizeQuantizer(quant_[c], min_quant_[c], q_bias_, c, &tables);
      }
      if (adaptive) { adapt(); if (!adapt_ok) return Fail("adaptive-quantization analysis failed on the device"); }
      float result;
      if (hook->for_size) {
        const HuffSpec* pdc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
        const HuffSpec* pac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
        HuffSpec popt[4];
        uint32_t freq[2][272];
        if (optimize) {
          if (trellis) {                             // what this pass is priced with; its blocks stay for a replay
            memcpy(pass_rate, tables.trellis_len, sizeof(pass_rate));
            tables.flags |= SJPEG_HIP_QUANT_KEEP;
          }
          if (!symbol_stats(freq)) return false;
          tables.flags &= ~SJPEG_HIP_QUANT_KEEP;
          for (int t = 0; t < ntables; ++t) {
            sjpeg_host::BuildOptimalSpec(freq[t] + 256, 12, &popt[t]);
            sjpeg_host::BuildOptimalSpec(freq[t], 256, &popt[2 + t]);
            pdc[t] = &popt[t]; pac[t] = &popt[2 + t];
          }
        }
        sjpeg_host::InstallCodes(pdc, pac, ntables, &tables);
        if (trellis) {                             // InitCodes(true) after CompileEntropyStats (src/dichotomy.cc:152)
          for (int t = 0; t < ntables; ++t) {
            for (int i = 0; i < popt[2 + t].nsyms; ++i) {
              const int sym = popt[2 + t].syms[i];
              tables.trellis_len[t][sym] = static_cast<uint8_t>(tables.ac_codes[t][sym] & 0xff);
            }
          }
          memcpy(pass_specs, popt, sizeof(pass_specs));
        }
        // HeaderSize() with the reference's own accounting (src/dichotomy.cc:210-241)
        size_t size = 20 + meta_.app_markers.size();
        if (!meta_.exif.empty()) size += 8 + meta_.exif.size();
        if (!meta_.iccp.empty()) {
          const size_t kMax = 0xffff - 12 - 4;
          size += ((meta_.iccp.size() - 1) / kMax + 1) * (12 + 4 + 2) + meta_.iccp.size();
        }
        if (!meta_.xmp.empty()) size += 2 + 2 + 29 + meta_.xmp.size();
        size += ntables * 65 + 2 + 2;
        size += 8 + 3 * nb_comps + 2;
        size += 6 + 2 * nb_comps + 2;
        size += 2;
        for (int t = 0; t < ntables; ++t) size += (2 + 3 + 16 + pdc[t]->nsyms) + (2 + 3 + 16 + pac[t]->nsyms);
        size *= 8;
        if (optimize) {                              // EntropySize(), src/entropy.cc:230-245
          for (int t = 0; t < ntables; ++t) {
            for (int len = 0; len < 12; ++len) {
              if (freq[t][256 + len]) size += static_cast<size_t>(freq[t][256 + len]) * ((tables.dc_codes[t][len] & 0xff) + len);
            }
            for (int sym = 0; sym < 256; ++sym) {
              if (freq[t][sym]) size += static_cast<size_t>(freq[t][sym]) * ((tables.ac_codes[t][sym] & 0xff) + (sym & 0x0f));
            }
          }
        } else {
          // BitCounter (src/bit_writer.h:292-365): coded bits + 8 per 0xFF among COMPLETED bytes.
          // One real coding pass gives both: entropy bits, and the escapes through the size.
          const size_t bound = sjpeg_hip_frame_bound(W_, H_, mode, 0);
          if (bound == 0) return false;
          uint64_t bits = 0, bytes = 0;
          for (size_t cap = ctx.FirstCapacity(W_, H_, mode, 0, bound); bytes == 0; cap = bound) {
            if (!ctx.Ensure(&ctx.d_out, &ctx.out_cap, cap)) return false;
            if (sjpeg_hip_encode_scan_src(ctx.engine, &dsrc, W_, H_, mode, 1, &tables, nullptr, 0, 0,
                                          ctx.d_out, cap, ctx.d_size, ctx.stream) != 0) {
              return FailHip("sjpeg_hip_encode_scan");
            }
            if (!ctx.ToHost(&bytes, ctx.d_size, sizeof(bytes))) return Fail("size pass failed");
            if (bytes == 0 && cap == bound) return Fail("size pass failed");
          }
          if (sjpeg_hip_engine_entropy_bits(ctx.engine, &bits, 1) != 0) return FailHip("entropy_bits");
          uint64_t escapes = bytes - (bits + 7) / 8;
          if ((bits & 7) != 0 && bytes >= 2) {       // a padded last byte that became 0xFF is not counted
            uint8_t tail[2];
            if (!ctx.ToHost(tail, static_cast<const uint8_t*>(ctx.d_out) + bytes - 2, 2)) {
              return Fail("size pass failed");
            }
            if (tail[0] == 0xff && tail[1] == 0x00) --escapes;
          }
          size += bits + 8 * escapes;
        }
        result = size / 8.f;
      } else {
        // ComputePSNR (src/dichotomy.cc:295-323)
        uint64_t err = 0;
        if (sjpeg_hip_scan_quant_error_src(ctx.engine, &dsrc, W_, H_, mode, 1, &tables,
                                           reinterpret_cast<uint64_t*>(ctx.d_stats), ctx.stream) != 0) {
          return FailHip("sjpeg_hip_scan_quant_error");
        }
        if (!ctx.ToHost(&err, ctx.d_stats, sizeof(err))) return Fail("error pass failed");
        sjpeg_host::FrameLayout L;
        sjpeg_host::LayoutFor(mode, &L);
        const uint64_t nb_mbs = static_cast<uint64_t>((W_ + L.block_w - 1) / L.block_w) * ((H_ + L.block_h - 1) / L.block_h);
        const uint64_t n = 64ull * nb_mbs * L.mcu_blocks;
        result = (err > 0 && n > 0) ? 4.3429448f * log(n / (err / 255. / 255.)) : 99.f;
      }
      last_is_best = (p == 0 || fabs(result - hook->target) < best);
      if (last_is_best) {
        memcpy(opt_quants, quant_, sizeof(opt_quants));
        best = fabs(result - hook->target);
        best_q = hook->q;
        best_result = result;
      }
      if (hook->Update(result)) break;
    }
    // transfer back the best matrices; they are final (no further adaptation)
    sjpeg_host::ScaleMatrix(opt_quants[0], 100.f, quant_[0]);
    sjpeg_host::ScaleMatrix(opt_quants[1], 100.f, quant_[1]);
    for (int c = 0; c < 2; ++c) sjpeg_host::FinalizeQuantizer(quant_[c], min_quant_[c], q_bias_, c, &tables);
    hook->q = best_q;
    hook->value = best_result;
    if (trellis && hook->for_size && last_is_best) {
      // src/dichotomy.cc:188: nothing is quantized again, the last pass' run/levels are written
      memcpy(tables.trellis_len, pass_rate, sizeof(pass_rate));
      final_tables_known = true;
    }
  } else if (adaptive) {
    adapt();
    if (!adapt_ok) return Fail("adaptive-quantization analysis failed on the device");
  }

  // Huffman tables: Annex K defaults (src/enc.cc:399) or optimised for this picture
  const HuffSpec* dc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
  const HuffSpec* ac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
  HuffSpec opt[4];
  bool replay = false;                              // trellis: the statistics pass keeps its blocks, the encode pass replays them
  if (final_tables_known) {
    for (int t = 0; t < ntables; ++t) { dc[t] = &pass_specs[t]; ac[t] = &pass_specs[2 + t]; }
    replay = true;                                  // the last pass' blocks are still in the engine
  } else if (optimize) {
    // statistics half of SinglePassScanOptimized on the GPU (src/enc.cc:323-372),
    // CompileEntropyStats on the host (src/entropy.cc:432-444)
    uint32_t freq[2][272];
    if (trellis) tables.flags |= SJPEG_HIP_QUANT_KEEP;
    if (!symbol_stats(freq)) return false;
    if (trellis) { tables.flags &= ~SJPEG_HIP_QUANT_KEEP; replay = true; }
    for (int t = 0; t < ntables; ++t) {
      sjpeg_host::BuildOptimalSpec(freq[t] + 256, 12, &opt[t]);
      sjpeg_host::BuildOptimalSpec(freq[t], 256, &opt[2 + t]);
      dc[t] = &opt[t];
      ac[t] = &opt[2 + t];
    }
  }
  if (replay) tables.flags |= SJPEG_HIP_QUANT_REPLAY;
  sjpeg_host::InstallCodes(dc, ac, ntables, &tables);

  // headers: SOI/APP0, metadata, DQT, SOF, DHT, SOS (src/enc.cc:415-443)
  std::vector<uint8_t> header;
  if (!sjpeg_host::AppendHeaders(W_, H_, mode, quant_, dc, ac, &meta_, &header)) {
    return Fail("invalid metadata (EXIF > 64 KiB, ICC >= 256 chunks or XMP too large)");
  }
  // one host allocation goes through the caller's MemoryManager, and its failure is
  // fatal, as in the reference (tests/unit_test.cc:373-454 rely on both).
  uint8_t* const staged_header = static_cast<uint8_t*>(mem_->Alloc(header.size()));
  if (staged_header == nullptr) return Fail("MemoryManager refused an allocation");
  memcpy(staged_header, header.data(), header.size());
  struct Guard {
    MemoryManager* m; void* p;
    ~Guard() { m->Free(p); }
  } guard = {mem_, staged_header};

  const size_t bound = sjpeg_hip_frame_bound(W_, H_, mode, header.size());
  if (bound == 0) return Fail("internal: no bound for this geometry");
  const bool mailed = bound <= DeviceContext::kMailData;          // small picture: straight into pinned memory
  volatile uint64_t* const h_size = reinterpret_cast<volatile uint64_t*>(ctx.h_mail);
  uint64_t size = 0;
  // The output buffer (and with it the engine's scratch, which follows out_stride) is sized for
  // what pictures code to, not for the 6.75 bytes per pixel of the worst case; a frame that does not
  // fit reports size 0 and is coded again against the bound (deterministic: same bytes either way).
  for (size_t cap = mailed ? bound : ctx.FirstCapacity(W_, H_, mode, header.size(), bound); size == 0; cap = bound) {
    if (!mailed && !ctx.Ensure(&ctx.d_out, &ctx.out_cap, cap)) return false;
    void* const d_stream = mailed ? static_cast<void*>(ctx.d_mail + 64) : ctx.d_out;
    *h_size = 0;
    if (sjpeg_hip_encode_scan_src(ctx.engine, &dsrc, W_, H_, mode, 1, &tables,
                                  staged_header, header.size(), /*append_eoi=*/1, d_stream, cap,
                                  reinterpret_cast<uint64_t*>(ctx.d_mail), ctx.stream) != 0) {
      return FailHip("sjpeg_hip_encode_scan");
    }
    if (hipStreamSynchronize(ctx.stream) != hipSuccess) {
      return Fail(std::string("device execution failed: ") + hipGetErrorString(hipGetLastError()));
    }
    size = *h_size;
    if (size == 0 && cap == bound) return Fail("internal: coded frame exceeded its bound");
  }

  // device -> sink, straight into the sink's own storage
  uint8_t* dst = nullptr;
  if (!sink_->Commit(0, size, &dst) || dst == nullptr) { sink_->Reset(); return Fail("sink refused the output"); }
  if (mailed) {
    memcpy(dst, ctx.h_mail + 64, size);
  } else if (!ctx.ToHost(dst, ctx.d_out, size)) {
    sink_->Reset();
    return Fail("hipMemcpy(device -> host) failed");
  }
  if (!sink_->Commit(size, 0, &dst) || !sink_->Finalize()) { sink_->Reset(); return Fail("sink failed"); }
  return true;
}

// ---- EncoderParam (reference: src/api.cc:74-143) ---------------------------------------------

EncoderParam::EncoderParam() : search_hook(nullptr), memory(nullptr) { Init(kDefaultQuality); }
EncoderParam::EncoderParam(float quality_factor) : search_hook(nullptr), memory(nullptr) {
  Init(quality_factor);
}

void EncoderParam::Init(float quality_factor) {
  yuv_mode = SJPEG_YUV_AUTO;
  Huffman_compress = true;
  adaptive_quantization = true;
  adaptive_bias = false;
  use_trellis = false;
  target_mode = TARGET_NONE;
  target_value = 0;
  passes = 1;
  tolerance = 1.;
  qmin = 0.;
  qmax = 100.;
  quantization_bias = kDefaultBias;
  qdelta_max_luma = kDefaultDeltaMaxLuma;
  qdelta_max_chroma = kDefaultDeltaMaxChroma;
  use_min_quant_ = false;
  min_quant_tolerance_ = 0;
  memset(min_quant_, 0, sizeof(min_quant_));
  SetQuality(quality_factor);
}

void EncoderParam::SetQuality(float quality_factor) {
  const float s = sjpeg_host::QualityToScale(quality_factor);
  sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[0], s, quant_[0]);
  sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[1], s, quant_[1]);
}

void EncoderParam::SetQuantization(const uint8_t m[2][64], float reduction) {
  if (reduction <= 1.f) reduction = 1.f;
  if (m == nullptr) return;
  for (int c = 0; c < 2; ++c) {
    for (int i = 0; i < 64; ++i) {
      // double arithmetic, as src/api.cc:115
      const int v = static_cast<int>(m[c][i] * 100. / reduction + .5);
      quant_[c][i] = static_cast<uint8_t>(v > 255 ? 255 : v < 1 ? 1 : v);
    }
  }
}

void EncoderParam::SetLimitQuantization(bool limit_quantization, int tolerance) {
  use_min_quant_ = limit_quantization;
  if (limit_quantization) SetMinQuantization(quant_, tolerance);
}

void EncoderParam::SetMinQuantization(const uint8_t m[2][64], int tolerance) {
  use_min_quant_ = true;
  memcpy(min_quant_[0], m[0], 64);
  memcpy(min_quant_[1], m[1], 64);
  min_quant_tolerance_ = tolerance < 0 ? 0 : tolerance > 100 ? 100 : tolerance;
}

void EncoderParam::ResetMetadata() {
  iccp.clear(); exif.clear(); app_markers.clear(); xmp.clear();
  xmp_split_point = 0u;
}

// ---- SearchHook defaults (reference: src/dichotomy.cc:41-75) ------------------------------

bool SearchHook::Setup(const EncoderParam& param) {
  for_size = (param.target_mode == EncoderParam::TARGET_SIZE);
  target = param.target_value;
  tolerance = param.tolerance / 100.;
  qmin = (param.qmin < 0) ? 0 : param.qmin;
  qmax = (param.qmax > 100) ? 100 : (param.qmax < param.qmin) ? param.qmin : param.qmax;
  const float q0 = SjpegEstimateQuality(param.GetQuantMatrix(0), false);
  q = q0 < qmin ? qmin : q0 > qmax ? qmax : q0;
  value = 0;
  pass = 0;
  return true;
}

bool SearchHook::Update(float result) {
  value = result;
  if (std::fabs(value - target) < tolerance * target) return true;
  if (value > target) qmax = q; else qmin = q;
  const float last_q = q;
  q = (qmin + qmax) / 2.;
  return std::fabs(q - last_q) < 0.15;
}

void SearchHook::NextMatrix(int idx, uint8_t dst[64]) {
  sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[idx], sjpeg_host::QualityToScale(q), dst);
}

// ---- entry points (reference: src/api.cc:183-304) ---------------------------------------------

bool Encode(const uint8_t* rgb, int width, int height, int stride,
            const EncoderParam& param, ByteSink* sink) {
  if (rgb == nullptr || sink == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0 || std::abs(stride) < 3 * width) return Fail("bad dimensions or stride");
  Encoder enc(rgb, width, height, stride, sink, param.memory);
  enc.InitFromParam(param);
  return enc.Run();
}

size_t Encode(const uint8_t* rgb, int width, int height, int stride,
              const EncoderParam& param, uint8_t** out_data) {
  if (out_data == nullptr) return 0;
  NewArraySink sink;
  if (!Encode(rgb, width, height, stride, param, &sink)) return 0;
  return sink.Release(out_data);
}

bool Encode(const uint8_t* rgb, int width, int height, int stride,
            const EncoderParam& param, std::string* output) {
  if (output == nullptr) return false;
  output->clear();
  ContainerSink<std::string> sink(output);
  return Encode(rgb, width, height, stride, param, &sink);
}

// ---- other input layouts (reference: src/api.cc:201-304, src/encoders.cc:346-490) ------------

static bool EncodeSource(const HostSource& src, int width, int height, const EncoderParam& param,
                         ByteSink* sink) {
  Encoder enc(src, width, height, sink, param.memory);
  enc.InitFromParam(param);
  return enc.Run();
}

bool EncodeBGRA(const uint8_t* bgra, int width, int height, int stride,
                const EncoderParam& param, ByteSink* sink) {
  if (bgra == nullptr || sink == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0 || std::abs(stride) < 4 * width) return Fail("bad dimensions or stride");
  return EncodeSource(HostSource{SJPEG_HIP_SRC_BGRA, {bgra, nullptr, nullptr}, {stride, 0, 0}}, width, height, param, sink);
}

bool EncodeRGBA(const uint8_t* rgba, int width, int height, int stride,
                const EncoderParam& param, ByteSink* sink) {
  if (rgba == nullptr || sink == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0 || std::abs(stride) < 4 * width) return Fail("bad dimensions or stride");
  return EncodeSource(HostSource{SJPEG_HIP_SRC_RGBA, {rgba, nullptr, nullptr}, {stride, 0, 0}}, width, height, param, sink);
}

bool EncodeGray(const uint8_t* gray, int width, int height, int stride,
                const EncoderParam& param, ByteSink* sink) {
  if (gray == nullptr || sink == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0 || std::abs(stride) < width) return Fail("bad dimensions or stride");
  return EncodeSource(HostSource{SJPEG_HIP_SRC_GRAY, {gray, nullptr, nullptr}, {stride, 0, 0}}, width, height, param, sink);
}

static bool EncodeNV(const uint8_t* y, int y_stride, const uint8_t* uv, int uv_stride, int width,
                     int height, int format, const EncoderParam& param, ByteSink* sink) {
  if (y == nullptr || uv == nullptr || sink == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0) return Fail("bad dimensions");
  if (std::abs(y_stride) < width || std::abs(uv_stride) < 2 * ((width + 1) / 2)) return Fail("bad stride");
  return EncodeSource(HostSource{format, {y, uv, nullptr}, {y_stride, uv_stride, 0}}, width, height, param, sink);
}

bool EncodeNV12(const uint8_t* y, int y_stride, const uint8_t* uv, int uv_stride,
                int width, int height, const EncoderParam& param, ByteSink* output) {
  return EncodeNV(y, y_stride, uv, uv_stride, width, height, SJPEG_HIP_SRC_NV12, param, output);
}

bool EncodeNV21(const uint8_t* y, int y_stride, const uint8_t* vu, int vu_stride,
                int width, int height, const EncoderParam& param, ByteSink* output) {
  return EncodeNV(y, y_stride, vu, vu_stride, width, height, SJPEG_HIP_SRC_NV21, param, output);
}

bool EncodeYUV444(const uint8_t* Y, int Y_stride, const uint8_t* U, int U_stride,
                  const uint8_t* V, int V_stride, int width, int height,
                  const EncoderParam& param, ByteSink* output) {
  if (Y == nullptr || U == nullptr || V == nullptr || output == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0) return Fail("bad dimensions");
  if (std::abs(Y_stride) < width || std::abs(U_stride) < width || std::abs(V_stride) < width) return Fail("bad stride");
  return EncodeSource(HostSource{SJPEG_HIP_SRC_YUV444, {Y, U, V}, {Y_stride, U_stride, V_stride}}, width, height, param, output);
}

bool EncodeYUV420(const uint8_t* Y, int Y_stride, const uint8_t* U, int U_stride,
                  const uint8_t* V, int V_stride, int width, int height,
                  const EncoderParam& param, ByteSink* output) {
  if (Y == nullptr || U == nullptr || V == nullptr || output == nullptr) return Fail("null argument");
  if (width <= 0 || height <= 0) return Fail("bad dimensions");
  const int cw = (width + 1) / 2;
  if (std::abs(Y_stride) < width || std::abs(U_stride) < cw || std::abs(V_stride) < cw) return Fail("bad stride");
  return EncodeSource(HostSource{SJPEG_HIP_SRC_YUV420, {Y, U, V}, {Y_stride, U_stride, V_stride}}, width, height, param, output);
}

#define SJPEG_STRING_VARIANT(NAME)                                                                  \
  bool NAME(const uint8_t* px, int width, int height, int stride, const EncoderParam& param,        \
            std::string* output) {                                                                   \
    if (output == nullptr) return false;                                                             \
    output->clear();                                                                                 \
    ContainerSink<std::string> sink(output);                                                         \
    return NAME(px, width, height, stride, param, &sink);                                            \
  }
SJPEG_STRING_VARIANT(EncodeBGRA)
SJPEG_STRING_VARIANT(EncodeRGBA)
SJPEG_STRING_VARIANT(EncodeGray)
#undef SJPEG_STRING_VARIANT

std::shared_ptr<ByteSink> MakeByteSink(std::string* output) {
  return std::shared_ptr<ByteSink>(new (std::nothrow) ContainerSink<std::string>(output));
}
template<> std::shared_ptr<ByteSink> MakeByteSink(std::vector<uint8_t>* output) {
  return std::shared_ptr<ByteSink>(new (std::nothrow) ContainerSink<std::vector<uint8_t> >(output));
}

}  // namespace sjpeg

// ---- plain-C entry points (reference: src/api.cc:32-67) ---------------------------------------

extern "C" {

uint32_t SjpegVersion() { return SJPEG_VERSION; }

const char* SjpegHipLastError() { return g_api_error.c_str(); }

size_t SjpegEncode(const uint8_t* rgb, int width, int height, int stride, uint8_t** out_data,
                   float quality, int method, SjpegYUVMode yuv_mode) {
  if (rgb == nullptr || out_data == nullptr) return 0;
  if (width <= 0 || height <= 0 || std::abs(stride) < 3 * width) return 0;
  *out_data = nullptr;
  NewArraySink sink;
  sjpeg::Encoder enc(rgb, width, height, stride, &sink, nullptr);
  enc.SetYuvMode(yuv_mode);
  enc.SetQuality(quality);
  enc.SetMethod(method);
  if (!enc.Run()) return 0;
  return sink.Release(out_data);
}

size_t SjpegCompress(const uint8_t* rgb, int width, int height, float quality, uint8_t** out_data) {
  return SjpegEncode(rgb, width, height, 3 * width, out_data, quality, 4, SJPEG_YUV_AUTO);
}

void SjpegFreeBuffer(const uint8_t* buffer) { delete[] buffer; }

// ---- C-ABI host helpers declared in sjpeg_hip.h --------------------------------------------

void sjpeg_hip_quality_matrices(float quality, uint8_t quant[2][64]) {
  const float s = sjpeg_host::QualityToScale(quality);
  sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[0], s, quant[0]);
  sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[1], s, quant[1]);
}

void sjpeg_hip_finalize_quant(uint8_t quant[2][64], const uint8_t* min_quant, int q_bias,
                              sjpeg_hip_scan_tables* tables) {
  uint8_t ones[64];
  memset(ones, 1, sizeof(ones));
  for (int c = 0; c < 2; ++c) {
    uint8_t m[64];
    sjpeg_host::ScaleMatrix(quant[c], 100.f, m);          // 0 -> 1, as src/enc.cc:106-109
    memcpy(quant[c], m, 64);
    sjpeg_host::FinalizeQuantizer(quant[c], min_quant ? min_quant + 64 * c : ones, q_bias, c, tables);
  }
}

void sjpeg_hip_default_huffman(sjpeg_hip_scan_tables* tables) {
  const HuffSpec* dc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
  const HuffSpec* ac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
  sjpeg_host::InstallCodes(dc, ac, 2, tables);
}

size_t sjpeg_hip_make_header(int width, int height, int yuv_mode, const uint8_t quant[2][64],
                             uint8_t* buf, size_t cap) {
  if (buf == nullptr || width <= 0 || height <= 0 || width > 65535 || height > 65535) return 0;
  const HuffSpec* dc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
  const HuffSpec* ac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
  std::vector<uint8_t> h;
  if (!sjpeg_host::AppendHeaders(width, height, yuv_mode, quant, dc, ac, nullptr, &h)) return 0;
  if (h.size() > cap) return 0;
  memcpy(buf, h.data(), h.size());
  return h.size();
}

void sjpeg_hip_adapt_quant(const uint32_t* hist, int yuv_mode, uint8_t quant[2][64],
                           const uint8_t* min_quant, int q_bias, int qdelta_max_luma,
                           int qdelta_max_chroma, sjpeg_hip_scan_tables* tables) {
  uint8_t mq[2][64];
  if (min_quant != nullptr) memcpy(mq, min_quant, sizeof(mq)); else memset(mq, 1, sizeof(mq));
  const int nb_comps = (yuv_mode == SJPEG_HIP_YUV400) ? 1 : 3;
  sjpeg_host::AdaptQuantMatrices(reinterpret_cast<const uint32_t(*)[64][128]>(hist), nb_comps, quant, mq,
                                 qdelta_max_luma, qdelta_max_chroma);
  for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
    sjpeg_host::FinalizeQuantizer(quant[idx], mq[idx], q_bias, idx, tables);
  }
}

void sjpeg_hip_adapt_quant_sums(const int64_t* sums, const int32_t* totlast, int yuv_mode,
                                uint8_t quant[2][64], const uint8_t* min_quant, int q_bias,
                                int qdelta_max_luma, int qdelta_max_chroma, sjpeg_hip_scan_tables* tables) {
  uint8_t mq[2][64];
  if (min_quant != nullptr) memcpy(mq, min_quant, sizeof(mq)); else memset(mq, 1, sizeof(mq));
  const int nb_comps = (yuv_mode == SJPEG_HIP_YUV400) ? 1 : 3;
  sjpeg_host::AdaptDecide(reinterpret_cast<const int64_t(*)[64][sjpeg_host::kAdaptDeltas][2]>(sums),
                          reinterpret_cast<const int32_t(*)[64][2]>(totlast), nb_comps, quant,
                          qdelta_max_luma, qdelta_max_chroma);
  for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
    sjpeg_host::FinalizeQuantizer(quant[idx], mq[idx], q_bias, idx, tables);
  }
}

void sjpeg_hip_optimize_huffman(const uint32_t* freq, int yuv_mode, sjpeg_hip_huffman_spec specs[4],
                                sjpeg_hip_scan_tables* tables) {
  const int ntables = (yuv_mode == SJPEG_HIP_YUV400) ? 1 : 2;
  const HuffSpec* dc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
  const HuffSpec* ac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
  for (int t = 0; t < ntables; ++t) {
    sjpeg_host::BuildOptimalSpec(freq + t * 272 + 256, 12, &specs[t]);
    sjpeg_host::BuildOptimalSpec(freq + t * 272, 256, &specs[2 + t]);
    dc[t] = &specs[t];
    ac[t] = &specs[2 + t];
  }
  sjpeg_host::InstallCodes(dc, ac, ntables, tables);
}

size_t sjpeg_hip_make_header_ex(int width, int height, int yuv_mode, const uint8_t quant[2][64],
                                const sjpeg_hip_huffman_spec* specs, uint8_t* buf, size_t cap) {
  if (buf == nullptr || width <= 0 || height <= 0 || width > 65535 || height > 65535) return 0;
  const HuffSpec* dc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
  const HuffSpec* ac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
  if (specs != nullptr) {
    const int ntables = (yuv_mode == SJPEG_HIP_YUV400) ? 1 : 2;
    for (int t = 0; t < ntables; ++t) { dc[t] = &specs[t]; ac[t] = &specs[2 + t]; }
  }
  std::vector<uint8_t> h;
  if (!sjpeg_host::AppendHeaders(width, height, yuv_mode, quant, dc, ac, nullptr, &h)) return 0;
  if (h.size() > cap) return 0;
  memcpy(buf, h.data(), h.size());
  return h.size();
}

int sjpeg_hip_set_riskiness_table(const uint8_t* table, size_t size) {
  if (table == nullptr || size != SJPEG_HIP_RISKINESS_TABLE_SIZE) return SJPEG_HIP_EINVAL;
  std::lock_guard<std::mutex> lock(g_risk_mutex);
  g_risk_table.assign(table, table + size);
  ++g_risk_generation;
  return 0;
}

int sjpeg_hip_has_riskiness_table(void) {
  std::lock_guard<std::mutex> lock(g_risk_mutex);
  return LoadRiskTableFromEnv() ? 1 : 0;
}

// reference: src/jpeg_tools.cc:177-236
SjpegYUVMode SjpegRiskiness(const uint8_t* rgb, int width, int height, int stride, float* risk) {
  if (risk != nullptr) *risk = -1.f;
  const int abs_stride = stride < 0 ? -stride : stride;
  if (rgb == nullptr || width <= 0 || height <= 0 || abs_stride < 3 * width) {
    Fail("SjpegRiskiness: bad arguments");
    return SJPEG_YUV_AUTO;
  }
  DeviceContext& ctx = g_ctx;
  if (!ctx.Init() || hipSetDevice(ctx.device) != hipSuccess) return SJPEG_YUV_AUTO;
  struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{ctx.stream};
  const size_t pitch = (3 * static_cast<size_t>(width) + 15) & ~static_cast<size_t>(15);
  if (!ctx.Ensure(&ctx.d_in, &ctx.in_cap, pitch * height + 64)) return SJPEG_YUV_AUTO;
  const uint8_t* lowest = stride < 0 ? rgb + static_cast<long long>(height - 1) * stride : rgb;
  if (!ctx.ToDevice2D(ctx.d_in, pitch, lowest, abs_stride, 3 * static_cast<size_t>(width), height)) {
    Fail("hipMemcpy2D(host -> device) failed");
    return SJPEG_YUV_AUTO;
  }
  sjpeg_hip_source dsrc;
  memset(&dsrc, 0, sizeof(dsrc));
  dsrc.format = SJPEG_HIP_SRC_RGB;
  uint8_t* d = static_cast<uint8_t*>(ctx.d_in);
  dsrc.plane[0] = stride < 0 ? d + pitch * (height - 1) : d;
  dsrc.row_stride[0] = stride < 0 ? -static_cast<long long>(pitch) : static_cast<long long>(pitch);
  SjpegYUVMode mode = SJPEG_YUV_AUTO;
  if (!ctx.Riskiness(dsrc, width, height, &mode, risk)) {
    if (risk != nullptr) *risk = -1.f;
    return SJPEG_YUV_AUTO;
  }
  return mode;
}

size_t sjpeg_hip_make_header_meta(int width, int height, int yuv_mode, const uint8_t quant[2][64],
                                  const sjpeg_hip_huffman_spec* specs, const sjpeg_hip_metadata* meta,
                                  uint8_t* buf, size_t cap) {
  if (buf == nullptr || width <= 0 || height <= 0 || width > 65535 || height > 65535) return 0;
  const HuffSpec* dc[2] = {&sjpeg_host::DefaultHuff(0, 0), &sjpeg_host::DefaultHuff(0, 1)};
  const HuffSpec* ac[2] = {&sjpeg_host::DefaultHuff(1, 0), &sjpeg_host::DefaultHuff(1, 1)};
  if (specs != nullptr) {
    const int ntables = (yuv_mode == SJPEG_HIP_YUV400) ? 1 : 2;
    for (int t = 0; t < ntables; ++t) { dc[t] = &specs[t]; ac[t] = &specs[2 + t]; }
  }
  sjpeg_host::Metadata m;
  if (meta != nullptr) {
    if (meta->app_markers != nullptr) m.app_markers.assign(static_cast<const char*>(meta->app_markers), meta->app_markers_size);
    if (meta->exif != nullptr) m.exif.assign(static_cast<const char*>(meta->exif), meta->exif_size);
    if (meta->iccp != nullptr) m.iccp.assign(static_cast<const char*>(meta->iccp), meta->iccp_size);
    if (meta->xmp != nullptr) m.xmp.assign(static_cast<const char*>(meta->xmp), meta->xmp_size);
    m.xmp_split = meta->xmp_split_point;
  }
  std::vector<uint8_t> h;
  if (!sjpeg_host::AppendHeaders(width, height, yuv_mode, quant, dc, ac, &m, &h)) return 0;
  if (h.size() > cap) return 0;
  memcpy(buf, h.data(), h.size());
  return h.size();
}

size_t sjpeg_hip_host_trim(void) {
  DeviceContext& ctx = g_ctx;
  if (ctx.engine == nullptr) return 0;
  const size_t held = ctx.CachedBytes();
  ctx.Trim();
  return held - ctx.CachedBytes();
}

}  // extern "C"

bool SjpegCompress(const uint8_t* rgb, int width, int height, float quality, std::string* output) {
  sjpeg::EncoderParam param;
  param.SetQuality(quality);
  return sjpeg::Encode(rgb, width, height, 3 * width, param, output);
}
