// C++ API conformance program for include/sjpeg.h (the drop-in surface), modelled on WHAT the
// reference's tests/unit_test.cc checks (argument validation, strides, fault injection through
// MemoryManager / ByteSink, parsers, thread determinism) -- own code, black box through the header.
// Needs a GPU.  Usage: api_test <outdir>: runs the checks (exit code = number of failures) and
// writes <outdir>/<case>.jpg for the parity cases that tests/test_cxx_api.py compares with the oracle.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "sjpeg.h"

static int g_failures = 0;
#define CHECK(cond) do { if (!(cond)) { ++g_failures; fprintf(stderr, "CHECK failed %s:%d: %s  [%s]\n", \
    __FILE__, __LINE__, #cond, SjpegHipLastError()); } } while (0)

static uint32_t g_seed;
static uint8_t Rand8() { g_seed = 1103515245u * g_seed + 12345u; return static_cast<uint8_t>(g_seed >> 16); }
static std::vector<uint8_t> Picture(int w, int h, uint32_t seed) {     // == oracle/synth.py g_struct
  g_seed = seed;
  std::vector<uint8_t> rgb(3 * static_cast<size_t>(w) * h);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
    uint8_t* p = &rgb[3 * (x + static_cast<size_t>(y) * w)];
    p[0] = static_cast<uint8_t>(x * 5 + (Rand8() >> 3));
    p[1] = static_cast<uint8_t>(y * 3 + (Rand8() >> 4));
    p[2] = static_cast<uint8_t>(((x / 8) ^ (y / 8)) * 51);
  }
  return rgb;
}

static void Save(const std::string& dir, const std::string& name, const std::string& data) {
  FILE* f = fopen((dir + "/" + name + ".jpg").c_str(), "wb");
  if (f == nullptr) { ++g_failures; return; }
  fwrite(data.data(), 1, data.size(), f);
  fclose(f);
}

struct CountingMemory : public sjpeg::MemoryManager {
  int allocs = 0, frees = 0, refuse_after = -1, refused = 0;
  void* Alloc(size_t size) override {
    if (refuse_after >= 0 && allocs >= refuse_after) { ++refused; return nullptr; }
    ++allocs;
    return malloc(size);
  }
  void Free(void* const ptr) override { if (ptr != nullptr) { ++frees; free(ptr); } }
};

struct FlakySink : public sjpeg::ByteSink {
  std::string data;
  size_t pos = 0;
  int commits = 0, fail_at;
  bool reset_called = false;
  explicit FlakySink(int fail) : fail_at(fail) {}
  bool Commit(size_t used, size_t extra, uint8_t** out) override {
    if (commits++ == fail_at) return false;
    pos += used;
    data.resize(pos + extra);
    *out = extra ? reinterpret_cast<uint8_t*>(&data[pos]) : nullptr;
    return true;
  }
  bool Finalize() override { data.resize(pos); return true; }
  void Reset() override { data.clear(); pos = 0; reset_called = true; }
};

// api_test <outdir> --auto <rgb file> <w> <h> [expect-fail]: the out-of-the-box calls of the reference
// (SjpegCompress, and sjpeg::Encode with a default EncoderParam = SJPEG_YUV_AUTO), with nothing but the
// library's own discovery of the riskiness table (file next to the library / SJPEG_HIP_RISKINESS_TABLE).
static int AutoMode(const std::string& dir, const char* path, int w, int h, bool expect_fail) {
  std::vector<uint8_t> rgb(3 * static_cast<size_t>(w) * h);
  FILE* f = fopen(path, "rb");
  if (f == nullptr || fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) { fprintf(stderr, "cannot read %s\n", path); return 99; }
  fclose(f);
  uint8_t* out = nullptr;
  const size_t n = SjpegCompress(rgb.data(), w, h, 75.f, &out);
  std::string dflt;
  const bool ok = sjpeg::Encode(rgb.data(), w, h, 3 * w, sjpeg::EncoderParam(), &dflt);
  if (expect_fail) {
    // no table anywhere: both must fail, loudly, never fall back to another colour mode
    CHECK(n == 0 && out == nullptr && !ok);
    CHECK(strstr(SjpegHipLastError(), "riskiness") != nullptr);
  } else {
    CHECK(n > 0 && out != nullptr && ok);
    if (n > 0) Save(dir, "compress_c1", std::string(reinterpret_cast<const char*>(out), n));
    Save(dir, "default_param_auto", dflt);
  }
  SjpegFreeBuffer(out);
  return g_failures;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  CHECK(SjpegVersion() == 0x000101);
  if (argc >= 6 && strcmp(argv[2], "--auto") == 0) {
    return AutoMode(dir, argv[3], atoi(argv[4]), atoi(argv[5]), argc >= 7 && strcmp(argv[6], "expect-fail") == 0);
  }

  // ---- parity cases through EncoderParam (default = adaptive quantization + optimised Huffman)
  const int W = 141, H = 99;
  const std::vector<uint8_t> rgb = Picture(W, H, 4242);
  const SjpegYUVMode modes[3] = {SJPEG_YUV_420, SJPEG_YUV_444, SJPEG_YUV_400};
  const char* mode_names[3] = {"420", "444", "400"};
  for (int m = 0; m < 3; ++m) {
    {
      sjpeg::EncoderParam param;                         // all defaults except the colour mode
      param.yuv_mode = modes[m];
      std::string out;
      CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
      Save(dir, std::string("default_") + mode_names[m], out);
    }
    {
      sjpeg::EncoderParam param(33.f);
      param.yuv_mode = modes[m];
      param.Huffman_compress = false;
      param.adaptive_quantization = true;
      param.quantization_bias = 0x60;
      param.qdelta_max_luma = 7;
      param.qdelta_max_chroma = 3;
      std::string out;
      CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
      Save(dir, std::string("q33_adaptive_bias60_d7_3_") + mode_names[m], out);
    }
  }
  {   // SetQuantization + SetLimitQuantization (the recompress recipe), vector sink
    uint8_t m[2][64];
    for (int i = 0; i < 64; ++i) { m[0][i] = static_cast<uint8_t>(3 + i); m[1][i] = static_cast<uint8_t>(5 + 2 * i); }
    sjpeg::EncoderParam param;
    param.yuv_mode = SJPEG_YUV_420;
    param.SetQuantization(m, 80.f);
    param.SetLimitQuantization(true);
    std::vector<uint8_t> out;
    std::shared_ptr<sjpeg::ByteSink> sink = sjpeg::MakeByteSink(&out);
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, sink.get()));
    Save(dir, "setquant_r80_limit_420", std::string(out.begin(), out.end()));
    CHECK(param.GetQuantMatrix(0)[0] == 4 && param.GetQuantMatrix(1)[63] == 164);
  }
  {   // metadata segments are written verbatim in front of the tables
    sjpeg::EncoderParam param(80.f);
    param.yuv_mode = SJPEG_YUV_444;
    param.Huffman_compress = false;
    param.adaptive_quantization = false;
    param.exif = std::string("II*\0fake-exif-payload", 20);
    param.iccp = std::string(70000, 'i');              // two APP2 chunks
    param.xmp = "<x:xmpmeta/>";
    param.app_markers = std::string("\xff\xe5\x00\x04zz", 6);
    std::string out;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
    Save(dir, "metadata_444", out);
    param.exif = std::string(70000, 'e');               // > 64 KiB: must fail
    CHECK(!sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
  }

  {   // use_trellis: only with Huffman_compress + adaptive_quantization (method 4 -> 7, src/api.cc:153-157)
    sjpeg::EncoderParam param(70.f);
    param.yuv_mode = SJPEG_YUV_420;
    param.use_trellis = true;
    std::string out, plain;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
    Save(dir, "trellis_q70_420", out);
    param.use_trellis = false;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &plain));
    CHECK(out != plain && out.size() <= plain.size());
    param.use_trellis = true;
    param.Huffman_compress = false;                    // method 3: the flag is ignored
    std::string a, b;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &a));
    param.use_trellis = false;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &b));
    CHECK(a == b);
  }

  // ---- multi-pass size / PSNR search (reference: src/dichotomy.cc; unit_test.cc TargetSize idea)
  {
    int idx = 0;
    const float size_targets[3] = {1500.f, 4000.f, 9000.f}, psnr_targets[3] = {30.f, 38.f, 45.f};
    for (int m = 0; m < 3; ++m) for (int huff = 0; huff < 2; ++huff) for (int adapt = 0; adapt < 2; ++adapt)
      for (int tm = 1; tm <= 2; ++tm) for (int t = 0; t < 3; ++t) for (int passes = 2; passes <= 6; passes += 4) {
        sjpeg::EncoderParam param(60.f);
        param.yuv_mode = modes[m];
        param.Huffman_compress = (huff != 0);
        param.adaptive_quantization = (adapt != 0);
        param.target_mode = static_cast<sjpeg::EncoderParam::TargetMode>(tm);
        param.target_value = (tm == 1) ? size_targets[t] : psnr_targets[t];
        param.passes = passes;
        param.tolerance = (tm == 1) ? 1.f : 0.1f;
        std::string out;
        CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
        char name[32];
        snprintf(name, sizeof(name), "search_%03d", idx++);
        Save(dir, name, out);
      }
    // the same search with trellis quantization (rate table accumulating over the passes)
    for (int m = 0; m < 3; ++m) for (int tm = 1; tm <= 2; ++tm) for (int t = 0; t < 3; ++t) for (int passes = 2; passes <= 6; passes += 4) {
      sjpeg::EncoderParam param(60.f);
      param.yuv_mode = modes[m];
      param.use_trellis = true;
      param.target_mode = static_cast<sjpeg::EncoderParam::TargetMode>(tm);
      param.target_value = (tm == 1) ? size_targets[t] : psnr_targets[t];
      param.passes = passes;
      param.tolerance = (tm == 1) ? 1.f : 0.1f;
      std::string out;
      CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
      char name[40];
      snprintf(name, sizeof(name), "search_trellis_%03d", idx++);
      Save(dir, name, out);
    }
    // a 10-pass size search lands close to the request, and closer than a single pass at the seed
    sjpeg::EncoderParam param(60.f);
    param.yuv_mode = SJPEG_YUV_420;
    param.target_mode = sjpeg::EncoderParam::TARGET_SIZE;
    param.target_value = 5000.f;
    param.passes = 10;
    param.tolerance = 1.f;
    std::string hit;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &hit));
    CHECK(hit.size() > 4500 && hit.size() < 5500);
    // user hook: called once per pass for both matrices, q/value reported back, Setup() veto is fatal
    struct CountingHook : public sjpeg::SearchHook {
      int setups = 0, matrices = 0, updates = 0;
      bool veto = false;
      bool Setup(const sjpeg::EncoderParam& p) override { ++setups; return !veto && sjpeg::SearchHook::Setup(p); }
      void NextMatrix(int i, uint8_t dst[64]) override { ++matrices; sjpeg::SearchHook::NextMatrix(i, dst); }
      bool Update(float r) override { ++updates; return sjpeg::SearchHook::Update(r); }
    } hook;
    param.search_hook = &hook;
    param.passes = 4;
    std::string hooked;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &hooked));
    CHECK(hook.setups == 1 && hook.updates >= 1 && hook.updates <= 4 && hook.matrices == 2 * hook.updates);
    CHECK(hook.value > 0.f && hook.q >= 0.f && hook.q <= 100.f && hook.for_size);
    Save(dir, "search_hooked", hooked);
    hook.veto = true;
    CHECK(!sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &hooked));
  }

  // ---- the other input layouts of the API (reference: src/sjpeg.h:300-349)
  {
    const int w = 37, h = 23, cw = (w + 1) / 2, ch = (h + 1) / 2;
    const std::vector<uint8_t> px = Picture(w, h, 777);          // reused as raw byte material
    std::vector<uint8_t> bgra(4 * w * h), yp(w * h), up(cw * ch), vp(cw * ch), uv(2 * cw * ch), u4(w * h), v4(w * h);
    for (int i = 0; i < w * h; ++i) {
      bgra[4 * i] = px[3 * i + 2]; bgra[4 * i + 1] = px[3 * i + 1]; bgra[4 * i + 2] = px[3 * i]; bgra[4 * i + 3] = 0x5a;
      yp[i] = px[3 * i]; u4[i] = px[3 * i + 1]; v4[i] = px[3 * i + 2];
    }
    for (int i = 0; i < cw * ch; ++i) { up[i] = px[5 * i % px.size()]; vp[i] = px[7 * i % px.size()]; uv[2 * i] = up[i]; uv[2 * i + 1] = vp[i]; }
    sjpeg::EncoderParam param(66.f);                              // default method 4
    std::string out;
    param.yuv_mode = SJPEG_YUV_444;
    CHECK(sjpeg::EncodeBGRA(bgra.data(), w, h, 4 * w, param, &out));
    Save(dir, "bgra_444_q66", out);
    std::vector<uint8_t> vec;
    std::shared_ptr<sjpeg::ByteSink> sink = sjpeg::MakeByteSink(&vec);
    param.yuv_mode = SJPEG_YUV_420;
    CHECK(sjpeg::EncodeGray(yp.data(), w, h, w, param, &out));
    Save(dir, "gray_q66", out);
    CHECK(sjpeg::EncodeNV12(yp.data(), w, uv.data(), 2 * cw, w, h, param, sink.get()));
    Save(dir, "nv12_q66", std::string(vec.begin(), vec.end()));
    sink = sjpeg::MakeByteSink(&vec);
    CHECK(sjpeg::EncodeNV21(yp.data(), w, uv.data(), 2 * cw, w, h, param, sink.get()));
    Save(dir, "nv21_q66", std::string(vec.begin(), vec.end()));
    sink = sjpeg::MakeByteSink(&vec);
    CHECK(sjpeg::EncodeYUV420(yp.data(), w, up.data(), cw, vp.data(), cw, w, h, param, sink.get()));
    Save(dir, "yuv420_q66", std::string(vec.begin(), vec.end()));
    sink = sjpeg::MakeByteSink(&vec);
    CHECK(sjpeg::EncodeYUV444(yp.data(), w, u4.data(), w, v4.data(), w, w, h, param, sink.get()));
    Save(dir, "yuv444_q66", std::string(vec.begin(), vec.end()));
    CHECK(!sjpeg::EncodeBGRA(bgra.data(), w, h, 4 * w - 1, param, &out));       // stride too small
    CHECK(!sjpeg::EncodeNV12(yp.data(), w, uv.data(), 2 * cw - 1, w, h, param, sink.get()));
    CHECK(!sjpeg::EncodeYUV420(yp.data(), w, nullptr, cw, vp.data(), cw, w, h, param, sink.get()));
  }

  // ---- quality ordering, method clamping
  {
    sjpeg::EncoderParam lo(30.f), hi(95.f);
    lo.yuv_mode = hi.yuv_mode = SJPEG_YUV_420;
    std::string a, b;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, lo, &a) && sjpeg::Encode(rgb.data(), W, H, 3 * W, hi, &b));
    CHECK(a.size() < b.size());
    uint8_t *p0 = nullptr, *pm = nullptr;
    const size_t n0 = SjpegEncode(rgb.data(), W, H, 3 * W, &p0, 75.f, 0, SJPEG_YUV_420);
    const size_t nm = SjpegEncode(rgb.data(), W, H, 3 * W, &pm, 75.f, -1, SJPEG_YUV_420);
    CHECK(n0 > 0 && n0 == nm && memcmp(p0, pm, n0) == 0);            // method < 0 clamps to 0
    delete[] p0;                                                      // new[] ownership
    SjpegFreeBuffer(pm);
  }

  // ---- invalid arguments (reference: unit_test.cc:165-193)
  {
    sjpeg::EncoderParam param;
    param.yuv_mode = SJPEG_YUV_420;
    std::string out;
    uint8_t* buf = nullptr;
    CHECK(!sjpeg::Encode(nullptr, W, H, 3 * W, param, &out));
    CHECK(!sjpeg::Encode(rgb.data(), 0, H, 3 * W, param, &out));
    CHECK(!sjpeg::Encode(rgb.data(), W, -3, 3 * W, param, &out));
    CHECK(!sjpeg::Encode(rgb.data(), W, H, 3 * W - 1, param, &out));
    CHECK(!sjpeg::Encode(rgb.data(), W, H, 3 * W, param, static_cast<std::string*>(nullptr)));
    CHECK(!sjpeg::Encode(rgb.data(), W, H, 3 * W, param, static_cast<sjpeg::ByteSink*>(nullptr)));
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, static_cast<uint8_t**>(nullptr)) == 0);
    CHECK(SjpegEncode(rgb.data(), W, H, 3 * W, &buf, 75.f, 0, static_cast<SjpegYUVMode>(11)) == 0);
    // SJPEG_YUV_AUTO works where the riskiness table is installed (next to the library, environment,
    // or setter) and fails loudly where it is not; never a silent other mode (--auto checks the bytes)
    param.yuv_mode = SJPEG_YUV_AUTO;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out) || strstr(SjpegHipLastError(), "riskiness") != nullptr);
  }

  // ---- large dimensions (reference: unit_test.cc:393-409): 65535 is legal
  {
    std::vector<uint8_t> row(3 * 65535, 0x55);
    uint8_t* buf = nullptr;
    const size_t n = SjpegEncode(row.data(), 65535, 1, 3 * 65535, &buf, 50.f, 0, SJPEG_YUV_444);
    int w = 0, h = 0;
    CHECK(n > 0 && SjpegDimensions(buf, n, &w, &h, nullptr) && w == 65535 && h == 1);
    SjpegFreeBuffer(buf);
  }

  // ---- strides: padding never leaks, bottom-up == flipped (reference: unit_test.cc:246-342)
  {
    sjpeg::EncoderParam param;
    param.yuv_mode = SJPEG_YUV_420;
    const int w = 17, h = 13, stride = 3 * w + 11;
    const std::vector<uint8_t> tight = Picture(w, h, 9);
    std::vector<uint8_t> padded(static_cast<size_t>(stride) * h, 0xEE), flipped(tight.size());
    for (int y = 0; y < h; ++y) {
      memcpy(&padded[static_cast<size_t>(y) * stride], &tight[3 * w * y], 3 * w);
      memcpy(&flipped[3 * w * (h - 1 - y)], &tight[3 * w * y], 3 * w);
    }
    std::string a, b, c;
    CHECK(sjpeg::Encode(tight.data(), w, h, 3 * w, param, &a));
    CHECK(sjpeg::Encode(padded.data(), w, h, stride, param, &b));
    CHECK(sjpeg::Encode(flipped.data() + 3 * w * (h - 1), w, h, -3 * w, param, &c));
    CHECK(a == b && a == c);
  }

  // ---- MemoryManager: used, balanced, and a refusal is fatal (reference: unit_test.cc:346-454)
  {
    sjpeg::EncoderParam param;
    param.yuv_mode = SJPEG_YUV_420;
    CountingMemory mem;
    param.memory = &mem;
    std::string out;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
    CHECK(mem.allocs > 0 && mem.allocs == mem.frees);
    for (int k = 0; k < 3; ++k) {
      CountingMemory failing;
      failing.refuse_after = k;
      param.memory = &failing;
      const bool ok = sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out);
      CHECK(ok == (failing.refused == 0));
      CHECK(failing.allocs == failing.frees);
    }
  }

  // ---- a failing sink fails the encode and gets Reset() (reference: unit_test.cc:564-603)
  {
    sjpeg::EncoderParam param;
    param.yuv_mode = SJPEG_YUV_444;
    for (int fail_at = 0; fail_at < 3; ++fail_at) {
      FlakySink sink(fail_at);
      const bool ok = sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &sink);
      CHECK(ok == (sink.commits <= fail_at));
      if (!ok) CHECK(sink.reset_called && sink.data.empty());
    }
  }

  // ---- parsers (reference: unit_test.cc:456-484, 625-646)
  {
    sjpeg::EncoderParam param(62.f);
    param.yuv_mode = SJPEG_YUV_420;
    param.adaptive_quantization = false;
    std::string out;
    CHECK(sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out));
    int w = 0, h = 0, is420 = 0;
    CHECK(SjpegDimensions(out, &w, &h, &is420) && w == W && h == H && is420 == 1);
    for (size_t cut = 0; cut < 660 && cut < out.size(); ++cut) {
      (void)SjpegDimensions(reinterpret_cast<const uint8_t*>(out.data()), cut, &w, &h, &is420);
    }
    uint8_t q[2][64];
    CHECK(SjpegFindQuantizer(out, q) == 2);
    CHECK(memcmp(q[0], param.GetQuantMatrix(0), 64) == 0 && memcmp(q[1], param.GetQuantMatrix(1), 64) == 0);
    const float est = SjpegEstimateQuality(q[0], false);
    CHECK(est >= 61.f && est <= 63.f);
  }

  // ---- concurrent encoders are deterministic (reference: unit_test.cc:114-131)
  {
    const int kThreads = 8;
    std::vector<std::string> out(kThreads);
    std::vector<std::thread> th;
    for (int t = 0; t < kThreads; ++t) {
      th.emplace_back([&, t]() {
        sjpeg::EncoderParam param(72.f);
        param.yuv_mode = SJPEG_YUV_420;
        sjpeg::Encode(rgb.data(), W, H, 3 * W, param, &out[t]);
      });
    }
    for (auto& t : th) t.join();
    for (int t = 0; t < kThreads; ++t) CHECK(!out[t].empty() && out[t] == out[0]);
    Save(dir, "threads_q72_420", out[0]);
  }

  printf("api_test: %d failure(s)\n", g_failures);
  return g_failures;
}
