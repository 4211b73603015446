// The batch entry of the C-ABI (sjpeg_hip_encode_batch_src) from a plain C++ process: batches of device-resident frames
// with every analysis method, against the host API's single-picture encode of the same frame (SjpegEncode: another code
// path of the library, byte-identical by contract).  Built twice by tools/san_engine.sh: as is, and against a library
// whose engine (scan_engine.hip's host half: buffers, streams, child engines, the lanes' state machine) is compiled with
// AddressSanitizer.  The environment decides how the batch is cut (SJPEG_HIP_BATCH_JOB_MPIX / _LANES / _NJOBS).
//   batch_lanes_test [rounds]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sjpeg.h"
#include "sjpeg_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)

static uint32_t g_seed = 12345;
static uint32_t Rnd() { g_seed = g_seed * 1103515245u + 12345u; return (g_seed >> 16) & 0x7fff; }

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 6;
  sjpeg_hip_engine* eng = nullptr;
  if (sjpeg_hip_engine_create(0, &eng) != 0) { fprintf(stderr, "engine: %s\n", sjpeg_hip_last_error()); return 2; }
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int bad = 0, frames_done = 0;
  for (int r = 0; r < rounds; ++r) {
    const int w = 24 + static_cast<int>(Rnd() % 300), h = 16 + static_cast<int>(Rnd() % 200);
    const int n = 1 + static_cast<int>(Rnd() % 29);
    const int mode = (r % 3 == 0) ? SJPEG_HIP_YUV420 : (r % 3 == 1) ? SJPEG_HIP_YUV444 : SJPEG_HIP_YUV400;
    const int method = 1 + r % 6;
    const float q = 20.f + static_cast<float>(Rnd() % 78);
    const size_t fbytes = static_cast<size_t>(w) * h * 3;
    std::vector<uint8_t> px(fbytes * n);
    for (int f = 0; f < n; ++f) {
      uint8_t* p = &px[fbytes * f];
      const int kind = static_cast<int>(Rnd() % 3);
      for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x, p += 3) {
          if (kind == 0) { p[0] = Rnd() & 255; p[1] = Rnd() & 255; p[2] = Rnd() & 255; }
          else if (kind == 1) { p[0] = (x * 5 + (Rnd() & 31)) & 255; p[1] = (y * 3 + (Rnd() & 15)) & 255; p[2] = (((x / 8) ^ (y / 8)) * 51) & 255; }
          else { p[0] = p[1] = p[2] = static_cast<uint8_t>(37 * f); }
        }
      }
    }
    uint8_t* d_px = nullptr;
    CHECK(hipMalloc(&d_px, px.size()));
    CHECK(hipMemcpy(d_px, px.data(), px.size(), hipMemcpyHostToDevice));
    const size_t stride = (sjpeg_hip_frame_bound(w, h, mode, 2048) + 15) & ~size_t(15);
    uint8_t* d_out = nullptr;
    uint64_t* d_sizes = nullptr;
    CHECK(hipMalloc(&d_out, stride * n));
    CHECK(hipMalloc(&d_sizes, sizeof(uint64_t) * n));
    sjpeg_hip_source src;
    memset(&src, 0, sizeof(src));
    src.format = SJPEG_HIP_SRC_RGB;
    src.plane[0] = d_px; src.row_stride[0] = 3ll * w; src.frame_stride[0] = static_cast<int64_t>(fbytes);
    uint8_t qm[2][64];
    sjpeg_hip_quality_matrices(q, qm);
    for (int rep = 0; rep < 2; ++rep) {            // (back to back, no host wait between the two)
      if (sjpeg_hip_encode_batch_src(eng, &src, w, h, mode, n, qm, nullptr, 0x78, method, 12, 1, d_out, stride, d_sizes, st) != 0) {
        fprintf(stderr, "batch: %s\n", sjpeg_hip_last_error());
        return 2;
      }
    }
    CHECK(hipStreamSynchronize(st));
    std::vector<uint64_t> sizes(n);
    CHECK(hipMemcpy(sizes.data(), d_sizes, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
    std::vector<uint8_t> got;
    for (int f = 0; f < n; ++f) {
      got.resize(sizes[f]);
      if (sizes[f] == 0) { ++bad; fprintf(stderr, "frame %d: size 0\n", f); continue; }
      CHECK(hipMemcpy(got.data(), d_out + stride * f, sizes[f], hipMemcpyDeviceToHost));
      uint8_t* ref = nullptr;
      const SjpegYUVMode ym = mode == SJPEG_HIP_YUV420 ? SJPEG_YUV_420 : mode == SJPEG_HIP_YUV444 ? SJPEG_YUV_444 : SJPEG_YUV_400;
      const size_t rs = SjpegEncode(&px[fbytes * f], w, h, 3 * w, &ref, q, method, ym);
      if (rs != sizes[f] || memcmp(ref, got.data(), rs) != 0) { ++bad; fprintf(stderr, "MISMATCH round %d frame %d (%dx%d mode %d method %d)\n", r, f, w, h, mode, method); }
      SjpegFreeBuffer(ref);
      ++frames_done;
    }
    if (r % 2 == 1) sjpeg_hip_engine_trim(eng);     // (the child engines' scratch goes and comes back)
    CHECK(hipFree(d_px)); CHECK(hipFree(d_out)); CHECK(hipFree(d_sizes));
  }
  sjpeg_hip_engine_destroy(eng);
  CHECK(hipStreamDestroy(st));
  printf("batch lanes test: %d rounds, %d frames, mismatches: %d\n", rounds, frames_done, bad);
  return bad == 0 ? 0 : 1;
}
