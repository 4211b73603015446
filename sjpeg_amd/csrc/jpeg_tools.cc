// jpeg_tools.cc -- host-side JPEG parsing helpers of the public API: pure CPU utilities
// callers use BEFORE encoding (e.g. the recompress recipe, reference examples/sjpeg.cc:262-286).
// Behaviour follows /root/reference/src/jpeg_tools.cc:52-167; never reads out of bounds.
#include <string.h>

#include <string>

#include "jpeg_host.h"
#include "sjpeg.h"

namespace {

inline uint32_t Be16(const uint8_t* p) { return (static_cast<uint32_t>(p[0]) << 8) | p[1]; }

// Position of the first SOF0/SOF1 marker, or `size` if none; at least 8 readable bytes are
// guaranteed behind a returned position (reference: GetSOFData, jpeg_tools.cc:35-50).
size_t FindSof(const uint8_t* d, size_t size) {
  if (d == nullptr || size < 10) return size;
  const size_t limit = size - 8;
  size_t pos = 2;                                   // skip SOI
  while (pos < limit && d[pos] != 0xff) ++pos;
  while (pos < limit) {
    const uint32_t marker = Be16(d + pos);
    if (marker == 0xffc0 || marker == 0xffc1) return pos;
    pos += 2 + Be16(d + pos + 2);
  }
  return size;
}

}  // namespace

extern "C" {

bool SjpegDimensions(const uint8_t* data, size_t size, int* width, int* height, int* is_yuv420) {
  const size_t pos = FindSof(data, size);
  if (pos >= size) return false;
  const uint8_t* s = data + pos;
  const size_t left = size - pos;
  if (left < 11) return false;
  if (height != nullptr) *height = static_cast<int>(Be16(s + 5));
  if (width != nullptr) *width = static_cast<int>(Be16(s + 7));
  if (is_yuv420 != nullptr) {
    const size_t ncomp = s[9];
    *is_yuv420 = (ncomp == 3);
    if (left < 11 + 3 * ncomp) return false;
    for (int c = 0; *is_yuv420 && c < 3; ++c) {
      *is_yuv420 &= (s[11 + 3 * c] == (c == 0 ? 0x22 : 0x11));
    }
  }
  return true;
}

int SjpegFindQuantizer(const uint8_t* d, size_t size, uint8_t quant[2][64]) {
  memset(quant[0], 0, 64);
  memset(quant[1], 0, 64);
  if (d == nullptr || size < 69 || d[0] != 0xff || d[1] != 0xd8) return 0;
  const size_t limit = size - 8;
  size_t pos = 2;
  while (pos < limit && d[pos] != 0xff) ++pos;
  unsigned seen = 0;
  while (pos < limit) {
    const uint32_t marker = Be16(d + pos);
    const size_t seg = 2 + Be16(d + pos + 2);
    if (pos + seg > limit) break;
    if (marker == 0xffda) break;                    // tables precede the first scan
    if (marker == 0xffdb) {
      size_t i = 4;
      while (i + 1 < seg) {
        const int pq = d[pos + i] >> 4, tq = d[pos + i] & 0x0f;
        if (pq > 1 || tq > 3) return 0;             // ITU T.81 B.2.4.1
        const size_t msize = 64 * pq + 65;
        if (i + msize > seg) return 0;
        if (tq < 2) {
          for (int j = 0; j < 64; ++j) {
            int v = pq ? static_cast<int>(Be16(d + pos + i + 1 + 2 * j)) : d[pos + i + 1 + j];
            if (v > 255) v = 255;                   // 16-bit tables are clamped
            quant[tq][sjpeg_host::kZigzag[j]] = static_cast<uint8_t>(v < 1 ? 1 : v);
          }
        }
        seen |= 1u << tq;
        i += msize;
      }
    }
    pos += seg;
  }
  return static_cast<int>((seen & 1) + ((seen >> 1) & 1) + ((seen >> 2) & 1) + ((seen >> 3) & 1));
}

void SjpegQuantMatrix(float quality, bool for_chroma, uint8_t matrix[64]) {
  sjpeg_host::ScaleMatrix(sjpeg_host::kAnnexK1[for_chroma ? 1 : 0],
                          sjpeg_host::QualityToScale(quality), matrix);
}

float SjpegEstimateQuality(const uint8_t matrix[64], bool for_chroma) {
  // exhaustive over the 101 integer qualities, squared error, early exit
  int best_q = 0;
  float best = 256.f * 256 * 64 + 1;
  for (int q = 0; q <= 100; ++q) {
    uint8_t m[64];
    SjpegQuantMatrix(static_cast<float>(q), for_chroma, m);
    float score = 0;
    for (int i = 0; i < 64; ++i) {
      const float d = static_cast<float>(m[i]) - static_cast<float>(matrix[i]);
      score += d * d;
      if (score > best) break;
    }
    if (score < best) { best = score; best_q = q; }
  }
  return static_cast<float>(best_q);
}

}  // extern "C"

bool SjpegDimensions(const std::string& jpeg_data, int* width, int* height, int* is_yuv420) {
  return SjpegDimensions(reinterpret_cast<const uint8_t*>(jpeg_data.data()), jpeg_data.size(),
                         width, height, is_yuv420);
}

int SjpegFindQuantizer(const std::string& jpeg_data, uint8_t quant[2][64]) {
  return SjpegFindQuantizer(reinterpret_cast<const uint8_t*>(jpeg_data.data()), jpeg_data.size(), quant);
}
