// jpeg_host.cc -- see jpeg_host.h.  Host-only, negligible cost; bytes must equal the
// reference's.  References are to /root/reference/src.
#include "jpeg_host.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace sjpeg_host {

const uint8_t kZigzag[64] = {       // JPEG Figure A.6 (quantize.cc:32-41)
  0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

const uint8_t kAnnexK1[2][64] = {   // JPEG Annex K.1 (quantize.cc:57-75)
  {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
  {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
   24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};

namespace {

// JPEG Annex K.3 typical Huffman tables (entropy.cc:31-82): BITS, then HUFFVAL for AC
// (the DC symbol list is simply 0..11).
const uint8_t kDcBitsLuma[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcBitsChroma[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kAcBitsLuma[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
const uint8_t kAcBitsChroma[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
const uint8_t kAcValsLuma[162] = {
  0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
  0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
  0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
  0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
  0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
  0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
  0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
  0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
  0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
  0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
  0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcValsChroma[162] = {
  0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
  0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
  0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
  0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
  0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63,
  0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
  0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
  0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
  0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
  0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
  0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

HuffSpec MakeSpec(const uint8_t bits[16], const uint8_t* vals, int n) {
  HuffSpec s;
  memset(&s, 0, sizeof(s));
  memcpy(s.bits, bits, 16);
  if (vals != nullptr) memcpy(s.syms, vals, n);
  else for (int i = 0; i < n; ++i) s.syms[i] = static_cast<uint8_t>(i);   // DC: 0..11
  s.nsyms = n;
  return s;
}

void Put16(std::vector<uint8_t>* o, uint32_t v) {
  o->push_back(static_cast<uint8_t>(v >> 8));
  o->push_back(static_cast<uint8_t>(v));
}
void Put32(std::vector<uint8_t>* o, uint32_t v) { Put16(o, v >> 16); Put16(o, v & 0xffff); }
void PutBytes(std::vector<uint8_t>* o, const void* p, size_t n) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  o->insert(o->end(), b, b + n);
}

}  // namespace

const HuffSpec& DefaultHuff(int type, int comp) {
  static const HuffSpec specs[4] = {
    MakeSpec(kDcBitsLuma, nullptr, 12), MakeSpec(kDcBitsChroma, nullptr, 12),
    MakeSpec(kAcBitsLuma, kAcValsLuma, 162), MakeSpec(kAcBitsChroma, kAcValsChroma, 162)};
  return specs[type * 2 + comp];
}

float QualityToScale(float q) {
  // libjpeg-6b mapping, float + floorf exactly as quantize.cc:77-82
  float s;
  if (q <= 0) s = 5000;
  else if (q < 50) s = 5000 / q;
  else if (q < 100) s = 2 * (100 - q);
  else s = 0;
  return floorf(s);
}

void ScaleMatrix(const uint8_t in[64], float scale_percent, uint8_t out[64]) {
  const float f = scale_percent / 100.f;            // quantize.cc:90
  for (int i = 0; i < 64; ++i) {
    const int v = static_cast<int>(in[i] * f + .5f);
    out[i] = static_cast<uint8_t>(v < 1 ? 1 : v > 255 ? 255 : v);
  }
}

void MinMatrix(const uint8_t in[64], int tolerance, uint8_t out[64]) {
  for (int i = 0; i < 64; ++i) {
    const int v = (in[i] * (256 - tolerance)) >> 8;
    out[i] = static_cast<uint8_t>(v < 1 ? 1 : v > 255 ? 255 : v);
  }
}

void FinalizeQuantizer(uint8_t quant[64], const uint8_t min_quant[64], int q_bias, int idx,
                       sjpeg_hip_scan_tables* t) {
  for (int i = 0; i < 64; ++i) quant[i] = std::max(quant[i], min_quant[i]);
  for (int i = 0; i < 64; ++i) {
    const uint32_t v = quant[i];
    // 16-bit reciprocal; v == 1 cannot be represented, so it becomes 0xffff with the
    // neutral bias 0x80 (quantize.cc:128-139).  DC always uses 0x80.
    const uint32_t recip = (v == 1) ? 0xffffu : ((1u << 16) + v / 2) / v;
    const uint32_t bias8 = (v == 1 || i == 0) ? 0x80u : static_cast<uint32_t>(q_bias);
    t->quant[idx][i] = static_cast<uint8_t>(v);
    t->iquant[idx][i] = static_cast<uint16_t>(recip);
    t->bias[idx][i] = static_cast<uint16_t>((((bias8 * v) << 4) + 128) >> 8);
  }
}

int BuildCodes(const HuffSpec& spec, uint32_t* tab) {
  uint32_t code = 0;
  int k = 0;
  for (int len = 1; len <= 16; ++len, code <<= 1) {
    for (int n = spec.bits[len - 1]; n > 0; --n, ++code) {
      tab[spec.syms[k++]] = (code << 16) | static_cast<uint32_t>(len);
    }
  }
  return k;
}

void InstallCodes(const HuffSpec* dc[2], const HuffSpec* ac[2], int ntables,
                  sjpeg_hip_scan_tables* t) {
  memset(t->dc_codes, 0, sizeof(t->dc_codes));
  memset(t->ac_codes, 0, sizeof(t->ac_codes));
  for (int c = 0; c < ntables; ++c) {
    BuildCodes(*dc[c], t->dc_codes[c]);
    BuildCodes(*ac[c], t->ac_codes[c]);
  }
}

bool LayoutFor(int yuv_mode, FrameLayout* L) {
  memset(L, 0, sizeof(*L));
  L->block_w = L->block_h = 8;
  L->sampling[0] = L->sampling[1] = L->sampling[2] = 0x11;
  L->quant_idx[1] = L->quant_idx[2] = 1;
  switch (yuv_mode) {
    case SJPEG_HIP_YUV420:
      L->nb_comps = 3; L->mcu_blocks = 6; L->block_w = L->block_h = 16; L->sampling[0] = 0x22;
      return true;
    case SJPEG_HIP_YUV444: L->nb_comps = 3; L->mcu_blocks = 3; return true;
    case SJPEG_HIP_YUV400: L->nb_comps = 1; L->mcu_blocks = 1; return true;
    default: return false;
  }
}

bool AppendHeaders(int W, int H, int yuv_mode, const uint8_t quant[2][64],
                   const HuffSpec* dc[2], const HuffSpec* ac[2], const Metadata* meta,
                   std::vector<uint8_t>* o) {
  FrameLayout L;
  if (!LayoutFor(yuv_mode, &L)) return false;
  // SOI + JFIF APP0: v1.01, density 1:1 (no units), no thumbnail (headers.cc:48-55)
  static const uint8_t kJfif[20] = {0xff, 0xd8, 0xff, 0xe0, 0x00, 0x10, 'J', 'F', 'I', 'F',
                                    0x00, 0x01, 0x01, 0x00, 0x00, 0x01, 0x00, 0x01, 0x00, 0x00};
  PutBytes(o, kJfif, sizeof(kJfif));
  if (meta != nullptr) {
    // raw application markers, verbatim (headers.cc:63-70)
    if (!meta->app_markers.empty()) PutBytes(o, meta->app_markers.data(), meta->app_markers.size());
    // EXIF in one APP1 (headers.cc:72-85)
    if (!meta->exif.empty()) {
      const size_t seg = meta->exif.size() + 6 + 2;
      if (seg > 0xffff) return false;
      Put16(o, 0xffe1); Put16(o, static_cast<uint32_t>(seg));
      PutBytes(o, "Exif\0\0", 6);
      PutBytes(o, meta->exif.data(), meta->exif.size());
    }
    // ICC profile in numbered APP2 chunks (headers.cc:87-113)
    if (!meta->iccp.empty()) {
      const size_t kMax = 0xffff - 12 - 4;
      const size_t nchunks = (meta->iccp.size() + kMax - 1) / kMax;
      if (nchunks >= 256) return false;
      size_t pos = 0;
      for (size_t seq = 1; pos < meta->iccp.size(); ++seq) {
        const size_t n = std::min(kMax, meta->iccp.size() - pos);
        Put16(o, 0xffe2); Put16(o, static_cast<uint32_t>(n + 12 + 4));
        PutBytes(o, "ICC_PROFILE", 12);
        o->push_back(static_cast<uint8_t>(seq));
        o->push_back(static_cast<uint8_t>(nchunks));
        PutBytes(o, meta->iccp.data() + pos, n);
        pos += n;
      }
    }
    // XMP: a single APP1 when it fits (headers.cc:162-180)
    if (!meta->xmp.empty()) {
      static const char kXmp[] = "http://ns.adobe.com/xap/1.0/";
      const size_t seg = 2 + meta->xmp.size() + sizeof(kXmp);
      if (seg > 0xffff) return false;      // extended XMP: not supported by this build
      Put16(o, 0xffe1); Put16(o, static_cast<uint32_t>(seg));
      PutBytes(o, kXmp, sizeof(kXmp));
      PutBytes(o, meta->xmp.data(), meta->xmp.size());
    }
  }
  // DQT, 8-bit precision, zig-zag order (headers.cc:182-196)
  const int nq = (yuv_mode == SJPEG_HIP_YUV400) ? 1 : 2;
  Put16(o, 0xffdb); Put16(o, nq * 65 + 2);
  for (int n = 0; n < nq; ++n) {
    o->push_back(static_cast<uint8_t>(n));
    for (int i = 0; i < 64; ++i) o->push_back(quant[n][kZigzag[i]]);
  }
  // SOF0 (headers.cc:202-219)
  Put16(o, 0xffc0); Put16(o, 3 * L.nb_comps + 8);
  o->push_back(8); Put16(o, H); Put16(o, W); o->push_back(static_cast<uint8_t>(L.nb_comps));
  for (int c = 0; c < L.nb_comps; ++c) {
    o->push_back(static_cast<uint8_t>(c + 1));
    o->push_back(static_cast<uint8_t>(L.sampling[c]));
    o->push_back(static_cast<uint8_t>(L.quant_idx[c]));
  }
  // DHT: one segment per table, DC then AC, luma then chroma (headers.cc:221-238)
  const int nt = (L.nb_comps == 1) ? 1 : 2;
  for (int c = 0; c < nt; ++c) {
    for (int type = 0; type <= 1; ++type) {
      const HuffSpec& h = type ? *ac[c] : *dc[c];
      Put16(o, 0xffc4); Put16(o, 3 + 16 + h.nsyms);
      o->push_back(static_cast<uint8_t>((type << 4) | c));
      PutBytes(o, h.bits, 16);
      PutBytes(o, h.syms, h.nsyms);
    }
  }
  // SOS, one scan with all components, full spectral range (headers.cc:242-258)
  Put16(o, 0xffda); Put16(o, 6 + 2 * L.nb_comps); o->push_back(static_cast<uint8_t>(L.nb_comps));
  for (int c = 0; c < L.nb_comps; ++c) {
    o->push_back(static_cast<uint8_t>(c + 1));
    o->push_back(static_cast<uint8_t>(L.quant_idx[c] * 0x11));
  }
  o->push_back(0); o->push_back(63); o->push_back(0);
  (void)Put32;
  return true;
}

}  // namespace sjpeg_host

// ------------------------------------------------------------------------------------------
// Adaptive quantization (method >= 3) and optimised Huffman tables (method 1, 4..): host-side
// analysis of statistics gathered on the GPU.  Floating-point expressions keep the reference's
// types and evaluation order so that every integer decision comes out identical.

#include <float.h>
#include <stdlib.h>

namespace sjpeg_host {

namespace {
enum { kQDeltaMin = -12, kQDeltaMax = 12, kQSize = kQDeltaMax + 1 - kQDeltaMin };   // sjpegi.h:269-273
// Gaussian weights, sigma ~ 3, centred on delta 0 (histogram.cc:117-124)
const float kDeltaWeight[kQSize] = {0, 0, 0, 0, 0, 1, 5, 16, 43, 94, 164, 228, 255,
                                    228, 164, 94, 43, 16, 5, 1, 0, 0, 0, 0, 0};
int BitLength(int v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
}  // namespace

void AdaptQuantMatrices(const uint32_t hist[2][64][128], int nb_comps, uint8_t quant[2][64],
                        const uint8_t min_quant[2][64], int qdelta_max_luma, int qdelta_max_chroma) {
  const double r_limit = 0.5;                       // kCorrelationThreshold
  for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
    const int delta_max = ((idx == 0) ? qdelta_max_luma : qdelta_max_chroma) - kQDeltaMin;
    static thread_local float sizes[64][kQSize];
    static thread_local float distortions[64][kQSize];
    double num = 0., den = 0.;
    uint64_t omit = 0x103ull;                       // DC and its two neighbours are never touched
    for (int pos = 0; pos < 64; ++pos) {
      if (omit & (1ull << pos)) continue;
      const int dq0 = quant[idx][pos];
      const int min_dq0 = min_quant[idx][pos];
      const int bias = 1 << 16 >> 1;
      const uint32_t* const h = hist[idx][pos];
      int total = 0, last = 0;
      for (int i = 0; i < 128; ++i) {
        total += static_cast<int>(h[i]);
        if (h[i]) last = i + 1;
      }
      if (total < 0.5 * last) {                     // kDensityThreshold
        omit |= 1ull << pos;
        continue;
      }
      double sw = 0., sx = 0., sxx = 0., syy1 = 0., sy1 = 0., sxy1 = 0., sy2 = 0., sxy2 = 0.;
      for (int delta = 0; delta < kQSize; ++delta) {
        double bsum = 0., dsum = 0.;
        const int dq = dq0 + (delta + kQDeltaMin);
        if (dq >= min_dq0 && dq <= 255) {
          const int idq = ((1 << 16) + dq - 1) / dq;
          for (int i = 0; i < last; ++i) {
            if (h[i]) {
              const int hi = static_cast<int>(h[i]);
              const int v = (i << 2) + 2;           // bin centroid: HSHIFT = 2, HHALF = 2
              const int qv = (v * idq + bias) >> 16;
              if (qv) {
                const int bits = BitLength(qv);
                const int dqv = qv * dq;
                const int error = (v - dqv) * (v - dqv);
                bsum += hi * bits;
                dsum += hi * error;
              } else {
                dsum += hi * v * v;
              }
            }
          }
          distortions[pos][delta] = static_cast<float>(dsum);
          sizes[pos][delta] = static_cast<float>(bsum);
          const double w = kDeltaWeight[delta];
          if (w > 0.) {
            const double x = static_cast<double>(delta + kQDeltaMin);
            sw += w;
            sx += w * x;
            sxx += w * x * x;
            sy1 += w * dsum;
            syy1 += w * dsum * dsum;
            sy2 += w * bsum;
            sxy1 += w * dsum * x;
            sxy2 += w * bsum * x;
          }
        } else {
          distortions[pos][delta] = FLT_MAX;
          sizes[pos][delta] = 0;
        }
      }
      const double cov_xy1 = sw * sxy1 - sx * sy1;
      if (cov_xy1 * cov_xy1 < r_limit * (sw * sxx - sx * sx) * (sw * syy1 - sy1 * sy1)) {
        omit |= 1ull << pos;
        continue;
      }
      num += cov_xy1;
      den += sw * sxy2 - sx * sy2;
    }
    double lambda = 0x80;                           // HLAMBDA
    if (num > 1000. && den < -10.) {
      lambda = -num / den;
      if (lambda < 1.) lambda = 1.;
    }
    for (int pos = 0; pos < 64; ++pos) {
      if (omit & (1ull << pos)) continue;
      float best_score = FLT_MAX;
      int best_dq = 0;
      for (int delta = 0; delta <= delta_max; ++delta) {
        if (distortions[pos][delta] < FLT_MAX) {
          const float score = distortions[pos][delta] + lambda * sizes[pos][delta];
          if (score < best_score) {
            best_score = score;
            best_dq = delta + kQDeltaMin;
          }
        }
      }
      quant[idx][pos] = static_cast<uint8_t>(quant[idx][pos] + best_dq);
    }
  }
}

void BuildOptimalSpec(const uint32_t* freq, int size, HuffSpec* out) {
  enum { kMaxBits = 32, kMaxCodeSize = 16 };
  int codesizes[257], chain[257], chain_end[257];   // chain_end: index of the tail of i's chain
  uint64_t sorted[257];
  int nb_syms = 0;
  for (int i = 0; i < size; ++i) {
    if (freq[i] > 0) sorted[nb_syms++] = (static_cast<uint64_t>(freq[i]) << 9) | static_cast<uint64_t>(i);
    codesizes[i] = 0; chain[i] = -1; chain_end[i] = i;
  }
  out->nsyms = nb_syms;
  // decreasing (frequency, symbol): keys are unique, so any correct sort gives the same order
  std::sort(sorted, sorted + nb_syms, [](uint64_t a, uint64_t b) { return a > b; });
  // pseudo symbol of lowest frequency: takes the all-ones code, which JPEG forbids
  sorted[nb_syms++] = (1ull << 9) | static_cast<uint64_t>(size);
  codesizes[size] = 0; chain[size] = -1; chain_end[size] = size;
  for (int nb = nb_syms - 1; nb >= 1; --nb) {       // Huffman merging, least frequent pair first
    const uint64_t s1 = sorted[nb - 1], s2 = sorted[nb];
    int i = static_cast<int>(s1 & 0x1ff);
    const int j = static_cast<int>(s2 & 0x1ff);
    chain[chain_end[i]] = j;
    chain_end[i] = chain_end[j];
    for (int t = i; t >= 0; t = chain[t]) ++codesizes[t];
    const uint64_t merged = s1 + (s2 & ~0x1ffull);
    int k = nb - 1;
    while (k > 0 && sorted[k - 1] < merged) { sorted[k] = sorted[k - 1]; --k; }
    sorted[k] = merged;
  }
  uint8_t bits[kMaxBits];
  memset(bits, 0, sizeof(bits));
  int max_bits = 0;
  for (int i = 0; i <= size; ++i) {
    int s = codesizes[i];
    if (s > 0) {
      if (s > kMaxBits) { s = kMaxBits; codesizes[i] = kMaxBits; }
      ++bits[s - 1];
      if (s > max_bits) max_bits = s;
    }
  }
  int start[kMaxBits], position = 0;
  for (int i = 0; i < max_bits; ++i) { start[i] = position; position += bits[i]; }
  memset(out->syms, 0, sizeof(out->syms));
  for (int sym = 0; sym < size; ++sym) {            // symbols by increasing code length
    const int s = codesizes[sym];
    if (s > 0) out->syms[start[s - 1]++] = static_cast<uint8_t>(sym);
  }
  for (int l = max_bits - 1; l >= kMaxCodeSize; --l) {   // limit to 16 bits (Annex K.2 style)
    while (bits[l] > 0) {
      int k = l - 2;
      while (bits[k] == 0) --k;
      bits[l] -= 2; bits[l - 1] += 1; bits[k] -= 1; bits[k + 1] += 2;
    }
  }
  max_bits = kMaxCodeSize;
  while (bits[--max_bits] == 0) {}
  --bits[max_bits];                                 // drop the pseudo symbol
  for (int i = 0; i < kMaxCodeSize; ++i) out->bits[i] = bits[i];
}

}  // namespace sjpeg_host
