// jpeg_host.cc -- see jpeg_host.h.  Host-only, negligible cost; bytes must equal the
// reference's.  References are to /root/reference/src.
#include "jpeg_host.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

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

// MD5 (RFC 1321) of the extension part of a long XMP packet: its upper-case hex digest is the
// GUID that ties the APP1 extension chunks to the main packet (headers.cc:114-160).
static void Md5Hex(const uint8_t* data, size_t size, char out[32]) {
  static const uint32_t kSine[64] = {
      0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u,
      0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u,
      0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u,
      0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au,
      0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
      0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u,
      0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u,
      0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};
  static const int kRot[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
  uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
  const uint64_t bit_len = static_cast<uint64_t>(size) * 8;
  const size_t padded = ((size + 8) / 64 + 1) * 64;
  std::vector<uint8_t> msg(padded, 0);
  if (size > 0) memcpy(msg.data(), data, size);
  msg[size] = 0x80;
  for (int i = 0; i < 8; ++i) msg[padded - 8 + i] = static_cast<uint8_t>(bit_len >> (8 * i));
  for (size_t off = 0; off < padded; off += 64) {
    uint32_t w[16];
    for (int i = 0; i < 16; ++i) {
      const uint8_t* p = &msg[off + 4 * i];
      w[i] = p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24);
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      switch (i >> 4) {
        case 0: f = (b & c) | (~b & d); g = i; break;
        case 1: f = (d & b) | (~d & c); g = (5 * i + 1) & 15; break;
        case 2: f = b ^ c ^ d; g = (3 * i + 5) & 15; break;
        default: f = c ^ (b | ~d); g = (7 * i) & 15; break;
      }
      const uint32_t t = a + f + kSine[i] + w[g];
      const int r = kRot[i >> 4][i & 3];
      a = d; d = c; c = b;
      b = b + ((t << r) | (t >> (32 - r)));
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
  }
  static const char kHex[] = "0123456789ABCDEF";
  for (int i = 0; i < 16; ++i) {
    const uint8_t byte = static_cast<uint8_t>(h[i >> 2] >> (8 * (i & 3)));
    out[2 * i] = kHex[byte >> 4];
    out[2 * i + 1] = kHex[byte & 15];
  }
}

// WriteXMP / WriteXMPExtended (headers.cc:114-180)
static bool AppendXmp(const std::string& xmp, uint16_t xmp_split, std::vector<uint8_t>* o) {
  static const char kXmp[] = "http://ns.adobe.com/xap/1.0/";
  const size_t seg = 2 + xmp.size() + sizeof(kXmp);
  if (seg <= 0xffff) {
    Put16(o, 0xffe1); Put16(o, static_cast<uint32_t>(seg));
    PutBytes(o, kXmp, sizeof(kXmp));
    PutBytes(o, xmp.data(), xmp.size());
    return true;
  }
  // too long for one APP1: main packet (its HasExtendedXMP attribute receives the MD5 of the
  // rest) + numbered extension chunks
  const size_t kMainSize = 65503;
  if (xmp.size() > (1u << 31)) return false;
  size_t split = (xmp_split == 0) ? kMainSize : xmp_split;
  split = std::min(split, xmp.size());
  static const char kNote[] = "xmpNote:HasExtendedXMP=\"";
  const size_t note = xmp.find(kNote);
  if (note == std::string::npos) return false;                     // no extension attribute
  if (note + 24 + 32 + 1 > split) return false;                     // ill-formed
  if (xmp[note + 24 + 32] != '"') return false;
  std::string main_part(xmp, 0, split);
  const std::string ext(xmp, split);
  char guid[32];
  Md5Hex(reinterpret_cast<const uint8_t*>(ext.data()), ext.size(), guid);
  memcpy(&main_part[note + 24], guid, 32);
  const size_t main_seg = 2 + main_part.size() + sizeof(kXmp);
  if (main_seg > 0xffff) return false;
  Put16(o, 0xffe1); Put16(o, static_cast<uint32_t>(main_seg));
  PutBytes(o, kXmp, sizeof(kXmp));
  PutBytes(o, main_part.data(), main_part.size());
  static const char kXmpExt[] = "http://ns.adobe.com/xmp/extension/";
  const size_t kBuf = 65458;
  const size_t head = sizeof(kXmpExt) + 40;                         // + GUID, total size, offset
  const size_t nchunks = ext.size() / kBuf + 1;
  size_t pos = 0;
  for (size_t chunk = 0; chunk < nchunks; ++chunk) {
    const size_t n = std::min(kBuf, ext.size() - pos);
    Put16(o, 0xffe1); Put16(o, static_cast<uint32_t>(2 + head + n));
    PutBytes(o, kXmpExt, sizeof(kXmpExt));
    PutBytes(o, guid, 32);
    Put16(o, static_cast<uint32_t>(ext.size() >> 16)); Put16(o, static_cast<uint32_t>(ext.size() & 0xffff));
    Put16(o, static_cast<uint32_t>(pos >> 16)); Put16(o, static_cast<uint32_t>(pos & 0xffff));
    PutBytes(o, ext.data() + pos, n);
    pos += n;
  }
  return true;
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
    // XMP: a single APP1 when it fits (headers.cc:162-180), else main + extension chunks
    if (!meta->xmp.empty() && !AppendXmp(meta->xmp, meta->xmp_split, o)) return false;
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
static_assert(kQSize == kAdaptDeltas, "candidate steps");
// Gaussian weights, sigma ~ 3, centred on delta 0 (histogram.cc:117-124)
const float kDeltaWeight[kQSize] = {0, 0, 0, 0, 0, 1, 5, 16, 43, 94, 164, 228, 255,
                                    228, 164, 94, 43, 16, 5, 1, 0, 0, 0, 0, 0};
int BitLength(int v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
}  // namespace

// The bin loops of AnalyseHisto (src/histogram.cc:150-205) for one table: for every position and
// candidate step the rate and distortion sums, plus the population / highest occupied bin of the
// position.  All terms are (wrapping, like the reference's `int` products) 32-bit integers and
// their running sums stay far below 2^53, so the reference's double accumulators hold exactly
// these integers: the sums can be made anywhere -- here, or on the device
// (sjpeg_hip_adapt_sums) -- and the float part below sees the same values.
void AdaptSums(const uint32_t hist[64][128], const uint8_t quant[64], const uint8_t min_quant[64],
               int64_t sums[64][kAdaptDeltas][2], int32_t totlast[64][2]) {
  for (int pos = 0; pos < 64; ++pos) {
    const uint32_t* const h = hist[pos];
    int total = 0, last = 0;
    for (int i = 0; i < 128; ++i) {
      total += static_cast<int>(h[i]);
      if (h[i]) last = i + 1;
    }
    totlast[pos][0] = total; totlast[pos][1] = last;
    const int dq0 = quant[pos], min_dq0 = min_quant[pos];
    const int bias = 1 << 16 >> 1;
    for (int delta = 0; delta < kAdaptDeltas; ++delta) {
      const int dq = dq0 + (delta + kQDeltaMin);
      if (dq < min_dq0 || dq > 255) {
        sums[pos][delta][0] = 0; sums[pos][delta][1] = INT64_MIN;     // not a candidate
        continue;
      }
      const int idq = ((1 << 16) + dq - 1) / dq;
      int64_t bsum = 0, dsum = 0;
      for (int i = 0; i < last; ++i) {
        const uint32_t hi = h[i];
        const uint32_t v = (static_cast<uint32_t>(i) << 2) + 2;       // bin centroid: HSHIFT = 2, HHALF = 2
        const uint32_t qv = (v * static_cast<uint32_t>(idq) + bias) >> 16;
        const uint32_t bits = qv ? static_cast<uint32_t>(BitLength(static_cast<int>(qv))) : 0u;
        const uint32_t d = v - qv * static_cast<uint32_t>(dq);        // (qv == 0: v itself)
        bsum += static_cast<int32_t>(hi * bits);
        dsum += static_cast<int32_t>(hi * (d * d));
      }
      sums[pos][delta][0] = bsum; sums[pos][delta][1] = dsum;
    }
  }
}

// The float / double half of AnalyseHisto (src/histogram.cc:169-312) on those sums.  The expression
// of every accumulated term and of the score is the reference's (operation order is normative for
// the rounding); the structure is this library's: a weighted two-cloud line fit per coefficient
// position, then one step choice per position.
namespace {

// Weighted moments of the clouds {step, distortion} and {step, rate} of one coefficient position.
struct StepFit {
  double w = 0., x = 0., xx = 0.;            // weights, steps
  double d = 0., dd = 0., xd = 0.;           // distortion cloud
  double r = 0., xr = 0.;                    // rate cloud
  void Add(double weight, double step, double distortion, double rate) {
    w += weight;
    x += weight * step;
    xx += weight * step * step;
    d += weight * distortion;
    dd += weight * distortion * distortion;
    r += weight * rate;
    xd += weight * distortion * step;
    xr += weight * rate * step;
  }
  double CovDistortion() const { return w * xd - x * d; }
  double CovRate() const { return w * xr - x * r; }
  // is distortion a (nearly) linear function of the step?  r^2 >= limit, without the division
  bool Correlated(double limit) const {
    const double c = CovDistortion();
    return !(c * c < limit * (w * xx - x * x) * (w * dd - d * d));
  }
};

struct StepCosts {                            // per candidate step of one position
  float distortion[kQSize];                   // FLT_MAX: not a candidate
  float rate[kQSize];
};

}  // namespace

void AdaptDecide(const int64_t sums[2][64][kAdaptDeltas][2], const int32_t totlast[2][64][2], int nb_comps,
                 uint8_t quant[2][64], int qdelta_max_luma, int qdelta_max_chroma) {
  constexpr double kMinCorrelation = 0.5, kMinDensity = 0.5, kFallbackLambda = 0x80;
  constexpr uint64_t kNeverTouched = 0x103ull;      // DC and its two neighbours
  static thread_local StepCosts costs[64];
  for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
    uint64_t live = ~kNeverTouched;                 // positions that take part
    double slope_distortion = 0., slope_rate = 0.;  // summed over the live positions
    for (int pos = 0; pos < 64; ++pos) {
      if (!((live >> pos) & 1u)) continue;
      if (totlast[idx][pos][0] < kMinDensity * totlast[idx][pos][1]) {   // sparse histogram
        live &= ~(1ull << pos);
        continue;
      }
      StepFit fit;
      StepCosts& c = costs[pos];
      for (int k = 0; k < kQSize; ++k) {
        const int64_t* const s = sums[idx][pos][k];
        if (s[1] == INT64_MIN) {                    // quantizer out of range
          c.distortion[k] = FLT_MAX;
          c.rate[k] = 0;
          continue;
        }
        const double rate = static_cast<double>(s[0]), distortion = static_cast<double>(s[1]);
        c.distortion[k] = static_cast<float>(distortion);
        c.rate[k] = static_cast<float>(rate);
        if (kDeltaWeight[k] > 0.) fit.Add(kDeltaWeight[k], static_cast<double>(k + kQDeltaMin), distortion, rate);
      }
      if (!fit.Correlated(kMinCorrelation)) {
        live &= ~(1ull << pos);
        continue;
      }
      slope_distortion += fit.CovDistortion();
      slope_rate += fit.CovRate();
    }
    // lambda = -d(distortion) / d(rate) around the current matrix, if the fit is well conditioned
    double lambda = kFallbackLambda;
    if (slope_distortion > 1000. && slope_rate < -10.) {
      lambda = -slope_distortion / slope_rate;
      if (lambda < 1.) lambda = 1.;
    }
    const int last_step = ((idx == 0) ? qdelta_max_luma : qdelta_max_chroma) - kQDeltaMin;
    for (int pos = 0; pos < 64; ++pos) {
      if (!((live >> pos) & 1u)) continue;
      const StepCosts& c = costs[pos];
      float best = FLT_MAX;
      int step = 0;
      for (int k = 0; k <= last_step; ++k) {
        if (!(c.distortion[k] < FLT_MAX)) continue;
        const float score = c.distortion[k] + lambda * c.rate[k];
        if (score < best) { best = score; step = k + kQDeltaMin; }
      }
      quant[idx][pos] = static_cast<uint8_t>(quant[idx][pos] + step);
    }
  }
}

void AdaptQuantMatrices(const uint32_t hist[2][64][128], int nb_comps, uint8_t quant[2][64],
                        const uint8_t min_quant[2][64], int qdelta_max_luma, int qdelta_max_chroma) {
  static thread_local int64_t sums[2][64][kAdaptDeltas][2];
  static thread_local int32_t totlast[2][64][2];
  for (int idx = 0; idx < (nb_comps > 1 ? 2 : 1); ++idx) AdaptSums(hist[idx], quant[idx], min_quant[idx], sums[idx], totlast[idx]);
  AdaptDecide(sums, totlast, nb_comps, quant, qdelta_max_luma, qdelta_max_chroma);
}

// Optimised Huffman table for one alphabet (what src/entropy.cc:254-430 produces): code lengths from
// Huffman's merging with a reserved leaf of weight 1 behind the real symbols (it ends up with the
// all-ones code, which JPEG forbids for real symbols), the length histogram limited to 16 bits
// (T.81 K.2), symbols listed by (length, value).  The reference keeps ties out of the merge order with
// a combined key (weight << 9 | id), a merged node inheriting the id of its heavier child: the same
// strict order is used here, on an explicit tree with a heap of the open nodes.
namespace {

struct TreeNode {
  uint64_t weight;
  int id;                                     // leaf: symbol value; inner node: id of its heavier child
  int parent;                                 // index into the node array, -1 for the root
};

// The open nodes as a binary min-heap of node indices on a fixed array (no allocation: a batch codes a
// table per component and frame between two device passes).  Order: (weight, id), both ascending.
struct OpenNodes {
  const TreeNode* nodes;
  int heap[258];
  int n = 0;
  bool Before(int a, int b) const {
    const TreeNode& x = nodes[a];
    const TreeNode& y = nodes[b];
    return x.weight != y.weight ? x.weight < y.weight : x.id < y.id;
  }
  void Push(int v) {
    int i = n++;
    while (i > 0 && Before(v, heap[(i - 1) >> 1])) { heap[i] = heap[(i - 1) >> 1]; i = (i - 1) >> 1; }
    heap[i] = v;
  }
  int Pop() {                                 // the lightest
    const int top = heap[0], v = heap[--n];
    int i = 0;
    for (;;) {
      int c = 2 * i + 1;
      if (c >= n) break;
      if (c + 1 < n && Before(heap[c + 1], heap[c])) ++c;
      if (!Before(heap[c], v)) break;
      heap[i] = heap[c];
      i = c;
    }
    if (n > 0) heap[i] = v;
    return top;
  }
};

// T.81 K.2 (Figure K.3): no code longer than `limit` bits.  count[l] = codes of length l + 1.
// false: the histogram was no prefix code (see below) -- the caller falls back to a flat code.
bool LimitCodeLengths(int* count, int longest, int limit) {
  for (int len = longest; len > limit; --len) {
    while (count[len - 1] > 0) {
      int shorter = len - 2;                  // a leaf at least two levels up becomes an inner node
      while (shorter >= 1 && count[shorter - 1] == 0) --shorter;
      // (only a histogram that is no prefix code any more gets here: depths beyond 32 bits were
      // clamped, which takes Fibonacci-like counts over 33+ symbols; the reference reads in front
      // of its array in that case, src/entropy.cc:396-398)
      // (a prefix code has its deepest leaves in pairs; a lone one is the clamped histogram again)
      if (shorter < 1 || count[len - 1] < 2) return false;
      count[len - 1] -= 2;                    // a pair of deepest leaves: one moves up one level,
      count[len - 2] += 1;
      count[shorter - 1] -= 1;                // the other joins the split leaf one level below it
      count[shorter] += 2;
    }
  }
  return true;
}

}  // namespace

void BuildOptimalSpec(const uint32_t* freq, int size, HuffSpec* out) {
  constexpr int kLongest = 32, kLimit = 16;
  memset(out->syms, 0, sizeof(out->syms));
  memset(out->bits, 0, sizeof(out->bits));
  TreeNode nodes[2 * 257];
  int nnodes = 0;
  for (int sym = 0; sym < size; ++sym) {
    if (freq[sym] > 0) nodes[nnodes++] = TreeNode{freq[sym], sym, -1};
  }
  const int used = nnodes;
  out->nsyms = used;
  if (used == 0) return;
  nodes[nnodes++] = TreeNode{1, size, -1};    // the reserved leaf
  const int leaves = used + 1;
  {
    OpenNodes open;
    open.nodes = nodes;
    for (int i = 0; i < used; ++i) open.Push(i);
    // (the reserved leaf is not ranked by its key: it is the lighter half of the FIRST merge whatever
    // the weights, src/entropy.cc:304-305 appends it behind the sorted symbols)
    for (int light = used; open.n > 0; light = -1) {
      if (light < 0) {
        if (open.n == 1) break;
        light = open.Pop();
      }
      const int heavy = open.Pop();
      const int parent = nnodes;
      nodes[nnodes++] = TreeNode{nodes[heavy].weight + nodes[light].weight, nodes[heavy].id, -1};
      nodes[light].parent = nodes[heavy].parent = parent;
      open.Push(parent);
    }
  }
  // depth of every node, root first (a parent is always created after its children)
  int depth[2 * 257];
  depth[nnodes - 1] = 0;
  for (int i = nnodes - 2; i >= 0; --i) depth[i] = depth[nodes[i].parent] + 1;
  // count[l]: leaves with a code of l + 1 bits.  int, not the DHT's bytes: with the reserved leaf still counted a
  // level can hold 256 (255 symbols of equal weight), which only becomes 255 once that leaf is taken out below
  int count[kLongest];
  memset(count, 0, sizeof(count));
  int length_of[257];
  for (int i = 0; i <= size; ++i) length_of[i] = 0;
  int longest = 0;
  for (int i = 0; i < leaves; ++i) {
    const int len = depth[i] < kLongest ? depth[i] : kLongest;
    length_of[nodes[i].id] = len;
    ++count[len - 1];
    if (len > longest) longest = len;
  }
  // the symbol list: by code length, then by value -- with the lengths of the unlimited tree
  // (counting sort: where the symbols of every length start, then one pass over the symbols)
  int next_of[kLongest + 1];
  int fill = 0;
  for (int len = 1; len <= longest; ++len) {
    next_of[len] = fill;
    fill += count[len - 1];
  }
  // (the reserved leaf is not listed: it is the last of the longest length)
  for (int sym = 0; sym < size; ++sym) {
    if (length_of[sym] > 0) out->syms[next_of[length_of[sym]]++] = static_cast<uint8_t>(sym);
  }
  if (!LimitCodeLengths(count, longest, kLimit)) {
    // Depths beyond 32 bits were clamped and the histogram is no prefix code (Fibonacci-like counts over 33 or
    // more symbols; the reference's behaviour is undefined there).  A VALID table instead of a wrong
    // one: a complete code of two lengths over all leaves, the reserved one included -- with f = ceil(log2(leaves)),
    // 2^f - leaves of them get f - 1 bits and the rest f (Kraft sum exactly 1) --, the symbols in the order of
    // their values.  257 leaves: 255 codes of 8 bits + 2 of 9, and the reserved leaf is one of the two (ADVICE r04:
    // one length for all of them put 256 / 257 into a byte).
    memset(count, 0, sizeof(count));
    int flat = 1;
    while ((1 << flat) < leaves) ++flat;      // <= 9 bits for 257 leaves
    const int n_short = (1 << flat) - leaves;
    if (flat > 1) count[flat - 2] = n_short;
    count[flat - 1] = leaves - n_short;
    int k = 0;
    for (int sym = 0; sym < size; ++sym) if (freq[sym] > 0) out->syms[k++] = static_cast<uint8_t>(sym);
  }
  int last = kLimit;                          // the reserved leaf is the last code of the longest length
  while (last > 1 && count[last - 1] == 0) --last;
  --count[last - 1];
  for (int l = 0; l < kLimit; ++l) out->bits[l] = static_cast<uint8_t>(count[l]);
}

}  // namespace sjpeg_host
