// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Thin extern "C" shim over the *real* reference encoder (webmproject/sjpeg),
// compiled together with the reference's own sources where they lie under
// /root/reference (see oracle/Makefile).  The resulting oracle/_ref/libsjpeg_ref.so
// is (a) what pins oracle/sjpeg_oracle.c, (b) what generates tests/golden/, and
// (c) the "reference" cpu_baseline leg of bench.py.  Nothing here restates
// reference code: it only calls the reference's public API (src/sjpeg.h) and
// the two linkable internal seams SURVEY.md §8c names (sjpegi.h:80-81,105-108).
#include <stdint.h>
#include <string.h>
#include <string>

#include "sjpeg.h"
#include "sjpegi.h"

namespace sjpeg {
extern bool ForceSlowCImplementation;   // src/enc.cc:157-158
}

extern "C" {

// sjpeg::Encode() with an EncoderParam assembled from plain scalars.
// quant == NULL -> SetQuality(quality); else SetQuantization(quant, reduction).
// Returns size (0 on failure); *out must be released with ref_free().
size_t ref_encode_param(const uint8_t* rgb, int w, int h, int stride,
                        float quality, int yuv_mode, int huffman, int adaptive,
                        int trellis, const uint8_t* quant, float reduction,
                        int limit_quant, int quant_bias, uint8_t** out) {
  sjpeg::EncoderParam param(quality);
  if (quant != nullptr) {
    uint8_t m[2][64];
    memcpy(m, quant, sizeof(m));
    param.SetQuantization(m, reduction);
  }
  if (limit_quant) param.SetLimitQuantization(true);
  param.yuv_mode = static_cast<SjpegYUVMode>(yuv_mode);
  param.Huffman_compress = (huffman != 0);
  param.adaptive_quantization = (adaptive != 0);
  param.use_trellis = (trellis != 0);
  if (quant_bias >= 0) param.quantization_bias = quant_bias;
  return sjpeg::Encode(rgb, w, h, stride, param, out);
}

// sjpeg::Encode() with metadata (src/sjpeg.h:258-266), method 0.
size_t ref_encode_meta(const uint8_t* rgb, int w, int h, int stride, float quality, int yuv_mode,
                       const char* app, size_t app_size, const char* exif, size_t exif_size,
                       const char* iccp, size_t iccp_size, const char* xmp, size_t xmp_size,
                       int xmp_split, uint8_t** out) {
  sjpeg::EncoderParam param(quality);
  param.yuv_mode = static_cast<SjpegYUVMode>(yuv_mode);
  param.Huffman_compress = false;
  param.adaptive_quantization = false;
  if (app != nullptr) param.app_markers.assign(app, app_size);
  if (exif != nullptr) param.exif.assign(exif, exif_size);
  if (iccp != nullptr) param.iccp.assign(iccp, iccp_size);
  if (xmp != nullptr) param.xmp.assign(xmp, xmp_size);
  param.xmp_split_point = static_cast<uint16_t>(xmp_split);
  return sjpeg::Encode(rgb, w, h, stride, param, out);
}

// sjpeg::Encode() with the size/PSNR search enabled (src/dichotomy.cc:113-205).
// target_mode: 1 = size (bytes), 2 = PSNR (dB).  q_out / value_out: what the default hook found.
size_t ref_encode_search(const uint8_t* rgb, int w, int h, int stride, float quality, int yuv_mode,
                         int huffman, int adaptive, int target_mode, float target_value, int passes,
                         float tolerance, float qmin, float qmax, int trellis, uint8_t** out) {
  sjpeg::EncoderParam param(quality);
  param.use_trellis = (trellis != 0);
  param.yuv_mode = static_cast<SjpegYUVMode>(yuv_mode);
  param.Huffman_compress = (huffman != 0);
  param.adaptive_quantization = (adaptive != 0);
  param.target_mode = static_cast<sjpeg::EncoderParam::TargetMode>(target_mode);
  param.target_value = target_value;
  param.passes = passes;
  param.tolerance = tolerance;
  param.qmin = qmin;
  param.qmax = qmax;
  return sjpeg::Encode(rgb, w, h, stride, param, out);
}

size_t ref_encode(const uint8_t* rgb, int w, int h, int stride, float quality,
                  int method, int yuv_mode, uint8_t** out) {
  return SjpegEncode(rgb, w, h, stride, out, quality, method,
                     static_cast<SjpegYUVMode>(yuv_mode));
}

size_t ref_compress(const uint8_t* rgb, int w, int h, float quality,
                    uint8_t** out) {
  return SjpegCompress(rgb, w, h, quality, out);
}

void ref_free(uint8_t* p) { SjpegFreeBuffer(p); }

// The other input layouts of the public API (src/sjpeg.h:300-349), all through a string sink.
// format: 1 BGRA, 2 RGBA, 3 gray, 4 planar 4:4:4, 5 planar 4:2:0, 6 NV12, 7 NV21 (= oracle ORC_SRC_*).
size_t ref_encode_src(int format, const uint8_t* p0, const uint8_t* p1, const uint8_t* p2,
                      int s0, int s1, int s2, int w, int h, float quality, int yuv_mode,
                      int huffman, int adaptive, uint8_t** out) {
  sjpeg::EncoderParam param(quality);
  param.yuv_mode = static_cast<SjpegYUVMode>(yuv_mode);
  param.Huffman_compress = (huffman != 0);
  param.adaptive_quantization = (adaptive != 0);
  std::string str;
  std::shared_ptr<sjpeg::ByteSink> sink = sjpeg::MakeByteSink(&str);
  bool ok = false;
  switch (format) {
    case 1: ok = sjpeg::EncodeBGRA(p0, w, h, s0, param, sink.get()); break;
    case 2: ok = sjpeg::EncodeRGBA(p0, w, h, s0, param, sink.get()); break;
    case 3: ok = sjpeg::EncodeGray(p0, w, h, s0, param, sink.get()); break;
    case 4: ok = sjpeg::EncodeYUV444(p0, s0, p1, s1, p2, s2, w, h, param, sink.get()); break;
    case 5: ok = sjpeg::EncodeYUV420(p0, s0, p1, s1, p2, s2, w, h, param, sink.get()); break;
    case 6: ok = sjpeg::EncodeNV12(p0, s0, p1, s1, w, h, param, sink.get()); break;
    case 7: ok = sjpeg::EncodeNV21(p0, s0, p1, s1, w, h, param, sink.get()); break;
    default: break;
  }
  *out = nullptr;
  if (!ok || str.empty()) return 0;
  *out = new uint8_t[str.size()];
  memcpy(*out, str.data(), str.size());
  return str.size();
}

// Stage seams (SURVEY.md §8c "Stage-level access").
void ref_get_block(int yuv_mode, const uint8_t* rgb, int step, int16_t* out) {
  sjpeg::GetBlockFunc(static_cast<SjpegYUVMode>(yuv_mode))(rgb, step, out);
}
void ref_fdct(int16_t* coeffs, int num_blocks) {
  sjpeg::GetFdct()(coeffs, num_blocks);
}
void ref_force_slow_c(int on) { sjpeg::ForceSlowCImplementation = (on != 0); }

void ref_quant_matrix(float quality, int for_chroma, uint8_t* m) {
  SjpegQuantMatrix(quality, for_chroma != 0, m);
}
int ref_find_quantizer(const uint8_t* data, size_t size, uint8_t* quant) {
  uint8_t q[2][64];
  memset(q, 0, sizeof(q));
  const int n = SjpegFindQuantizer(data, size, q);
  memcpy(quant, q, sizeof(q));
  return n;
}
int ref_dimensions(const uint8_t* data, size_t size, int* w, int* h, int* is420) {
  return SjpegDimensions(data, size, w, h, is420) ? 1 : 0;
}
float ref_estimate_quality(const uint8_t* m, int for_chroma) {
  return SjpegEstimateQuality(m, for_chroma != 0);
}
int ref_riskiness(const uint8_t* rgb, int w, int h, int stride, float* risk) {
  return static_cast<int>(SjpegRiskiness(rgb, w, h, stride, risk));
}
uint32_t ref_version() { return SjpegVersion(); }

}  // extern "C"
