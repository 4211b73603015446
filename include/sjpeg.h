// sjpeg.h -- public API of the MI355X-native sjpeg-compatible JPEG encoder.
//
// Source-level drop-in for the reference's only installed header
// (/root/reference/src/sjpeg.h): same names, same argument meaning, same ownership and
// error conventions (0 / false on failure, never throws, never aborts).  The per-block
// hot path (colour conversion, forward DCT, quantization, Huffman coding, bit packing) runs
// as HIP kernels on a gfx950 device (see sjpeg_hip.h); quantizer/table/header preparation
// stays on the host and mirrors the reference bit for bit.
//
// What this build runs on the GPU: YUV 4:2:0 / 4:4:4 / 4:0:0 from every input layout of the
// API, compression methods 0..8 (standard or optimised Huffman tables, fixed or adaptive
// quantization, trellis quantization), and the multi-pass size / PSNR search (every pass is a GPU
// pass over the resident picture), SJPEG_YUV_SHARP (the sharp
// conversion runs on the device).  SJPEG_YUV_AUTO, SjpegCompress() and SjpegRiskiness() need the
// reference's trained score table: it SHIPS with the library as riskiness.bin (reference data,
// unmodified, Apache-2.0: riskiness.NOTICE) and is found next to libsjpeg_amd.so; a table from
// elsewhere goes in through SJPEG_HIP_RISKINESS_TABLE or sjpeg_hip_set_riskiness_table()
// (sjpeg_hip.h).  A library installed WITHOUT the file fails those three requests (0 / false).
// There is no CPU fallback for anything.
// The reason of the last failure on the calling thread: SjpegHipLastError().
#ifndef SJPEG_AMD_SJPEG_H_
#define SJPEG_AMD_SJPEG_H_

#include <inttypes.h>
#include <stddef.h>

#include <memory>
#include <string>
#include <vector>

#define SJPEG_VERSION 0x000101   // same bitstream-level version as the reference (0.1.1)

#if defined(__cplusplus) || defined(c_plusplus)
extern "C" {
#endif

// reference: src/sjpeg.h:34
uint32_t SjpegVersion();

// reference: src/sjpeg.h:54-60
typedef enum {
  SJPEG_YUV_AUTO = 0,   // decide between 420 / sharp / 444 from the picture
  SJPEG_YUV_420,        // 4:2:0
  SJPEG_YUV_SHARP,      // 4:2:0 through the "sharp" converter
  SJPEG_YUV_444,        // 4:4:4
  SJPEG_YUV_400         // luma only
} SjpegYUVMode;

// One-call encode, reference: src/sjpeg.h:45 (== SjpegEncode(method 4, SJPEG_YUV_AUTO)).
// *out_data is allocated with new[]; free with delete[] or SjpegFreeBuffer().
size_t SjpegCompress(const uint8_t* rgb, int width, int height, float quality, uint8_t** out_data);

// reference: src/sjpeg.h:104-109.  'stride' in bytes, |stride| >= 3*width, may be negative.
// compression_method 0..8 as tabulated in the reference header (clamped).
size_t SjpegEncode(const uint8_t* rgb, int width, int height, int stride, uint8_t** out_data,
    float quality, int compression_method, SjpegYUVMode yuv_mode);

// reference: src/sjpeg.h:113
void SjpegFreeBuffer(const uint8_t* buffer);

// JPEG-parsing helpers, reference: src/sjpeg.h:122-150 (pure host code)
bool SjpegDimensions(const uint8_t* data, size_t size, int* width, int* height, int* is_yuv420);
int SjpegFindQuantizer(const uint8_t* data, size_t size, uint8_t quant[2][64]);
float SjpegEstimateQuality(const uint8_t matrix[64], bool for_chroma);
void SjpegQuantMatrix(float quality, bool for_chroma, uint8_t matrix[64]);
// reference: src/sjpeg.h:149.  Needs the riskiness score table (sjpeg_hip.h): without it the
// result is SJPEG_YUV_AUTO ("undecided"), *risk = -1 and SjpegHipLastError() says why.
SjpegYUVMode SjpegRiskiness(const uint8_t* rgb, int width, int height, int stride, float* risk);

// Not part of the reference: text of the last failure on this thread ("" if none).
const char* SjpegHipLastError();

#if defined(__cplusplus) || defined(c_plusplus)
}    // extern "C"
#endif

// std::string flavours, reference: src/sjpeg.h:159-165
bool SjpegCompress(const uint8_t* rgb, int width, int height, float quality, std::string* output);
bool SjpegDimensions(const std::string& jpeg_data, int* width, int* height, int* is_yuv420);
int SjpegFindQuantizer(const std::string& jpeg_data, uint8_t quant[2][64]);

namespace sjpeg {

struct Encoder;         // internal
struct SearchHook;
struct ByteSink;
struct MemoryManager;

// Encoding parameters, reference: src/sjpeg.h:187-275 (field order and defaults kept).
struct EncoderParam {
  EncoderParam();
  explicit EncoderParam(float quality_factor);

  void SetQuality(float quality_factor);
  void SetQuantization(const uint8_t m[2][64], float reduction = 100.f);
  const uint8_t* GetQuantMatrix(int idx) const { return quant_[idx]; }
  void SetLimitQuantization(bool limit_quantization = true, int tolerance = 0);
  void SetMinQuantization(const uint8_t m[2][64], int min_quant_tolerance = 0);

  SjpegYUVMode yuv_mode;
  bool Huffman_compress;
  bool adaptive_quantization;
  bool adaptive_bias;
  bool use_trellis;

  typedef enum { TARGET_NONE = 0, TARGET_SIZE = 1, TARGET_PSNR = 2 } TargetMode;
  TargetMode target_mode;
  float target_value;
  int passes;
  float tolerance;
  float qmin, qmax;

  int quantization_bias;
  int qdelta_max_luma;
  int qdelta_max_chroma;

  sjpeg::SearchHook* search_hook;

  std::string exif;
  std::string iccp;
  std::string app_markers;
  std::string xmp;
  uint16_t xmp_split_point = 0u;
  void ResetMetadata();

  sjpeg::MemoryManager* memory;

 protected:
  uint8_t quant_[2][64];
  uint8_t min_quant_[2][64];
  bool use_min_quant_;
  int min_quant_tolerance_;

 protected:
  void Init(float quality_factor);
  friend struct sjpeg::Encoder;
};

// reference: src/sjpeg.h:280-292
bool Encode(const uint8_t* rgb, int width, int height, int stride, const EncoderParam& param,
    std::string* output);
size_t Encode(const uint8_t* rgb, int width, int height, int stride, const EncoderParam& param,
    uint8_t** out_data);
bool Encode(const uint8_t* rgb, int width, int height, int stride, const EncoderParam& param,
    sjpeg::ByteSink* sink);

// Other input layouts, reference: src/sjpeg.h:300-349.  BGRA/RGBA: 4 bytes per pixel, alpha
// ignored, stride >= 4*width.  Gray: luma samples as they are (YUV 4:0:0).  NV12/NV21: luma
// plane + interleaved U,V (resp. V,U) plane of (width+1)/2 x (height+1)/2 pairs (YUV 4:2:0).
// YUV444 / YUV420: three planes.  The colour mode is implied by the layout for all but
// BGRA/RGBA (where param.yuv_mode selects 420 / 444 / 400).
bool EncodeBGRA(const uint8_t* bgra, int width, int height, int stride, const EncoderParam& param,
    sjpeg::ByteSink* sink);
bool EncodeBGRA(const uint8_t* bgra, int width, int height, int stride, const EncoderParam& param,
    std::string* output);
bool EncodeRGBA(const uint8_t* rgba, int width, int height, int stride, const EncoderParam& param,
    sjpeg::ByteSink* sink);
bool EncodeRGBA(const uint8_t* rgba, int width, int height, int stride, const EncoderParam& param,
    std::string* output);
bool EncodeGray(const uint8_t* gray, int width, int height, int stride, const EncoderParam& param,
    sjpeg::ByteSink* sink);
bool EncodeGray(const uint8_t* gray, int width, int height, int stride, const EncoderParam& param,
    std::string* output);
bool EncodeNV21(const uint8_t* y, int y_stride, const uint8_t* vu, int vu_stride, int width,
    int height, const EncoderParam& param, sjpeg::ByteSink* output);
bool EncodeNV12(const uint8_t* y, int y_stride, const uint8_t* uv, int uv_stride, int width,
    int height, const EncoderParam& param, sjpeg::ByteSink* output);
bool EncodeYUV444(const uint8_t* Y, int Y_stride, const uint8_t* U, int U_stride, const uint8_t* V,
    int V_stride, int width, int height, const EncoderParam& param, sjpeg::ByteSink* output);
bool EncodeYUV420(const uint8_t* Y, int Y_stride, const uint8_t* U, int U_stride, const uint8_t* V,
    int V_stride, int width, int height, const EncoderParam& param, sjpeg::ByteSink* output);

// reference: src/sjpeg.h:355-373
struct SearchHook {
  float q;
  float qmin, qmax;
  float target;
  float tolerance;
  bool for_size;
  float value;
  int pass;
  virtual bool Setup(const EncoderParam& param);
  virtual void NextMatrix(int idx, uint8_t dst[64]);
  virtual bool Update(float result);
  virtual ~SearchHook() {}
};

// Streaming output, reference: src/sjpeg.h:378-398.
//   Commit(used, extra, &ptr): 'used' bytes were written since the last call; make
//   'extra' more available at *ptr.   Finalize(): no more commits.   Reset(): drop all.
struct ByteSink {
 public:
  virtual ~ByteSink() {}
  virtual bool Commit(size_t used_size, size_t extra_size, uint8_t** data) = 0;
  virtual bool Finalize() = 0;
  virtual void Reset() = 0;
};

std::shared_ptr<ByteSink> MakeByteSink(std::string* output);
template<typename T>
std::shared_ptr<ByteSink> MakeByteSink(std::vector<T>* output);
template<> std::shared_ptr<ByteSink> MakeByteSink(std::vector<uint8_t>* output);

// Host allocations of the codec go through this, reference: src/sjpeg.h:410-415.
struct MemoryManager {
 public:
  virtual ~MemoryManager() {}
  virtual void* Alloc(size_t size) = 0;
  virtual void Free(void* const ptr) = 0;
};

}  // namespace sjpeg

// courtesy alias for the name the reference's README still uses
typedef sjpeg::EncoderParam SjpegEncodeParam;

#endif    // SJPEG_AMD_SJPEG_H_
