// riskiness.hip -- the stencil behind SjpegRiskiness() / SJPEG_YUV_AUTO on gfx950.
//
// Reference: /root/reference/src/jpeg_tools.cc:170-236 (SjpegRiskiness) and
// src/colors_rgb.cc:1085-1122 (pixel -> 7x7x7 YUV cell index).  Every pixel becomes the index of
// its cell; every position (i, j) with a right and a lower neighbour looks three pairs of
// cells up in a 343 x 343 score table and the picture's verdict is made of three sums.  The
// score table is trained data of the reference (src/score_7.cc); it ships beside the library as
// riskiness.bin (riskiness.NOTICE), or comes from sjpeg_hip_set_riskiness_table / SJPEG_HIP_RISKINESS_TABLE.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sjpeg_hip.h"

namespace {

constexpr int kCells = 7;                         // kRGBSize
constexpr int kCells3 = kCells * kCells * kCells;
constexpr int kNoiseLevel = 4;

__device__ __forceinline__ uint32_t clip8(int v) { return v < 0 ? 0u : v > 255 ? 255u : static_cast<uint32_t>(v); }
__device__ __forceinline__ uint32_t cell(uint32_t v) { return (v * (0x0101u * (kCells - 1))) >> 16; }   // ~ v * 6 / 255

__device__ __forceinline__ int yuv_index(const uint8_t* p, int ro, int go, int bo) {   // colors_rgb.cc:1104-1112
  const int r = p[ro], g = p[go], b = p[bo];
  const uint32_t y = cell(static_cast<uint32_t>((19595 * r + 38469 * g + 7471 * b + 32768) >> 16));
  const uint32_t u = cell(clip8(128 + ((-11059 * r - 21709 * g + 32768 * b + 32768) >> 16)));
  const uint32_t v = cell(clip8(128 + ((32768 * r - 27439 * g - 5329 * b + 32768) >> 16)));
  return static_cast<int>(y + u * kCells + v * kCells * kCells);
}

struct RiskArgs {
  const uint8_t* rgb;
  long long row_stride, frame_stride;
  int pix_step, r_off, g_off, b_off;
  int W, H;
  const uint8_t* table;                           // [343 * 343]
  unsigned long long* out;                        // [nframes][3]: score_sum, score_num, gray_num
};

// One workgroup = 256 columns x a BAND of rows (the grid's y dimension cuts the picture into at most 64 bands): a
// thread walks down its column, keeps the cell index of the pixel it stands on for the next row (two conversions per
// position instead of three), and the three sums leave the workgroup as ONE atomic each.  (Round 3's kernel was one
// row per workgroup and three 64-bit atomics per WAVE on the same three addresses: 390 000 serialised atomics for a
// 4K picture, 4.3 ms of the 5.3 ms SjpegCompress() took.)
__global__ __launch_bounds__(256) void risk_scan(const RiskArgs a) {
  __shared__ unsigned long long part[4][3];
  const int frame = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int rows = a.H - 1;                                    // positions (i, j), j = 1 .. H - 1 (the row below)
  const int per = (rows + static_cast<int>(gridDim.y) - 1) / static_cast<int>(gridDim.y);
  const int j0 = 1 + static_cast<int>(blockIdx.y) * per, j1 = min(j0 + per, a.H);
  unsigned long long s_sum = 0;
  uint32_t s_num = 0, g_num = 0;
  if (i < a.W - 1 && j0 < j1) {
    const uint8_t* row = a.rgb + frame * a.frame_stride + static_cast<long long>(j0 - 1) * a.row_stride;
    const long long o0 = static_cast<long long>(i) * a.pix_step, o1 = o0 + a.pix_step;
    int idx0 = yuv_index(row + o0, a.r_off, a.g_off, a.b_off);
    constexpr int gray = (kCells / 2) * (1 + kCells) * kCells;
    constexpr int gray_min = gray - gray % kCells;             // idx = y + 7 * (u + 7 * v): neutral chroma <=> [gray_min, gray_min + 7)
    for (int j = j0; j < j1; ++j) {
      const int idx1 = yuv_index(row + o1, a.r_off, a.g_off, a.b_off);
      row += a.row_stride;
      const int idx2 = yuv_index(row + o0, a.r_off, a.g_off, a.b_off);
      const int score = a.table[idx0 + kCells3 * idx1] + a.table[idx0 + kCells3 * idx2] + a.table[idx1 + kCells3 * idx2];
      if (score > kNoiseLevel) { s_sum += static_cast<unsigned long long>(score); ++s_num; }
      g_num += (idx0 >= gray_min && idx0 < gray_min + kCells) ? 1u : 0u;
      idx0 = idx2;
    }
  }
  unsigned long long n_sum = s_num, gn_sum = g_num;
  for (int d = 32; d > 0; d >>= 1) {
    s_sum += __shfl_down(s_sum, d, 64); n_sum += __shfl_down(n_sum, d, 64); gn_sum += __shfl_down(gn_sum, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6][0] = s_sum; part[threadIdx.x >> 6][1] = n_sum; part[threadIdx.x >> 6][2] = gn_sum;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (v) atomicAdd(&a.out[static_cast<size_t>(frame) * 3 + threadIdx.x], v);
  }
}

}  // namespace

extern "C" int sjpeg_hip_riskiness_sums(const sjpeg_hip_source* src, int width, int height, int nframes,
                                        const uint8_t* d_table, uint64_t* d_sums, void* stream) {
  if (src == nullptr || src->plane[0] == nullptr || d_table == nullptr || d_sums == nullptr ||
      width <= 0 || height <= 0 || nframes <= 0 || nframes > 65535 || height > 65536) {
    return SJPEG_HIP_EINVAL;
  }
  RiskArgs a;
  switch (src->format) {
    case SJPEG_HIP_SRC_RGB: a.pix_step = 3; a.r_off = 0; a.g_off = 1; a.b_off = 2; break;
    case SJPEG_HIP_SRC_BGRA: a.pix_step = 4; a.r_off = 2; a.g_off = 1; a.b_off = 0; break;
    case SJPEG_HIP_SRC_RGBA: a.pix_step = 4; a.r_off = 0; a.g_off = 1; a.b_off = 2; break;
    default: return SJPEG_HIP_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  a.rgb = static_cast<const uint8_t*>(src->plane[0]);
  a.row_stride = src->row_stride[0]; a.frame_stride = src->frame_stride[0];
  a.W = width; a.H = height;
  a.table = d_table;
  a.out = reinterpret_cast<unsigned long long*>(d_sums);
  if (hipMemsetAsync(d_sums, 0, static_cast<size_t>(nframes) * 3 * sizeof(uint64_t), st) != hipSuccess) return SJPEG_HIP_ERUNTIME;
  if (width < 2 || height < 2) return 0;                       // no (i, j) has both neighbours
  const int bands = height - 1 < 64 ? height - 1 : 64;
  hipLaunchKernelGGL(risk_scan, dim3((width - 1 + 255) / 256, bands, nframes), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? 0 : SJPEG_HIP_ERUNTIME;
}
