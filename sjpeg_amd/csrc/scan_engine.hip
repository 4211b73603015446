// scan_engine.hip -- the MI355X (gfx950 / CDNA4) scan engine behind include/sjpeg_hip.h.
//
// Replaces the reference's per-MCU hot loop (Encoder::SinglePassScan,
// /root/reference/src/enc.cc:276-307) for whole frames / batches of frames resident in HBM.
// Written for wave64 + LDS from scratch; integer-only, no MFMA (8-point integer butterflies
// and a per-coefficient reciprocal multiply are not a dense contraction).
//
// Pipeline per batch (all on one stream, no host round trip):
//
//   K1 scan_segments   one workgroup per SEGMENT (= run of consecutive MCUs of one frame):
//        P1 colour  : RGB rows -> level-shifted Y/Cb/Cr int16 blocks in LDS
//                     (reference: src/colors_rgb.cc:785-879, edge replication
//                      src/colors_rgb.cc:1212-1232 == coordinate clamping)
//        P2 block   : one THREAD per 8x8 block, whole block in registers:
//                     AverageExtraLuma fix-up (src/encoders.cc:107-125), forward DCT
//                     (src/fdct.cc:67-144,174-209,596-609), quantization
//                     (src/quantize.cc:119-121,288-320) -> zig-zag int16 + non-zero mask
//        P3 entropy : DC prediction (src/entropy.cc:133-150) through LDS, per-thread
//                     run/size Huffman coding (src/entropy.cc:161-198), workgroup prefix
//                     scan of block bit lengths, MSB-first bit packing into an LDS window
//        P4 flush   : coalesced store of the segment's packed words + its bit length
//   K2 scan_seg_offsets   per frame: exclusive scan of segment bit lengths
//   K3 concat_chunks      gather: every 4 KiB chunk of the frame's single continuous
//                         (un-stuffed) bit stream is assembled from the segments at their
//                         bit offsets; final byte padded with 1-bits
//                         (src/bit_writer.cc:107-116); counts 0xFF bytes per chunk
//   K4 scan_chunk_offsets per frame: exclusive scan of 0xFF counts, final size
//   K5 stuff_chunks       0xFF -> 0xFF00 byte stuffing (src/bit_writer.h:172-196) into
//                         the caller's output slot, header in front, FF D9 behind
//
// The reference never emits restart markers, so bit-exactness needs exactly this
// bit-level stitching.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

#include "sjpeg_hip.h"

namespace {

// ------------------------------------------------------------------------------------
// geometry

constexpr int kThreads = 256;        // 4 waves
constexpr int kSlotBytes = 144;      // 64 int16 + 16 B pad: conflict-free ds_read_b128 per lane
constexpr int kWinWords = 2048;      // LDS bit window, 32-bit MSB-first words (8 KiB)
constexpr int kMaxBlockBits = 1728;  // 22 (DC) + 63*27 (AC) rounded up; reference bound enc.cc:206-209
constexpr int kChunkWords = 1024;    // K3/K5 chunk: 4 KiB of un-stuffed stream
constexpr int kChunkBytes = kChunkWords * 4;

template <int MODE> struct Geo;
template <> struct Geo<SJPEG_HIP_YUV420> {
  static constexpr int kBpm = 6, kMcuPx = 16, kSegMcus = 41;    // (41 + 1 halo) * 6 = 252 threads
};
template <> struct Geo<SJPEG_HIP_YUV444> {
  static constexpr int kBpm = 3, kMcuPx = 8, kSegMcus = 84;     // 85 * 3 = 255
};
template <> struct Geo<SJPEG_HIP_YUV400> {
  static constexpr int kBpm = 1, kMcuPx = 8, kSegMcus = 255;    // 256
};

// device copy of sjpeg_hip_scan_tables, pre-digested
struct DevTables {
  uint2 q[2][64];          // {iquant, bias*iquant}, natural order
  uint32_t dc[2][12];
  uint32_t ac[2][256];
};

struct ScanArgs {
  const uint8_t* rgb;
  long long row_stride, frame_stride;
  int W, H, mb_w, n_mcus, nseg, has_clip;
  const DevTables* tables;
  uint32_t* seg_words;     // [nframes*nseg][slot_words]
  uint32_t slot_words;
  uint32_t* seg_nbits;     // [nframes*nseg]
  int16_t* coeffs;         // optional tap (TAP instantiation only)
  int ablate;              // profiling knob (env SJPEG_HIP_ABLATE): stop after phase 1/2/3; 0 = full
};

// LDS carve (bytes), all offsets multiples of 16
constexpr int kSamplesBytes = kThreads * kSlotBytes;            // 36864
constexpr int kOffWin = kSamplesBytes;
constexpr int kOffQ = kOffWin + kWinWords * 4;
constexpr int kOffAc = kOffQ + 2 * 64 * 8;
constexpr int kOffDc = kOffAc + 2 * 256 * 4;
constexpr int kOffDcs = kOffDc + 128;                           // int dcs[kThreads]
constexpr int kOffMisc = kOffDcs + kThreads * 4;                // scan scratch
constexpr int kLdsBytes = kOffMisc + 64;

// ------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// (a*b) >> 16 with 24-bit operands: the column pass' 16-bit fixed-point multiply
__device__ __forceinline__ int mulhi16(int a, int b) { return __mul24(a, b) >> 16; }

// In-place 8-point column transform on 8 registers; operation order of
// src/fdct.cc:67-144 (plain-C macro set :148-157).  Outputs land in natural frequency order.
__device__ __forceinline__ void fdct_col8(int& x0, int& x1, int& x2, int& x3,
                                          int& x4, int& x5, int& x6, int& x7) {
  int d07 = x0 - x7, s07 = x0 + x7;
  int d25 = x2 - x5, s25 = x2 + x5;
  int d34 = x3 - x4, s34 = x3 + x4;
  int d16 = x1 - x6, s16 = x1 + x6;
  int ed = s07 - s34, es = s07 + s34;
  int fd = s16 - s25, fs = s16 + s25;
  const int a = es << 3, b = fs << 3;
  x0 = a + b;
  x4 = a - b;
  ed <<= 3; fd <<= 3; d34 <<= 3; d07 <<= 3;
  x2 = mulhi16(27146, fd) + ed;
  x6 = mulhi16(27146, ed) - fd;
  d25 <<= 4; d16 <<= 4;
  const int od = mulhi16(d16 - d25, 23170);
  const int os = mulhi16(d16 + d25, 23170);
  const int p3 = d34 - od, p1 = d34 + od;
  const int p0 = d07 - os, p2 = d07 + os;
  const int t3 = mulhi16(p3, -21746) + p3 + 1;
  const int t1 = mulhi16(p1, 13036) + p2 + 1;
  const int t4 = mulhi16(-21746, p0) + p0;
  const int t5 = mulhi16(13036, p2);
  x1 = t1;
  x3 = p0 - t3;
  x5 = p3 + t4;
  x7 = t5 - p1;
}

// Row transform with compile-time table (src/fdct.cc:174-209); products are 24x16 bit.
template <int C1, int C2, int C3, int C4, int C5, int C6, int C7>
__device__ __forceinline__ void fdct_row8(int* r) {
  const int a0 = r[0] + r[7], b0 = r[0] - r[7];
  const int a1 = r[1] + r[6], b1 = r[1] - r[6];
  const int a2 = r[2] + r[5], b2 = r[2] - r[5];
  const int a3 = r[3] + r[4], b3 = r[3] - r[4];
  const int c0 = a0 + a3, c1 = a0 - a3, c2 = a1 + a2, c3 = a1 - a2;
  r[0] = mul24(C4, c0 + c2) >> 16;
  r[4] = mul24(C4, c0 - c2) >> 16;
  r[2] = (mul24(C2, c1) + mul24(C6, c3)) >> 16;
  r[6] = (mul24(C6, c1) - mul24(C2, c3)) >> 16;
  r[1] = (mul24(C1, b0) + mul24(C3, b1) + mul24(C5, b2) + mul24(C7, b3)) >> 16;
  r[3] = (mul24(C3, b0) - mul24(C7, b1) - mul24(C1, b2) - mul24(C5, b3)) >> 16;
  r[5] = (mul24(C5, b0) - mul24(C1, b1) + mul24(C7, b2) + mul24(C3, b3)) >> 16;
  r[7] = (mul24(C7, b0) - mul24(C5, b1) + mul24(C3, b2) - mul24(C1, b3)) >> 16;
}

// 64 samples (row-major, registers) -> 64 coefficients, x16 scaled (src/fdct.cc:596-609)
__device__ __forceinline__ void fdct_block(int* v) {
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    fdct_col8(v[x], v[8 + x], v[16 + x], v[24 + x], v[32 + x], v[40 + x], v[48 + x], v[56 + x]);
  }
  // cos(k*pi/16)/sqrt(2) tables, rows 1/7, 2/6, 3/5 pre-scaled (src/fdct.cc:28-35)
  fdct_row8<22725, 21407, 19266, 16384, 12873, 8867, 4520>(v + 0);
  fdct_row8<31521, 29692, 26722, 22725, 17855, 12299, 6270>(v + 8);
  fdct_row8<29692, 27969, 25172, 21407, 16819, 11585, 5906>(v + 16);
  fdct_row8<26722, 25172, 22654, 19266, 15137, 10426, 5315>(v + 24);
  fdct_row8<22725, 21407, 19266, 16384, 12873, 8867, 4520>(v + 32);
  fdct_row8<26722, 25172, 22654, 19266, 15137, 10426, 5315>(v + 40);
  fdct_row8<29692, 27969, 25172, 21407, 16819, 11585, 5906>(v + 48);
  fdct_row8<31521, 29692, 26722, 22725, 17855, 12299, 6270>(v + 56);
}

// zig-zag position -> natural index (JPEG Figure A.6)
__device__ constexpr int kZig(int i) {
  constexpr int z[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                         12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                         58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return z[i];
}

__device__ __forceinline__ uint32_t pack16(int lo, int hi) {
  return (static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16);
}

// BT.601 full-range 16.16 fixed point (src/colors_rgb.cc:17-19,31-32)
__device__ __forceinline__ int luma16(int r, int g, int b) {
  return (mul24(19595, r) + mul24(38469, g) + mul24(7471, b) + (32768 - (128 << 16))) >> 16;
}
__device__ __forceinline__ int cb16(int r, int g, int b, int rnd, int sh) {
  return (mul24(-11059, r) - mul24(21709, g) + mul24(32768, b) + rnd) >> sh;
}
__device__ __forceinline__ int cr16(int r, int g, int b, int rnd, int sh) {
  return (mul24(32768, r) - mul24(27439, g) - mul24(5329, b) + rnd) >> sh;
}

__device__ __forceinline__ int byte_of(const uint32_t* w, int i) {
  return static_cast<int>((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
}

// 24 bytes (8 pixels) of one row; coordinates clamp to the picture (edge replication)
__device__ __forceinline__ void load_row8(const uint8_t* frame, long long row_stride, int W, int H,
                                          int x0, int y, bool inside, uint32_t* w) {
  if (inside) {
    const uint8_t* p = frame + y * row_stride + 3ll * x0;
    __builtin_memcpy(w, p, 24);
  } else {
    const int yy = y < H ? y : H - 1;
    const uint8_t* row = frame + yy * row_stride;
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int xx = (x0 + i) < W ? (x0 + i) : W - 1;
      const uint8_t* p = row + 3ll * xx;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int bi = 3 * i + c;
        w[bi >> 2] |= static_cast<uint32_t>(p[c]) << (8 * (bi & 3));
      }
    }
  }
}

// workgroup exclusive scan of one uint32 per thread; returns exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t x, uint32_t* scratch /*>=8 u32*/,
                                                      uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    const uint32_t s = scratch[w];
    if (w < wave) base += s;
    sum += s;
  }
  __syncthreads();
  *total = sum;
  return base + incl - x;
}

// ------------------------------------------------------------------------------------
// K1: colour + fDCT + quantize + entropy-code one segment

template <int MODE, bool TAP>
__global__ __launch_bounds__(kThreads) void scan_segments(const ScanArgs a) {
  using G = Geo<MODE>;
  constexpr int BPM = G::kBpm;
  constexpr int PX = G::kMcuPx;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* const win = reinterpret_cast<uint32_t*>(smem + kOffWin);
  uint2* const lq = reinterpret_cast<uint2*>(smem + kOffQ);
  uint32_t* const lac = reinterpret_cast<uint32_t*>(smem + kOffAc);
  uint32_t* const ldc = reinterpret_cast<uint32_t*>(smem + kOffDc);
  int* const dcs = reinterpret_cast<int*>(smem + kOffDcs);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + kOffMisc);

  const int tid = threadIdx.x;
  const int seg = blockIdx.x, frame = blockIdx.y;
  const int m_first = seg * G::kSegMcus;                       // first coded MCU of the segment
  const int n_coded = min(G::kSegMcus, a.n_mcus - m_first);
  const int halo = m_first > 0 ? 1 : 0;                        // previous MCU: DC predictors only
  const uint8_t* const frame_px = a.rgb + frame * a.frame_stride;

  // tables -> LDS
  {
    const DevTables* t = a.tables;
    if (tid < 128) lq[tid] = (&t->q[0][0])[tid];
    for (int i = tid; i < 512; i += kThreads) lac[i] = (&t->ac[0][0])[i];
    if (tid < 24) ldc[tid] = (&t->dc[0][0])[tid];
  }

  // ---- P1: colour conversion, strips of 8 pixels (x2 rows for 4:2:0) --------------------
  // local MCU index ml: 0 = halo, 1..n_coded = coded MCUs; block slot = ml*BPM + k
  {
    const int ml_lo = 1 - halo;
    const int n_proc = n_coded + halo;
    constexpr int kRowsPerStrip = (MODE == SJPEG_HIP_YUV420) ? 2 : 1;
    constexpr int kStripsX = PX / 8;                           // strips per MCU row
    const int per_row = kStripsX * n_proc;
    const int nstrips = 8 * per_row;
    const int rnd_y = 0;
    (void)rnd_y;
    for (int s = tid; s < nstrips; s += kThreads) {
      const int yp = s / per_row;
      const int rem = s - yp * per_row;
      const int ml = ml_lo + rem / kStripsX;
      const int xs = rem % kStripsX;
      const int mcu = m_first - 1 + ml;
      const int mb_y = mcu / a.mb_w;
      const int mb_x = mcu - mb_y * a.mb_w;
      const int x0 = mb_x * PX + xs * 8;
      const int y0 = mb_y * PX + yp * kRowsPerStrip;
      const bool inside = (x0 + 8 <= a.W) && (y0 + kRowsPerStrip <= a.H);
      uint32_t w0[6];
      load_row8(frame_px, a.row_stride, a.W, a.H, x0, y0, inside, w0);
      if (MODE == SJPEG_HIP_YUV420) {
        uint32_t w1[6];
        load_row8(frame_px, a.row_stride, a.W, a.H, x0, y0 + 1, inside, w1);
        int ya[8], yb[8];
        int U[4], V[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int R = 0, Gs = 0, B = 0;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = 2 * c + e;
            const int r0 = byte_of(w0, 3 * i), g0 = byte_of(w0, 3 * i + 1), b0 = byte_of(w0, 3 * i + 2);
            const int r1 = byte_of(w1, 3 * i), g1 = byte_of(w1, 3 * i + 1), b1 = byte_of(w1, 3 * i + 2);
            ya[i] = luma16(r0, g0, b0);
            yb[i] = luma16(r1, g1, b1);
            R += r0 + r1; Gs += g0 + g1; B += b0 + b1;
          }
          U[c] = cb16(R, Gs, B, 32768 << 2, 18);
          V[c] = cr16(R, Gs, B, 32768 << 2, 18);
        }
        const int k = (yp >> 2) * 2 + xs;
        const int row = (yp & 3) * 2;
        unsigned char* ys = smem + (ml * BPM + k) * kSlotBytes + row * 16;
        *reinterpret_cast<uint4*>(ys) =
            make_uint4(pack16(ya[0], ya[1]), pack16(ya[2], ya[3]), pack16(ya[4], ya[5]), pack16(ya[6], ya[7]));
        *reinterpret_cast<uint4*>(ys + 16) =
            make_uint4(pack16(yb[0], yb[1]), pack16(yb[2], yb[3]), pack16(yb[4], yb[5]), pack16(yb[6], yb[7]));
        unsigned char* us = smem + (ml * BPM + 4) * kSlotBytes + yp * 16 + xs * 8;
        *reinterpret_cast<uint2*>(us) = make_uint2(pack16(U[0], U[1]), pack16(U[2], U[3]));
        *reinterpret_cast<uint2*>(us + kSlotBytes) = make_uint2(pack16(V[0], V[1]), pack16(V[2], V[3]));
      } else {
        int yv[8], uv[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = byte_of(w0, 3 * i), g = byte_of(w0, 3 * i + 1), b = byte_of(w0, 3 * i + 2);
          yv[i] = luma16(r, g, b);
          if (MODE == SJPEG_HIP_YUV444) {
            uv[i] = cb16(r, g, b, 32768, 16);
            vv[i] = cr16(r, g, b, 32768, 16);
          }
        }
        unsigned char* ys = smem + (ml * BPM) * kSlotBytes + yp * 16;
        *reinterpret_cast<uint4*>(ys) =
            make_uint4(pack16(yv[0], yv[1]), pack16(yv[2], yv[3]), pack16(yv[4], yv[5]), pack16(yv[6], yv[7]));
        if (MODE == SJPEG_HIP_YUV444) {
          *reinterpret_cast<uint4*>(ys + kSlotBytes) =
              make_uint4(pack16(uv[0], uv[1]), pack16(uv[2], uv[3]), pack16(uv[4], uv[5]), pack16(uv[6], uv[7]));
          *reinterpret_cast<uint4*>(ys + 2 * kSlotBytes) =
              make_uint4(pack16(vv[0], vv[1]), pack16(vv[2], vv[3]), pack16(vv[4], vv[5]), pack16(vv[6], vv[7]));
        }
      }
    }
  }
  __syncthreads();
  if (a.ablate == 1) return;

  // ---- P2: one thread per block: fix-up, fDCT, quantize ---------------------------------
  const int ml = tid / BPM;                    // local MCU (0 = halo)
  const int k = tid - ml * BPM;                // block inside the MCU
  const bool has_block = (ml <= n_coded) && (ml >= 1 || halo);
  const bool emits = (ml >= 1) && (ml <= n_coded);
  const int tbl = (MODE == SJPEG_HIP_YUV420) ? (k >= 4) : (MODE == SJPEG_HIP_YUV444 ? (k >= 1) : 0);
  unsigned char* const slot = smem + tid * kSlotBytes;

  int v[64];
  if (has_block) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint4 q = *reinterpret_cast<const uint4*>(slot + 16 * r);
      const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[8 * r + 2 * c] = static_cast<int>(static_cast<int16_t>(u[c] & 0xffffu));
        v[8 * r + 2 * c + 1] = static_cast<int>(u[c]) >> 16;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = 0;
  }

  if (MODE == SJPEG_HIP_YUV420 && a.has_clip) {
    // AverageExtraLuma (src/encoders.cc:107-125): luma blocks wholly outside the picture
    // become flat at (sum + 32) >> 6 of a neighbouring real block.
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) sum += v[i];
    dcs[tid] = sum;
    __syncthreads();
    if (has_block && k >= 1 && k <= 3) {
      const int mcu = m_first - 1 + ml;
      const int mb_y = mcu / a.mb_w;
      const int mb_x = mcu - mb_y * a.mb_w;
      const int sub_w = a.W - mb_x * 16, sub_h = a.H - mb_y * 16;
      int src = -1;
      if (k == 1) {
        if (sub_w <= 8) src = 0;
      } else if (sub_h <= 8) {
        src = (sub_w > 8) ? 1 : 0;
      } else if (k == 3 && sub_w <= 8) {
        src = 2;
      }
      if (src >= 0) {
        const int flat = (dcs[ml * BPM + src] + 32) >> 6;
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = flat;
      }
    }
    __syncthreads();
  }

  fdct_block(v);

  // quantize: level = ((|c| + bias) * iquant) >> 20 == (|c|*iquant + bias*iquant) >> 20.
  // The reference's qthresh test is implied: |c| >= qthresh <=> level > 0 (quantize.cc:144-145).
  uint32_t nz_lo = 0, nz_hi = 0;
  int dc_val;
  {
    const uint2* qt = lq + tbl * 64;
    int q[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const uint2 e = qt[j];
      const int c = v[j];
      const int m = c >> 31;
      const uint32_t mag = static_cast<uint32_t>((c ^ m) - m);
      const uint32_t lvl = (__umul24(mag, e.x) + e.y) >> 20;
      q[j] = (static_cast<int>(lvl) ^ m) - m;
    }
    dc_val = q[0];
    // zig-zag reorder: 4 int16 per ds_write_b64; non-zero mask over AC positions
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
      const int c0 = q[kZig(i)], c1 = q[kZig(i + 1)], c2 = q[kZig(i + 2)], c3 = q[kZig(i + 3)];
      *reinterpret_cast<uint2*>(slot + 2 * i) = make_uint2(pack16(c0, c1), pack16(c2, c3));
      const uint32_t bits = (c0 != 0 ? 1u : 0u) | (c1 != 0 ? 2u : 0u) | (c2 != 0 ? 4u : 0u) | (c3 != 0 ? 8u : 0u);
      if (i < 32) nz_lo |= bits << i; else nz_hi |= bits << (i - 32);
    }
    nz_lo &= ~1u;                               // DC is coded separately
  }
  if (TAP) {
    if (emits) {
      const long long nblk_frame = static_cast<long long>(a.n_mcus) * BPM;
      const long long blk = frame * nblk_frame + static_cast<long long>(m_first - 1 + ml) * BPM + k;
      const uint4* src = reinterpret_cast<const uint4*>(slot);
      uint4* dst = reinterpret_cast<uint4*>(a.coeffs + blk * 64);
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = src[i];
    }
  }

  if (a.ablate == 2) { if (nz_lo + nz_hi + dc_val == 0x7fffffff) a.seg_nbits[0] = 1; return; }

  // ---- P3: entropy coding ----------------------------------------------------------------
  dcs[tid] = dc_val;
  __syncthreads();
  int pred = 0;
  {
    int prev;   // slot holding the previous block of the same component, stream order
    if (MODE == SJPEG_HIP_YUV420) prev = (k == 0) ? tid - 3 : (k <= 3 ? tid - 1 : tid - 6);
    else prev = tid - BPM;
    const bool prev_in_halo = prev < BPM;
    if (emits && !(prev_in_halo && !halo)) pred = dcs[prev];
  }
  const uint32_t* const ac = lac + tbl * 256;
  uint32_t dc_bits, dc_len;
  {
    const int diff = dc_val - pred;
    const int ad = diff < 0 ? -diff : diff;
    const int n = 32 - __clz(ad);                 // 0 for diff == 0 (clz(0) == 32)
    const uint32_t suffix = static_cast<uint32_t>(diff < 0 ? diff - 1 : diff) & ((1u << n) - 1u);
    const uint32_t code = ldc[tbl * 12 + n];
    dc_bits = ((code >> 16) << n) | suffix;
    dc_len = (code & 0xffu) + n;
  }
  const unsigned long long nz = (static_cast<unsigned long long>(nz_hi) << 32) | nz_lo;
  const int16_t* const zz = reinterpret_cast<const int16_t*>(slot);
  const uint32_t zrl = ac[0xf0], eob = ac[0x00];

  // pass 1: bit length of the block
  uint32_t len = 0;
  if (emits) {
    len = dc_len;
    unsigned long long m = nz;
    int prev = 1;
    while (m) {
      const int i = __builtin_ctzll(m);
      m &= m - 1;
      const int c = zz[i];
      const int run = i - prev;
      prev = i + 1;
      const int mag = c < 0 ? -c : c;
      const int n = 32 - __clz(mag);
      len += (run >> 4) * (zrl & 0xffu) + (ac[((run & 15) << 4) | n] & 0xffu) + n;
    }
    if (prev <= 63) len += eob & 0xffu;          // last non-zero index < 63
  }
  uint32_t total;
  const uint32_t start = wg_exclusive_scan(len, misc, &total);
  const uint32_t end = start + len;
  if (a.ablate == 3) { if (tid == 0) a.seg_nbits[static_cast<size_t>(frame) * a.nseg + seg] = total; return; }

  // pass 2: emit into the LDS window, round by round (one round unless the segment
  // overflows the window); words are MSB-first, flushed coalesced to the segment's slot.
  uint32_t* const out_words = a.seg_words + (static_cast<size_t>(frame) * a.nseg + seg) * a.slot_words;
  uint32_t base = 0;                               // bit position of window word 0, multiple of 32
  uint32_t carry = 0;
  bool done = !emits;
  for (;;) {
    for (int i = tid; i < kWinWords; i += kThreads) win[i] = 0;
    if (tid == 0) misc[8] = total;
    __syncthreads();
    if (tid == 0 && carry != 0) atomicOr(&win[0], carry);
    const bool fits = !done && (end <= base + kWinWords * 32u);
    if (!done && !fits) atomicMin(&misc[8], start);
    __syncthreads();
    // blocks are in stream order, so the set that fits is a prefix of the remaining ones:
    // everything that starts before the first non-fitting block is emitted this round.
    const uint32_t limit = misc[8];
    if (fits && start < limit) {
      uint32_t pos = start - base;
      uint32_t wi = pos >> 5;
      int o = pos & 31;
      unsigned long long acc = 0;
      auto put = [&](uint32_t bits, int nb) {
        acc |= static_cast<unsigned long long>(bits) << (64 - o - nb);
        o += nb;
        if (o >= 32) {
          atomicOr(&win[wi], static_cast<uint32_t>(acc >> 32));
          acc <<= 32;
          o -= 32;
          ++wi;
        }
      };
      put(dc_bits, dc_len);
      unsigned long long m = nz;
      int prev = 1;
      while (m) {
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        const int c = zz[i];
        int run = i - prev;
        prev = i + 1;
        while (run >= 16) { put(zrl >> 16, zrl & 0xffu); run -= 16; }
        const int mag = c < 0 ? -c : c;
        const int n = 32 - __clz(mag);
        const uint32_t suffix = static_cast<uint32_t>(c < 0 ? c - 1 : c) & ((1u << n) - 1u);
        const uint32_t code = ac[(run << 4) | n];
        put(((code >> 16) << n) | suffix, (code & 0xffu) + n);
      }
      if (prev <= 63) put(eob >> 16, eob & 0xffu);
      if (o > 0) atomicOr(&win[wi], static_cast<uint32_t>(acc >> 32));
      done = true;
    }
    __syncthreads();
    const uint32_t filled = limit - base;          // bits valid in the window
    const bool last = (limit == total);
    const uint32_t nfull = last ? (filled + 31) >> 5 : filled >> 5;
    for (uint32_t i = tid; i < nfull; i += kThreads) out_words[(base >> 5) + i] = win[i];
    if (last) break;
    carry = win[filled >> 5];                      // partial word carried into the next window
    base += filled & ~31u;
    __syncthreads();
  }
  if (tid == 0) a.seg_nbits[static_cast<size_t>(frame) * a.nseg + seg] = total;
}

// ------------------------------------------------------------------------------------
// K2: per frame, exclusive scan of segment bit lengths

struct StitchArgs {
  int nseg, nframes;
  const uint32_t* seg_nbits;
  unsigned long long* seg_off;       // [nframes][nseg+1]
  const uint32_t* seg_words;
  uint32_t slot_words;
  uint32_t* ubuf;                    // [nframes][ubuf_words] un-stuffed stream, MSB-first words
  size_t ubuf_words;
  uint32_t* chunk_ff;                // [nframes][max_chunks]
  unsigned long long* chunk_off;     // [nframes][max_chunks]
  uint32_t max_chunks;
  const uint8_t* header;
  uint32_t header_size;
  int append_eoi;
  uint8_t* out;
  size_t out_stride;
  unsigned long long* sizes;
};

__global__ __launch_bounds__(kThreads) void scan_seg_offsets(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  const int frame = blockIdx.x;
  const uint32_t* nb = a.seg_nbits + static_cast<size_t>(frame) * a.nseg;
  unsigned long long* off = a.seg_off + static_cast<size_t>(frame) * (a.nseg + 1);
  unsigned long long running = 0;
  for (int base = 0; base < a.nseg; base += kThreads) {
    const int i = base + threadIdx.x;
    const uint32_t x = i < a.nseg ? nb[i] : 0u;
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan(x, scratch, &total);
    if (i < a.nseg) off[i] = running + ex;
    running += total;
  }
  if (threadIdx.x == 0) off[a.nseg] = running;
}

// ------------------------------------------------------------------------------------
// K3: gather the continuous bit stream, 4 KiB chunks, and count 0xFF bytes

__device__ __forceinline__ uint32_t count_ff(uint32_t w, int nbytes /*valid leading bytes, MSB first*/) {
  uint32_t n = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b < nbytes && ((w >> (24 - 8 * b)) & 0xffu) == 0xffu) ++n;
  }
  return n;
}

__global__ __launch_bounds__(kThreads) void concat_chunks(const StitchArgs a) {
  __shared__ uint32_t red[8];
  const int frame = blockIdx.y;
  const unsigned long long* off = a.seg_off + static_cast<size_t>(frame) * (a.nseg + 1);
  const unsigned long long T = off[a.nseg];                 // total bits
  const unsigned long long U = (T + 7) >> 3;                // bytes incl. 1-bit padding
  const uint32_t nchunks = static_cast<uint32_t>((U + kChunkBytes - 1) / kChunkBytes);
  const uint32_t* segw = a.seg_words + static_cast<size_t>(frame) * a.nseg * a.slot_words;
  uint32_t* ub = a.ubuf + static_cast<size_t>(frame) * a.ubuf_words;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const unsigned long long w0 = static_cast<unsigned long long>(chunk) * kChunkWords + threadIdx.x * 4;
    unsigned long long pos = w0 * 32;
    uint32_t ffs = 0;
    if (pos < U * 8) {
      // segment containing bit `pos`: largest s with off[s] <= pos
      int lo = 0, hi = a.nseg - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= pos) lo = mid; else hi = mid - 1;
      }
      int s = lo;
      unsigned long long s_beg = off[s], s_end = off[s + 1];
      uint32_t words[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t outw = 0;
        int need = 32;
        while (need > 0 && pos < T) {
          while (pos >= s_end) { ++s; s_beg = s_end; s_end = off[s + 1]; }
          const unsigned long long avail = s_end - pos;
          const int take = avail < static_cast<unsigned long long>(need) ? static_cast<int>(avail) : need;
          const uint32_t r = static_cast<uint32_t>(pos - s_beg);
          const uint32_t* p = segw + static_cast<size_t>(s) * a.slot_words + (r >> 5);
          const unsigned long long two = (static_cast<unsigned long long>(p[0]) << 32) | p[1];
          const uint32_t bits = static_cast<uint32_t>((two << (r & 31)) >> (64 - take));
          outw |= bits << (need - take);
          need -= take;
          pos += take;
        }
        if (need > 0) {                                    // past the end: pad with 1-bits
          outw |= (need == 32) ? 0xffffffffu : ((1u << need) - 1u);
          pos += need;
        }
        words[j] = outw;
        const unsigned long long byte0 = (w0 + j) * 4;
        const int valid = byte0 >= U ? 0 : (U - byte0 >= 4 ? 4 : static_cast<int>(U - byte0));
        ffs += count_ff(outw, valid);
      }
      *reinterpret_cast<uint4*>(ub + w0) = make_uint4(words[0], words[1], words[2], words[3]);
    }
    // workgroup sum of 0xFF counts
    for (int d = 32; d > 0; d >>= 1) ffs += __shfl_down(ffs, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ffs;
    __syncthreads();
    if (threadIdx.x == 0) {
      a.chunk_ff[static_cast<size_t>(frame) * a.max_chunks + chunk] = red[0] + red[1] + red[2] + red[3];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------
// K4: per frame, exclusive scan of per-chunk 0xFF counts; final stream size

__global__ __launch_bounds__(kThreads) void scan_chunk_offsets(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  const int frame = blockIdx.x;
  const unsigned long long T = a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg];
  const unsigned long long U = (T + 7) >> 3;
  const uint32_t nchunks = static_cast<uint32_t>((U + kChunkBytes - 1) / kChunkBytes);
  const uint32_t* ff = a.chunk_ff + static_cast<size_t>(frame) * a.max_chunks;
  unsigned long long* co = a.chunk_off + static_cast<size_t>(frame) * a.max_chunks;
  unsigned long long running = 0;
  for (uint32_t base = 0; base < nchunks; base += kThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t x = i < nchunks ? ff[i] : 0u;
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan(x, scratch, &total);
    if (i < nchunks) co[i] = running + ex;
    running += total;
  }
  // a frame that does not fit the caller's slot reports size 0 and is not written
  const unsigned long long body = U + running;
  const unsigned long long size = a.header_size + body + (a.append_eoi ? 2 : 0);
  const bool fits = size <= a.out_stride;
  uint8_t* dst = a.out + static_cast<size_t>(frame) * a.out_stride;
  if (threadIdx.x == 0) {
    if (fits && a.append_eoi) {
      dst[a.header_size + body] = 0xff;
      dst[a.header_size + body + 1] = 0xd9;
    }
    a.sizes[frame] = fits ? size : 0ull;
  }
  // header bytes in front of the entropy segment
  if (fits) {
    for (uint32_t i = threadIdx.x; i < a.header_size; i += kThreads) dst[i] = a.header[i];
  }
}

// ------------------------------------------------------------------------------------
// K5: byte stuffing into the caller's slot

__global__ __launch_bounds__(kThreads) void stuff_chunks(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  const int frame = blockIdx.y;
  const unsigned long long T = a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg];
  const unsigned long long U = (T + 7) >> 3;
  const uint32_t nchunks = static_cast<uint32_t>((U + kChunkBytes - 1) / kChunkBytes);
  const uint32_t* ub = a.ubuf + static_cast<size_t>(frame) * a.ubuf_words;
  const unsigned long long* co = a.chunk_off + static_cast<size_t>(frame) * a.max_chunks;
  uint8_t* const dst0 = a.out + static_cast<size_t>(frame) * a.out_stride + a.header_size;
  if (a.sizes[frame] == 0) return;                          // did not fit (see K4)
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const unsigned long long w0 = static_cast<unsigned long long>(chunk) * kChunkWords + threadIdx.x * 4;
    const unsigned long long byte0 = w0 * 4;
    uint4 q = make_uint4(0, 0, 0, 0);
    int valid = 0;
    if (byte0 < U) {
      q = *reinterpret_cast<const uint4*>(ub + w0);
      valid = (U - byte0 >= 16) ? 16 : static_cast<int>(U - byte0);
    }
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t ffs = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) ffs += count_ff(w[j], valid - 4 * j);
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan(ffs, scratch, &total);
    uint8_t* d = dst0 + byte0 + co[chunk] + ex;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < valid) {
        const uint8_t b = static_cast<uint8_t>(w[j >> 2] >> (24 - 8 * (j & 3)));
        *d++ = b;
        if (b == 0xff) *d++ = 0x00;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// host side

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    const hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                             \
      return fail(e_ == hipErrorOutOfMemory ? SJPEG_HIP_ENOMEM : SJPEG_HIP_ERUNTIME,    \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    }                                                                                   \
  } while (0)

struct FrameGeo {
  int bpm, px, seg_mcus, mb_w, mb_h, n_mcus, nseg;
  uint32_t slot_words;
};

bool frame_geo(int W, int H, int mode, FrameGeo* g) {
  if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return false;   // src/enc.cc:406
  switch (mode) {
    case SJPEG_HIP_YUV420: g->bpm = 6; g->px = 16; g->seg_mcus = Geo<SJPEG_HIP_YUV420>::kSegMcus; break;
    case SJPEG_HIP_YUV444: g->bpm = 3; g->px = 8; g->seg_mcus = Geo<SJPEG_HIP_YUV444>::kSegMcus; break;
    case SJPEG_HIP_YUV400: g->bpm = 1; g->px = 8; g->seg_mcus = Geo<SJPEG_HIP_YUV400>::kSegMcus; break;
    default: return false;
  }
  g->mb_w = (W + g->px - 1) / g->px;                              // src/enc.cc:410-411
  g->mb_h = (H + g->px - 1) / g->px;
  g->n_mcus = g->mb_w * g->mb_h;
  g->nseg = (g->n_mcus + g->seg_mcus - 1) / g->seg_mcus;
  g->slot_words = (static_cast<uint32_t>(g->seg_mcus) * g->bpm * kMaxBlockBits + 31) / 32 + 2;
  return true;
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;     // elements
  int ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (e != hipSuccess) {
      return fail(SJPEG_HIP_ENOMEM, std::string("hipMalloc(") + std::to_string(n * sizeof(T)) +
                                        "): " + hipGetErrorString(e));
    }
    cap = n;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct sjpeg_hip_engine {
  int device = 0;
  DevBuf<DevTables> tables;
  DevBuf<uint8_t> header;
  DevBuf<uint32_t> seg_words, seg_nbits, ubuf, chunk_ff;
  DevBuf<unsigned long long> seg_off, chunk_off;
  bool timing = false;
  int ablate = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ev_valid = false;
};

namespace {

void digest_tables(const sjpeg_hip_scan_tables* t, DevTables* d) {
  for (int c = 0; c < 2; ++c) {
    for (int j = 0; j < 64; ++j) {
      d->q[c][j].x = t->iquant[c][j];
      d->q[c][j].y = static_cast<uint32_t>(t->bias[c][j]) * t->iquant[c][j];
    }
  }
  memcpy(d->dc, t->dc_codes, sizeof(d->dc));
  memcpy(d->ac, t->ac_codes, sizeof(d->ac));
}

template <bool TAP>
int launch_scan(int mode, dim3 grid, hipStream_t st, const ScanArgs& a) {
  switch (mode) {
    case SJPEG_HIP_YUV420:
      hipLaunchKernelGGL((scan_segments<SJPEG_HIP_YUV420, TAP>), grid, dim3(kThreads), kLdsBytes, st, a);
      break;
    case SJPEG_HIP_YUV444:
      hipLaunchKernelGGL((scan_segments<SJPEG_HIP_YUV444, TAP>), grid, dim3(kThreads), kLdsBytes, st, a);
      break;
    default:
      hipLaunchKernelGGL((scan_segments<SJPEG_HIP_YUV400, TAP>), grid, dim3(kThreads), kLdsBytes, st, a);
      break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int prepare_scan(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride, int64_t frame_stride,
                 int W, int H, int mode, int nframes, const sjpeg_hip_scan_tables* tables,
                 hipStream_t st, FrameGeo* g, ScanArgs* a) {
  if (e == nullptr || d_rgb == nullptr || tables == nullptr || nframes <= 0) {
    return fail(SJPEG_HIP_EINVAL, "null argument or nframes <= 0");
  }
  if (!frame_geo(W, H, mode, g)) return fail(SJPEG_HIP_EINVAL, "bad dimensions or yuv_mode");
  const int64_t abs_stride = row_stride < 0 ? -row_stride : row_stride;
  if (abs_stride < 3ll * W) return fail(SJPEG_HIP_EINVAL, "|row_stride| < 3*width");
  if (nframes > 65535) return fail(SJPEG_HIP_EINVAL, "nframes > 65535");
  HIP_TRY(hipSetDevice(e->device));
  int rc;
  if ((rc = e->tables.ensure(1))) return rc;
  const size_t total_segs = static_cast<size_t>(nframes) * g->nseg;
  if ((rc = e->seg_words.ensure(total_segs * g->slot_words))) return rc;
  if ((rc = e->seg_nbits.ensure(total_segs))) return rc;
  DevTables host_tables;
  digest_tables(tables, &host_tables);
  HIP_TRY(hipMemcpyAsync(e->tables.p, &host_tables, sizeof(DevTables), hipMemcpyHostToDevice, st));
  a->rgb = static_cast<const uint8_t*>(d_rgb);
  a->row_stride = row_stride;
  a->frame_stride = frame_stride;
  a->W = W; a->H = H; a->mb_w = g->mb_w; a->n_mcus = g->n_mcus; a->nseg = g->nseg;
  a->has_clip = (W % g->px != 0) || (H % g->px != 0);
  a->tables = e->tables.p;
  a->seg_words = e->seg_words.p;
  a->slot_words = g->slot_words;
  a->seg_nbits = e->seg_nbits.p;
  a->coeffs = nullptr;
  a->ablate = e->ablate;
  return 0;
}

}  // namespace

extern "C" {

int sjpeg_hip_abi_version(void) { return SJPEG_HIP_ABI_VERSION; }

const char* sjpeg_hip_last_error(void) { return g_last_error.c_str(); }

int sjpeg_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sjpeg_hip_engine_create(int device, sjpeg_hip_engine** engine) {
  if (engine == nullptr) return fail(SJPEG_HIP_EINVAL, "engine == NULL");
  *engine = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    return fail(SJPEG_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
  }
  if (device < 0 || device >= n) return fail(SJPEG_HIP_EINVAL, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    return fail(SJPEG_HIP_ENODEV, std::string("device is ") + prop.gcnArchName +
                                      ", this library is built for gfx950 only");
  }
  sjpeg_hip_engine* e = new (std::nothrow) sjpeg_hip_engine;
  if (e == nullptr) return fail(SJPEG_HIP_ENOMEM, "host allocation failed");
  e->device = device;
  if (const char* ab = getenv("SJPEG_HIP_ABLATE")) e->ablate = atoi(ab);   // profiling only
  *engine = e;
  return 0;
}

void sjpeg_hip_engine_destroy(sjpeg_hip_engine* e) {
  if (e == nullptr) return;
  (void)hipSetDevice(e->device);
  e->tables.release(); e->header.release(); e->seg_words.release(); e->seg_nbits.release();
  e->ubuf.release(); e->chunk_ff.release(); e->seg_off.release(); e->chunk_off.release();
  for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  delete e;
}

size_t sjpeg_hip_frame_bound(int width, int height, int yuv_mode, size_t header_size) {
  FrameGeo g;
  if (!frame_geo(width, height, yuv_mode, &g)) return 0;
  // un-stuffed worst case = slots; stuffing at most doubles it
  const size_t unstuffed = static_cast<size_t>(g.nseg) * g.slot_words * 4;
  return header_size + 2 * unstuffed + 2 + 64;
}

int sjpeg_hip_engine_set_timing(sjpeg_hip_engine* e, int enable) {
  if (e == nullptr) return fail(SJPEG_HIP_EINVAL, "engine == NULL");
  HIP_TRY(hipSetDevice(e->device));
  if (enable && e->ev[0] == nullptr) {
    for (auto& ev : e->ev) HIP_TRY(hipEventCreate(&ev));
  }
  e->timing = enable != 0;
  e->ev_valid = false;
  return 0;
}

static float elapsed(sjpeg_hip_engine* e, int i0, int i1) {
  if (e == nullptr || !e->ev_valid) return -1.f;
  if (hipEventSynchronize(e->ev[i1]) != hipSuccess) return -1.f;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, e->ev[i0], e->ev[i1]) != hipSuccess) return -1.f;
  return ms;
}
float sjpeg_hip_engine_last_scan_ms(sjpeg_hip_engine* e) { return elapsed(e, 0, 1); }
float sjpeg_hip_engine_last_total_ms(sjpeg_hip_engine* e) { return elapsed(e, 0, 2); }

int sjpeg_hip_scan_coeffs(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride,
                          int64_t frame_stride, int width, int height, int yuv_mode, int nframes,
                          const sjpeg_hip_scan_tables* tables, int16_t* d_coeffs, void* stream) {
  if (d_coeffs == nullptr) return fail(SJPEG_HIP_EINVAL, "d_coeffs == NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameGeo g;
  ScanArgs a;
  const int rc = prepare_scan(e, d_rgb, row_stride, frame_stride, width, height, yuv_mode, nframes,
                              tables, st, &g, &a);
  if (rc) return rc;
  a.coeffs = d_coeffs;
  return launch_scan<true>(yuv_mode, dim3(g.nseg, nframes), st, a);
}

int sjpeg_hip_encode_scan(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride,
                          int64_t frame_stride, int width, int height, int yuv_mode, int nframes,
                          const sjpeg_hip_scan_tables* tables, const void* header,
                          size_t header_size, int append_eoi, void* d_out, size_t out_stride,
                          uint64_t* d_sizes, void* stream) {
  if (d_out == nullptr || d_sizes == nullptr) return fail(SJPEG_HIP_EINVAL, "d_out/d_sizes == NULL");
  if (header == nullptr) header_size = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameGeo g;
  ScanArgs a;
  int rc = prepare_scan(e, d_rgb, row_stride, frame_stride, width, height, yuv_mode, nframes,
                        tables, st, &g, &a);
  if (rc) return rc;
  if (out_stride < header_size + 2 + 64) {
    return fail(SJPEG_HIP_ECAPACITY, "out_stride " + std::to_string(out_stride) + " too small");
  }
  const size_t ubuf_words = (static_cast<size_t>(g.nseg) * g.slot_words + kChunkWords + 3) & ~size_t(3);
  const uint32_t max_chunks = static_cast<uint32_t>((ubuf_words + kChunkWords - 1) / kChunkWords);
  if ((rc = e->seg_off.ensure(static_cast<size_t>(nframes) * (g.nseg + 1)))) return rc;
  if ((rc = e->ubuf.ensure(static_cast<size_t>(nframes) * ubuf_words))) return rc;
  if ((rc = e->chunk_ff.ensure(static_cast<size_t>(nframes) * max_chunks))) return rc;
  if ((rc = e->chunk_off.ensure(static_cast<size_t>(nframes) * max_chunks))) return rc;
  if ((rc = e->header.ensure(header_size > 0 ? header_size : 1))) return rc;
  if (header_size > 0) {
    HIP_TRY(hipMemcpyAsync(e->header.p, header, header_size, hipMemcpyHostToDevice, st));
  }

  StitchArgs s;
  s.nseg = g.nseg; s.nframes = nframes;
  s.seg_nbits = e->seg_nbits.p; s.seg_off = e->seg_off.p;
  s.seg_words = e->seg_words.p; s.slot_words = g.slot_words;
  s.ubuf = e->ubuf.p; s.ubuf_words = ubuf_words;
  s.chunk_ff = e->chunk_ff.p; s.chunk_off = e->chunk_off.p; s.max_chunks = max_chunks;
  s.header = e->header.p; s.header_size = static_cast<uint32_t>(header_size);
  s.append_eoi = append_eoi;
  s.out = static_cast<uint8_t*>(d_out); s.out_stride = out_stride;
  s.sizes = reinterpret_cast<unsigned long long*>(d_sizes);

  if (e->timing) HIP_TRY(hipEventRecord(e->ev[0], st));
  if ((rc = launch_scan<false>(yuv_mode, dim3(g.nseg, nframes), st, a))) return rc;
  if (e->timing) HIP_TRY(hipEventRecord(e->ev[1], st));

  hipLaunchKernelGGL(scan_seg_offsets, dim3(nframes), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  uint32_t gx = 4096u / static_cast<uint32_t>(nframes);
  if (gx < 64) gx = 64;
  if (gx > max_chunks) gx = max_chunks;
  hipLaunchKernelGGL(concat_chunks, dim3(gx, nframes), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(scan_chunk_offsets, dim3(nframes), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(stuff_chunks, dim3(gx, nframes), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  if (e->timing) {
    HIP_TRY(hipEventRecord(e->ev[2], st));
    e->ev_valid = true;
  }
  return 0;
}

}  // extern "C"
