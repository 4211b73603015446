// sharp_yuv.hip -- SJPEG_YUV_SHARP: the iterative "sharp" RGB -> YUV 4:2:0 conversion on gfx950.
//
// Reference: /root/reference/src/yuv_convert.cc (ApplySharpYUVConversion / PreprocessARGB).  The
// picture is held as a 10-bit luma-like plane W plus chroma differences (R-W, G-W, B-W) at half
// resolution; up to four sweeps upsample the chroma (9:3:3:1), compare the result with the
// gamma-correct targets and feed the differences back.  A sweep walks the row pairs top to bottom
// and updates the chroma rows IN PLACE -- the row above a pair already belongs to this sweep, the
// row below still to the previous one -- so the row pairs of one sweep are inherently
// sequential; the parallelism is across the width of a row pair (sharp_sweeps_strips, round 6: strips of
// columns, a workgroup each on a CU of its own, which meet every 32 row pairs; before that one workgroup per
// picture and sweep with barriers between row pairs), across the SWEEPS of a picture (sweep t + 1 runs a few
// row pairs behind sweep t in workgroups of its own, on versioned planes instead of in place) and across
// the pictures of a batch.  The import and the final conversion have no such dependency and run one
// thread per chroma sample.
//
// Everything is integer arithmetic on the reference's fixed-point formats; the two gamma tables
// are built on the host with libm's pow() exactly as the reference builds them (:114-152).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "sjpeg_hip.h"

namespace {

constexpr int kSfix = 2;                          // extra precision bits of W and RGB
constexpr int kMaxY = (256 << kSfix) - 1;
constexpr int kGammaTab = 32;
constexpr int kG2LBits = 14;
constexpr int kSweepThreads = 1024;

struct GammaTables {
  uint32_t g2l[kMaxY + 1];                        // gamma -> linear, 14 fractional bits
  uint32_t l2g[kGammaTab + 2];                    // linear -> gamma, interpolated
};

struct SharpArgs {
  const uint8_t* rgb;
  long long row_stride, frame_stride;
  int pix_step, r_off, g_off, b_off;              // bytes per pixel and channel positions
  int W, H, w, h, uv_w, uv_h;                     // w, h: padded to even
  const GammaTables* tab;
  uint16_t* best_y;  uint16_t* target_y;          // [nframes][h][w]
  int16_t* best_uv;  int16_t* target_uv;          // [nframes][uv_h][3][uv_w]
  int16_t* row_uv;                                // [nframes][3][uv_w]: chroma of the row pair in flight
  // Pipelined sweeps (sharp_sweeps_piped): plane p of frame f of the W / chroma buffers is at
  // best_y + (p * nframes + f) * w * h (plane 0 = what the import wrote = best_y itself), sweep t reads plane
  // t % 3 and writes plane (t + 1) % 3; ctrl[f][32]: progress[4] (row pairs done), done[4], sum[4] (two words
  // each), cancel, final sweep.  nplanes == 1: the in-place kernels, everything in plane 0.
  uint32_t* ctrl;
  int ctrl_words;                                 // per picture: 32, or 32 + 4 * strips (sharp_sweeps_strips: progress[4][strips] behind the 32)
  int nframes, nplanes;
  int frame0;                                     // sharp_sweeps_piped: first frame of this launch (batches go in chunks)
  uint8_t* y; uint8_t* u; uint8_t* v;
  long long y_frame_stride, uv_frame_stride;
  int stress;                                     // race stress builds only (SJPEG_HIP_ABLATE), else 0
};

__device__ __forceinline__ uint32_t lin2gamma(const uint32_t* l2g, uint32_t value) {   // :158-171
  const uint32_t v = value * kGammaTab;
  const uint32_t pos = v >> kG2LBits;
  const uint32_t x = v - (pos << kG2LBits);
  const uint32_t v0 = l2g[pos], v1 = l2g[pos + 1];
  return v0 + (((v1 - v0) * x) >> kG2LBits);
}
__device__ __forceinline__ int clip_y(int y) { return y < 0 ? 0 : y > kMaxY ? kMaxY : y; }
__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
__device__ __forceinline__ uint32_t gray(uint32_t r, uint32_t g, uint32_t b) {          // :435-438
  return (13933u * r + 46871u * g + 4732u * b + (1u << 16 >> 1)) >> 16;
}

// W targets of a 2x2 group of 10-bit RGB pixels (UpdateW, :467-475) and its chroma target
// (UpdateChroma / ScaleDown, :440-465).  px[row][col][channel].
__device__ __forceinline__ void eval_group(const uint32_t* g2l, const uint32_t* l2g, const int px[2][2][3],
                                           int wout[2][2], int uv[3]) {
  uint32_t lin[2][2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int k = 0; k < 3; ++k) lin[r][c][k] = g2l[px[r][c][k]];
      wout[r][c] = static_cast<int>(lin2gamma(l2g, gray(lin[r][c][0], lin[r][c][1], lin[r][c][2])));
    }
  }
  uint32_t ch[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ch[k] = lin2gamma(l2g, (lin[0][0][k] + lin[0][1][k] + lin[1][0][k] + lin[1][1][k] + 2) >> 2);
  }
  const int Wv = static_cast<int>(gray(ch[0], ch[1], ch[2]));
#pragma unroll
  for (int k = 0; k < 3; ++k) uv[k] = static_cast<int16_t>(static_cast<int>(ch[k]) - Wv);
}

// ---- import: 8 -> 10 bits, W and chroma targets (:492-510,608-632); one thread per chroma sample
__global__ __launch_bounds__(256) void sharp_import(const SharpArgs a) {
  const int frame = blockIdx.z;
  const int c = blockIdx.x * 256 + threadIdx.x, ry = blockIdx.y;
  if (c >= a.uv_w) return;
  const uint8_t* base = a.rgb + frame * a.frame_stride;
  int px[2][2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int yy = min(2 * ry + r, a.H - 1);                 // bottom replication
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int xx = min(2 * c + cc, a.W - 1);               // right replication
      const uint8_t* p = base + yy * a.row_stride + static_cast<long long>(xx) * a.pix_step;
      px[r][cc][0] = (p[a.r_off] << kSfix) | (1 << kSfix >> 1);
      px[r][cc][1] = (p[a.g_off] << kSfix) | (1 << kSfix >> 1);
      px[r][cc][2] = (p[a.b_off] << kSfix) | (1 << kSfix >> 1);
    }
  }
  int wt[2][2], uv[3];
  eval_group(a.tab->g2l, a.tab->l2g, px, wt, uv);
  const size_t yo = static_cast<size_t>(frame) * a.w * a.h;
  const size_t uo = (static_cast<size_t>(frame) * a.uv_h + ry) * 3 * a.uv_w;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const size_t o = yo + static_cast<size_t>(2 * ry + r) * a.w + 2 * c + cc;
      a.best_y[o] = static_cast<uint16_t>(gray(px[r][cc][0], px[r][cc][1], px[r][cc][2]));   // StoreGray
      a.target_y[o] = static_cast<uint16_t>(wt[r][cc]);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a.target_uv[uo + k * a.uv_w + c] = static_cast<int16_t>(uv[k]);
    a.best_uv[uo + k * a.uv_w + c] = static_cast<int16_t>(uv[k]);
  }
}

// Race stress build (make STRESS=1|2): SHARP_RACE_POINT(n) holds the waves w with (w & 3) == k of the
// workgroup back at point n (or lets only them run on: bit 7) when SJPEG_HIP_ABLATE is
// 0x5a000000 | count << 16 | n << 8 | k -- the code of K1's RACE_POINT (scan_segments.h), points 32..47
// (tools/race_sweep.py).
#ifdef SJPEG_HIP_PRIO_STRESS
#define SHARP_RACE_POINT(n) sharp_race_point(a.stress, n)
__device__ __forceinline__ void sharp_race_point(int code, int n) {
  if ((code >> 24) == 0x5a && ((code >> 8) & 255) == n && ((((threadIdx.x >> 6) & 3) == (code & 3)) != ((code & 0x80) != 0))) {
    for (int i = 0; i < ((code >> 16) & 255); ++i) __builtin_amdgcn_s_sleep(127);
  }
}
#else
#define SHARP_RACE_POINT(n)
#endif

// ---- the sweeps (:634-668): one workgroup per picture
__global__ __launch_bounds__(kSweepThreads) void sharp_sweeps(const SharpArgs a) {
  __shared__ uint32_t g2l[kMaxY + 1];
  __shared__ uint32_t l2g[kGammaTab + 2];
  __shared__ unsigned long long red[kSweepThreads / 64];
  __shared__ int stop;
  const int frame = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i <= kMaxY; i += kSweepThreads) g2l[i] = a.tab->g2l[i];
  if (tid < kGammaTab + 2) l2g[tid] = a.tab->l2g[tid];
  uint16_t* const best_y = a.best_y + static_cast<size_t>(frame) * a.w * a.h;
  const uint16_t* const target_y = a.target_y + static_cast<size_t>(frame) * a.w * a.h;
  int16_t* const best_uv = a.best_uv + static_cast<size_t>(frame) * a.uv_h * 3 * a.uv_w;
  const int16_t* const target_uv = a.target_uv + static_cast<size_t>(frame) * a.uv_h * 3 * a.uv_w;
  int16_t* const row_uv = a.row_uv + static_cast<size_t>(frame) * 3 * a.uv_w;
  const int w = a.w, h = a.h, uv_w = a.uv_w;
  const unsigned long long threshold = static_cast<unsigned long long>(3.0 * w * h);
  unsigned long long prev_diff = ~0ull;
  SHARP_RACE_POINT(32);
  __syncthreads();
  for (int iter = 0; iter < 4; ++iter) {
    unsigned long long diff = 0;
    for (int j = 0; j < h; j += 2) {
      SHARP_RACE_POINT(33);
      const int ry = j >> 1;
      const int16_t* const cur = best_uv + static_cast<size_t>(ry) * 3 * uv_w;
      const int16_t* const prev = best_uv + static_cast<size_t>(ry > 0 ? ry - 1 : 0) * 3 * uv_w;
      const int16_t* const next = best_uv + static_cast<size_t>(j < h - 2 ? ry + 1 : ry) * 3 * uv_w;
      // the chroma this sweep measures goes to a side row until all reads of the row pair are done
      for (int c = tid; c < uv_w; c += kSweepThreads) {
        // chroma upsampled 9:3:3:1 onto the four pixels of the group, added to W
        // (InterpolateTwoRows / SharpFilterRow / Filter2, :195-203,485-541)
        int px[2][2][3];
        const int cl = c > 0 ? c - 1 : 0, cr = c < uv_w - 1 ? c + 1 : uv_w - 1;
        const int wy[2][2] = {{best_y[static_cast<size_t>(j) * w + 2 * c], best_y[static_cast<size_t>(j) * w + 2 * c + 1]},
                              {best_y[static_cast<size_t>(j + 1) * w + 2 * c], best_y[static_cast<size_t>(j + 1) * w + 2 * c + 1]}};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int A = cur[k * uv_w + c], Al = cur[k * uv_w + cl], Ar = cur[k * uv_w + cr];
          const int P = prev[k * uv_w + c], Pl = prev[k * uv_w + cl], Pr = prev[k * uv_w + cr];
          const int N = next[k * uv_w + c], Nl = next[k * uv_w + cl], Nr = next[k * uv_w + cr];
          int up0, up1, dn0, dn1;
          if (c == 0) { up0 = (A * 3 + P + 2) >> 2; dn0 = (A * 3 + N + 2) >> 2; }
          else { up0 = (A * 9 + Al * 3 + P * 3 + Pl + 8) >> 4; dn0 = (A * 9 + Al * 3 + N * 3 + Nl + 8) >> 4; }
          if (c == uv_w - 1) { up1 = (A * 3 + P + 2) >> 2; dn1 = (A * 3 + N + 2) >> 2; }
          else { up1 = (A * 9 + Ar * 3 + P * 3 + Pr + 8) >> 4; dn1 = (A * 9 + Ar * 3 + N * 3 + Nr + 8) >> 4; }
          px[0][0][k] = clip_y(wy[0][0] + up0); px[0][1][k] = clip_y(wy[0][1] + up1);
          px[1][0][k] = clip_y(wy[1][0] + dn0); px[1][1][k] = clip_y(wy[1][1] + dn1);
        }
        int wt[2][2], uv[3];
        eval_group(g2l, l2g, px, wt, uv);
        // SharpUpdateY (:175-185) on the four pixels
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const size_t o = static_cast<size_t>(j + r) * w + 2 * c + cc;
            const int d = static_cast<int>(target_y[o]) - wt[r][cc];
            best_y[o] = static_cast<uint16_t>(clip_y(wy[r][cc] + d));
            diff += static_cast<unsigned long long>(d < 0 ? -d : d);
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) row_uv[k * uv_w + c] = static_cast<int16_t>(uv[k]);
      }
      SHARP_RACE_POINT(34);
      __syncthreads();                              // all neighbours have read this chroma row
      SHARP_RACE_POINT(35);
      // SharpUpdateRGB (:187-193): the row becomes this sweep's
      for (int c = tid; c < uv_w; c += kSweepThreads) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const size_t o = static_cast<size_t>(ry) * 3 * uv_w + k * uv_w + c;
          best_uv[o] = static_cast<int16_t>(best_uv[o] + (target_uv[o] - row_uv[k * uv_w + c]));
        }
      }
      SHARP_RACE_POINT(36);
      __syncthreads();
    }
    // exit test (:660-666): sum of |dW| over the picture
    for (int d = 32; d > 0; d >>= 1) diff += __shfl_down(diff, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = diff;
    __syncthreads();
    if (tid == 0) {
      unsigned long long sum = 0;
      for (int i = 0; i < kSweepThreads / 64; ++i) sum += red[i];
      stop = (iter > 0 && (sum < threshold || sum > prev_diff)) ? 1 : 0;
      red[0] = sum;
    }
    __syncthreads();
    prev_diff = red[0];
    const int s = stop;
    __syncthreads();
    if (s) break;
  }
}

// ---- the same sweeps for pictures up to 4096 pixels wide: nothing a step needs from global
// memory is requested in that step.  The chroma rows `cur` and `next` of a row pair are last
// sweep's values and the W / target rows do not depend on the neighbours, so they are loaded one
// row pair AHEAD into registers; the one thing that does depend on the step before -- the row
// above, just updated by the neighbours -- travels through a double-buffered LDS row.  One barrier
// per row pair, and it does not wait for the loads in flight.
constexpr int kFastCols = 2;                      // chroma columns per thread
__global__ __launch_bounds__(kSweepThreads) void sharp_sweeps_fast(const SharpArgs a) {
  __shared__ uint32_t g2l[kMaxY + 1];
  __shared__ uint32_t l2g[kGammaTab + 2];
  __shared__ unsigned long long red[kSweepThreads / 64];
  __shared__ int stop;
  __shared__ int16_t above[2][3][kFastCols * kSweepThreads];   // the updated row above, ping-pong
  const int frame = blockIdx.x, tid = threadIdx.x;
#ifdef SJPEG_HIP_PRIO_STRESS
  // race stress build (Makefile STRESS=1|2): the 16 waves run at four different priorities
  switch (((tid >> 6) + SJPEG_HIP_PRIO_STRESS) & 3) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(0); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(1);
  }
#endif
  for (int i = tid; i <= kMaxY; i += kSweepThreads) g2l[i] = a.tab->g2l[i];
  if (tid < kGammaTab + 2) l2g[tid] = a.tab->l2g[tid];
  uint16_t* const best_y = a.best_y + static_cast<size_t>(frame) * a.w * a.h;
  const uint16_t* const target_y = a.target_y + static_cast<size_t>(frame) * a.w * a.h;
  int16_t* const best_uv = a.best_uv + static_cast<size_t>(frame) * a.uv_h * 3 * a.uv_w;
  const int16_t* const target_uv = a.target_uv + static_cast<size_t>(frame) * a.uv_h * 3 * a.uv_w;
  const int w = a.w, h = a.h, uv_w = a.uv_w, uv_h = a.uv_h;
  const unsigned long long threshold = static_cast<unsigned long long>(3.0 * w * h);
  unsigned long long prev_diff = ~0ull;
  // a barrier that orders LDS traffic only: global loads stay in flight across it
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  struct RowData {                                  // what one row pair needs from global memory, per column
    int uv[3][3];                                   // chroma row: [channel][left, centre, right]
    uint32_t wy[2], ty[2];                          // W and its target: two pixels per word, two rows
    int tuv[3];                                     // chroma target
  };
  auto load_uv = [&](int row, int c, int (&dst)[3][3]) {
    const int cl = c > 0 ? c - 1 : 0, cr = c < uv_w - 1 ? c + 1 : uv_w - 1;
    const int16_t* r = best_uv + static_cast<size_t>(row) * 3 * uv_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) { dst[k][0] = r[k * uv_w + cl]; dst[k][1] = r[k * uv_w + c]; dst[k][2] = r[k * uv_w + cr]; }
  };
  auto load_rest = [&](int ry, int c, RowData& d) {
    const uint32_t* by0 = reinterpret_cast<const uint32_t*>(best_y + static_cast<size_t>(2 * ry) * w);
    const uint32_t* by1 = reinterpret_cast<const uint32_t*>(best_y + static_cast<size_t>(2 * ry + 1) * w);
    const uint32_t* ty0 = reinterpret_cast<const uint32_t*>(target_y + static_cast<size_t>(2 * ry) * w);
    const uint32_t* ty1 = reinterpret_cast<const uint32_t*>(target_y + static_cast<size_t>(2 * ry + 1) * w);
    d.wy[0] = by0[c]; d.wy[1] = by1[c]; d.ty[0] = ty0[c]; d.ty[1] = ty1[c];
    const int16_t* t = target_uv + static_cast<size_t>(ry) * 3 * uv_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) d.tuv[k] = t[k * uv_w + c];
  };
  SHARP_RACE_POINT(40);
  __syncthreads();
  for (int iter = 0; iter < 4; ++iter) {
    SHARP_RACE_POINT(41);
    unsigned long long diff = 0;
    RowData now[kFastCols], ahead[kFastCols];
    int nxt[kFastCols][3][3];                       // chroma row ry + 1 (last sweep's values)
#pragma unroll
    for (int s = 0; s < kFastCols; ++s) {
      const int c = tid + s * kSweepThreads;
      if (c < uv_w) {
        load_uv(0, c, now[s].uv);
        load_rest(0, c, now[s]);
        load_uv(uv_h > 1 ? 1 : 0, c, nxt[s]);
#pragma unroll
        for (int k = 0; k < 3; ++k) above[0][k][c] = static_cast<int16_t>(now[s].uv[k][1]);   // row pair 0: "above" is the row itself
      }
    }
    SHARP_RACE_POINT(42);
    lds_barrier();
    for (int ry = 0; ry < uv_h; ++ry) {
      const int pp = ry & 1;
      SHARP_RACE_POINT(43);
      // request what the NEXT row pair needs
#pragma unroll
      for (int s = 0; s < kFastCols; ++s) {
        const int c = tid + s * kSweepThreads;
        if (c < uv_w && ry + 1 < uv_h) {
          load_rest(ry + 1, c, ahead[s]);
          load_uv(ry + 2 < uv_h ? ry + 2 : ry + 1, c, ahead[s].uv);     // becomes `nxt` of the next step
        }
      }
#pragma unroll
      for (int s = 0; s < kFastCols; ++s) {
        const int c = tid + s * kSweepThreads;
        if (c < uv_w) {
          const int cl = c > 0 ? c - 1 : 0, cr = c < uv_w - 1 ? c + 1 : uv_w - 1;
          const int wy[2][2] = {{static_cast<int>(now[s].wy[0] & 0xffffu), static_cast<int>(now[s].wy[0] >> 16)},
                                {static_cast<int>(now[s].wy[1] & 0xffffu), static_cast<int>(now[s].wy[1] >> 16)}};
          const int ty[2][2] = {{static_cast<int>(now[s].ty[0] & 0xffffu), static_cast<int>(now[s].ty[0] >> 16)},
                                {static_cast<int>(now[s].ty[1] & 0xffffu), static_cast<int>(now[s].ty[1] >> 16)}};
          int px[2][2][3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int A = now[s].uv[k][1], Al = now[s].uv[k][0], Ar = now[s].uv[k][2];
            const int P = above[pp][k][c], Pl = above[pp][k][cl], Pr = above[pp][k][cr];
            const bool has_next = ry + 1 < uv_h;                   // last row pair: next == cur
            const int N = has_next ? nxt[s][k][1] : A, Nl = has_next ? nxt[s][k][0] : Al, Nr = has_next ? nxt[s][k][2] : Ar;
            int up0, up1, dn0, dn1;
            if (c == 0) { up0 = (A * 3 + P + 2) >> 2; dn0 = (A * 3 + N + 2) >> 2; }
            else { up0 = (A * 9 + Al * 3 + P * 3 + Pl + 8) >> 4; dn0 = (A * 9 + Al * 3 + N * 3 + Nl + 8) >> 4; }
            if (c == uv_w - 1) { up1 = (A * 3 + P + 2) >> 2; dn1 = (A * 3 + N + 2) >> 2; }
            else { up1 = (A * 9 + Ar * 3 + P * 3 + Pr + 8) >> 4; dn1 = (A * 9 + Ar * 3 + N * 3 + Nr + 8) >> 4; }
            px[0][0][k] = clip_y(wy[0][0] + up0); px[0][1][k] = clip_y(wy[0][1] + up1);
            px[1][0][k] = clip_y(wy[1][0] + dn0); px[1][1][k] = clip_y(wy[1][1] + dn1);
          }
          int wt[2][2], uv[3];
          eval_group(g2l, l2g, px, wt, uv);
          uint32_t newy[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            int ny[2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
              const int d = ty[r][cc] - wt[r][cc];
              ny[cc] = clip_y(wy[r][cc] + d);
              diff += static_cast<unsigned long long>(d < 0 ? -d : d);
            }
            newy[r] = static_cast<uint32_t>(ny[0]) | (static_cast<uint32_t>(ny[1]) << 16);
          }
          reinterpret_cast<uint32_t*>(best_y + static_cast<size_t>(2 * ry) * w)[c] = newy[0];
          reinterpret_cast<uint32_t*>(best_y + static_cast<size_t>(2 * ry + 1) * w)[c] = newy[1];
          // SharpUpdateRGB: the row becomes this sweep's; the neighbours of the next row pair
          // read it from the other LDS buffer
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int16_t nv = static_cast<int16_t>(now[s].uv[k][1] + (now[s].tuv[k] - uv[k]));
            best_uv[static_cast<size_t>(ry) * 3 * uv_w + k * uv_w + c] = nv;
            above[pp ^ 1][k][c] = nv;
          }
        }
      }
      SHARP_RACE_POINT(44);
      lds_barrier();
      SHARP_RACE_POINT(45);
      // rotate: cur <- next (old values), next <- the row requested above
#pragma unroll
      for (int s = 0; s < kFastCols; ++s) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
          for (int t = 0; t < 3; ++t) { now[s].uv[k][t] = nxt[s][k][t]; nxt[s][k][t] = ahead[s].uv[k][t]; }
          now[s].tuv[k] = ahead[s].tuv[k];
        }
        now[s].wy[0] = ahead[s].wy[0]; now[s].wy[1] = ahead[s].wy[1];
        now[s].ty[0] = ahead[s].ty[0]; now[s].ty[1] = ahead[s].ty[1];
      }
    }
    // exit test (:660-666): sum of |dW| over the picture; full barriers: the next sweep reads
    // what this one stored
    for (int d = 32; d > 0; d >>= 1) diff += __shfl_down(diff, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = diff;
    SHARP_RACE_POINT(46);
    __syncthreads();
    if (tid == 0) {
      unsigned long long sum = 0;
      for (int i = 0; i < kSweepThreads / 64; ++i) sum += red[i];
      stop = (iter > 0 && (sum < threshold || sum > prev_diff)) ? 1 : 0;
      red[0] = sum;
    }
    __syncthreads();
    prev_diff = red[0];
    const int sflag = stop;
    SHARP_RACE_POINT(47);
    __syncthreads();
    if (sflag) break;
  }
}

// ---- the sweeps as a PIPELINE of workgroups (pictures up to 2 x 1024 chroma columns wide).  A sweep is sequential
// down the picture, but sweep t + 1 needs of sweep t only the rows down to two below the row pair it is at: the four
// sweeps of a picture run as four workgroups, each a few row pairs behind the one before.  No sweep updates in
// place any more: sweep t reads plane t % 3 of the W / chroma buffers and writes plane (t + 1) % 3 (its own row
// above travels through LDS as before).  Plane (t + 1) % 3 is also what sweep t - 2 read -- rows that sweep passed
// four row pairs ago -- and what sweep t + 3 would write, which does not exist.  The reference stops after the first
// sweep t >= 1 whose sum of |dW| is below a threshold or above its predecessor's (:660-666): here every sweep starts
// speculatively, the first one that meets the condition names itself the final one and raises `cancel`, which the
// later ones poll with their row dependencies; what they wrote by then lies in planes that are not the final
// sweep's (it would take sweep s + 3 to overwrite the output of sweep s, and sweep 0 never stops).
// Hand-over between workgroups: a counter of finished row pairs per sweep, released (agent scope) behind a full
// barrier of the producer and acquired by every thread of the consumer behind its own barrier.
// Grid (8, 4, ceil(nframes / 8)): frame = z * 8 + x, sweep = y -- a producer always has a smaller linear index than
// its consumers (it is resident or done when they start spinning), and with the round-robin placement of
// workgroups on the eight XCDs the four sweeps of a picture share one L2.
// COLS: chroma columns per thread -- 1 for pictures up to 2048 pixels wide (half the registers: no spills), else 2
template <int COLS>
__global__ __launch_bounds__(kSweepThreads) void sharp_sweeps_piped(const SharpArgs a) {
  __shared__ uint32_t g2l[kMaxY + 1];
  __shared__ uint32_t l2g[kGammaTab + 2];
  __shared__ unsigned long long red[kSweepThreads / 64];
  __shared__ int go;
  __shared__ int16_t above[2][3][COLS * kSweepThreads];   // the updated row above, ping-pong
  const int frame = a.frame0 + blockIdx.z * 8 + blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
  if (frame >= a.nframes) return;
  for (int i = tid; i <= kMaxY; i += kSweepThreads) g2l[i] = a.tab->g2l[i];
  if (tid < kGammaTab + 2) l2g[tid] = a.tab->l2g[tid];
  const int w = a.w, h = a.h, uv_w = a.uv_w, uv_h = a.uv_h;
  const size_t ysz = static_cast<size_t>(w) * h, usz = static_cast<size_t>(uv_h) * 3 * uv_w;
  const int pin = t % 3, pout = (t + 1) % 3;
  const uint16_t* const in_y = a.best_y + (static_cast<size_t>(pin) * a.nframes + frame) * ysz;
  uint16_t* const out_y = a.best_y + (static_cast<size_t>(pout) * a.nframes + frame) * ysz;
  const int16_t* const in_uv = a.best_uv + (static_cast<size_t>(pin) * a.nframes + frame) * usz;
  int16_t* const out_uv = a.best_uv + (static_cast<size_t>(pout) * a.nframes + frame) * usz;
  const uint16_t* const target_y = a.target_y + static_cast<size_t>(frame) * ysz;
  const int16_t* const target_uv = a.target_uv + static_cast<size_t>(frame) * usz;
  uint32_t* const ctrl = a.ctrl + static_cast<size_t>(frame) * a.ctrl_words;
  const unsigned long long threshold = static_cast<unsigned long long>(3.0 * w * h);

  struct RowData {                                  // what one row pair needs from global memory, per column
    int uv[3][3];                                   // chroma row: [channel][left, centre, right]
    uint32_t wy[2], ty[2];                          // W and its target: two pixels per word, two rows
    int tuv[3];                                     // chroma target
  };
  auto load_uv = [&](int row, int c, int (&dst)[3][3]) {
    const int cl = c > 0 ? c - 1 : 0, cr = c < uv_w - 1 ? c + 1 : uv_w - 1;
    const int16_t* r = in_uv + static_cast<size_t>(row) * 3 * uv_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) { dst[k][0] = r[k * uv_w + cl]; dst[k][1] = r[k * uv_w + c]; dst[k][2] = r[k * uv_w + cr]; }
  };
  auto load_rest = [&](int ry, int c, RowData& d) {
    const uint32_t* by0 = reinterpret_cast<const uint32_t*>(in_y + static_cast<size_t>(2 * ry) * w);
    const uint32_t* by1 = reinterpret_cast<const uint32_t*>(in_y + static_cast<size_t>(2 * ry + 1) * w);
    const uint32_t* ty0 = reinterpret_cast<const uint32_t*>(target_y + static_cast<size_t>(2 * ry) * w);
    const uint32_t* ty1 = reinterpret_cast<const uint32_t*>(target_y + static_cast<size_t>(2 * ry + 1) * w);
    d.wy[0] = by0[c]; d.wy[1] = by1[c]; d.ty[0] = ty0[c]; d.ty[1] = ty1[c];
    const int16_t* tu = target_uv + static_cast<size_t>(ry) * 3 * uv_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) d.tuv[k] = tu[k * uv_w + c];
  };
  // Waits until the sweep before this one has finished `need` row pairs (all of them visible here), or a sweep
  // before it has been named the final one.  Returns false in that case: this sweep's work is not wanted.
  // A look at the other workgroup's counter is a round trip through the memory fabric (about as long as a row pair
  // takes to compute), so it is not taken per row pair: the sweep waits until its producer is kAhead row pairs
  // further than it needs and then runs that far on what it knows.
  constexpr int kAhead = 16;
  int known = t == 0 ? uv_h : 0;                    // row pairs of the producer known to be done (uniform)
  auto wait_for = [&](int need) -> bool {
    if (need > uv_h) need = uv_h;
    if (known >= need) return true;
    const int want = need + kAhead < uv_h ? need + kAhead : uv_h;
    if (tid == 0) {
      int seen = -1;
      for (;;) {
        if (__hip_atomic_load(&ctrl[16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        const int p = static_cast<int>(__hip_atomic_load(&ctrl[t - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
        if (p >= want) { seen = p; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      go = seen;
    }
    __syncthreads();
    const int g = go;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // every wave's loads behind this see the producer's rows
    __syncthreads();                                // (`go` is rewritten by the next call)
    if (g < 0) return false;
    known = g;
    return true;
  };
  SHARP_RACE_POINT(40);
  __syncthreads();
  if (tid == 0) ctrl[18 + 2 * t] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());   // (debug: SJPEG_HIP_SHARP_DEBUG)
  unsigned long long diff = 0;
  bool wanted = wait_for(2);                        // rows 0 and 1 of the input plane
  if (wanted) {
    RowData now[COLS], ahead[COLS];
    int nxt[COLS][3][3];                       // chroma row ry + 1 (the sweep before's values)
#pragma unroll
    for (int s = 0; s < COLS; ++s) {
      const int c = tid + s * kSweepThreads;
      if (c < uv_w) {
        load_uv(0, c, now[s].uv);
        load_rest(0, c, now[s]);
        load_uv(uv_h > 1 ? 1 : 0, c, nxt[s]);
#pragma unroll
        for (int k = 0; k < 3; ++k) above[0][k][c] = static_cast<int16_t>(now[s].uv[k][1]);   // row pair 0: "above" is the row itself
      }
    }
    SHARP_RACE_POINT(42);
    __syncthreads();
    for (int ry = 0; ry < uv_h; ++ry) {
      const int pp = ry & 1;
      SHARP_RACE_POINT(43);
      // what the NEXT row pair needs: rows ry + 1 and ry + 2 of the input plane
      if (ry + 1 < uv_h) {
        wanted = wait_for(ry + 3);
        if (!wanted) break;
#pragma unroll
        for (int s = 0; s < COLS; ++s) {
          const int c = tid + s * kSweepThreads;
          if (c < uv_w) {
            load_rest(ry + 1, c, ahead[s]);
            load_uv(ry + 2 < uv_h ? ry + 2 : ry + 1, c, ahead[s].uv);     // becomes `nxt` of the next step
          }
        }
      }
#pragma unroll
      for (int s = 0; s < COLS; ++s) {
        const int c = tid + s * kSweepThreads;
        if (c < uv_w) {
          const int cl = c > 0 ? c - 1 : 0, cr = c < uv_w - 1 ? c + 1 : uv_w - 1;
          const int wy[2][2] = {{static_cast<int>(now[s].wy[0] & 0xffffu), static_cast<int>(now[s].wy[0] >> 16)},
                                {static_cast<int>(now[s].wy[1] & 0xffffu), static_cast<int>(now[s].wy[1] >> 16)}};
          const int ty[2][2] = {{static_cast<int>(now[s].ty[0] & 0xffffu), static_cast<int>(now[s].ty[0] >> 16)},
                                {static_cast<int>(now[s].ty[1] & 0xffffu), static_cast<int>(now[s].ty[1] >> 16)}};
          int px[2][2][3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int A = now[s].uv[k][1], Al = now[s].uv[k][0], Ar = now[s].uv[k][2];
            const int P = above[pp][k][c], Pl = above[pp][k][cl], Pr = above[pp][k][cr];
            const bool has_next = ry + 1 < uv_h;                   // last row pair: next == cur
            const int N = has_next ? nxt[s][k][1] : A, Nl = has_next ? nxt[s][k][0] : Al, Nr = has_next ? nxt[s][k][2] : Ar;
            int up0, up1, dn0, dn1;
            if (c == 0) { up0 = (A * 3 + P + 2) >> 2; dn0 = (A * 3 + N + 2) >> 2; }
            else { up0 = (A * 9 + Al * 3 + P * 3 + Pl + 8) >> 4; dn0 = (A * 9 + Al * 3 + N * 3 + Nl + 8) >> 4; }
            if (c == uv_w - 1) { up1 = (A * 3 + P + 2) >> 2; dn1 = (A * 3 + N + 2) >> 2; }
            else { up1 = (A * 9 + Ar * 3 + P * 3 + Pr + 8) >> 4; dn1 = (A * 9 + Ar * 3 + N * 3 + Nr + 8) >> 4; }
            px[0][0][k] = clip_y(wy[0][0] + up0); px[0][1][k] = clip_y(wy[0][1] + up1);
            px[1][0][k] = clip_y(wy[1][0] + dn0); px[1][1][k] = clip_y(wy[1][1] + dn1);
          }
          int wt[2][2], uv[3];
          eval_group(g2l, l2g, px, wt, uv);
          uint32_t newy[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            int ny[2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
              const int d = ty[r][cc] - wt[r][cc];
              ny[cc] = clip_y(wy[r][cc] + d);
              diff += static_cast<unsigned long long>(d < 0 ? -d : d);
            }
            newy[r] = static_cast<uint32_t>(ny[0]) | (static_cast<uint32_t>(ny[1]) << 16);
          }
          reinterpret_cast<uint32_t*>(out_y + static_cast<size_t>(2 * ry) * w)[c] = newy[0];
          reinterpret_cast<uint32_t*>(out_y + static_cast<size_t>(2 * ry + 1) * w)[c] = newy[1];
          // SharpUpdateRGB: the row becomes this sweep's; the neighbours of the next row pair read it from the
          // other LDS buffer, the next sweep from this sweep's output plane
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int16_t nv = static_cast<int16_t>(now[s].uv[k][1] + (now[s].tuv[k] - uv[k]));
            out_uv[static_cast<size_t>(ry) * 3 * uv_w + k * uv_w + c] = nv;
            above[pp ^ 1][k][c] = nv;
          }
        }
      }
      SHARP_RACE_POINT(44);
      // Row pairs are handed to the next sweep eight at a time: a hand-over needs every thread's stores to be DONE (a
      // full barrier: it waits for the memory counter, i.e. a store's round trip, about as long as the row pair's
      // arithmetic), the seven steps between only order the LDS row above and leave stores and loads in flight.
      const bool hand_over = (ry & 7) == 7 || ry + 1 == uv_h;
      if (hand_over) {
        // EVERY wave waits for its own stores in front of the barrier (`s_waitcnt vmcnt(0)`: they have reached the
        // XCD's L2): a workgroup-scope barrier alone compiles to `s_waitcnt lgkmcnt(0); s_barrier` -- it does not wait
        // for the other waves' global stores, and thread 0's release below would cover only wave 0's (the
        // rows of the other fifteen waves could still be in flight when the counter was published).  Behind the
        // barrier ONE agent-scope release (thread 0's store of the counter: `buffer_wbl2 sc1` writes the L2's dirty
        // lines back, whoever wrote them, then `s_waitcnt vmcnt(0)`) publishes them all.  (Round 5: the first fix
        // had every one of the sixteen waves write the L2 back -- 1.9 -> 2.35 ms per 1080p picture; this form 2.0.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      SHARP_RACE_POINT(45);
      if (hand_over && tid == 0) {
        __hip_atomic_store(&ctrl[t], static_cast<uint32_t>(ry + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      // rotate: cur <- next (old values), next <- the row requested above
#pragma unroll
      for (int s = 0; s < COLS; ++s) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
          for (int q = 0; q < 3; ++q) { now[s].uv[k][q] = nxt[s][k][q]; nxt[s][k][q] = ahead[s].uv[k][q]; }
          now[s].tuv[k] = ahead[s].tuv[k];
        }
        now[s].wy[0] = ahead[s].wy[0]; now[s].wy[1] = ahead[s].wy[1];
        now[s].ty[0] = ahead[s].ty[0]; now[s].ty[1] = ahead[s].ty[1];
      }
    }
  }
  if (tid == 0) ctrl[19 + 2 * t] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  if (!wanted) return;                              // (uniform: an earlier sweep is the final one)
  // exit test (:660-666): sum of |dW| over the picture, against the sweep before
  for (int d = 32; d > 0; d >>= 1) diff += __shfl_down(diff, d, 64);
  if ((tid & 63) == 0) red[tid >> 6] = diff;
  SHARP_RACE_POINT(46);
  __syncthreads();
  if (tid == 0) {
    unsigned long long sum = 0;
    for (int i = 0; i < kSweepThreads / 64; ++i) sum += red[i];
    bool cancelled = false;
    unsigned long long prev = ~0ull;
    if (t > 0) {
      // the sweep before has finished (its last row was waited for) -- but its verdict comes behind its rows
      while (__hip_atomic_load(&ctrl[4 + t - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        if (__hip_atomic_load(&ctrl[16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        __builtin_amdgcn_s_sleep(4);
      }
      cancelled = __hip_atomic_load(&ctrl[16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      prev = static_cast<unsigned long long>(ctrl[8 + 2 * (t - 1)]) | (static_cast<unsigned long long>(ctrl[9 + 2 * (t - 1)]) << 32);
    }
    if (!cancelled) {
      const bool stop = t > 0 && (sum < threshold || sum > prev);
      ctrl[8 + 2 * t] = static_cast<uint32_t>(sum);
      ctrl[9 + 2 * t] = static_cast<uint32_t>(sum >> 32);
      if (stop || t == 3) {
        ctrl[17] = static_cast<uint32_t>(t);
        __hip_atomic_store(&ctrl[16], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(&ctrl[4 + t], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- the pipelined sweeps ACROSS THE WIDTH as well (round 6).  A row pair of a sweep was one workgroup's: sixteen waves
// on ONE CU, ~300 instructions and 26 gamma-table lookups a column -- a step's time was that CU's issue rate and grew
// with the width (1080p 2.9 us, 4K 6.1 us a row pair, x 540 / x 1080 row pairs).  The columns of a row pair do not
// depend on each other; what a column needs that is NEW in this sweep is the row above at columns c - 1 .. c + 1.  So a
// sweep is cut into STRIPS of kStripOwn chroma columns, a workgroup of kStripThreads each, which carries kStripHalo more
// columns on either side: it computes them as well (they are the neighbour strip's, computed there too -- the same
// integers from the same inputs), and each row pair one more column from either edge has no valid row above any
// more.  After kStripHalo row pairs what is left valid is exactly the strip's own columns: there, and only there, the strips
// of a sweep meet -- every strip has published its rows (the counter of the hand-over to the next sweep), waits for its two
// neighbours' counters and takes the halo's row above from their output.  Everything else a column reads is the sweep
// before's plane, whose strips s - 1 .. s + 1 are waited for as the one workgroup was.  The exit test (:660-666) is the
// last strip's of a sweep to arrive: the strips add their sums of |dW| up in the control block.
// ctrl[f][32 + 4 * strips]: arrivals[4], done[4], sum[4] (two words each), cancel, final sweep, stamps[8]; progress[4][strips] at 32.
// Grid (strips, 4, pictures of the chunk): a workgroup only ever waits for workgroups of a smaller linear index or
// for its neighbour strips, which are dispatched next to it; a launch holds no more workgroups than are resident at once.
constexpr int kStripThreads = 256, kStripHalo = 32, kStripOwn = kStripThreads - 2 * kStripHalo;
__global__ __launch_bounds__(kStripThreads) void sharp_sweeps_strips(const SharpArgs a) {
  __shared__ uint32_t g2l[kMaxY + 1];
  __shared__ uint32_t l2g[kGammaTab + 2];
  __shared__ unsigned long long red[kStripThreads / 64];
  __shared__ int go;
  __shared__ int16_t above[2][3][kStripThreads];    // the updated row above, ping-pong
  const int strip = blockIdx.x, nstrips = gridDim.x, t = blockIdx.y, tid = threadIdx.x;
  const int frame = a.frame0 + blockIdx.z;
  for (int i = tid; i <= kMaxY; i += kStripThreads) g2l[i] = a.tab->g2l[i];
  if (tid < kGammaTab + 2) l2g[tid] = a.tab->l2g[tid];
  const int w = a.w, h = a.h, uv_w = a.uv_w, uv_h = a.uv_h;
  const size_t ysz = static_cast<size_t>(w) * h, usz = static_cast<size_t>(uv_h) * 3 * uv_w;
  const int pin = t % 3, pout = (t + 1) % 3;
  const uint16_t* const in_y = a.best_y + (static_cast<size_t>(pin) * a.nframes + frame) * ysz;
  uint16_t* const out_y = a.best_y + (static_cast<size_t>(pout) * a.nframes + frame) * ysz;
  const int16_t* const in_uv = a.best_uv + (static_cast<size_t>(pin) * a.nframes + frame) * usz;
  int16_t* const out_uv = a.best_uv + (static_cast<size_t>(pout) * a.nframes + frame) * usz;
  const uint16_t* const target_y = a.target_y + static_cast<size_t>(frame) * ysz;
  const int16_t* const target_uv = a.target_uv + static_cast<size_t>(frame) * usz;
  uint32_t* const ctrl = a.ctrl + static_cast<size_t>(frame) * a.ctrl_words;
  uint32_t* const progress = ctrl + 32;             // [4][nstrips]
  const unsigned long long threshold = static_cast<unsigned long long>(3.0 * w * h);
  const int own0 = strip * kStripOwn, own1 = own0 + kStripOwn < uv_w ? own0 + kStripOwn : uv_w;
  const int c = own0 - kStripHalo + tid;            // this thread's chroma column
  const bool live = c >= 0 && c < uv_w;
  const bool owned = c >= own0 && c < own1;
  // the neighbours' places in the LDS row (a column at the picture's edge is its own neighbour, as in the reference;
  // one at the workgroup's edge has none -- it is the first to go invalid, whatever it reads)
  const int tl = (c > 0 && tid > 0) ? tid - 1 : tid, tr = (c < uv_w - 1 && tid < kStripThreads - 1) ? tid + 1 : tid;
  const int s_lo = strip > 0 ? strip - 1 : 0, s_hi = strip < nstrips - 1 ? strip + 1 : strip;

  struct RowData { int uv[3][3]; uint32_t wy[2], ty[2]; int tuv[3]; };
  auto load_uv = [&](int row, int (&dst)[3][3]) {
    const int cl = c > 0 ? c - 1 : 0, cr = c < uv_w - 1 ? c + 1 : uv_w - 1;
    const int16_t* r = in_uv + static_cast<size_t>(row) * 3 * uv_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) { dst[k][0] = r[k * uv_w + cl]; dst[k][1] = r[k * uv_w + c]; dst[k][2] = r[k * uv_w + cr]; }
  };
  auto load_rest = [&](int ry, RowData& d) {
    d.wy[0] = reinterpret_cast<const uint32_t*>(in_y + static_cast<size_t>(2 * ry) * w)[c];
    d.wy[1] = reinterpret_cast<const uint32_t*>(in_y + static_cast<size_t>(2 * ry + 1) * w)[c];
    d.ty[0] = reinterpret_cast<const uint32_t*>(target_y + static_cast<size_t>(2 * ry) * w)[c];
    d.ty[1] = reinterpret_cast<const uint32_t*>(target_y + static_cast<size_t>(2 * ry + 1) * w)[c];
    const int16_t* tu = target_uv + static_cast<size_t>(ry) * 3 * uv_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) d.tuv[k] = tu[k * uv_w + c];
  };
  // Waits until strips s_lo .. s_hi of sweep `tt` have all finished `want` row pairs (visible here), or a sweep has been
  // named the final one (returns -1).  Returns the least of the three counters.
  auto wait_strips = [&](int tt, int want, bool self_too) -> int {
    if (tid == 0) {
      int seen = 0x7fffffff;
      for (int s = s_lo; s <= s_hi && seen >= 0; ++s) {
        if (!self_too && s == strip) continue;
        for (;;) {
          // (relaxed looks: an acquire load is a load AND a cache invalidate, per look and waiting workgroup; the one
          // acquire that matters is the fence behind the barrier below)
          if (__hip_atomic_load(&ctrl[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { seen = -1; break; }
          const int p = static_cast<int>(__hip_atomic_load(&progress[tt * nstrips + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (p >= want) { seen = p < seen ? p : seen; break; }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      go = seen;
    }
    __syncthreads();
    const int g = go;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // every wave's loads behind this see the producers' rows
    __syncthreads();                                // (`go` is rewritten by the next call)
    return g;
  };
  constexpr int kAhead = 16;
  int known = t == 0 ? uv_h : 0;                    // row pairs of the sweep before known to be done in all three strips
  auto wait_for = [&](int need) -> bool {
    if (need > uv_h) need = uv_h;
    if (known >= need) return true;
    const int g = wait_strips(t - 1, need + kAhead < uv_h ? need + kAhead : uv_h, true);
    if (g < 0) return false;
    known = g;
    return true;
  };
  SHARP_RACE_POINT(40);
  __syncthreads();
  if (tid == 0 && strip == 0) ctrl[18 + 2 * t] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  unsigned long long diff = 0;
  bool wanted = wait_for(3);
  if (wanted) {
    // (a step is shorter than a trip to memory now: what a row pair needs is asked for TWO steps ahead)
    RowData now, ahead, ahead2;
    int nxt[3][3];
    if (live) {
      load_uv(0, now.uv);
      load_rest(0, now);
      load_uv(uv_h > 1 ? 1 : 0, nxt);
      if (uv_h > 1) { load_rest(1, ahead); load_uv(uv_h > 2 ? 2 : 1, ahead.uv); }
#pragma unroll
      for (int k = 0; k < 3; ++k) above[0][k][tid] = static_cast<int16_t>(now.uv[k][1]);   // row pair 0: "above" is the row itself
    }
    SHARP_RACE_POINT(42);
    __syncthreads();
    for (int ry = 0; ry < uv_h; ++ry) {
      const int pp = ry & 1;
      SHARP_RACE_POINT(43);
      if (ry > 0 && (ry % kStripHalo) == 0 && nstrips > 1) {
        // the strips of this sweep meet: the halo's row above comes from the neighbours' output (row ry - 1)
        SHARP_RACE_POINT(47);
        if (wait_strips(t, ry, false) < 0) { wanted = false; break; }
        if (live && !owned) {
          const int16_t* const r = out_uv + static_cast<size_t>(ry - 1) * 3 * uv_w;
#pragma unroll
          for (int k = 0; k < 3; ++k) above[pp][k][tid] = r[k * uv_w + c];       // (behind wait_strips' acquire)
        }
        __syncthreads();
      }
      if (ry + 2 < uv_h) {
        wanted = wait_for(ry + 4);
        if (!wanted) break;
        if (live) {
          load_rest(ry + 2, ahead2);
          load_uv(ry + 3 < uv_h ? ry + 3 : ry + 2, ahead2.uv);     // becomes `nxt` two steps on
        }
      }
      if (live) {
        const int wy[2][2] = {{static_cast<int>(now.wy[0] & 0xffffu), static_cast<int>(now.wy[0] >> 16)},
                              {static_cast<int>(now.wy[1] & 0xffffu), static_cast<int>(now.wy[1] >> 16)}};
        const int ty[2][2] = {{static_cast<int>(now.ty[0] & 0xffffu), static_cast<int>(now.ty[0] >> 16)},
                              {static_cast<int>(now.ty[1] & 0xffffu), static_cast<int>(now.ty[1] >> 16)}};
        int px[2][2][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int A = now.uv[k][1], Al = now.uv[k][0], Ar = now.uv[k][2];
          const int P = above[pp][k][tid], Pl = above[pp][k][tl], Pr = above[pp][k][tr];
          const bool has_next = ry + 1 < uv_h;
          const int N = has_next ? nxt[k][1] : A, Nl = has_next ? nxt[k][0] : Al, Nr = has_next ? nxt[k][2] : Ar;
          int up0, up1, dn0, dn1;
          if (c == 0) { up0 = (A * 3 + P + 2) >> 2; dn0 = (A * 3 + N + 2) >> 2; }
          else { up0 = (A * 9 + Al * 3 + P * 3 + Pl + 8) >> 4; dn0 = (A * 9 + Al * 3 + N * 3 + Nl + 8) >> 4; }
          if (c == uv_w - 1) { up1 = (A * 3 + P + 2) >> 2; dn1 = (A * 3 + N + 2) >> 2; }
          else { up1 = (A * 9 + Ar * 3 + P * 3 + Pr + 8) >> 4; dn1 = (A * 9 + Ar * 3 + N * 3 + Nr + 8) >> 4; }
          px[0][0][k] = clip_y(wy[0][0] + up0); px[0][1][k] = clip_y(wy[0][1] + up1);
          px[1][0][k] = clip_y(wy[1][0] + dn0); px[1][1][k] = clip_y(wy[1][1] + dn1);
        }
        int wt[2][2], uv[3];
        eval_group(g2l, l2g, px, wt, uv);
        uint32_t newy[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          int ny[2];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const int d = ty[r][cc] - wt[r][cc];
            ny[cc] = clip_y(wy[r][cc] + d);
            if (owned) diff += static_cast<unsigned long long>(d < 0 ? -d : d);
          }
          newy[r] = static_cast<uint32_t>(ny[0]) | (static_cast<uint32_t>(ny[1]) << 16);
        }
        // (the rows go out as agent-scope stores -- write-through, `sc1` -- so that publishing them needs no release
        // fence: that is an L2 write-back of whatever is dirty, per workgroup and hand-over, and with twenty times the
        // workgroups of the one-per-sweep kernel it halved the throughput of a batch)
        if (owned) {
          __hip_atomic_store(&reinterpret_cast<uint32_t*>(out_y + static_cast<size_t>(2 * ry) * w)[c], newy[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&reinterpret_cast<uint32_t*>(out_y + static_cast<size_t>(2 * ry + 1) * w)[c], newy[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int16_t nv = static_cast<int16_t>(now.uv[k][1] + (now.tuv[k] - uv[k]));
          if (owned) __hip_atomic_store(&out_uv[static_cast<size_t>(ry) * 3 * uv_w + k * uv_w + c], nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          above[pp ^ 1][k][tid] = nv;
        }
      }
      SHARP_RACE_POINT(44);
      // hand-over (to the next sweep, and to the neighbour strips of this one): every wave's stores are acknowledged in
      // front of the barrier (written through, see above), the counter behind it is a store of the same kind
      const bool hand_over = (ry & 7) == 7 || ry + 1 == uv_h;
      if (hand_over) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      SHARP_RACE_POINT(45);
      if (hand_over && tid == 0) {
        __hip_atomic_store(&progress[t * nstrips + strip], static_cast<uint32_t>(ry + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { now.uv[k][q] = nxt[k][q]; nxt[k][q] = ahead.uv[k][q]; ahead.uv[k][q] = ahead2.uv[k][q]; }
        now.tuv[k] = ahead.tuv[k]; ahead.tuv[k] = ahead2.tuv[k];
      }
      now.wy[0] = ahead.wy[0]; now.wy[1] = ahead.wy[1];
      now.ty[0] = ahead.ty[0]; now.ty[1] = ahead.ty[1];
      ahead.wy[0] = ahead2.wy[0]; ahead.wy[1] = ahead2.wy[1];
      ahead.ty[0] = ahead2.ty[0]; ahead.ty[1] = ahead2.ty[1];
    }
  }
  if (tid == 0 && strip == 0) ctrl[19 + 2 * t] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  if (!wanted) return;                              // (uniform: an earlier sweep is the final one)
  // exit test (:660-666): the sweep's sum of |dW| over the picture = the strips' sums; the last strip to arrive takes it
  for (int d = 32; d > 0; d >>= 1) diff += __shfl_down(diff, d, 64);
  if ((tid & 63) == 0) red[tid >> 6] = diff;
  SHARP_RACE_POINT(46);
  __syncthreads();
  if (tid == 0) {
    unsigned long long mine = 0;
    for (int i = 0; i < kStripThreads / 64; ++i) mine += red[i];
    unsigned long long* const sum_t = reinterpret_cast<unsigned long long*>(ctrl + 8 + 2 * t);
    __hip_atomic_fetch_add(sum_t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t arrived = __hip_atomic_fetch_add(&ctrl[t], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived + 1u == static_cast<uint32_t>(nstrips)) {
      const unsigned long long sum = __hip_atomic_load(sum_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool cancelled = false;
      unsigned long long prev = ~0ull;
      if (t > 0) {
        while (__hip_atomic_load(&ctrl[4 + t - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          if (__hip_atomic_load(&ctrl[16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          __builtin_amdgcn_s_sleep(4);
        }
        cancelled = __hip_atomic_load(&ctrl[16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        prev = __hip_atomic_load(reinterpret_cast<unsigned long long*>(ctrl + 8 + 2 * (t - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!cancelled) {
        const bool stop = t > 0 && (sum < threshold || sum > prev);
        if (stop || t == 3) {
          ctrl[17] = static_cast<uint32_t>(t);
          __hip_atomic_store(&ctrl[16], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&ctrl[4 + t], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ---- back to 8-bit planes (:543-575; this file's own -11058 / -5328 constants)
__global__ __launch_bounds__(256) void sharp_export(const SharpArgs a) {
  const int frame = blockIdx.z;
  const int c = blockIdx.x * 256 + threadIdx.x, ry = blockIdx.y;
  if (c >= a.uv_w) return;
  // (pipelined sweeps: the plane the last sweep that counts wrote)
  const int plane = a.nplanes == 3 ? static_cast<int>((a.ctrl[static_cast<size_t>(frame) * a.ctrl_words + 17] + 1u) % 3u) : 0;
  const size_t pf = static_cast<size_t>(plane) * a.nframes + frame;
  const size_t uo = (pf * a.uv_h + ry) * 3 * a.uv_w;
  const int r = a.best_uv[uo + c], g = a.best_uv[uo + a.uv_w + c], b = a.best_uv[uo + 2 * a.uv_w + c];
  const int rnd = 1 << 18 >> 1;
  const int cw = (a.W + 1) >> 1;
  if (c < cw && ry < ((a.H + 1) >> 1)) {
    uint8_t* up = a.u + frame * a.uv_frame_stride + static_cast<size_t>(ry) * cw;
    uint8_t* vp = a.v + frame * a.uv_frame_stride + static_cast<size_t>(ry) * cw;
    up[c] = static_cast<uint8_t>(clip8(128 + ((-11058 * r - 21709 * g + 32768 * b + rnd) >> 18)));
    vp[c] = static_cast<uint8_t>(clip8(128 + ((32768 * r - 27439 * g - 5328 * b + rnd) >> 18)));
  }
  const size_t yo = pf * a.w * a.h;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int x = 2 * c + cc, y = 2 * ry + rr;
      if (x < a.W && y < a.H) {
        const int Wv = a.best_y[yo + static_cast<size_t>(y) * a.w + x];
        a.y[frame * a.y_frame_stride + static_cast<size_t>(y) * a.W + x] =
            static_cast<uint8_t>(clip8((19595 * (r + Wv) + 38469 * (g + Wv) + 7471 * (b + Wv) + rnd) >> 18));
      }
    }
  }
}

// ---- pictures too small for the iterative conversion (:57-100,674-690): plain averaging
__global__ __launch_bounds__(64) void sharp_small(const SharpArgs a) {
  const int frame = blockIdx.x;
  const uint8_t* base = a.rgb + frame * a.frame_stride;
  const int cw = (a.W + 1) >> 1, ch = (a.H + 1) >> 1;
  for (int i = threadIdx.x; i < a.W * a.H; i += 64) {
    const int x = i % a.W, y = i / a.W;
    const uint8_t* p = base + y * a.row_stride + static_cast<long long>(x) * a.pix_step;
    const int v = 19595 * p[a.r_off] + 38469 * p[a.g_off] + 7471 * p[a.b_off];
    a.y[frame * a.y_frame_stride + i] = static_cast<uint8_t>((v + (1 << 16 >> 1)) >> 16);
  }
  for (int i = threadIdx.x; i < cw * ch; i += 64) {
    const int cx = i % cw, cy = i / cw;
    const int y0 = 2 * cy, y1 = min(2 * cy + 1, a.H - 1);
    const int x0 = 2 * cx, x1 = 2 * cx + 1;
    const uint8_t* p00 = base + y0 * a.row_stride + static_cast<long long>(x0) * a.pix_step;
    const uint8_t* p10 = base + y1 * a.row_stride + static_cast<long long>(x0) * a.pix_step;
    int r, g, b;
    if (x1 < a.W) {
      const uint8_t* p01 = p00 + a.pix_step;
      const uint8_t* p11 = p10 + a.pix_step;
      r = p00[a.r_off] + p01[a.r_off] + p10[a.r_off] + p11[a.r_off];
      g = p00[a.g_off] + p01[a.g_off] + p10[a.g_off] + p11[a.g_off];
      b = p00[a.b_off] + p01[a.b_off] + p10[a.b_off] + p11[a.b_off];
    } else {
      r = 2 * (p00[a.r_off] + p10[a.r_off]); g = 2 * (p00[a.g_off] + p10[a.g_off]); b = 2 * (p00[a.b_off] + p10[a.b_off]);
    }
    const int rnd = 1 << 18 >> 1;
    a.u[frame * a.uv_frame_stride + i] = static_cast<uint8_t>(clip8(128 + ((-11058 * r - 21709 * g + 32768 * b + rnd) >> 18)));
    a.v[frame * a.uv_frame_stride + i] = static_cast<uint8_t>(clip8(128 + ((32768 * r - 27439 * g - 5328 * b + rnd) >> 18)));
  }
}

GammaTables g_tables;
std::once_flag g_tables_once;

void build_tables() {                               // reference InitGammaTablesF, :114-152
  const double norm = 1. / kMaxY, scale = 1. / kGammaTab;
  const double a = 0.099, thresh = 0.018, gamma = 1. / 0.45;
  const double final_scale = 1 << kG2LBits;
  for (int v = 0; v <= kMaxY; ++v) {
    const double g = norm * v;
    double value;
    if (g <= thresh * 4.5) {
      value = g / 4.5;
    } else {
      const double a_rec = 1. / (1. + a);
      value = pow(a_rec * (g + a), gamma);
    }
    g_tables.g2l[v] = static_cast<uint32_t>(value * final_scale + .5);
  }
  for (int v = 0; v <= kGammaTab; ++v) {
    const double g = scale * v;
    double value;
    if (g <= thresh) value = 4.5 * g;
    else value = (1. + a) * pow(g, 1. / gamma) - a;
    g_tables.l2g[v] = static_cast<uint32_t>(kMaxY * value) + (1 << kG2LBits >> 1);
  }
  g_tables.l2g[kGammaTab + 1] = g_tables.l2g[kGammaTab];
}

size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

}  // namespace

extern "C" {

size_t sjpeg_hip_sharp_workspace(int width, int height, int nframes) {
  if (width <= 0 || height <= 0 || width > 65535 || height > 65535 || nframes <= 0) return 0;
  const size_t w = (static_cast<size_t>(width) + 1) & ~size_t(1), h = (static_cast<size_t>(height) + 1) & ~size_t(1);
  // (three planes of W and chroma for the pipelined sweeps + the two target arrays, the side row of the in-place
  // kernel, the sweeps' control words)
  const size_t strips = (w / 2 + kStripOwn - 1) / kStripOwn;
  const size_t per = 4 * align256(w * h * 2) + 4 * align256(3 * (w / 2) * (h / 2) * 2) + align256(3 * (w / 2) * 2) + align256((32 + 4 * strips) * 4);
  return align256(sizeof(GammaTables)) + per * static_cast<size_t>(nframes);
}

int sjpeg_hip_sharp_yuv(const sjpeg_hip_source* src, int width, int height, int nframes,
                        uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, int64_t y_frame_stride,
                        int64_t uv_frame_stride, void* d_workspace, size_t workspace_size, void* stream) {
  if (src == nullptr || src->plane[0] == nullptr || d_y == nullptr || d_u == nullptr || d_v == nullptr ||
      d_workspace == nullptr) {
    return SJPEG_HIP_EINVAL;
  }
  if (workspace_size < sjpeg_hip_sharp_workspace(width, height, nframes) || nframes > 65535) return SJPEG_HIP_EINVAL;
  SharpArgs a;
  memset(&a, 0, sizeof(a));
  switch (src->format) {
    case SJPEG_HIP_SRC_RGB: a.pix_step = 3; a.r_off = 0; a.g_off = 1; a.b_off = 2; break;
    case SJPEG_HIP_SRC_BGRA: a.pix_step = 4; a.r_off = 2; a.g_off = 1; a.b_off = 0; break;
    case SJPEG_HIP_SRC_RGBA: a.pix_step = 4; a.r_off = 0; a.g_off = 1; a.b_off = 2; break;
    default: return SJPEG_HIP_EINVAL;                // the sharp conversion starts from RGB
  }
  const int64_t st_abs = src->row_stride[0] < 0 ? -src->row_stride[0] : src->row_stride[0];
  if (st_abs < static_cast<int64_t>(a.pix_step) * width) return SJPEG_HIP_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  a.rgb = static_cast<const uint8_t*>(src->plane[0]);
  a.row_stride = src->row_stride[0]; a.frame_stride = src->frame_stride[0];
  a.W = width; a.H = height;
  a.w = (width + 1) & ~1; a.h = (height + 1) & ~1; a.uv_w = a.w >> 1; a.uv_h = a.h >> 1;
  a.y = d_y; a.u = d_u; a.v = d_v;
  a.y_frame_stride = y_frame_stride; a.uv_frame_stride = uv_frame_stride;
#ifdef SJPEG_HIP_PRIO_STRESS
  if (const char* ab = getenv("SJPEG_HIP_ABLATE")) a.stress = atoi(ab);   // (read per call: the sweep changes it)
#endif
  if (width <= 4 || height <= 4) {
    hipLaunchKernelGGL(sharp_small, dim3(nframes), dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : SJPEG_HIP_ERUNTIME;
  }
  std::call_once(g_tables_once, build_tables);
  uint8_t* p = static_cast<uint8_t*>(d_workspace);
  a.tab = reinterpret_cast<const GammaTables*>(p);
  if (hipMemcpyAsync(p, &g_tables, sizeof(GammaTables), hipMemcpyHostToDevice, st) != hipSuccess) return SJPEG_HIP_ERUNTIME;
  p += align256(sizeof(GammaTables));
  const size_t ysz = align256(static_cast<size_t>(a.w) * a.h * 2) , usz = align256(static_cast<size_t>(3) * a.uv_w * a.uv_h * 2);
  // per-frame arrays are addressed as [frame][...] with the un-padded sizes: keep them contiguous
  // (plane p of frame f: base + (p * nframes + f) * elements of a frame; the in-place kernels use plane 0 only)
  a.nframes = nframes;
  a.best_y = reinterpret_cast<uint16_t*>(p); p += 3 * ysz * nframes;
  a.target_y = reinterpret_cast<uint16_t*>(p); p += ysz * nframes;
  a.best_uv = reinterpret_cast<int16_t*>(p); p += 3 * usz * nframes;
  a.target_uv = reinterpret_cast<int16_t*>(p); p += usz * nframes;
  a.row_uv = reinterpret_cast<int16_t*>(p); p += align256(static_cast<size_t>(3) * a.uv_w * 2) * nframes;
  a.ctrl = reinterpret_cast<uint32_t*>(p);
  const dim3 grid((a.uv_w + 255) / 256, a.uv_h, nframes);
  // SJPEG_HIP_SHARP_INPLACE=1: round 3's one-workgroup-per-picture sweeps (A/B, and the fallback for very wide pictures)
  static const bool inplace = getenv("SJPEG_HIP_SHARP_INPLACE") != nullptr && atoi(getenv("SJPEG_HIP_SHARP_INPLACE")) != 0;
  // SJPEG_HIP_SHARP_STRIPS=0: round 4's one workgroup per picture and sweep (A/B)
  static const bool no_strips = getenv("SJPEG_HIP_SHARP_STRIPS") != nullptr && atoi(getenv("SJPEG_HIP_SHARP_STRIPS")) == 0;
  const int nstrips = (a.uv_w + kStripOwn - 1) / kStripOwn;
  const bool strips = !inplace && !no_strips;
  const bool piped = !inplace && (strips || a.uv_w <= kFastCols * kSweepThreads);
  a.nplanes = piped ? 3 : 1;
  a.ctrl_words = strips ? 32 + 4 * nstrips : 32;
  if (piped && hipMemsetAsync(a.ctrl, 0, static_cast<size_t>(nframes) * a.ctrl_words * sizeof(uint32_t), st) != hipSuccess) return SJPEG_HIP_ERUNTIME;
  hipLaunchKernelGGL(sharp_import, grid, dim3(256), 0, st, a);
  a.frame0 = 0;
  if (strips) {
    // (as below: a launch never has more workgroups than are resident at once -- 4 * strips small workgroups a picture,
    // as many a CU as the occupancy query says)
    static const int kSlots = [] {
      if (const char* e = getenv("SJPEG_HIP_SHARP_SLOTS")) { const int v = atoi(e); if (v > 0) return v; }   // (A/B)
      int dev = 0, cus = 0, per_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sharp_sweeps_strips, kStripThreads, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 16;
      }
      const int all = cus * per_cu;               // what the device holds of this kernel at once; three quarters of it a launch
      return all < 32 ? 16 : all / 4 * 3;
    }();
    const int chunk = kSlots / (4 * nstrips) < 1 ? 1 : kSlots / (4 * nstrips);
    for (int f0 = 0; f0 < nframes; f0 += chunk) {
      a.frame0 = f0;
      hipLaunchKernelGGL(sharp_sweeps_strips, dim3(nstrips, 4, nframes - f0 < chunk ? nframes - f0 : chunk), dim3(kStripThreads), 0, st, a);
    }
    a.frame0 = 0;
  } else if (piped) {
    // A sweep spins on the counter of the sweep before it, another workgroup of the same launch: that only ends if
    // the producer is resident.  A launch therefore never has more workgroups than the device holds at once --
    // chunks of pictures, four workgroups of 1024 threads each, one per CU on at most half the chip (the whole
    // batch in one grid relied on dispatch order for batches beyond that) --; the chunks follow each other on the stream.
    // (from the device's CU count, not a constant: a grid of (8, 4, pictures / 8) has four workgroups of 1024 threads
    // per picture, one workgroup fills a CU's wave slots for this kernel -- a quarter of the CUs' worth of pictures per
    // launch leaves every producer resident on any device, half the chip on MI355X as before: 256 / 8 = 32)
    static const int kChunk = [] {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
      const int c = (cus / 8) & ~7;               // whole groups of eight pictures (the grid's z dimension)
      return c < 8 ? 8 : c > 32 ? 32 : c;
    }();
    for (int f0 = 0; f0 < nframes; f0 += kChunk) {
      const int nf = nframes - f0 < kChunk ? nframes - f0 : kChunk;
      a.frame0 = f0;
      if (a.uv_w <= kSweepThreads) hipLaunchKernelGGL(sharp_sweeps_piped<1>, dim3(8, 4, (nf + 7) / 8), dim3(kSweepThreads), 0, st, a);
      else hipLaunchKernelGGL(sharp_sweeps_piped<2>, dim3(8, 4, (nf + 7) / 8), dim3(kSweepThreads), 0, st, a);
    }
    a.frame0 = 0;
  }
  else if (a.uv_w <= kFastCols * kSweepThreads) hipLaunchKernelGGL(sharp_sweeps_fast, dim3(nframes), dim3(kSweepThreads), 0, st, a);
  else hipLaunchKernelGGL(sharp_sweeps, dim3(nframes), dim3(kSweepThreads), 0, st, a);
  hipLaunchKernelGGL(sharp_export, grid, dim3(256), 0, st, a);
  if (piped && getenv("SJPEG_HIP_SHARP_DEBUG") != nullptr) {        // when the four sweeps of frame 0 ran (10 ns ticks), who was final
    uint32_t c[36];
    if (hipMemcpyAsync(c, a.ctrl, sizeof(c), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
      if (strips) fprintf(stderr, "sharp sweeps, %d strips: final %u, strips arrived %u %u %u %u;", nstrips, c[17], c[0], c[1], c[2], c[3]);
      else fprintf(stderr, "sharp sweeps: final %u, rows done %u %u %u %u of %d;", c[17], c[0], c[1], c[2], c[3], a.uv_h);
      for (int t = 0; t < 4; ++t) fprintf(stderr, " [%d] %.1f..%.1f us", t, (c[18 + 2 * t] - c[18]) / 100.0, (c[19 + 2 * t] - c[18]) / 100.0);
      fprintf(stderr, "\n");
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : SJPEG_HIP_ERUNTIME;
}

}  // extern "C"
