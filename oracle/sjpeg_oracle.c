/* TEST INFRASTRUCTURE ONLY -- see sjpeg_oracle.h.  Plain C99, scalar, single-threaded.
 * A restatement of the reference's *plain-C* functions (the normative spec, SURVEY.md
 * §0 fact 5), written from their described behaviour; each function cites file:line
 * under /root/reference.  Never linked into, loaded by, or shipped with the product. */
#include "sjpeg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- constant tables */

/* zig-zag scan position -> natural index (JPEG Figure A.6; src/quantize.cc:32-41) */
const uint8_t orc_zigzag[64] = {
  0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* JPEG Annex K.1 luminance / chrominance matrices (src/quantize.cc:57-75) */
static const uint8_t kAnnexK1[2][64] = {
  {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
  {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
   24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};

/* JPEG Annex K.3 Huffman specifications (src/entropy.cc:31-82): BITS then HUFFVAL */
static const uint8_t kDcBits[2][16] = {
  {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
  {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcBits[2][16] = {
  {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125},
  {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}};
static const uint8_t kAcVals[2][162] = {
  {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51,
   0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
   0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
   0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
   0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57,
   0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
   0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92,
   0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
   0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
   0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
   0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
   0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa},
  {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07,
   0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
   0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
   0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38,
   0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
   0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
   0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
   0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
   0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
   0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
   0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
   0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa}};

/* ---------------------------------------------------------------- geometry */

typedef struct {
  int nb_comps, mcu_blocks, block_w, block_h;
  int nb_blocks[3], quant_idx[3], sampling[3];
} orc_layout;

/* src/encoders.cc:32-88 */
static int layout_for(int yuv_mode, orc_layout* L) {
  memset(L, 0, sizeof(*L));
  if (yuv_mode == ORC_YUV_420) {
    L->nb_comps = 3; L->mcu_blocks = 6; L->block_w = L->block_h = 16;
    L->nb_blocks[0] = 4; L->nb_blocks[1] = L->nb_blocks[2] = 1;
    L->sampling[0] = 0x22; L->sampling[1] = L->sampling[2] = 0x11;
    L->quant_idx[1] = L->quant_idx[2] = 1;
  } else if (yuv_mode == ORC_YUV_444) {
    L->nb_comps = 3; L->mcu_blocks = 3; L->block_w = L->block_h = 8;
    L->nb_blocks[0] = L->nb_blocks[1] = L->nb_blocks[2] = 1;
    L->sampling[0] = L->sampling[1] = L->sampling[2] = 0x11;
    L->quant_idx[1] = L->quant_idx[2] = 1;
  } else if (yuv_mode == ORC_YUV_400) {
    L->nb_comps = 1; L->mcu_blocks = 1; L->block_w = L->block_h = 8;
    L->nb_blocks[0] = 1; L->sampling[0] = 0x11;
  } else {
    return 0;
  }
  return 1;
}

/* ---------------------------------------------------------------- quantizer set-up */

/* src/quantize.cc:77-82 -- libjpeg-6b quality mapping, float, floorf */
float orc_qfactor(float q) {
  float s;
  if (q <= 0) s = 5000;
  else if (q < 50) s = 5000 / q;
  else if (q < 100) s = 2 * (100 - q);
  else s = 0;
  return floorf(s);
}

/* src/quantize.cc:88-96 -- float scale, +.5f, truncate, clamp to [1,255] */
void orc_set_quant_matrix(const uint8_t in[64], float q_factor, uint8_t out[64]) {
  const float scale = q_factor / 100.f;
  for (int i = 0; i < 64; ++i) {
    const int v = (int)(in[i] * scale + .5f);
    out[i] = (uint8_t)(v < 1 ? 1 : v > 255 ? 255 : v);
  }
}

/* src/enc.cc:100-104 */
void orc_quality_matrices(float quality, uint8_t out[2][64]) {
  const float qf = orc_qfactor(quality);
  orc_set_quant_matrix(kAnnexK1[0], qf, out[0]);
  orc_set_quant_matrix(kAnnexK1[1], qf, out[1]);
}

/* src/quantize.cc:115-148 -- reciprocal, bias and dead-zone threshold per coefficient.
 * FP_BITS = 16, AC_BITS = 4 (src/sjpegi.h:165). */
void orc_finalize_quant(orc_quantizer* q, int q_bias) {
  for (int i = 0; i < 64; ++i) {
    if (q->quant[i] < q->min_quant[i]) q->quant[i] = q->min_quant[i];
  }
  for (int i = 0; i < 64; ++i) {
    const uint32_t v = q->quant[i];
    uint32_t iq, b8;
    if (v == 1) { iq = 0xffffu; b8 = 0x80; }       /* 1/1 does not fit 16 bits */
    else { iq = ((1u << 16) + v / 2) / v; b8 = (i == 0) ? 0x80u : (uint32_t)q_bias; }
    const uint16_t ibias = (uint16_t)((((b8 * v) << 4) + 128) >> 8);
    const uint16_t iquant = (uint16_t)iq;
    const uint16_t thresh = (uint16_t)(((1u << 20) + iquant - 1) / iquant - ibias);
    q->iquant[i] = iquant;
    q->bias[i] = ibias;
    q->qthresh[i] = thresh;
  }
}

/* ---------------------------------------------------------------- Huffman tables */

/* src/entropy.cc:98-112 -- canonical code assignment, packed (code << 16) | length */
int orc_build_huffman(const uint8_t bits[16], const uint8_t* syms, uint32_t* tab) {
  uint32_t code = 0;
  int total = 0;
  for (int len = 1; len <= 16; ++len) {
    for (int k = 0; k < bits[len - 1]; ++k) {
      tab[*syms++] = (code << 16) | (uint32_t)len;
      ++code;
      ++total;
    }
    code <<= 1;
  }
  return total;
}

void orc_default_codes(uint32_t dc[2][12], uint32_t ac[2][256]) {
  memset(dc, 0, 2 * 12 * sizeof(uint32_t));
  memset(ac, 0, 2 * 256 * sizeof(uint32_t));
  for (int c = 0; c < 2; ++c) {
    orc_build_huffman(kDcBits[c], kDcVals, dc[c]);
    orc_build_huffman(kAcBits[c], kAcVals[c], ac[c]);
  }
}

/* ---------------------------------------------------------------- colour conversion */

/* 16.16 fixed-point BT.601 full range; src/colors_rgb.cc:17-19,31-32,785-828.
 * Y is level-shifted by -128 inside the rounding constant. */
#define ORC_HALF (1 << 15)
static int16_t luma_of(int r, int g, int b) {
  return (int16_t)((19595 * r + 38469 * g + 7471 * b + ORC_HALF - (128 << 16)) >> 16);
}
static int16_t cb_of(int r, int g, int b, int shift, int rnd) {
  return (int16_t)((-11059 * r - 21709 * g + 32768 * b + rnd) >> shift);
}
static int16_t cr_of(int r, int g, int b, int shift, int rnd) {
  return (int16_t)((32768 * r - 27439 * g - 5329 * b + rnd) >> shift);
}

/* 8x8 pixels -> Y,U,V blocks (src/colors_rgb.cc:830-838) or Y only (:840-848) */
static void block8_from_rgb(const uint8_t* p, int step, int16_t* out, int with_chroma) {
  for (int y = 0; y < 8; ++y, p += step) {
    for (int x = 0; x < 8; ++x) {
      const int r = p[3 * x], g = p[3 * x + 1], b = p[3 * x + 2];
      out[y * 8 + x] = luma_of(r, g, b);
      if (with_chroma) {
        out[64 + y * 8 + x] = cb_of(r, g, b, 16, ORC_HALF);
        out[128 + y * 8 + x] = cr_of(r, g, b, 16, ORC_HALF);
      }
    }
  }
}

/* 16x16 pixels -> Y0 Y1 Y2 Y3 U V (src/colors_rgb.cc:850-879): chroma is ONE transform
 * of the 2x2 r/g/b sums with rounding HALF<<2 and shift 18. */
static void mcu420_from_rgb(const uint8_t* p, int step, int16_t* out) {
  for (int y = 0; y < 16; ++y) {
    for (int x = 0; x < 16; ++x) {
      const uint8_t* px = p + y * step + 3 * x;
      const int blk = (y >> 3) * 2 + (x >> 3);
      out[blk * 64 + (y & 7) * 8 + (x & 7)] = luma_of(px[0], px[1], px[2]);
    }
  }
  for (int cy = 0; cy < 8; ++cy) {
    for (int cx = 0; cx < 8; ++cx) {
      const uint8_t* a = p + (2 * cy) * step + 6 * cx;
      const uint8_t* b = a + step;
      const int R = a[0] + a[3] + b[0] + b[3];
      const int G = a[1] + a[4] + b[1] + b[4];
      const int B = a[2] + a[5] + b[2] + b[5];
      out[4 * 64 + cy * 8 + cx] = cb_of(R, G, B, 18, ORC_HALF << 2);
      out[5 * 64 + cy * 8 + cx] = cr_of(R, G, B, 18, ORC_HALF << 2);
    }
  }
}

static int block_avg(const int16_t* b) {      /* src/encoders.cc:95-99 */
  int s = 0;
  for (int i = 0; i < 64; ++i) s += b[i];
  return (s + 32) >> 6;
}
static void block_fill(int16_t* b, int v) {
  for (int i = 0; i < 64; ++i) b[i] = (int16_t)v;
}

/* 8x8 block of an 8-bit plane, level shifted; coordinates clamp to the plane (this is what
 * Convert8To16bClipped / Replicate8b do: src/colors_rgb.cc:1212-1260). step = bytes per sample. */
static void plane_block(const uint8_t* p, int stride, int step, int pw, int ph, int x0, int y0,
                        int16_t* out) {
  for (int y = 0; y < 8; ++y) {
    const int sy = (y0 + y < ph) ? y0 + y : ph - 1;
    for (int x = 0; x < 8; ++x) {
      const int sx = (x0 + x < pw) ? x0 + x : pw - 1;
      out[y * 8 + x] = (int16_t)(p[(ptrdiff_t)sy * stride + (ptrdiff_t)sx * step] - 128);
    }
  }
}

static void luma_fixup_420(int sub_w, int sub_h, int16_t* out) {
  /* src/encoders.cc:107-125: luma blocks lying wholly outside the picture become flat at
   * the average of a neighbouring real block. */
  int dc = block_avg(out);
  if (sub_w <= 8) block_fill(out + 64, dc);
  if (sub_h <= 8) {
    if (sub_w > 8) dc = block_avg(out + 64);
    block_fill(out + 128, dc);
    block_fill(out + 192, dc);
  } else if (sub_w <= 8) {
    block_fill(out + 192, block_avg(out + 128));
  }
}

void orc_get_samples_src(const orc_source* S, int yuv_mode, int W, int H, int mb_x, int mb_y,
                         int16_t* out) {
  orc_layout L;
  if (!layout_for(yuv_mode, &L)) return;
  const int bw = L.block_w, bh = L.block_h;
  /* src/enc.cc:280-281,292: an MCU is "clipped" iff it is the partial last column/row */
  const int clipped = (mb_y == H / bh) || (mb_x == W / bw);
  const int sub_w = W - mb_x * bw, sub_h = H - mb_y * bh;
  if (S->format <= ORC_SRC_RGBA) {
    /* packed colour: 3 or 4 bytes per pixel (src/encoders.cc:157-253, colors_rgb.cc:882-1025) */
    const int px = (S->format == ORC_SRC_RGB) ? 3 : 4;
    const int ro = (S->format == ORC_SRC_BGRA) ? 2 : 0, bo = (S->format == ORC_SRC_BGRA) ? 0 : 2;
    const int stride = S->stride[0];
    const uint8_t* src = S->plane[0] + (ptrdiff_t)(px * mb_x) * bw + (ptrdiff_t)mb_y * stride * bh;
    /* gather the MCU as tight RGB, clamping coordinates when clipped (Replicate8b) */
    uint8_t tmp[16 * 16 * 3];
    const int vw = (clipped && sub_w < bw) ? sub_w : bw, vh = (clipped && sub_h < bh) ? sub_h : bh;
    for (int y = 0; y < bh; ++y) {
      const int sy = y < vh ? y : vh - 1;
      for (int x = 0; x < bw; ++x) {
        const int sx = x < vw ? x : vw - 1;
        const uint8_t* q = src + (ptrdiff_t)sy * stride + px * sx;
        uint8_t* d = tmp + (y * bw + x) * 3;
        d[0] = q[ro]; d[1] = q[1]; d[2] = q[bo];
      }
    }
    if (yuv_mode == ORC_YUV_420) {
      mcu420_from_rgb(tmp, 3 * bw, out);
      if (clipped) luma_fixup_420(sub_w, sub_h, out);
    } else {
      block8_from_rgb(tmp, 3 * bw, out, yuv_mode == ORC_YUV_444);
    }
    return;
  }
  /* 8-bit planes: samples are used as they are, minus 128 (src/encoders.cc:256-490) */
  if (S->format == ORC_SRC_GRAY) {
    plane_block(S->plane[0], S->stride[0], 1, W, H, mb_x * 8, mb_y * 8, out);
  } else if (S->format == ORC_SRC_YUV444) {
    for (int c = 0; c < 3; ++c) plane_block(S->plane[c], S->stride[c], 1, W, H, mb_x * 8, mb_y * 8, out + 64 * c);
  } else {
    for (int k = 0; k < 4; ++k) {
      plane_block(S->plane[0], S->stride[0], 1, W, H, mb_x * 16 + 8 * (k & 1), mb_y * 16 + 8 * (k >> 1), out + 64 * k);
    }
    if (clipped) luma_fixup_420(sub_w, sub_h, out);
    const int cw = (W + 1) >> 1, ch = (H + 1) >> 1;
    if (S->format == ORC_SRC_YUV420) {
      plane_block(S->plane[1], S->stride[1], 1, cw, ch, mb_x * 8, mb_y * 8, out + 4 * 64);
      plane_block(S->plane[2], S->stride[2], 1, cw, ch, mb_x * 8, mb_y * 8, out + 5 * 64);
    } else {   /* NV12: U,V,U,V...  NV21: V,U,V,U... */
      const int uo = (S->format == ORC_SRC_NV12) ? 0 : 1;
      plane_block(S->plane[1] + uo, S->stride[1], 2, cw, ch, mb_x * 8, mb_y * 8, out + 4 * 64);
      plane_block(S->plane[1] + (1 - uo), S->stride[1], 2, cw, ch, mb_x * 8, mb_y * 8, out + 5 * 64);
    }
  }
}

void orc_get_samples(int yuv_mode, const uint8_t* rgb, int W, int H, int stride,
                     int mb_x, int mb_y, int16_t* out) {
  orc_source S;
  memset(&S, 0, sizeof(S));
  S.format = ORC_SRC_RGB; S.plane[0] = rgb; S.stride[0] = stride;
  orc_get_samples_src(&S, yuv_mode, W, H, mb_x, mb_y, out);
}

/* ---------------------------------------------------------------- forward DCT */

/* 16-bit fixed-point multiply used by the column pass: (a*b) >> 16 (src/fdct.cc:150) */
static int32_t mulhi(int32_t a, int32_t b) { return (a * b) >> 16; }

/* Column pass on one column (stride 8); src/fdct.cc:67-144 read with the plain-C
 * macro set of :148-157.  int32 temporaries, int16 stores, exact operation order. */
static void fdct_column(int16_t* c) {
  const int32_t x0 = c[0], x1 = c[8], x2 = c[16], x3 = c[24];
  const int32_t x4 = c[32], x5 = c[40], x6 = c[48], x7 = c[56];
  /* first butterflies: differences d, sums s */
  int32_t d07 = x0 - x7, s07 = x0 + x7;
  int32_t d25 = x2 - x5, s25 = x2 + x5;
  int32_t d34 = x3 - x4, s34 = x3 + x4;
  int32_t d16 = x1 - x6, s16 = x1 + x6;
  /* even part */
  int32_t e_d = s07 - s34, e_s = s07 + s34;     /* (m7, m4) */
  int32_t f_d = s16 - s25, f_s = s16 + s25;     /* (m6, m5) */
  int32_t a = e_s * 8, b = f_s * 8;
  c[0] = (int16_t)(a + b);
  c[32] = (int16_t)(a - b);
  e_d *= 8; f_d *= 8; d34 *= 8; d07 *= 8;
  c[16] = (int16_t)(mulhi(27146, f_d) + e_d);   /* kTan2 */
  c[48] = (int16_t)(mulhi(27146, e_d) - f_d);
  /* odd part */
  d25 *= 16; d16 *= 16;                          /* <<(3+1): k2Sqrt2 carries the extra 1/2 */
  int32_t od = mulhi(d16 - d25, 23170);          /* m1 */
  int32_t os = mulhi(d16 + d25, 23170);          /* m2 */
  int32_t p3 = d34 - od, p1 = d34 + od;          /* butterfly(m3, m1) */
  int32_t p0 = d07 - os, p2 = d07 + os;          /* butterfly(m0, m2) */
  int32_t t3 = mulhi(p3, -21746) + p3 + 1;       /* kTan3m1, CORRECT_LSB */
  int32_t t1 = mulhi(p1, 13036) + p2 + 1;        /* kTan1,   CORRECT_LSB */
  int32_t t4 = mulhi(-21746, p0) + p0;
  int32_t t5 = mulhi(13036, p2);
  c[8] = (int16_t)t1;
  c[24] = (int16_t)(p0 - t3);
  c[40] = (int16_t)(p3 + t4);
  c[56] = (int16_t)(t5 - p1);
}

/* Row pass; src/fdct.cc:174-209.  Products accumulate in 32 bits with wrap-around
 * (computed unsigned here to keep the C well defined), arithmetic >>16, no rounding. */
static int16_t descale(uint32_t v) { return (int16_t)((int32_t)v >> 16); }
static void fdct_row(int16_t* r, const int16_t* t) {
  const int32_t a0 = r[0] + r[7], b0 = r[0] - r[7];
  const int32_t a1 = r[1] + r[6], b1 = r[1] - r[6];
  const int32_t a2 = r[2] + r[5], b2 = r[2] - r[5];
  const int32_t a3 = r[3] + r[4], b3 = r[3] - r[4];
  const uint32_t C1 = t[0], C2 = t[1], C3 = t[2], C4 = t[3], C5 = t[4], C6 = t[5], C7 = t[6];
  const uint32_t c0 = a0 + a3, c1 = a0 - a3, c2 = a1 + a2, c3 = a1 - a2;
  const uint32_t B0 = b0, B1 = b1, B2 = b2, B3 = b3;
  r[0] = descale(C4 * (c0 + c2));
  r[4] = descale(C4 * (c0 - c2));
  r[2] = descale(C2 * c1 + C6 * c3);
  r[6] = descale(C6 * c1 - C2 * c3);
  r[1] = descale(C1 * B0 + C3 * B1 + C5 * B2 + C7 * B3);
  r[3] = descale(C3 * B0 - C7 * B1 - C1 * B2 - C5 * B3);
  r[5] = descale(C5 * B0 - C1 * B1 + C7 * B2 + C3 * B3);
  r[7] = descale(C7 * B0 - C5 * B1 + C3 * B2 - C1 * B3);
}

/* src/fdct.cc:28-35: cos(k*pi/16)/sqrt(2) in 15 bits, rows 1/7, 2/6, 3/5 pre-scaled */
static const int16_t kRowTab[4][7] = {
  {22725, 21407, 19266, 16384, 12873, 8867, 4520},
  {31521, 29692, 26722, 22725, 17855, 12299, 6270},
  {29692, 27969, 25172, 21407, 16819, 11585, 5906},
  {26722, 25172, 22654, 19266, 15137, 10426, 5315}};
static const uint8_t kRowSel[8] = {0, 1, 2, 3, 0, 3, 2, 1};     /* src/fdct.cc:599-606 */

void orc_fdct(int16_t* coeffs, int num_blocks) {
  for (int n = 0; n < num_blocks; ++n, coeffs += 64) {
    for (int x = 0; x < 8; ++x) fdct_column(coeffs + x);
    for (int y = 0; y < 8; ++y) fdct_row(coeffs + 8 * y, kRowTab[kRowSel[y]]);
  }
}

/* ---------------------------------------------------------------- quantization */

static int bit_length(uint32_t v) {   /* CalcLog2, src/sjpegi.h:186-197 */
  int n = 0;
  while (v) { ++n; v >>= 1; }
  return n;
}

/* ((a + bias) * iquant >> 16) >> 4, src/quantize.cc:119-121 */
static int quantize_abs(uint32_t a, uint32_t iq, uint32_t bias) {
  return (int)((((a + bias) * iq) >> 16) >> 4);
}

int orc_quantize_block(const int16_t in[64], const orc_quantizer* q, int16_t zz[64]) {
  for (int i = 1; i < 64; ++i) {
    const int j = orc_zigzag[i];
    const int v = in[j];
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    int lvl = 0;
    if (a >= q->qthresh[j]) lvl = quantize_abs(a, q->iquant[j], q->bias[j]);
    zz[i] = (int16_t)(v < 0 ? -lvl : lvl);
  }
  const int d = in[0];
  const int dc = d < 0 ? -quantize_abs((uint32_t)-d, q->iquant[0], q->bias[0])
                       : quantize_abs((uint32_t)d, q->iquant[0], q->bias[0]);
  zz[0] = (int16_t)dc;
  return dc;
}

/* Trellis quantization of one block: src/quantize.cc:325-457 (TrellisNode, SearchBestPrev,
 * Encoder::TrellisQuantizeBlock).  For every non-zero coefficient two candidate levels (the
 * rounded one, and the largest level of the next smaller size class) become nodes of a graph
 * whose edges cost distortion + lambda * bits; ac_codes = the (code << 16 | length) table the
 * rate is priced with.  Integer arithmetic throughout, 32-bit wrap-around like the reference. */
int orc_trellis_block(const int16_t in[64], const orc_quantizer* q, const uint32_t ac_codes[256],
                      int16_t zz[64]) {
  enum { kNodes = 1 + 2 * 63 };
  uint32_t n_score[kNodes], n_disto[kNodes];
  int n_level[kNodes], n_pos[kNodes], n_rank[kNodes], n_prev[kNodes], n_neg[kNodes];
  uint32_t disto0[64];
  int count = 1;                                   /* node 0 = the sink */
  n_score[0] = 0; n_disto[0] = 0; n_pos[0] = 0; n_rank[0] = 0;
  n_prev[0] = -1; n_level[0] = 0; n_neg[0] = 0;
  disto0[0] = 0;
  const uint32_t zrl_len = ac_codes[0xf0] & 0xff;
  for (int i = 1; i < 64; ++i) {
    const int j = orc_zigzag[i];
    const uint32_t qq = (uint32_t)q->quant[j] << 4;
    const uint32_t lambda = qq * qq / 32u;
    const int raw = in[j];
    const int neg = raw < 0;
    const int V = neg ? -raw : raw;
    disto0[i] = (uint32_t)(V * V) + disto0[i - 1];
    int v = quantize_abs((uint32_t)V, q->iquant[j], q->bias[j]);
    if (v == 0) continue;
    int nbits = bit_length((uint32_t)v);
    for (int k = 0; k < 2; ++k) {
      const int err = V - v * (int)qq;
      const int me = count;
      n_level[me] = v; n_neg[me] = neg; n_pos[me] = i;
      n_disto[me] = (uint32_t)(err * err);
      n_score[me] = 0xffffffffu;
      /* best predecessor, walking back towards the sink */
      int found = 0;
      const uint32_t base_disto = n_disto[me] + disto0[i - 1];
      for (int c = me - 1; c >= 0; --c) {
        const int run = i - 1 - n_pos[c];
        if (run < 0) continue;
        uint32_t bits = (uint32_t)nbits + (uint32_t)(run >> 4) * zrl_len;
        const uint32_t disto = base_disto - disto0[n_pos[c]];
        if (disto + lambda * bits >= n_score[me]) break;
        bits += ac_codes[((run & 15) << 4) | nbits] & 0xff;
        const uint32_t score = disto + lambda * bits + n_score[c];
        if (score < n_score[me]) {
          n_score[me] = score; n_disto[me] = disto;
          n_prev[me] = c; n_rank[me] = n_rank[c] + 1;
          found = 1;
        }
      }
      if (found) ++count;
      --nbits;
      if (nbits <= 0) break;
      v = (1 << nbits) - 1;
    }
  }
  /* best entry point, searched backwards (the EOB cost is the same for all but position 63) */
  int best = 0;
  if (count > 1) {
    uint32_t best_score = 0xffffffffu;
    for (int c = count - 1; c >= 0; --c) {
      const uint32_t disto = disto0[63] - disto0[n_pos[c]];
      n_disto[c] += disto;
      n_score[c] += disto;
      if (n_score[c] < best_score) { best = c; best_score = n_score[c]; }
    }
  }
  for (int i = 1; i < 64; ++i) zz[i] = 0;
  for (int c = best; c > 0; c = n_prev[c]) zz[n_pos[c]] = (int16_t)(n_neg[c] ? -n_level[c] : n_level[c]);
  const int d = in[0];
  const int dc = d < 0 ? -quantize_abs((uint32_t)-d, q->iquant[0], q->bias[0])
                       : quantize_abs((uint32_t)d, q->iquant[0], q->bias[0]);
  zz[0] = (int16_t)dc;
  return dc;
}

/* ---------------------------------------------------------------- bit writer */

typedef struct {
  uint8_t* buf;
  size_t size, cap;
  uint32_t acc;      /* pending bits, right-aligned */
  int nbits;
} orc_bw;

static void bw_byte(orc_bw* w, uint8_t b) {
  if (w->size == w->cap) {
    w->cap = w->cap ? 2 * w->cap : 4096;
    w->buf = (uint8_t*)realloc(w->buf, w->cap);
  }
  w->buf[w->size++] = b;
}
/* MSB-first; every completed 0xFF byte of entropy data is followed by 0x00
 * (src/bit_writer.h:172-209). */
static void bw_put(orc_bw* w, uint32_t bits, int n) {
  while (n > 0) {
    const int take = n > 8 ? 8 : n;      /* feed at most a byte at a time */
    const uint32_t chunk = (bits >> (n - take)) & ((1u << take) - 1);
    w->acc = (w->acc << take) | chunk;
    w->nbits += take;
    n -= take;
    while (w->nbits >= 8) {
      const uint8_t b = (uint8_t)(w->acc >> (w->nbits - 8));
      bw_byte(w, b);
      if (b == 0xff) bw_byte(w, 0x00);
      w->nbits -= 8;
    }
    w->acc &= (1u << w->nbits) - 1;
  }
}
static void bw_code(orc_bw* w, uint32_t packed) { bw_put(w, packed >> 16, (int)(packed & 0xff)); }
/* src/bit_writer.cc:107-116: pad with 1-bits to a byte boundary */
static void bw_pad(orc_bw* w) {
  const int pad = (8 - w->nbits) & 7;
  if (pad) bw_put(w, (1u << pad) - 1, pad);
}
static void bw_raw(orc_bw* w, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i) bw_byte(w, p[i]);
}

/* ---------------------------------------------------------------- entropy coding */

/* Emit one block given its quantized coefficients zz[] (zig-zag, zz[0] = DC value).
 * DC: src/entropy.cc:133-150 (difference category + suffix); AC: src/entropy.cc:161-198
 * with the run/level pairs src/quantize.cc:288-320 would have produced. */
static void code_block(orc_bw* w, const int16_t zz[64], int* dc_pred,
                       const uint32_t* dc_codes, const uint32_t* ac_codes) {
  const int diff = zz[0] - *dc_pred;
  *dc_pred = zz[0];
  if (diff == 0) {
    bw_code(w, dc_codes[0]);
  } else {
    const int n = bit_length((uint32_t)(diff < 0 ? -diff : diff));
    const uint32_t suffix = (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << n) - 1);
    bw_code(w, dc_codes[n]);
    bw_put(w, suffix, n);
  }
  int run = 0;
  for (int i = 1; i < 64; ++i) {
    const int v = zz[i];
    if (v == 0) { ++run; continue; }
    while (run >= 16) { bw_code(w, ac_codes[0xf0]); run -= 16; }
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    const int n = bit_length(a);
    const uint32_t suffix = (v < 0 ? ~a : a) & ((1u << n) - 1);
    bw_code(w, ac_codes[(run << 4) | n]);
    bw_put(w, suffix, n);
    run = 0;
  }
  if (run > 0) bw_code(w, ac_codes[0x00]);      /* EOB iff last nonzero index < 63 */
}

/* ---------------------------------------------------------------- sharp YUV 4:2:0 (SJPEG_YUV_SHARP)
 * src/yuv_convert.cc (the iterative "sharp" RGB -> YUV 4:2:0 conversion): the picture is held as
 * luma-like W (10-bit) + chroma differences (R-W, G-W, B-W at half resolution); four sweeps
 * upsample the chroma, compare with the gamma-correct targets and feed the differences back.  A
 * sweep walks the row pairs top to bottom and updates the chroma rows IN PLACE (the row above a
 * pair is already this sweep's, the row below is still the previous sweep's): that order is
 * normative.  The two gamma tables come from libm's pow() exactly as the reference builds them. */
#include <math.h>
enum { kSfix = 2, kMaxY = (256 << kSfix) - 1, kGammaTab = 32, kG2LBits = 14 };
static uint32_t g_g2l[kMaxY + 1];
static uint32_t g_l2g[kGammaTab + 2];
static int g_gamma_ready = 0;

static void sharp_init_tables(void) {                       /* src/yuv_convert.cc:114-152 */
  if (g_gamma_ready) return;
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
    g_g2l[v] = (uint32_t)(value * final_scale + .5);
  }
  for (int v = 0; v <= kGammaTab; ++v) {
    const double g = scale * v;
    double value;
    if (g <= thresh) value = 4.5 * g;
    else value = (1. + a) * pow(g, 1. / gamma) - a;
    g_l2g[v] = (uint32_t)(kMaxY * value) + (1 << kG2LBits >> 1);
  }
  g_l2g[kGammaTab + 1] = g_l2g[kGammaTab];
  g_gamma_ready = 1;
}

static uint32_t lin2gamma(uint32_t value) {                 /* src/yuv_convert.cc:158-171 */
  const uint32_t v = value * kGammaTab;
  const uint32_t pos = v >> kG2LBits;
  const uint32_t x = v - (pos << kG2LBits);
  const uint32_t v0 = g_l2g[pos], v1 = g_l2g[pos + 1];
  return v0 + (((v1 - v0) * x) >> kG2LBits);
}
static int sharp_clip_y(int y) { return y < 0 ? 0 : y > kMaxY ? kMaxY : y; }
static int sharp_clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static uint32_t sharp_gray(uint32_t r, uint32_t g, uint32_t b) {            /* :435-438 */
  return (13933u * r + 46871u * g + 4732u * b + (1u << 16 >> 1)) >> 16;
}
static uint32_t sharp_down(int a, int b, int c, int d) {                    /* :440-446 */
  return lin2gamma((g_g2l[a] + g_g2l[b] + g_g2l[c] + g_g2l[d] + 2) >> 2);
}

/* W and chroma targets of one row pair of (interpolated or imported) 10-bit RGB: rows[row][c][x] */
static void sharp_eval(const uint16_t* r0, const uint16_t* r1, int w, uint16_t* w0, uint16_t* w1,
                       int16_t* uv /* [3][uv_w] */) {
  const int uv_w = w >> 1;
  for (int i = 0; i < w; ++i) {                                              /* UpdateW, :467-475 */
    w0[i] = (uint16_t)lin2gamma(sharp_gray(g_g2l[r0[i]], g_g2l[r0[w + i]], g_g2l[r0[2 * w + i]]));
    w1[i] = (uint16_t)lin2gamma(sharp_gray(g_g2l[r1[i]], g_g2l[r1[w + i]], g_g2l[r1[2 * w + i]]));
  }
  for (int i = 0; i < uv_w; ++i) {                                           /* UpdateChroma, :448-465 */
    uint32_t c[3];
    for (int k = 0; k < 3; ++k) {
      c[k] = sharp_down(r0[k * w + 2 * i], r0[k * w + 2 * i + 1], r1[k * w + 2 * i], r1[k * w + 2 * i + 1]);
    }
    const int W = (int)sharp_gray(c[0], c[1], c[2]);
    for (int k = 0; k < 3; ++k) uv[k * uv_w + i] = (int16_t)((int)c[k] - W);
  }
}

void orc_sharp_yuv(const uint8_t* rgb, int W, int H, int stride, uint8_t* yp, uint8_t* up, uint8_t* vp) {
  const int uv_w_out = (W + 1) >> 1;
  if (W <= 4 || H <= 4) {
    /* too small for the iterative conversion: plain averaging with its own constants (:57-100,674-690) */
    for (int y = 0; y < H; y += 2) {
      const uint8_t* r1 = rgb + (long)y * stride;
      const uint8_t* r2 = (y < H - 1) ? r1 + stride : r1;
      for (int row = 0; row < 2 && y + row < H; ++row) {
        const uint8_t* rr = row ? r2 : r1;
        for (int i = 0; i < W; ++i) {
          const int v = 19595 * rr[3 * i] + 38469 * rr[3 * i + 1] + 7471 * rr[3 * i + 2];
          yp[(y + row) * W + i] = (uint8_t)((v + (1 << 16 >> 1)) >> 16);
        }
      }
      for (int i = 0; i < uv_w_out; ++i) {
        int r, g, b;
        if (2 * i + 1 < W) {
          r = r1[6 * i] + r1[6 * i + 3] + r2[6 * i] + r2[6 * i + 3];
          g = r1[6 * i + 1] + r1[6 * i + 4] + r2[6 * i + 1] + r2[6 * i + 4];
          b = r1[6 * i + 2] + r1[6 * i + 5] + r2[6 * i + 2] + r2[6 * i + 5];
        } else {
          r = 2 * (r1[6 * i] + r2[6 * i]); g = 2 * (r1[6 * i + 1] + r2[6 * i + 1]); b = 2 * (r1[6 * i + 2] + r2[6 * i + 2]);
        }
        const int rnd = 1 << 18 >> 1;
        up[(y >> 1) * uv_w_out + i] = (uint8_t)sharp_clip8(128 + ((-11058 * r - 21709 * g + 32768 * b + rnd) >> 18));
        vp[(y >> 1) * uv_w_out + i] = (uint8_t)sharp_clip8(128 + ((32768 * r - 27439 * g - 5328 * b + rnd) >> 18));
      }
    }
    return;
  }
  sharp_init_tables();
  const int w = (W + 1) & ~1, h = (H + 1) & ~1, uv_w = w >> 1, uv_h = h >> 1;
  uint16_t* rows = (uint16_t*)malloc((size_t)6 * w * sizeof(uint16_t));     /* two rows of [3][w] */
  uint16_t* best_y = (uint16_t*)malloc((size_t)w * h * sizeof(uint16_t));
  uint16_t* target_y = (uint16_t*)malloc((size_t)w * h * sizeof(uint16_t));
  uint16_t* cur_w = (uint16_t*)malloc((size_t)2 * w * sizeof(uint16_t));
  int16_t* best_uv = (int16_t*)malloc((size_t)3 * uv_w * uv_h * sizeof(int16_t));
  int16_t* target_uv = (int16_t*)malloc((size_t)3 * uv_w * uv_h * sizeof(int16_t));
  int16_t* cur_uv = (int16_t*)malloc((size_t)3 * uv_w * sizeof(int16_t));
  uint16_t* r0 = rows; uint16_t* r1 = rows + 3 * w;
  /* import: 8 -> 10 bits with a half, right / bottom replication (:492-510,608-632) */
  for (int j = 0; j < H; j += 2) {
    for (int row = 0; row < 2; ++row) {
      uint16_t* dst = row ? r1 : r0;
      const int yy = (j + row < H) ? j + row : j;
      const uint8_t* src = rgb + (long)yy * stride;
      for (int i = 0; i < w; ++i) {
        const int xx = i < W ? i : W - 1;
        for (int k = 0; k < 3; ++k) dst[k * w + i] = (uint16_t)((src[3 * xx + k] << kSfix) | (1 << kSfix >> 1));
      }
    }
    for (int i = 0; i < w; ++i) {                                            /* StoreGray */
      best_y[j * w + i] = (uint16_t)sharp_gray(r0[i], r0[w + i], r0[2 * w + i]);
      best_y[(j + 1) * w + i] = (uint16_t)sharp_gray(r1[i], r1[w + i], r1[2 * w + i]);
    }
    sharp_eval(r0, r1, w, target_y + j * w, target_y + (j + 1) * w, target_uv + (j >> 1) * 3 * uv_w);
  }
  memcpy(best_uv, target_uv, (size_t)3 * uv_w * uv_h * sizeof(int16_t));
  /* sweeps (:634-668) */
  const uint64_t threshold = (uint64_t)(3.0 * w * h);
  uint64_t prev_diff = ~(uint64_t)0;
  for (int iter = 0; iter < 4; ++iter) {
    uint64_t diff_sum = 0;
    for (int j = 0; j < h; j += 2) {
      const int ry = j >> 1;
      const int16_t* cur = best_uv + ry * 3 * uv_w;
      const int16_t* prev = best_uv + (ry > 0 ? ry - 1 : 0) * 3 * uv_w;
      const int16_t* next = best_uv + (j < h - 2 ? ry + 1 : ry) * 3 * uv_w;
      /* chroma upsampled 9:3:3:1 onto the two rows, added to W (InterpolateTwoRows, :512-541) */
      for (int k = 0; k < 3; ++k) {
        const int16_t *A = cur + k * uv_w, *P = prev + k * uv_w, *N = next + k * uv_w;
        for (int x = 0; x < w; ++x) {
          const int near = x >> 1;
          int v_up, v_dn;
          if (x == 0 || x == w - 1) {
            v_up = (A[near] * 3 + P[near] + 2) >> 2;
            v_dn = (A[near] * 3 + N[near] + 2) >> 2;
          } else {
            const int far = (x & 1) ? near + 1 : near - 1;
            v_up = (A[near] * 9 + A[far] * 3 + P[near] * 3 + P[far] + 8) >> 4;
            v_dn = (A[near] * 9 + A[far] * 3 + N[near] * 3 + N[far] + 8) >> 4;
          }
          r0[k * w + x] = (uint16_t)sharp_clip_y(best_y[j * w + x] + v_up);
          r1[k * w + x] = (uint16_t)sharp_clip_y(best_y[(j + 1) * w + x] + v_dn);
        }
      }
      sharp_eval(r0, r1, w, cur_w, cur_w + w, cur_uv);
      for (int i = 0; i < 2 * w; ++i) {                                      /* SharpUpdateY, :175-185 */
        const int d = (int)target_y[j * w + i] - (int)cur_w[i];
        best_y[j * w + i] = (uint16_t)sharp_clip_y((int)best_y[j * w + i] + d);
        diff_sum += (uint64_t)(d < 0 ? -d : d);
      }
      int16_t* bu = best_uv + ry * 3 * uv_w;                                 /* SharpUpdateRGB */
      const int16_t* tu = target_uv + ry * 3 * uv_w;
      for (int i = 0; i < 3 * uv_w; ++i) bu[i] = (int16_t)(bu[i] + (tu[i] - cur_uv[i]));
    }
    if (iter > 0) {
      if (diff_sum < threshold) break;
      if (diff_sum > prev_diff) break;
    }
    prev_diff = diff_sum;
  }
  /* back to 8-bit Y / U / V (:543-575; note the -11058 / -5328 constants of this file) */
  const int rnd = 1 << 18 >> 1;
  for (int j = 0; j < H; ++j) {
    const int16_t* uvr = best_uv + (j >> 1) * 3 * uv_w;
    for (int i = 0; i < W; ++i) {
      const int Wv = best_y[j * w + i];
      const int r = uvr[(i >> 1)] + Wv, g = uvr[uv_w + (i >> 1)] + Wv, b = uvr[2 * uv_w + (i >> 1)] + Wv;
      yp[j * W + i] = (uint8_t)sharp_clip8((19595 * r + 38469 * g + 7471 * b + rnd) >> 18);
    }
  }
  for (int j = 0; j < uv_h; ++j) {
    const int16_t* uvr = best_uv + j * 3 * uv_w;
    for (int i = 0; i < uv_w; ++i) {
      const int r = uvr[i], g = uvr[uv_w + i], b = uvr[2 * uv_w + i];
      up[j * uv_w + i] = (uint8_t)sharp_clip8(128 + ((-11058 * r - 21709 * g + 32768 * b + rnd) >> 18));
      vp[j * uv_w + i] = (uint8_t)sharp_clip8(128 + ((32768 * r - 27439 * g - 5328 * b + rnd) >> 18));
    }
  }
  free(rows); free(best_y); free(target_y); free(cur_w); free(best_uv); free(target_uv); free(cur_uv);
}

/* ---------------------------------------------------------------- SjpegRiskiness / SJPEG_YUV_AUTO
 * src/jpeg_tools.cc:170-236 with the pixel -> 7x7x7 cell index of src/colors_rgb.cc:1085-1122.
 * `table` = the reference's trained 343 x 343 score table (src/score_7.cc), supplied by the caller
 * (tests read it out of the built reference, oracle/_ref): it is not part of this repository. */
static int risk_index(const uint8_t* p) {
  const int r = p[0], g = p[1], b = p[2];
  const uint32_t y = (uint32_t)((19595 * r + 38469 * g + 7471 * b + 32768) >> 16);
  int u = 128 + ((-11059 * r - 21709 * g + 32768 * b + 32768) >> 16);
  int v = 128 + ((32768 * r - 27439 * g - 5329 * b + 32768) >> 16);
  u = u < 0 ? 0 : u > 255 ? 255 : u;
  v = v < 0 ? 0 : v > 255 ? 255 : v;
  const uint32_t k = 0x0101u * 6u;
  return (int)(((y * k) >> 16) + 7 * (((uint32_t)u * k) >> 16) + 49 * (((uint32_t)v * k) >> 16));
}

int orc_riskiness(const uint8_t* rgb, int W, int H, int stride, const uint8_t* table, float* risk) {
  int64_t score_sum = 0, score_num = 0, gray_num = 0;
  const int gray = (7 / 2) * (1 + 7) * 7, gray_min = gray - gray % 7;
  for (int j = 1; j < H; ++j) {
    const uint8_t* r1 = rgb + (long)(j - 1) * stride;
    const uint8_t* r2 = r1 + stride;
    for (int i = 0; i < W - 1; ++i) {
      const int idx0 = risk_index(r1 + 3 * i), idx1 = risk_index(r1 + 3 * i + 3), idx2 = risk_index(r2 + 3 * i);
      const int score = table[idx0 + 343 * idx1] + table[idx0 + 343 * idx2] + table[idx1 + 343 * idx2];
      if (score > 4) { score_sum += score; score_num += 1; }
      gray_num += (idx0 >= gray_min && idx0 < gray_min + 7);
    }
  }
  const double count = (double)score_num;
  double gray_count = (double)gray_num;
  double total = (count > 0) ? score_sum / count : 0.;
  const double num_samples = (W - 1.) * (H - 1.);
  if (num_samples > 0.) gray_count /= num_samples;
  const double frac = 100. * count / ((double)W * H);
  if (frac < 1.) total = 0.;
  total = (total > 25.) ? 100. : total * 100. / 25.;
  if (risk != NULL) *risk = (float)total;
  return (gray_count > 0.995) ? ORC_YUV_400 : (total < 40.0) ? ORC_YUV_420 : (total < 70.0) ? ORC_YUV_SHARP : ORC_YUV_444;
}

/* ---------------------------------------------------------------- scan drivers */

static orc_source rgb_source(const uint8_t* rgb, int stride) {
  orc_source S;
  memset(&S, 0, sizeof(S));
  S.format = ORC_SRC_RGB; S.plane[0] = rgb; S.stride[0] = stride;
  return S;
}


typedef struct {
  orc_layout L;
  orc_quantizer q[2];
  int W, H, mb_w, mb_h;
  const uint32_t (*trellis_ac)[256];   /* non-NULL: trellis quantization priced with these AC codes */
} orc_scan;

static int scan_quantize(const orc_scan* s, const int16_t* blk, int t, int16_t zz[64]) {
  if (s->trellis_ac != NULL) return orc_trellis_block(blk, &s->q[t], s->trellis_ac[t], zz);
  return orc_quantize_block(blk, &s->q[t], zz);
}

static int scan_init(orc_scan* s, int W, int H, int yuv_mode, const uint8_t quant[2][64],
                     const uint8_t* min_quant, int q_bias) {
  if (!layout_for(yuv_mode, &s->L)) return 0;
  if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return 0;    /* src/enc.cc:406 */
  for (int c = 0; c < 2; ++c) {
    /* Encoder::SetQuantMatrices re-applies SetQuantMatrix(m, 100) (src/enc.cc:106-109):
     * identity except that 0 becomes 1. */
    orc_set_quant_matrix(quant[c], 100.f, s->q[c].quant);
    if (min_quant) memcpy(s->q[c].min_quant, min_quant + 64 * c, 64);
    else memset(s->q[c].min_quant, 1, 64);
    orc_finalize_quant(&s->q[c], q_bias);
  }
  s->trellis_ac = NULL;
  s->W = W; s->H = H;
  s->mb_w = (W + s->L.block_w - 1) / s->L.block_w;     /* src/enc.cc:410-411 */
  s->mb_h = (H + s->L.block_h - 1) / s->L.block_h;
  return 1;
}

size_t orc_scan_coeffs(const uint8_t* rgb, int W, int H, int stride, int yuv_mode,
                       const uint8_t quant[2][64], int q_bias, int16_t* zz) {
  const orc_source S = rgb_source(rgb, stride);
  return orc_scan_coeffs_src(&S, W, H, yuv_mode, quant, q_bias, zz);
}

size_t orc_scan_coeffs_src(const orc_source* S, int W, int H, int yuv_mode,
                           const uint8_t quant[2][64], int q_bias, int16_t* zz) {
  orc_scan s;
  if (!scan_init(&s, W, H, yuv_mode, quant, NULL, q_bias)) return 0;
  size_t nb = 0;
  int16_t in[6 * 64];
  for (int my = 0; my < s.mb_h; ++my) {
    for (int mx = 0; mx < s.mb_w; ++mx) {
      orc_get_samples_src(S, yuv_mode, W, H, mx, my, in);
      orc_fdct(in, s.L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < s.L.nb_comps; ++c) {
        for (int i = 0; i < s.L.nb_blocks[c]; ++i, blk += 64, ++nb) {
          orc_quantize_block(blk, &s.q[s.L.quant_idx[c]], zz + 64 * nb);
        }
      }
    }
  }
  return nb;
}

/* The hot loop: src/enc.cc:276-307 */
static void scan_emit(orc_scan* s, const orc_source* S, int yuv_mode, orc_bw* w,
                      uint32_t dc_codes[2][12], uint32_t ac_codes[2][256]) {
  int pred[3] = {0, 0, 0};                       /* ResetDCs, src/entropy.cc:155-159 */
  int16_t in[6 * 64], zz[64];
  for (int my = 0; my < s->mb_h; ++my) {
    for (int mx = 0; mx < s->mb_w; ++mx) {
      orc_get_samples_src(S, yuv_mode, s->W, s->H, mx, my, in);
      orc_fdct(in, s->L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < s->L.nb_comps; ++c) {
        const int t = s->L.quant_idx[c];
        for (int i = 0; i < s->L.nb_blocks[c]; ++i, blk += 64) {
          scan_quantize(s, blk, t, zz);
          code_block(w, zz, &pred[c], dc_codes[t], ac_codes[t]);
        }
      }
    }
  }
  bw_pad(w);
}

size_t orc_scan_bits(const uint8_t* rgb, int W, int H, int stride, int yuv_mode,
                     const uint8_t quant[2][64], int q_bias, uint8_t** out) {
  orc_scan s;
  *out = NULL;
  if (!scan_init(&s, W, H, yuv_mode, quant, NULL, q_bias)) return 0;
  uint32_t dc[2][12], ac[2][256];
  orc_default_codes(dc, ac);
  orc_bw w;
  memset(&w, 0, sizeof(w));
  const orc_source S = rgb_source(rgb, stride);
  scan_emit(&s, &S, yuv_mode, &w, dc, ac);
  *out = w.buf;
  return w.size;
}

/* ---------------------------------------------------------------- headers */

static void put16(orc_bw* w, int v) { bw_byte(w, (uint8_t)(v >> 8)); bw_byte(w, (uint8_t)v); }

typedef struct { uint8_t bits[16]; uint8_t syms[256]; int nsyms; } orc_huff;

static void default_huff(orc_huff h[4]) {        /* DC luma, DC chroma, AC luma, AC chroma */
  for (int c = 0; c < 2; ++c) {
    memset(&h[c], 0, sizeof(h[c])); memset(&h[2 + c], 0, sizeof(h[2 + c]));
    memcpy(h[c].bits, kDcBits[c], 16); memcpy(h[c].syms, kDcVals, 12); h[c].nsyms = 12;
    memcpy(h[2 + c].bits, kAcBits[c], 16); memcpy(h[2 + c].syms, kAcVals[c], 162); h[2 + c].nsyms = 162;
  }
}

static void write_headers_huff(orc_bw* w, const orc_scan* s, int yuv_mode, const orc_huff h[4]);
static void write_headers(orc_bw* w, const orc_scan* s, int yuv_mode) {
  orc_huff h[4];
  default_huff(h);
  write_headers_huff(w, s, yuv_mode, h);
}

static void write_headers_huff(orc_bw* w, const orc_scan* s, int yuv_mode, const orc_huff h[4]) {
  /* SOI + JFIF APP0, v1.01, 1:1 aspect, no thumbnail (src/headers.cc:48-55) */
  static const uint8_t app0[20] = {0xff, 0xd8, 0xff, 0xe0, 0, 16, 'J', 'F', 'I', 'F', 0,
                                   1, 1, 0, 0, 1, 0, 1, 0, 0};
  bw_raw(w, app0, sizeof(app0));
  /* DQT: 8-bit tables in zig-zag order, ids 0 (and 1) (src/headers.cc:182-196) */
  const int nq = (yuv_mode == ORC_YUV_400) ? 1 : 2;
  put16(w, 0xffdb); put16(w, nq * 65 + 2);
  for (int n = 0; n < nq; ++n) {
    bw_byte(w, (uint8_t)n);
    for (int i = 0; i < 64; ++i) bw_byte(w, s->q[n].quant[orc_zigzag[i]]);
  }
  /* SOF0 (src/headers.cc:202-219) */
  put16(w, 0xffc0); put16(w, 3 * s->L.nb_comps + 8);
  bw_byte(w, 8); put16(w, s->H); put16(w, s->W); bw_byte(w, (uint8_t)s->L.nb_comps);
  for (int c = 0; c < s->L.nb_comps; ++c) {
    bw_byte(w, (uint8_t)(c + 1)); bw_byte(w, (uint8_t)s->L.sampling[c]);
    bw_byte(w, (uint8_t)s->L.quant_idx[c]);
  }
  /* DHT: one segment per table: DC-luma, AC-luma, DC-chroma, AC-chroma (src/headers.cc:221-238) */
  const int nt = (s->L.nb_comps == 1) ? 1 : 2;
  for (int c = 0; c < nt; ++c) {
    put16(w, 0xffc4); put16(w, 3 + 16 + h[c].nsyms); bw_byte(w, (uint8_t)c);
    bw_raw(w, h[c].bits, 16); bw_raw(w, h[c].syms, (size_t)h[c].nsyms);
    put16(w, 0xffc4); put16(w, 3 + 16 + h[2 + c].nsyms); bw_byte(w, (uint8_t)(0x10 | c));
    bw_raw(w, h[2 + c].bits, 16); bw_raw(w, h[2 + c].syms, (size_t)h[2 + c].nsyms);
  }
  /* SOS (src/headers.cc:242-258) */
  put16(w, 0xffda); put16(w, 6 + 2 * s->L.nb_comps); bw_byte(w, (uint8_t)s->L.nb_comps);
  for (int c = 0; c < s->L.nb_comps; ++c) {
    bw_byte(w, (uint8_t)(c + 1)); bw_byte(w, (uint8_t)(s->L.quant_idx[c] * 0x11));
  }
  bw_byte(w, 0); bw_byte(w, 63); bw_byte(w, 0);
}

size_t orc_headers(int W, int H, int yuv_mode, const uint8_t quant[2][64], uint8_t* buf) {
  orc_scan s;
  if (!scan_init(&s, W, H, yuv_mode, quant, NULL, 0x78)) return 0;
  orc_bw w;
  memset(&w, 0, sizeof(w));
  write_headers(&w, &s, yuv_mode);
  memcpy(buf, w.buf, w.size);
  free(w.buf);
  return w.size;
}

size_t orc_encode_matrices(const uint8_t* rgb, int W, int H, int stride,
                           const uint8_t quant[2][64], const uint8_t* min_quant,
                           int q_bias, int yuv_mode, uint8_t** out) {
  orc_scan s;
  *out = NULL;
  if (rgb == NULL || abs(stride) < 3 * W) return 0;          /* src/api.cc:35-36 */
  if (!scan_init(&s, W, H, yuv_mode, quant, min_quant, q_bias)) return 0;
  uint32_t dc[2][12], ac[2][256];
  orc_default_codes(dc, ac);
  orc_bw w;
  memset(&w, 0, sizeof(w));
  write_headers(&w, &s, yuv_mode);                          /* src/enc.cc:415-443 order */
  const orc_source S = rgb_source(rgb, stride);
  scan_emit(&s, &S, yuv_mode, &w, dc, ac);
  put16(&w, 0xffd9);                                        /* src/headers.cc:262-268 */
  *out = w.buf;
  return w.size;
}

/* ---------------------------------------------------------------- restart-marker variant
 * NOT the reference's output (it never writes DRI / RSTn, src/sjpegi.h:68-74): the same picture,
 * same tables, same coefficients, with the scan cut every `ri` MCUs the way ITU-T T.81 section B.2.4.4 /
 * F.1.2.3 and libjpeg do it: the interval's bits padded to a byte with 1-bits (stuffed like any
 * other byte), marker FF D0+(n & 7), DC predictors back to zero.  A DRI segment (FF DD 00 04 Ri)
 * stands in front of SOS.  Used to check the GPU's optional restart mode byte for byte; that it
 * decodes to the same pixels as the exact stream is checked with an independent decoder (tests). */
size_t orc_encode_rst(const uint8_t* rgb, int W, int H, int stride, float quality, int yuv_mode,
                      int ri, uint8_t** out) {
  orc_scan s;
  uint8_t m[2][64];
  *out = NULL;
  if (rgb == NULL || abs(stride) < 3 * W || ri <= 0 || ri > 65535) return 0;
  orc_quality_matrices(quality, m);
  if (!scan_init(&s, W, H, yuv_mode, m, NULL, 0x78)) return 0;
  uint32_t dc[2][12], ac[2][256];
  orc_default_codes(dc, ac);
  orc_bw w, hdr;
  memset(&w, 0, sizeof(w));
  memset(&hdr, 0, sizeof(hdr));
  write_headers(&hdr, &s, yuv_mode);
  /* DRI in front of the SOS segment (the last FF DA of the header) */
  size_t sos = hdr.size;
  while (sos >= 2 && !(hdr.buf[sos - 2] == 0xff && hdr.buf[sos - 1] == 0xda)) --sos;
  sos -= 2;
  bw_raw(&w, hdr.buf, sos);
  put16(&w, 0xffdd); put16(&w, 4); put16(&w, ri);
  bw_raw(&w, hdr.buf + sos, hdr.size - sos);
  free(hdr.buf);
  const orc_source S = rgb_source(rgb, stride);
  int pred[3] = {0, 0, 0};
  int16_t in[6 * 64], zz[64];
  int count = 0, nrst = 0;
  const int n_mcus = s.mb_w * s.mb_h;
  for (int my = 0; my < s.mb_h; ++my) {
    for (int mx = 0; mx < s.mb_w; ++mx) {
      orc_get_samples_src(&S, yuv_mode, s.W, s.H, mx, my, in);
      orc_fdct(in, s.L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < s.L.nb_comps; ++c) {
        const int t = s.L.quant_idx[c];
        for (int i = 0; i < s.L.nb_blocks[c]; ++i, blk += 64) {
          scan_quantize(&s, blk, t, zz);
          code_block(&w, zz, &pred[c], dc[t], ac[t]);
        }
      }
      ++count;
      if (count % ri == 0 && count < n_mcus) {
        bw_pad(&w);
        bw_byte(&w, 0xff); bw_byte(&w, (uint8_t)(0xd0 + (nrst++ & 7)));
        pred[0] = pred[1] = pred[2] = 0;
      }
    }
  }
  bw_pad(&w);
  put16(&w, 0xffd9);
  *out = w.buf;
  return w.size;
}

size_t orc_encode(const uint8_t* rgb, int W, int H, int stride, float quality,
                  int yuv_mode, uint8_t** out) {
  uint8_t m[2][64];
  orc_quality_matrices(quality, m);
  return orc_encode_matrices(rgb, W, H, stride, m, NULL, 0x78, yuv_mode, out);
}

/* ---------------------------------------------------------------- methods 1..6 */

/* src/histogram.cc:98-108 (plain-C variant: bins >= 128 are dropped), :317-339 */
void orc_histogram(const uint8_t* rgb, int W, int H, int stride, int yuv_mode, uint32_t* hist) {
  const orc_source S = rgb_source(rgb, stride);
  orc_histogram_src(&S, W, H, yuv_mode, hist);
}

void orc_histogram_src(const orc_source* S, int W, int H, int yuv_mode, uint32_t* hist) {
  orc_layout L;
  memset(hist, 0, 2 * 64 * 128 * sizeof(uint32_t));
  if (!layout_for(yuv_mode, &L)) return;
  const int mb_w = (W + L.block_w - 1) / L.block_w, mb_h = (H + L.block_h - 1) / L.block_h;
  int16_t in[6 * 64];
  for (int my = 0; my < mb_h; ++my) {
    for (int mx = 0; mx < mb_w; ++mx) {
      orc_get_samples_src(S, yuv_mode, W, H, mx, my, in);
      orc_fdct(in, L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < L.nb_comps; ++c) {
        uint32_t* h = hist + (size_t)L.quant_idx[c] * 64 * 128;
        for (int n = 0; n < L.nb_blocks[c]; ++n, blk += 64) {
          for (int i = 0; i < 64; ++i) {
            const int k = (blk[i] < 0 ? -blk[i] : blk[i]) >> 2;
            if (k < 128) ++h[i * 128 + k];
          }
        }
      }
    }
  }
}

/* src/histogram.cc:126-315.  double/float expressions in the reference's order. */
void orc_adapt_quant(const uint32_t* hist, int nb_comps, orc_quantizer q[2], int q_bias,
                     int qdelta_max_luma, int qdelta_max_chroma) {
  enum { DMIN = -12, NQ = 25 };
  static const float weight[NQ] = {0, 0, 0, 0, 0, 1, 5, 16, 43, 94, 164, 228, 255,
                                   228, 164, 94, 43, 16, 5, 1, 0, 0, 0, 0, 0};
  for (int idx = (nb_comps > 1 ? 1 : 0); idx >= 0; --idx) {
    const int delta_max = (idx == 0 ? qdelta_max_luma : qdelta_max_chroma) - DMIN;
    float sizes[64][NQ], dist[64][NQ];
    double num = 0., den = 0.;
    uint64_t omit = 0x103ull;
    for (int pos = 0; pos < 64; ++pos) {
      if (omit & (1ull << pos)) continue;
      const int dq0 = q[idx].quant[pos], min_dq0 = q[idx].min_quant[pos];
      const int bias = 1 << 16 >> 1;
      const uint32_t* h = hist + ((size_t)idx * 64 + pos) * 128;
      int total = 0, last = 0;
      for (int i = 0; i < 128; ++i) { total += (int)h[i]; if (h[i]) last = i + 1; }
      if (total < 0.5 * last) { omit |= 1ull << pos; continue; }
      double sw = 0., sx = 0., sxx = 0., syy1 = 0., sy1 = 0., sxy1 = 0., sy2 = 0., sxy2 = 0.;
      for (int d = 0; d < NQ; ++d) {
        double bsum = 0., dsum = 0.;
        const int dq = dq0 + (d + DMIN);
        if (dq >= min_dq0 && dq <= 255) {
          const int idq = ((1 << 16) + dq - 1) / dq;
          for (int i = 0; i < last; ++i) {
            if (h[i]) {
              const int hi = (int)h[i];
              const int v = (i << 2) + 2;
              const int qv = (v * idq + bias) >> 16;
              if (qv) {
                const int bits = bit_length((uint32_t)qv);
                const int dqv = qv * dq;
                const int error = (v - dqv) * (v - dqv);
                bsum += hi * bits;
                dsum += hi * error;
              } else {
                dsum += hi * v * v;
              }
            }
          }
          dist[pos][d] = (float)dsum;
          sizes[pos][d] = (float)bsum;
          const double w = weight[d];
          if (w > 0.) {
            const double x = (double)(d + DMIN);
            sw += w; sx += w * x; sxx += w * x * x;
            sy1 += w * dsum; syy1 += w * dsum * dsum; sy2 += w * bsum;
            sxy1 += w * dsum * x; sxy2 += w * bsum * x;
          }
        } else {
          dist[pos][d] = 3.402823466e+38F;     /* FLT_MAX */
          sizes[pos][d] = 0;
        }
      }
      const double cov = sw * sxy1 - sx * sy1;
      if (cov * cov < 0.5 * (sw * sxx - sx * sx) * (sw * syy1 - sy1 * sy1)) { omit |= 1ull << pos; continue; }
      num += cov;
      den += sw * sxy2 - sx * sy2;
    }
    double lambda = 0x80;
    if (num > 1000. && den < -10.) { lambda = -num / den; if (lambda < 1.) lambda = 1.; }
    for (int pos = 0; pos < 64; ++pos) {
      if (omit & (1ull << pos)) continue;
      float best = 3.402823466e+38F;
      int best_dq = 0;
      for (int d = 0; d <= delta_max; ++d) {
        if (dist[pos][d] < 3.402823466e+38F) {
          const float score = dist[pos][d] + lambda * sizes[pos][d];
          if (score < best) { best = score; best_dq = d + DMIN; }
        }
      }
      q[idx].quant[pos] = (uint8_t)(q[idx].quant[pos] + best_dq);
    }
    orc_finalize_quant(&q[idx], q_bias);
  }
}

static void block_stats(const int16_t zz[64], int* dc_pred, uint32_t* f /*[272]*/) {
  const int diff = zz[0] - *dc_pred;             /* src/entropy.cc:208-227 */
  *dc_pred = zz[0];
  ++f[256 + bit_length((uint32_t)(diff < 0 ? -diff : diff))];
  int run = 0;
  for (int i = 1; i < 64; ++i) {
    const int v = zz[i];
    if (v == 0) { ++run; continue; }
    if (run >> 4) f[0xf0] += (uint32_t)(run >> 4);
    ++f[((run & 15) << 4) | bit_length((uint32_t)(v < 0 ? -v : v))];
    run = 0;
  }
  if (run > 0) ++f[0x00];
}

static void scan_stats(orc_scan* s, const orc_source* S, int yuv_mode, uint32_t* freq) {
  int pred[3] = {0, 0, 0};
  int16_t in[6 * 64], zz[64];
  memset(freq, 0, 2 * 272 * sizeof(uint32_t));
  for (int my = 0; my < s->mb_h; ++my) {
    for (int mx = 0; mx < s->mb_w; ++mx) {
      orc_get_samples_src(S, yuv_mode, s->W, s->H, mx, my, in);
      orc_fdct(in, s->L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < s->L.nb_comps; ++c) {
        const int t = s->L.quant_idx[c];
        for (int i = 0; i < s->L.nb_blocks[c]; ++i, blk += 64) {
          scan_quantize(s, blk, t, zz);
          block_stats(zz, &pred[c], freq + 272 * t);
        }
      }
    }
  }
}

void orc_symbol_stats(const uint8_t* rgb, int W, int H, int stride, int yuv_mode,
                      const uint8_t quant[2][64], int q_bias, uint32_t* freq) {
  const orc_source S = rgb_source(rgb, stride);
  orc_symbol_stats_src(&S, W, H, yuv_mode, quant, q_bias, freq);
}

void orc_symbol_stats_src(const orc_source* S, int W, int H, int yuv_mode,
                          const uint8_t quant[2][64], int q_bias, uint32_t* freq) {
  orc_scan s;
  if (!scan_init(&s, W, H, yuv_mode, quant, NULL, q_bias)) return;
  scan_stats(&s, S, yuv_mode, freq);
}

/* src/entropy.cc:254-430: Huffman's merging with the all-ones code reserved and lengths
 * limited to 16 bits. */
int orc_build_optimal(const uint32_t* freq, int size, uint8_t out_bits[16], uint8_t syms[256]) {
  int codesizes[257], chain[257], tail[257];
  uint64_t sorted[257];
  int nb = 0;
  for (int i = 0; i < size; ++i) {
    if (freq[i] > 0) sorted[nb++] = ((uint64_t)freq[i] << 9) | (uint64_t)i;
    codesizes[i] = 0; chain[i] = -1; tail[i] = i;
  }
  const int nsyms = nb;
  for (int a = 1; a < nb; ++a) {                 /* decreasing order; keys are unique */
    const uint64_t key = sorted[a];
    int b = a;
    while (b > 0 && sorted[b - 1] < key) { sorted[b] = sorted[b - 1]; --b; }
    sorted[b] = key;
  }
  sorted[nb++] = (1ull << 9) | (uint64_t)size;   /* pseudo symbol -> forbidden all-ones code */
  codesizes[size] = 0; chain[size] = -1; tail[size] = size;
  for (int n = nb - 1; n >= 1; --n) {
    const uint64_t s1 = sorted[n - 1], s2 = sorted[n];
    const int i = (int)(s1 & 0x1ff), j = (int)(s2 & 0x1ff);
    chain[tail[i]] = j;
    tail[i] = tail[j];
    for (int t = i; t >= 0; t = chain[t]) ++codesizes[t];
    const uint64_t merged = s1 + (s2 & ~0x1ffull);
    int k = n - 1;
    while (k > 0 && sorted[k - 1] < merged) { sorted[k] = sorted[k - 1]; --k; }
    sorted[k] = merged;
  }
  uint8_t bits[32];
  memset(bits, 0, sizeof(bits));
  int max_bits = 0;
  for (int i = 0; i <= size; ++i) {
    int sz = codesizes[i];
    if (sz > 0) {
      if (sz > 32) { sz = 32; codesizes[i] = 32; }
      ++bits[sz - 1];
      if (sz > max_bits) max_bits = sz;
    }
  }
  int start[32], position = 0;
  for (int i = 0; i < max_bits; ++i) { start[i] = position; position += bits[i]; }
  memset(syms, 0, 256);
  for (int sym = 0; sym < size; ++sym) {
    if (codesizes[sym] > 0) syms[start[codesizes[sym] - 1]++] = (uint8_t)sym;
  }
  for (int l = max_bits - 1; l >= 16; --l) {
    while (bits[l] > 0) {
      int k = l - 2;
      while (bits[k] == 0) --k;
      bits[l] -= 2; bits[l - 1] += 1; bits[k] -= 1; bits[k + 1] += 2;
    }
  }
  max_bits = 16;
  while (bits[--max_bits] == 0) {}
  --bits[max_bits];
  memcpy(out_bits, bits, 16);
  return nsyms;
}

/* src/enc.cc:391-448 with the method flags of :121-129 (no trellis, single pass) */
size_t orc_encode_full(const uint8_t* rgb, int W, int H, int stride, const uint8_t quant[2][64],
                       const uint8_t* min_quant, int q_bias, int qdelta_max_luma,
                       int qdelta_max_chroma, int yuv_mode, int method, uint8_t** out) {
  *out = NULL;
  if (rgb == NULL || abs(stride) < 3 * W) return 0;
  const orc_source S = rgb_source(rgb, stride);
  return orc_encode_src(&S, W, H, quant, min_quant, q_bias, qdelta_max_luma, qdelta_max_chroma,
                        yuv_mode, method, out);
}

size_t orc_encode_src(const orc_source* S, int W, int H, const uint8_t quant[2][64],
                      const uint8_t* min_quant, int q_bias, int qdelta_max_luma,
                      int qdelta_max_chroma, int yuv_mode, int method, uint8_t** out) {
  orc_scan s;
  *out = NULL;
  if (yuv_mode == ORC_YUV_SHARP) {
    /* EncoderSharp420 (src/encoders.cc:512-541): the sharp conversion makes planes, the planar
     * 4:2:0 encoder takes them from there.  Packed RGB only (BGRA / RGBA are repacked first,
     * src/api.cc:208-224). */
    if (S->format != ORC_SRC_RGB) return 0;
    const int uv_w = (W + 1) >> 1, uv_h = (H + 1) >> 1;
    uint8_t* planes = (uint8_t*)malloc((size_t)W * H + 2 * (size_t)uv_w * uv_h);
    orc_sharp_yuv(S->plane[0], W, H, S->stride[0], planes, planes + (size_t)W * H,
                  planes + (size_t)W * H + (size_t)uv_w * uv_h);
    orc_source P;
    memset(&P, 0, sizeof(P));
    P.format = ORC_SRC_YUV420;
    P.plane[0] = planes; P.stride[0] = W;
    P.plane[1] = planes + (size_t)W * H; P.stride[1] = uv_w;
    P.plane[2] = P.plane[1] + (size_t)uv_w * uv_h; P.stride[2] = uv_w;
    const size_t n = orc_encode_src(&P, W, H, quant, min_quant, q_bias, qdelta_max_luma,
                                    qdelta_max_chroma, ORC_YUV_420, method, out);
    free(planes);
    return n;
  }
  if (S->format == ORC_SRC_GRAY) yuv_mode = ORC_YUV_400;
  else if (S->format == ORC_SRC_YUV444) yuv_mode = ORC_YUV_444;
  else if (S->format >= ORC_SRC_YUV420) yuv_mode = ORC_YUV_420;
  if (method < 0) method = 0;
  if (method > 8) method = 8;                     /* src/enc.cc:122 */
  if (!scan_init(&s, W, H, yuv_mode, quant, min_quant, q_bias)) return 0;
  const int adaptive = method >= 3, optimize = (method != 0 && method != 3);
  /* methods 7, 8: trellis quantization, priced with the standard AC tables (InitCodes(true) in
   * SinglePassScanOptimized, src/enc.cc:330-334) */
  uint32_t std_dc[2][12], std_ac[2][256];
  orc_default_codes(std_dc, std_ac);
  if (method >= 7) s.trellis_ac = (const uint32_t (*)[256])std_ac;
  if (adaptive) {
    uint32_t* hist = (uint32_t*)malloc(2 * 64 * 128 * sizeof(uint32_t));
    orc_histogram_src(S, W, H, yuv_mode, hist);
    orc_adapt_quant(hist, s.L.nb_comps, s.q, q_bias, qdelta_max_luma, qdelta_max_chroma);
    free(hist);
  }
  orc_huff h[4];
  default_huff(h);
  uint32_t dc[2][12], ac[2][256];
  if (optimize) {
    uint32_t freq[2][272];
    scan_stats(&s, S, yuv_mode, &freq[0][0]);
    const int nt = s.L.nb_comps == 1 ? 1 : 2;
    for (int t = 0; t < nt; ++t) {
      memset(&h[t], 0, sizeof(h[t])); memset(&h[2 + t], 0, sizeof(h[2 + t]));
      h[t].nsyms = orc_build_optimal(freq[t] + 256, 12, h[t].bits, h[t].syms);
      h[2 + t].nsyms = orc_build_optimal(freq[t], 256, h[2 + t].bits, h[2 + t].syms);
    }
  }
  memset(dc, 0, sizeof(dc)); memset(ac, 0, sizeof(ac));
  for (int t = 0; t < 2; ++t) {
    orc_build_huffman(h[t].bits, h[t].syms, dc[t]);
    orc_build_huffman(h[2 + t].bits, h[2 + t].syms, ac[t]);
  }
  orc_bw w;
  memset(&w, 0, sizeof(w));
  write_headers_huff(&w, &s, yuv_mode, h);
  scan_emit(&s, S, yuv_mode, &w, dc, ac);
  put16(&w, 0xffd9);
  *out = w.buf;
  return w.size;
}

size_t orc_encode_method(const uint8_t* rgb, int W, int H, int stride, float quality,
                         int yuv_mode, int method, uint8_t** out) {
  uint8_t m[2][64];
  orc_quality_matrices(quality, m);
  return orc_encode_full(rgb, W, H, stride, m, NULL, 0x78, 12, 1, yuv_mode, method, out);
}

/* ---------------------------------------------------------------- size / PSNR search */

static float estimate_quality(const uint8_t m[64]) {      /* src/jpeg_tools.cc:143-167, luma */
  int best_q = 0;
  float best = 256.f * 256 * 64 + 1;
  for (int q = 0; q <= 100; ++q) {
    uint8_t t[64];
    orc_set_quant_matrix(kAnnexK1[0], orc_qfactor((float)q), t);
    float score = 0;
    for (int i = 0; i < 64; ++i) {
      const float d = (float)t[i] - (float)m[i];
      score += d * d;
      if (score > best) break;
    }
    if (score < best) { best = score; best_q = q; }
  }
  return (float)best_q;
}

/* src/dichotomy.cc:302-323 + src/quantize.cc:553-568 */
static float psnr_and_error(const orc_source* S, const orc_scan* s, int yuv_mode, uint64_t* error_out) {
  uint64_t error = 0;
  int16_t in[6 * 64];
  for (int my = 0; my < s->mb_h; ++my) {
    for (int mx = 0; mx < s->mb_w; ++mx) {
      orc_get_samples_src(S, yuv_mode, s->W, s->H, mx, my, in);
      orc_fdct(in, s->L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < s->L.nb_comps; ++c) {
        const orc_quantizer* Q = &s->q[s->L.quant_idx[c]];
        for (int n = 0; n < s->L.nb_blocks[c]; ++n, blk += 64) {
          uint32_t err = 0;
          for (int j = 0; j < 64; ++j) {
            int32_t v0 = blk[j] < 0 ? -blk[j] : blk[j];
            const uint32_t v = Q->quant[j] * (uint32_t)quantize_abs((uint32_t)v0, Q->iquant[j], Q->bias[j]);
            v0 >>= 4;
            err += ((uint32_t)v0 - v) * ((uint32_t)v0 - v);
          }
          error += err;
        }
      }
    }
  }
  if (error_out != NULL) *error_out = error;
  const uint64_t size = 64ull * (uint64_t)s->mb_w * s->mb_h * s->L.mcu_blocks;
  return (error > 0 && size > 0) ? 4.3429448f * log(size / (error / 255. / 255.)) : 99.f;
}
static float psnr_of(const orc_source* S, const orc_scan* s, int yuv_mode) {
  return psnr_and_error(S, s, yuv_mode, NULL);
}
/* the two quantities one search pass measures, for direct kernel checks */
uint64_t orc_quant_error_src(const orc_source* S, int W, int H, int yuv_mode, const uint8_t quant[2][64],
                             int q_bias) {
  orc_scan s;
  uint64_t err = 0;
  if (!scan_init(&s, W, H, yuv_mode, quant, NULL, q_bias)) return 0;
  psnr_and_error(S, &s, yuv_mode, &err);
  return err;
}

/* bits of the entropy segment incl. 0xFF escapes of completed bytes (BitCounter, bit_writer.h:292-365) */
static size_t counted_bits(orc_scan* s, const orc_source* S, int yuv_mode, uint32_t dc[2][12], uint32_t ac[2][256]) {
  orc_bw w;
  memset(&w, 0, sizeof(w));
  /* emit without the final padding: scan_emit pads, so redo its loop here */
  int pred[3] = {0, 0, 0};
  int16_t in[6 * 64], zz[64];
  for (int my = 0; my < s->mb_h; ++my) {
    for (int mx = 0; mx < s->mb_w; ++mx) {
      orc_get_samples_src(S, yuv_mode, s->W, s->H, mx, my, in);
      orc_fdct(in, s->L.mcu_blocks);
      const int16_t* blk = in;
      for (int c = 0; c < s->L.nb_comps; ++c) {
        const int t = s->L.quant_idx[c];
        for (int i = 0; i < s->L.nb_blocks[c]; ++i, blk += 64) {
          orc_quantize_block(blk, &s->q[t], zz);
          code_block(&w, zz, &pred[c], dc[t], ac[t]);
        }
      }
    }
  }
  const size_t bits = 8 * w.size + (size_t)w.nbits;   /* escapes are already in w.size */
  free(w.buf);
  return bits;
}
uint64_t orc_counted_bits_src(const orc_source* S, int W, int H, int yuv_mode, const uint8_t quant[2][64],
                              int q_bias) {
  orc_scan s;
  uint32_t dc[2][12], ac[2][256];
  if (!scan_init(&s, W, H, yuv_mode, quant, NULL, q_bias)) return 0;
  orc_default_codes(dc, ac);
  return counted_bits(&s, S, yuv_mode, dc, ac);
}

size_t orc_encode_search(const orc_source* S, int W, int H, const uint8_t quant[2][64],
                         const uint8_t* min_quant, int q_bias, int qdelta_max_luma,
                         int qdelta_max_chroma, int yuv_mode, int huffman, int adaptive,
                         int target_mode, float target_value, int passes, float tolerance,
                         float qmin_in, float qmax_in, int trellis, uint8_t** out) {
  orc_scan s;
  *out = NULL;
  if (S->format == ORC_SRC_GRAY) yuv_mode = ORC_YUV_400;
  else if (S->format == ORC_SRC_YUV444) yuv_mode = ORC_YUV_444;
  else if (S->format >= ORC_SRC_YUV420) yuv_mode = ORC_YUV_420;
  if (!scan_init(&s, W, H, yuv_mode, quant, min_quant, q_bias)) return 0;
  passes = passes < 1 ? 1 : passes > 20 ? 20 : passes;            /* src/api.cc:169 */
  /* use_trellis only takes effect with Huffman_compress + adaptive_quantization (method 4 -> 7,
   * src/api.cc:153-157).  The trellis prices its rate with the AC tables that are CURRENT when a
   * pass starts (InitCodes(true) in StoreRunLevels, src/dichotomy.cc:86): the standard ones at
   * first, then whatever the previous size pass compiled. */
  trellis = trellis && huffman && adaptive;
  uint32_t rate_dc[2][12], rate_ac[2][256], pass_rate_ac[2][256];
  orc_default_codes(rate_dc, rate_ac);
  orc_huff pass_h[4];
  default_huff(pass_h);
  int last_is_best = 0;
  /* SearchHook::Setup, src/dichotomy.cc:41-52 */
  const int for_size = (target_mode == 1);
  const float target = target_value;
  const float tol = tolerance / 100.;
  float qmin = (qmin_in < 0) ? 0 : qmin_in;
  float qmax = (qmax_in > 100) ? 100 : (qmax_in < qmin_in) ? qmin_in : qmax_in;
  float q = estimate_quality(quant[0]);
  q = q < qmin ? qmin : q > qmax ? qmax : q;
  uint32_t* hist = NULL;
  if (adaptive) {
    hist = (uint32_t*)malloc(2 * 64 * 128 * sizeof(uint32_t));
    orc_histogram_src(S, W, H, yuv_mode, hist);
  }
  const int nt = s.L.nb_comps == 1 ? 1 : 2;
  uint8_t opt_quants[2][64];
  float best = 0.f;
  for (int p = 0; p < passes; ++p) {
    for (int c = 0; c < 2; ++c) {                  /* NextMatrix + FinalizeQuantMatrix */
      orc_set_quant_matrix(kAnnexK1[c], orc_qfactor(q), s.q[c].quant);
      orc_finalize_quant(&s.q[c], q_bias);
    }
    if (adaptive) orc_adapt_quant(hist, s.L.nb_comps, s.q, q_bias, qdelta_max_luma, qdelta_max_chroma);
    float result;
    if (for_size) {
      orc_huff h[4];
      default_huff(h);
      uint32_t freq[2][272];
      if (trellis) {
        memcpy(pass_rate_ac, rate_ac, sizeof(rate_ac));
        s.trellis_ac = (const uint32_t (*)[256])pass_rate_ac;
      }
      scan_stats(&s, S, yuv_mode, &freq[0][0]);
      if (huffman) {
        for (int t = 0; t < nt; ++t) {
          memset(&h[t], 0, sizeof(h[t])); memset(&h[2 + t], 0, sizeof(h[2 + t]));
          h[t].nsyms = orc_build_optimal(freq[t] + 256, 12, h[t].bits, h[t].syms);
          h[2 + t].nsyms = orc_build_optimal(freq[t], 256, h[2 + t].bits, h[2 + t].syms);
        }
      }
      uint32_t dc[2][12], ac[2][256];
      memset(dc, 0, sizeof(dc)); memset(ac, 0, sizeof(ac));
      for (int t = 0; t < 2; ++t) { orc_build_huffman(h[t].bits, h[t].syms, dc[t]); orc_build_huffman(h[2 + t].bits, h[2 + t].syms, ac[t]); }
      if (trellis) {
        /* the compiled tables are the current ones from here on.  InitCodes() writes the codes of the
         * symbols a table HAS over the encoder's code arrays (src/entropy.cc:98-128) and leaves the
         * rest as they were: the rate table accumulates, it is never cleared. */
        for (int t = 0; t < nt; ++t) orc_build_huffman(h[2 + t].bits, h[2 + t].syms, rate_ac[t]);
        memcpy(pass_h, h, sizeof(pass_h));
      }
      /* HeaderSize(), src/dichotomy.cc:210-241 (no metadata here) */
      size_t size = 20 + (size_t)nt * 65 + 2 + 2 + 8 + 3 * s.L.nb_comps + 2 + 6 + 2 * s.L.nb_comps + 2 + 2;
      for (int t = 0; t < nt; ++t) size += (2 + 3 + 16 + h[t].nsyms) + (2 + 3 + 16 + h[2 + t].nsyms);
      size *= 8;
      if (huffman) {                               /* EntropySize(), src/entropy.cc:230-245 */
        for (int t = 0; t < nt; ++t) {
          for (int len = 0; len < 12; ++len) if (freq[t][256 + len]) size += (size_t)freq[t][256 + len] * ((dc[t][len] & 0xff) + len);
          for (int sym = 0; sym < 256; ++sym) if (freq[t][sym]) size += (size_t)freq[t][sym] * ((ac[t][sym] & 0xff) + (sym & 0x0f));
        }
      } else {
        size += counted_bits(&s, S, yuv_mode, dc, ac);
      }
      result = size / 8.f;
    } else {
      result = psnr_of(S, &s, yuv_mode);
    }
    last_is_best = (p == 0 || fabs(result - target) < best);
    if (last_is_best) {
      memcpy(opt_quants[0], s.q[0].quant, 64);
      memcpy(opt_quants[1], s.q[1].quant, 64);
      best = fabs(result - target);
    }
    /* SearchHook::Update, src/dichotomy.cc:54-71 */
    int done = (fabs(result - target) < tol * target);
    if (done) break;
    if (result > target) qmax = q; else qmin = q;
    const float last_q = q;
    q = (qmin + qmax) / 2.;
    done = (fabs(q - last_q) < 0.15);
    if (done) break;
  }
  free(hist);
  if (!trellis) {
    /* final encode with the best matrices (no further adaptation) */
    return orc_encode_src(S, W, H, opt_quants, min_quant, q_bias, qdelta_max_luma, qdelta_max_chroma,
                          yuv_mode, huffman ? 1 : 0, out);
  }
  /* trellis (src/dichotomy.cc:176-204): the best matrices come back; if the last pass was a size
   * pass and also the best one its stored run/levels ARE the stream (quantized with the rate
   * table that pass started from, coded with the tables it compiled); otherwise the blocks are
   * quantized once more with the tables current now, and the tables compiled from that. */
  for (int c = 0; c < 2; ++c) {
    orc_set_quant_matrix(opt_quants[c], 100.f, s.q[c].quant);
    orc_finalize_quant(&s.q[c], q_bias);
  }
  orc_huff h[4];
  if (for_size && last_is_best) {
    s.trellis_ac = (const uint32_t (*)[256])pass_rate_ac;
    memcpy(h, pass_h, sizeof(h));
  } else {
    s.trellis_ac = (const uint32_t (*)[256])rate_ac;
    uint32_t freq[2][272];
    scan_stats(&s, S, yuv_mode, &freq[0][0]);
    default_huff(h);
    for (int t = 0; t < nt; ++t) {
      memset(&h[t], 0, sizeof(h[t])); memset(&h[2 + t], 0, sizeof(h[2 + t]));
      h[t].nsyms = orc_build_optimal(freq[t] + 256, 12, h[t].bits, h[t].syms);
      h[2 + t].nsyms = orc_build_optimal(freq[t], 256, h[2 + t].bits, h[2 + t].syms);
    }
  }
  uint32_t dc[2][12], ac[2][256];
  memset(dc, 0, sizeof(dc)); memset(ac, 0, sizeof(ac));
  for (int t = 0; t < 2; ++t) { orc_build_huffman(h[t].bits, h[t].syms, dc[t]); orc_build_huffman(h[2 + t].bits, h[2 + t].syms, ac[t]); }
  orc_bw w;
  memset(&w, 0, sizeof(w));
  write_headers_huff(&w, &s, yuv_mode, h);
  scan_emit(&s, S, yuv_mode, &w, dc, ac);
  put16(&w, 0xffd9);
  *out = w.buf;
  return w.size;
}

void orc_free(void* p) { free(p); }
