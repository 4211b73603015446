// Constants, launch arguments and device helpers shared by the kernels of scan_engine.hip
// (geometry, LDS carve, scans, packed int16 arithmetic, the colour conversion of every source layout).
// Part of the single translation unit scan_engine.hip: included there inside its anonymous
// namespace, after <hip/hip_runtime.h> and sjpeg_hip.h; not a stand-alone header.

// ------------------------------------------------------------------------------------
// geometry

constexpr int kThreads = 256;        // stitch kernels (K2..K5): 4 waves
constexpr int kScanThreads = 256;    // K1: 4 waves; 252 of them own a block (41 coded MCUs + 1 halo MCU in 4:2:0)
constexpr int kSlotBytes = 144;      // 64 int16 + 16 B pad: conflict-free ds_read_b128 per lane
constexpr int kMaxBlockBits = 1728;  // 22 (DC) + 63*27 (AC) rounded up; reference bound enc.cc:206-209
constexpr int kChunkWords = 1024;    // K3/K5 chunk: 4 KiB of un-stuffed stream
constexpr int kChunkBytes = kChunkWords * 4;

template <int MODE> struct Geo;
template <> struct Geo<SJPEG_HIP_YUV420> {
  static constexpr int kBpm = 6, kMcuPx = 16, kSegMcus = 41;    // (41 + 1 halo) * 6 = 252 threads
};
template <> struct Geo<SJPEG_HIP_YUV444> {
  static constexpr int kBpm = 3, kMcuPx = 8, kSegMcus = 82;     // 83 * 3 = 249; 246 coded blocks like 4:2:0 (the part list's size)
};
template <> struct Geo<SJPEG_HIP_YUV400> {
  static constexpr int kBpm = 1, kMcuPx = 8, kSegMcus = 246;    // 247; 246 coded blocks
};

// device copy of sjpeg_hip_scan_tables, pre-digested
// Laid out as the kernels stage it: two contiguous groups copied to LDS 16 bytes per thread.
struct alignas(16) DevTables {
  // group A (1152 B, lives in the idle bit window until the DC codes are done)
  uint4 q[2][32];          // per natural-order PAIR (2j, 2j+1): {iq0 | iq1<<16, bias0*iq0, bias1*iq1, q0 | q1<<16}
  uint32_t dc[2][12];
  // bits a block's AC entries (sign-magnitude pairs) must NOT have for the lean walk to be provably
  // in place: levels of n <= n_safe bits, n_safe = largest n with len(0, n') + n' <= 16 for all n' <= n
  uint32_t safe_mask[2];
  // the AC code words the lean walk needs beside the merged ones: {EOB, ZRL} per table ((code << 16) | length)
  uint32_t eob_zrl[2][2];
  uint32_t pad_a[2];
  // group B (3456 B)
  uint32_t ac[2][256];
  // Lean entropy walk, indexed [run & 15][clz(level) - 22] (level 1..1023 <=> clz 31..22):
  // (code << n) | (code length + n) << 27, i.e. everything of a run/size symbol but the n suffix
  // bits, in one word.  Run-major: the symbols a picture is made of (runs 0..3, 1..7 level bits) are
  // 28 words in a row, one LDS bank each -- size-major (rows of 16 runs) they shared eight banks.
  uint32_t acm[2][16][10];
  // 1, 2 or 3 ZRL codes as a left-aligned 64-bit pattern {high word, low word, bits, 0} (index 0 unused):
  // what the stitch puts in front of a part whose first run is 16 or longer
  uint4 zrlpat[2][4];
  uint8_t tlen[2][256];    // trellis quantization: AC code lengths the rate is priced with
};
constexpr int kTablesA16 = 72, kTablesB16 = 216;      // uint4 per group
static_assert(sizeof(uint4) * kTablesA16 == 1152 && sizeof(DevTables) == 1152 + 3456 + 512, "DevTables layout");

// source classes the colour phase is specialised for
enum { kSrcRgb24 = 0, kSrcRgbx32 = 1, kSrcPlanes = 2 };

struct ScanArgs {
  const uint8_t* plane[3];      // packed colour / gray: [0]; planar YUV: Y, U, V; NV12/NV21: Y, UV
  long long row_stride[3], frame_stride[3];
  int rsh, bsh;                 // kSrcRgbx32: bit position of R and B inside a pixel dword (0 / 16)
  int cstep, uoff, voff;        // kSrcPlanes: bytes per chroma sample (2 = interleaved) and U/V offsets
  int W, H, mb_w, n_mcus, nseg, has_clip;
  int seg_first;                // band mode: frame-level index of this launch's segment 0
  int rst;                      // restart mode: segments are restart intervals (SJPEG_HIP_RESTART_MARKERS)
  const DevTables* tables;
  int tables_stride;            // 0: every frame uses tables[0]; 1: frame f uses tables[f]
  uint32_t* seg_words;     // [nframes*nseg][slot_words]: the first slot_words words of every segment
  uint32_t slot_words;
  uint32_t* seg_nbits;     // [nframes*nseg]
  // per-frame pool: what of a segment does not fit its slot (seg_xbase: word offset, ~0 = nothing or
  // the pool was full), and the rows of the checked walk; pool_ctr[2f] = words taken, [2f + 1] = overran
  uint32_t* pool;          // [nframes][pool_words]
  uint32_t pool_words;
  uint32_t* pool_ctr;      // [nframes][2]
  uint32_t* seg_xbase;     // [nframes*nseg]
  uint32_t* replay;        // [nframes*nseg][kScanThreads][36]: quantized blocks kept by a statistics pass (or NULL)
  int16_t* coeffs;         // kKindTap: quantized coefficients
  uint32_t* partial;       // kKindHisto / kKindStats: per-workgroup partial statistics
  unsigned long long* stamps;  // profiling (env SJPEG_HIP_STAMPS): 8 cycle stamps per workgroup
  int stamp_real;              // ... taken from the device-wide 100 MHz counter instead
  int ablate;              // profiling knob (env SJPEG_HIP_ABLATE): stop after phase 1/2/3; 0 = full
  // small launches without a K2 (stitch_kernels.h, fused_k2): every workgroup zeroes its share of the frame's 0xFF
  // counters for K3 -- clear_per of the frame's clear_n, from index seg * clear_per on (NULL: K2 does it)
  uint32_t* clear_ff;
  uint32_t clear_per, clear_n;
};

// LDS carve of K1 (bytes, all offsets multiples of 16): the block slots, then the region R behind them.
// Two layouts.  ROOMY (every kind, every mode): 256 slots, an 8 KiB bit window that also takes the tables of
// P1 / P2 and the bookkeeping of the entropy phase while it is idle, the raw AC table, 48.6 KB: three workgroups
// per CU.  COMPACT (the plain encode kind of 4:2:0 from packed RGB -- the headline path): 252 slots (42 MCUs of
// six blocks: the four spare threads have none), R = 4 656 B, 40 944 B in all: FOUR workgroups per CU and, for the
// compiler, 128 registers.  What makes R that small: the raw AC table stays in global memory (the lean walk needs
// its EOB / ZRL words only, which ride in the padding of the DC codes; the checked walk -- q >= 97 noise -- reads
// the rest from global memory), the part lengths go back into the tail word the part was described by, the DC
// predictors are read from the neighbour's slot, the part list lies over the dead quantizer table, and the bit
// window (4 448 B: the ordinary segment of 2.7 KB still fits one round) lies over everything the stitch no
// longer needs.
template <bool COMPACT> struct Lds;
template <> struct Lds<false> {
  static constexpr int kSlots = kScanThreads;
  static constexpr int kSamplesBytes = kSlots * kSlotBytes;      // 36864
  static constexpr int kWinWords = 2048;                         // LDS bit window, 32-bit MSB-first words (8 KiB)
  static constexpr int kOffWin = kSamplesBytes;
  // The quantizer table (1 KiB) and the DC codes are only read before the bit window is first
  // touched (P2 / DC coding), so they live INSIDE the window region.
  static constexpr int kOffQ = kOffWin;                          // uint4[64]
  static constexpr int kOffDc = kOffWin + 1024;                  // uint32[24] + safe masks [2] + EOB / ZRL words [4]
  static constexpr int kOffTlen = kOffWin + 1152;                // uint8[2][256], trellis kinds only
  static constexpr int kOffTq = kOffWin + 1664;                  // uint2[2][64], trellis kinds only: the quantizer in ZIG-ZAG order
  // Entropy-phase bookkeeping inside the (still idle) bit window.  All of it lies behind the tables staged at
  // the front of the window (quantizer 0..1023, DC codes ..1151, trellis lengths ..1663 bytes): the DC codes are
  // still being read by slow waves when fast ones already write the part list -- no barrier between the two.
  static constexpr int kOffHist = kOffWin + 4400;                // u32 [32]: the sort's bins
  static constexpr int kOffList = kOffWin + 4672;                // u16 [1024]: block | quarter << 8
  static constexpr int kListBytes = 2048;
  static constexpr int kOffDcw = kOffWin + 6720;                 // u32 [256]: DC code words (length << 24 | bits)
  static constexpr int kOffAc = kOffWin + kWinWords * 4 + 16;    // +1 spare word (16 B keeps alignment)
  static constexpr int kOffAcm = kOffAc + 2 * 256 * 4;           // uint32[2][16][10]: merged code words of the lean walk
  static constexpr int kOffZrl = kOffAcm + 2 * 160 * 4;          // uint4[2][4]: ZRL patterns
  static constexpr int kOffMisc = kOffZrl + 128;                 // scan scratch
  static constexpr int kLdsBytes = kOffMisc + 64;                // 48592: three workgroups per CU
  // kKindStats only: the symbol counters u32[2][256 AC + 16 DC], in TWO copies picked by lane parity, behind everything
  static constexpr int kOffStats = kLdsBytes, kStatsCopies = 2;
  static_assert(kOffTq >= kOffTlen + 512 && kOffHist >= kOffTq + 1024 && kOffList >= kOffHist + 128 && kOffDcw >= kOffList + 2048 &&
                kOffDcw + 4 * kScanThreads <= kOffWin + kWinWords * 4, "bookkeeping behind the tables, inside the window");
};
template <> struct Lds<true> {
  static constexpr int kSlots = 252;
  static constexpr int kSamplesBytes = kSlots * kSlotBytes;      // 36288
  static constexpr int kWinWords = 1112;
  static constexpr int kOffWin = kSamplesBytes;
  static constexpr int kOffQ = kOffWin;                          // uint4[64] (P1 / P2) ...
  static constexpr int kOffList = kOffWin;                       // ... u16[984] (written behind the DC barrier: P2 is over)
  static constexpr int kListBytes = 1968;
  static constexpr int kOffHist = kOffWin + 1968;                // u32 [20]
  static constexpr int kOffDc = kOffWin + 2048;                  // uint32[24] + safe masks [2] + EOB / ZRL words [4]
  static constexpr int kOffTlen = -1, kOffAc = -1, kOffTq = -1;  // (no trellis kind, no raw AC table in this layout)
  static constexpr int kOffAcm = kOffWin + 2176;
  // kKindStats (which has no merged code words, DC code words, ZRL patterns or window): ONE copy of the counters
  static constexpr int kOffStats = kOffAcm, kStatsCopies = 1;
  static constexpr int kOffDcw = kOffWin + 3456;                 // u32 [252]
  static constexpr int kOffZrl = kOffWin + 4464;                 // (read by the stitch: behind the window, like misc)
  static constexpr int kOffMisc = kOffZrl + 128;
  static constexpr int kLdsBytes = kOffMisc + 64;                // 40944
  static_assert(kOffHist >= kOffList + kListBytes && kOffDc >= kOffHist + 80 && kOffDcw >= kOffAcm + 1280 &&
                kOffZrl >= kOffDcw + 4 * kSlots && (kWinWords + 1) * 4 <= kOffZrl - kOffWin, "compact carve");
  static_assert(4 * kLdsBytes <= 160 * 1024, "four workgroups per CU");
};
constexpr int kLdsBytesStats = Lds<false>::kOffStats + 2 * 2 * 272 * 4;     // (the roomy layout of the statistics kind)
static_assert(Lds<true>::kOffStats + 2 * 272 * 4 <= Lds<true>::kOffZrl, "compact carve, statistics kind");
constexpr int kSamplesBytes = Lds<false>::kSamplesBytes;
static_assert(3 * kLdsBytesStats <= 160 * 1024, "three workgroups per CU (statistics kind)");
static_assert(3 * Lds<false>::kLdsBytes <= 160 * 1024, "three workgroups per CU");

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


// ---- packed int16 arithmetic (two columns per register), the 16-bit-lane formulation the
// reference's own SIMD paths use and prove bit-identical to the plain-C one (src/fdct.cc:147).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 as_pk(uint32_t v) { return __builtin_bit_cast(s16x2, v); }

// per-lane (a*K) >> 16 on both halves: two 24-bit multiplies + one byte permute
__device__ __forceinline__ s16x2 pk_mulhi(s16x2 a, int K) {
  const int lo = __mul24(static_cast<int>(a.x), K), hi = __mul24(static_cast<int>(a.y), K);
  return as_pk(__builtin_amdgcn_perm(static_cast<uint32_t>(hi), static_cast<uint32_t>(lo), 0x07060302u));
}
__device__ __forceinline__ s16x2 pk_swap(s16x2 a) {
  return as_pk(__builtin_amdgcn_alignbit(as_u32(a), as_u32(a), 16));
}
__device__ __forceinline__ s16x2 pk_const(int lo, int hi) {
  return as_pk((static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16));
}

// Column transform of TWO adjacent columns at once; operation order of src/fdct.cc:67-144.
// RANGES: proven, not argued -- tools/int16_ranges.py writes every statement of this function, of fdct_row8_pk and
// of row_quant as an affine form of the block's 64 samples and bounds it at the corners of the sample box, for
// samples -128 .. 127 (luma, planar sources) and -127 .. 128 (chroma from RGB: pure blue / red are +128) -- NOT for
// their union, which does not fit.  Tightest values (profiles/r04/int16_ranges.txt, tests/test_packed_ranges.py; the
// GPU tests paint these corners: test_extremal_patterns_of_the_packed_fdct):
//   int16 lanes   column pass: outputs |r0|, |r4| <= 8192, all others <= 6290; every multiplier operand <= 4926,
//                 every 24-bit product < 2^28 (u3 / t4: 4926 * 43790 = 2.16e8)
//                 row pass: a_i, b_i of rows 0 / 4 <= 16384; c1 = a0 - a3, c3 = a1 - a2 of rows 0 / 4 <= 32 640
//                 (127 below the lane's limit: eight rows x 16 samples x 255); |coefficient| <= 16 385
//   32-bit sums   dot-product chains <= 2^30 (row 0, acc0)
//   quantizer     |c| * iquant + bias * iquant <= 1.08e9 for any table the host can make; level <= 1025
__device__ __forceinline__ void fdct_col8_pk(uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3,
                                             uint32_t& r4, uint32_t& r5, uint32_t& r6, uint32_t& r7) {
  const s16x2 x0 = as_pk(r0), x1 = as_pk(r1), x2 = as_pk(r2), x3 = as_pk(r3);
  const s16x2 x4 = as_pk(r4), x5 = as_pk(r5), x6 = as_pk(r6), x7 = as_pk(r7);
  s16x2 d07 = x0 - x7, s07 = x0 + x7;
  s16x2 d25 = x2 - x5, s25 = x2 + x5;
  s16x2 d34 = x3 - x4, s34 = x3 + x4;
  s16x2 d16 = x1 - x6, s16 = x1 + x6;
  s16x2 ed = s07 - s34, es = s07 + s34;
  s16x2 fd = s16 - s25, fs = s16 + s25;
  const s16x2 a = es << 3, b = fs << 3;
  r0 = as_u32(a + b);
  r4 = as_u32(a - b);
  ed = ed << 3; fd = fd << 3; d34 = d34 << 3; d07 = d07 << 3;
  r2 = as_u32(pk_mulhi(fd, 27146) + ed);
  r6 = as_u32(pk_mulhi(ed, 27146) - fd);
  // ((x << 4) * 23170) >> 16 == (x * (23170 << 4)) >> 16: the shift rides in the 24-bit multiplier
  // (|d16 -+ d25| <= 510, the product stays below 2^28: range table above)
  const s16x2 od = pk_mulhi(d16 - d25, 23170 << 4);
  const s16x2 os = pk_mulhi(d16 + d25, 23170 << 4);
  const s16x2 p3 = d34 - od, p1 = d34 + od;
  const s16x2 p0 = d07 - os, p2 = d07 + os;
  // ((p * K) >> 16) + p == (p * (K + 65536)) >> 16 exactly (|p| <= 4926: the product stays below 2^28; range table above)
  const s16x2 u3 = pk_mulhi(p3, 65536 - 21746);              // t3 - 1
  const s16x2 t4 = pk_mulhi(p0, 65536 - 21746);
  const s16x2 t5 = pk_mulhi(p2, 13036);
  // a + b + 1 == a - ~b on two's complement halves (the bitwise NOT is a full-rate op)
  r1 = as_u32(pk_mulhi(p1, 13036) - as_pk(~as_u32(p2)));     // t1 = mulhi + p2 + 1
  r3 = as_u32(p0 + as_pk(~as_u32(u3)));                       // p0 - (u3 + 1)
  r5 = as_u32(p3 + t4);
  r7 = as_u32(t5 - p1);
}

__device__ __forceinline__ int dot2(s16x2 a, int klo, int khi, int acc) {
  return __builtin_amdgcn_sdot2(a, pk_const(klo, khi), acc, false);
}
// the same with a zero accumulator: the three-operand encoding takes the 0 inline (the
// accumulate-in-place form the compiler picks would need a v_mov first)
__device__ __forceinline__ int dot2z(s16x2 a, int klo, int khi) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(as_u32(a)), "s"(as_u32(pk_const(klo, khi))));
  return d;
}

// Row transform of one row held as 4 packed pairs IN SLOT ORDER -- (x0,x1) (x3,x2) (x4,x5) (x7,x6):
// P1 writes the second and fourth pair of every row with their halves exchanged, which costs it
// nothing (operand order of the instruction that packs them) and leaves the butterflies below
// without a single half-swap; the column pass works on each half by itself and does not care.
// 32-bit wrap-around accumulation like src/fdct.cc:174-209 (and pmaddwd in its SSE2 twin).
// acc[i] >> 16 is coefficient i of the row.
template <int C1, int C2, int C3, int C4, int C5, int C6, int C7>
__device__ __forceinline__ void fdct_row8_pk(const uint32_t* row, int* acc) {
  const s16x2 p0 = as_pk(row[0]), q1 = as_pk(row[1]), q2 = as_pk(row[2]), q3 = as_pk(row[3]);
  const s16x2 A01 = p0 + q3, B01 = p0 - q3;       // (a0,a1), (b0,b1)
  const s16x2 A32 = q1 + q2, B32 = q1 - q2;       // (a3,a2), (b3,b2)
  // even part: (c1, c3) = (a0 - a3, a1 - a2), one product pair per output.  The sums c0 = a0 + a3 and
  // c2 = a1 + a2 are NOT formed in 16 bits: a chroma sample can be +128 (pure blue: Cb = (32768 * 255 +
  // 32768) >> 16; pure red: Cr), four all-128 columns make the column pass' DC terms 8192 each and their
  // sum 32768 -- one past int16 (solid red / blue pictures came out wrong up to round 3: found by the fuzz).
  // Their two outputs are accumulated from (a0, a1) and (a3, a2) in 32 bits instead: one dot product more.
  // The DIFFERENCES do fit: |a0 - a3| <= 8 * 16 * 255 = 32 640 for either sample range (tools/int16_ranges.py).
  const s16x2 C13 = A01 - A32;
  acc[0] = dot2(A32, C4, C4, dot2z(A01, C4, C4));
  acc[4] = dot2(A32, C4, -C4, dot2z(A01, C4, -C4));
  acc[2] = dot2z(C13, C2, C6);
  acc[6] = dot2z(C13, C6, -C2);
  acc[1] = dot2(B32, C7, C5, dot2z(B01, C1, C3));
  acc[3] = dot2(B32, -C5, -C1, dot2z(B01, C3, -C7));
  acc[5] = dot2(B32, C3, C7, dot2z(B01, C5, -C1));
  acc[7] = dot2(B32, -C1, C3, dot2z(B01, C7, -C5));
}

// D = a.u16[half] * b.u16[half] + c   (one VOP3 op on packed operands)
__device__ __forceinline__ uint32_t mad_u16_lo(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t mad_u16_hi(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// D = a.u16[0 or 1] * k + c with a uniform 16-bit multiplier k (scalar operand)
__device__ __forceinline__ uint32_t mad_u16_lo_k(uint32_t a, uint32_t k, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t mad_u16_hi_k(uint32_t a, uint32_t k, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "s"(k), "v"(c));
  return d;
}
// D = a.u16[1] * b.u16[0] + c
__device__ __forceinline__ uint32_t mad_u16_hl(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// a.u16[0] * b.u16[0] + a.u16[1] * b.u16[1] + c, modulo 2^32
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}
__device__ __forceinline__ uint32_t sdot2u(uint32_t a, int klo, int khi, uint32_t c) {
  return static_cast<uint32_t>(__builtin_amdgcn_sdot2(as_pk(a), pk_const(klo, khi), static_cast<int>(c), false));
}
// (x1 >> 16) << 16 | (x0 >> 16) & 0xffff: the upper halves of two 32-bit sums as an int16 pair
__device__ __forceinline__ uint32_t pk_top(uint32_t x0, uint32_t x1) {
  return __builtin_amdgcn_perm(x1, x0, 0x07060302u);
}

// natural index -> zig-zag position
__device__ constexpr int kInvZig(int j) {
  constexpr int z[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
                         3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                         10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
                         21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
  return z[j];
}

// One row: transform, quantize, pack.  Produces 4 dwords of sign-magnitude entries
// (bit 15 = negative, bits 0..14 = level) for natural positions 8*ROW .. 8*ROW+7 and ORs the
// non-zero flags into the zig-zag-ordered 64-bit mask.
//   level = ((|c| + bias) * iquant) >> 20 == (|c|*iquant + bias*iquant) >> 20
// The reference's qthresh test is implied: |c| >= qthresh <=> level > 0 (quantize.cc:144-145).
template <int ROW>
__device__ __forceinline__ void quant_row(const uint32_t* cps, const uint4* qt, uint32_t* ent, uint32_t* nzq) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t cp = cps[k];                   // coefficients 2k, 2k + 1 of the row as an int16 pair
    const s16x2 c = as_pk(cp);
    const uint32_t ap = as_u32(__builtin_elementwise_max(c, pk_const(0, 0) - c));
    const uint4 t = qt[4 * ROW + k];
    // both >> 20 at once: the upper halves of the two sums as a pair, then a packed >> 4
    const u16x2 tops = __builtin_bit_cast(u16x2, pk_top(mad_u16_lo(ap, t.x, t.y), mad_u16_hi(ap, t.x, t.z)));
    const uint32_t lv = __builtin_bit_cast(uint32_t, tops >> u16x2{4, 4});
    ent[k] = (cp & 0x80008000u) | lv;
    // non-zero flags: both at once (packed min), each dropped at its zig-zag position of the
    // 16-bit mask of its quarter by one multiply-add (every position is written exactly once)
    uint32_t f;                                   // (min(l0, 1), min(l1, 1))
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(lv), "v"(0x00010001u));
    const int z0 = kInvZig(8 * ROW + 2 * k), z1 = kInvZig(8 * ROW + 2 * k + 1);   // folds after unroll
    if ((z0 >> 4) == (z1 >> 4)) {                 // same quarter (24 of the 32 pairs): one dot product
      nzq[z0 >> 4] = udot2(f, (1u << (z0 & 15)) | (1u << ((z1 & 15) + 16)), nzq[z0 >> 4]);
    } else {
      nzq[z0 >> 4] = mad_u16_lo_k(f, 1u << (z0 & 15), nzq[z0 >> 4]);
      nzq[z1 >> 4] = mad_u16_hi_k(f, 1u << (z1 & 15), nzq[z1 >> 4]);
    }
  }
}
template <int ROW, int C1, int C2, int C3, int C4, int C5, int C6, int C7>
__device__ __forceinline__ void row_quant(const uint32_t* row, const uint4* qt, uint32_t* ent, uint32_t* nzq) {
  int acc[8];
  fdct_row8_pk<C1, C2, C3, C4, C5, C6, C7>(row, acc);
  uint32_t cps[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cps[k] = __builtin_amdgcn_perm(static_cast<uint32_t>(acc[2 * k + 1]), static_cast<uint32_t>(acc[2 * k]), 0x07060302u);
  }
  quant_row<ROW>(cps, qt, ent, nzq);
}

// One row, transform only: the raw coefficients (int16 pairs, natural order) for the trellis.
template <int C1, int C2, int C3, int C4, int C5, int C6, int C7>
__device__ __forceinline__ void row_raw(const uint32_t* row, uint32_t* ent) {
  int acc[8];
  fdct_row8_pk<C1, C2, C3, C4, C5, C6, C7>(row, acc);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ent[k] = __builtin_amdgcn_perm(static_cast<uint32_t>(acc[2 * k + 1]), static_cast<uint32_t>(acc[2 * k]), 0x07060302u);
  }
}

// zig-zag position -> natural index, as data (the trellis walks positions in a run-time loop)
__device__ const unsigned char kZigTab[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// byte-permute selector that builds (E16[a], E16[b]) from the dwords holding them
__device__ constexpr uint32_t kPairSel(int a, int b) {
  const uint32_t lo = (a & 1) ? 0x0302u : 0x0100u;       // from S1 (second operand)
  const uint32_t hi = (b & 1) ? 0x0706u : 0x0504u;       // from S0 (first operand)
  return lo | (hi << 16);
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

// Raw dwords of 8 consecutive pixels of row y (coordinates clamp to the picture): 6 dwords for
// packed RGB, 8 for the 4-byte layouts.
template <int SRC>
__device__ __forceinline__ void load_px8(const ScanArgs& a, const uint8_t* frame_px, int x0, int y,
                                         bool inside, uint32_t* w) {
  if (SRC == kSrcRgb24) {
    load_row8(frame_px, a.row_stride[0], a.W, a.H, x0, y, inside, w);
  } else {
    // 4 bytes per pixel (BGRA / RGBA, alpha ignored: src/colors_rgb.cc:882-1025)
    if (inside) {
      __builtin_memcpy(w, frame_px + y * a.row_stride[0] + 4ll * x0, 32);
    } else {
      const int Wc = a.W;
      const int yy = y < a.H ? y : a.H - 1;
      const uint8_t* row = frame_px + yy * a.row_stride[0];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int xx = (x0 + i) < Wc ? (x0 + i) : Wc - 1;
        __builtin_memcpy(&w[i], row + 4ll * xx, 4);
      }
    }
  }
}

// The same 8 pixels as packed 16-bit operands: rg[i] = r_i | g_i << 16 (i = 0..7),
// bb[j] = b_2j | b_(2j+1) << 16 (j = 0..3).  One byte-permute per register.
template <int SRC>
__device__ __forceinline__ void unpack_px8(const ScanArgs& a, const uint32_t* w, uint32_t* rg, uint32_t* bb) {
  if (SRC == kSrcRgb24) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int o = 3 * i, d = o >> 2, sl = o & 3;                  // r at byte o, g at o + 1
      const uint32_t sel = sl | 0x0c00u | (static_cast<uint32_t>(sl + 1) << 16) | 0x0c000000u;
      rg[i] = __builtin_amdgcn_perm(w[d + 1 < 6 ? d + 1 : 5], w[d], sel);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = 6 * j + 2, d = o >> 2, sl = o & 3;              // b at bytes o and o + 3
      const uint32_t sel = sl | 0x0c00u | (static_cast<uint32_t>(sl + 3) << 16) | 0x0c000000u;
      bb[j] = __builtin_amdgcn_perm(w[d + 1 < 6 ? d + 1 : 5], w[d], sel);
    }
  } else {
    const uint32_t rs = static_cast<uint32_t>(a.rsh) >> 3, bs = static_cast<uint32_t>(a.bsh) >> 3;
    const uint32_t sel_rg = rs | 0x0c00u | 0x00010000u | 0x0c000000u;
    const uint32_t sel_bb = bs | 0x0c00u | ((4u + bs) << 16) | 0x0c000000u;
#pragma unroll
    for (int i = 0; i < 8; ++i) rg[i] = __builtin_amdgcn_perm(0u, w[i], sel_rg);
#pragma unroll
    for (int j = 0; j < 4; ++j) bb[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], sel_bb);
  }
}

// BT.601 full-range 16.16 fixed point (src/colors_rgb.cc:17-19,31-32,785-828) on packed operands.
// All sums are the reference's, modulo 2^32; the int16 results are read off the upper halves.
constexpr uint32_t kLumaRG = 19595u | (38469u << 16);
constexpr uint32_t kLumaRound = static_cast<uint32_t>(32768 - (128 << 16));
// luma of pixels 2j and 2j + 1 as an int16 pair; `swapped`: (2j + 1, 2j), the slot order of odd pairs
__device__ __forceinline__ uint32_t luma_pair(uint32_t rg0, uint32_t rg1, uint32_t bbj, uint32_t k7471,
                                              uint32_t rnd, bool swapped) {
  const uint32_t y0 = udot2(rg0, kLumaRG, mad_u16_lo(bbj, k7471, rnd));
  const uint32_t y1 = udot2(rg1, kLumaRG, mad_u16_hl(bbj, k7471, rnd));
  return swapped ? pk_top(y1, y0) : pk_top(y0, y1);
}
// 32-bit Cb / Cr sums (before the final shift) from R | G << 16 and a PAIR of blue terms b | b' << 16
// (src/colors_rgb.cc: Cb = -11059 R - 21709 G + 32768 B, Cr = 32768 R - 27439 G - 5329 B, + rounding;
// everything modulo 2^32).  wb / wb': 1 for the blue terms that count (both for a 2x2 sum whose
// blue halves were added pairwise, one of them for a single pixel).
template <int WB0, int WB1>
__device__ __forceinline__ uint32_t cb_sum(uint32_t RG, uint32_t BB, uint32_t rnd) {
  return sdot2u(RG, -11059, -21709, udot2(BB, (WB0 ? 32768u : 0u) | (WB1 ? 32768u << 16 : 0u), rnd));
}
template <int WB0, int WB1>
__device__ __forceinline__ uint32_t cr_sum(uint32_t RG, uint32_t BB, uint32_t k32768, uint32_t rnd) {
  return sdot2u(BB, WB0 ? -5329 : 0, WB1 ? -5329 : 0, sdot2u(RG, 0, -27439, mad_u16_lo(RG, k32768, rnd)));
}

// 8 level-shifted samples of an 8-bit plane (sample pitch `step` bytes), clamped coordinates:
// what Convert8To16b[Clipped] / Replicate8b produce (src/colors_rgb.cc:1212-1260)
__device__ __forceinline__ void fetch_plane(const uint8_t* plane, long long stride, int step, int pw,
                                            int ph, int x0, int y, int n, int* out) {
  const int yy = y < ph ? y : ph - 1;
  const uint8_t* row = plane + yy * stride;
  if (step == 1 && n == 8 && x0 + 8 <= pw) {
    uint32_t w[2];
    __builtin_memcpy(w, row + x0, 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = byte_of(w, i) - 128;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < n) {
        const int xx = (x0 + i) < pw ? (x0 + i) : pw - 1;
        out[i] = static_cast<int>(row[static_cast<long long>(xx) * step]) - 128;
      }
    }
  }
}

#ifdef SJPEG_HIP_PRIO_STRESS
// wave priority by wave index (pattern 1: 3,0,2,1 then 0,3,1,2; pattern 2: the reverse order)
template <int PATTERN>
__device__ __forceinline__ void prio_stress(int phase) {
  const int k = (((threadIdx.x >> 6) & 3) + 2 * phase + (PATTERN == 2 ? 1 : 0)) & 3;
  switch (k) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(0); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(1);
  }
}
#endif

// inclusive prefix sum over the 64 lanes of a wave, DPP only (no LDS round trips): four shifts inside
// the rows of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xf, 0xf, false));   // row_shr:1
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xf, 0xf, false));   // row_shr:2
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xf, 0xf, false));   // row_shr:4
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xf, 0xf, false));   // row_shr:8
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xa, 0xf, false));   // row_bcast:15
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xc, 0xf, false));   // row_bcast:31
  return x;
}

// workgroup exclusive scan of one uint32 per thread; returns exclusive prefix, *total = sum
template <int NT, bool TRAILING_BARRIER = true>
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t x, uint32_t* scratch /*>=8 u32*/,
                                                      uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_inclusive_scan(x);
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t s = scratch[w];
    if (w < wave) base += s;
    sum += s;
  }
  if (TRAILING_BARRIER) __syncthreads();          // (scratch may be reused right away)
  *total = sum;
  return base + incl - x;
}

