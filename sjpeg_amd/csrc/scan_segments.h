// K1 scan_segments: colour + fDCT + quantize + entropy-code one segment per workgroup
// (and its statistics / histogram / error / trellis / replay kinds).
// Part of the single translation unit scan_engine.hip: included there inside its anonymous
// namespace, after <hip/hip_runtime.h> and sjpeg_hip.h; not a stand-alone header.
// ------------------------------------------------------------------------------------
// K1: colour + fDCT + quantize + entropy-code one segment

// experiment switches of the entropy phase (tools/build_variant.sh); the defaults are the shipped kernel
#ifndef SJPEG_WALK_PIPE
#define SJPEG_WALK_PIPE 2
#endif
#ifndef SJPEG_NO_MERGE
#define SJPEG_NO_MERGE 0
#endif
// wave priorities by phase (s_setprio): one 2-bit field per phase -- P1 (bits 0-1), P2 (2-3), P3 (4-5), P4 (6-7).
// Shipped: the entropy phase and the stitch run at priority 2, colour conversion and DCT at 0 -- the waves that
// wait for LDS round trips issue the moment their data is back, the dense arithmetic of the other workgroups of the
// CU fills the rest.  A/B on one box, six alternating runs each (profiles/r05/phase_prio_ab.txt): K1 0.8549 ->
// 0.8445 ms, step 0.8964 -> 0.8899; P1 at 3 (0x03) and P2 at 3 (0x0c): nothing.  (The race-stress builds skew the
// priorities themselves.)
#ifndef SJPEG_PHASE_PRIO
#ifdef SJPEG_HIP_PRIO_STRESS
#define SJPEG_PHASE_PRIO 0
#else
#define SJPEG_PHASE_PRIO 0xa0
#endif
#endif
#define PHASE_PRIO(k) do { if (SJPEG_PHASE_PRIO) __builtin_amdgcn_s_setprio((SJPEG_PHASE_PRIO >> (2 * (k))) & 3); } while (0)

enum { kKindEncode = 0, kKindTap = 1, kKindHisto = 2, kKindStats = 3, kKindError = 4,
       kKindEncodeTrellis = 5, kKindStatsTrellis = 6,     // the same two with trellis quantization
       kKindEncodeReplay = 7,     // entropy-code the coefficients a statistics pass left behind
       kKindStatsCoef = 8 };      // statistics from the DCT coefficients a histogram pass left behind
constexpr int kHistoWords = 2 * 64 * 32;          // words of u8 counters [2][64][128] a workgroup bins one segment into (LDS)
// The histogram kind is PERSISTENT: a workgroup bins the segments seg, seg + gridDim.x, ... of its frame and leaves ONE
// partial behind -- 16-bit counters, two words per word of 8-bit ones (scan_reduce.h reduce_partials16).
constexpr int kHistoPartialWords = 2 * kHistoWords;
constexpr int kHistoMaxSegsPerGroup = 256;        // 246 blocks a segment at most: 16 bits hold 266 of them
// LDS of the histogram kind behind P1 (over the block slots): 256 staged half blocks of 64 + 16 bytes, then the 8-bit
// counters.  A WORD of counters belongs to one bin of FOUR positions -- 2q, 2q + 1, 32 + 2q, 33 + 2q, q = 0..15: the
// pair of coefficients lane q of the binning loop reads from the first half of a block and from the second --, so the
// byte a coefficient bumps is known at compile time and its bin is nothing but the word's address.  129 words per
// (table, q): bins 0..127 and one word that swallows everything above.
constexpr int kHistoStageStride = 80;
constexpr int kHistoOffBins = 256 * kHistoStageStride;
constexpr int kHistoPosBytes = 129 * 4;           // one (table, q) group
constexpr int kHistoTblBytes = 16 * kHistoPosBytes;
constexpr int kHistoLdsBytes = kHistoOffBins + 2 * kHistoTblBytes;   // 37 376: four workgroups per CU
constexpr int kStatsWords = 2 * 272;              // per-workgroup partial: u32 [2][256 AC + 16 DC]

// Race stress build (make STRESS=1|2): RACE_POINT(n) holds ONE wave of every workgroup back at
// point n when the engine's SJPEG_HIP_ABLATE is 0x5a000000 | count << 16 | n << 8 | wave -- a
// wave that must not run ahead (or lag behind) without a barrier shows up as a parity failure
// (tools/race_sweep.py walks all points and waves).
#ifdef SJPEG_HIP_PRIO_STRESS
#define RACE_POINT(n) race_point(a.ablate, n)
__device__ __forceinline__ void race_point(int code, int n) {
  // (bit 7 set: every wave BUT that one is held back, i.e. the one wave runs ahead)
  if ((code >> 24) == 0x5a && ((code >> 8) & 255) == n && ((((threadIdx.x >> 6) & 3) == (code & 3)) != ((code & 0x80) != 0))) {
    for (int i = 0; i < ((code >> 16) & 255); ++i) __builtin_amdgcn_s_sleep(127);
  }
}
#else
#define RACE_POINT(n)
#endif

// (the histogram kind is compiled for the THREE persistent workgroups per CU it runs as: 168 registers.  Held to the 128
// of four per CU -- what rounds 4 and 5 shipped, from when it ran four -- it spilt ten registers' worth of loop-invariant
// values that every segment reloaded from private memory: pass 0.217 -> 0.204 ms per 16 4K frames, the default-parameter
// calls 2-2.5 % (round 6).  Packed RGB only: ROCm 7.2's clang crashes in its register allocator on the 4-byte-pixel
// instantiation with the smaller LDS block)
// the compact LDS layout (scan_device.h): four workgroups per CU
#ifndef SJPEG_HISTO_WGS
#define SJPEG_HISTO_WGS 3          // workgroups per CU the histogram kind is compiled for (A/B: 4)
#endif
template <int MODE, int KINDX, int SRC>
constexpr bool kCompactLds = (KINDX == kKindEncode || KINDX == kKindEncodeReplay || KINDX == kKindStats || KINDX == kKindStatsCoef);

template <int MODE, int KINDX, int SRC>
__global__ __launch_bounds__(kScanThreads, ((KINDX == kKindHisto && SRC == kSrcRgb24) ? SJPEG_HISTO_WGS : (kCompactLds<MODE, KINDX, SRC> ? 4 : 1))) void scan_segments(const ScanArgs a) {
  constexpr bool TRELLIS = (KINDX == kKindEncodeTrellis || KINDX == kKindStatsTrellis);
  constexpr bool REPLAY = (KINDX == kKindEncodeReplay);
  // the block's unquantized coefficients come from the histogram pass of the same call (the adaptive methods run
  // one before they know the quantizer): no second colour conversion / DCT
  constexpr bool COEF = (KINDX == kKindStatsCoef);
  constexpr int KIND = (KINDX == kKindEncodeTrellis || KINDX == kKindEncodeReplay) ? kKindEncode : (KINDX == kKindStatsTrellis || COEF) ? kKindStats : KINDX;
  constexpr bool COMPACT = kCompactLds<MODE, KINDX, SRC>;
  // The statistics kinds (but the trellis one) count a block's symbols straight out of the thread's registers, zig-zag
  // position by position -- no entries in LDS, no parts, no sort, no walk (below, "kKindStats, direct")
#ifndef SJPEG_STATS_DIRECT
#define SJPEG_STATS_DIRECT 1
#endif
  constexpr bool DIRECT = SJPEG_STATS_DIRECT && (KINDX == kKindStats || KINDX == kKindStatsCoef);
  using L = Lds<COMPACT>;
  constexpr int kWinWords = L::kWinWords;
  using G = Geo<MODE>;
  constexpr int BPM = G::kBpm;
  constexpr int PX = G::kMcuPx;
  static_assert(!COMPACT || (G::kSegMcus + 1) * BPM <= L::kSlots, "a slot for every block of the segment and its halo MCU");
  // (round 4: 83 MCUs of 4:4:4 made 996 parts at q 92 and the list ran into the sort's bins)
  static_assert(2 * 4 * G::kSegMcus * BPM <= L::kListBytes, "the part list holds four parts of every coded block");
  // static, not `extern __shared__`: the address of a dynamic block is resolved after instruction
  // selection and leaves a `+ 0` in ~65 address computations of this kernel
  __shared__ __attribute__((aligned(16))) unsigned char smem[(KIND == kKindStats && !COMPACT) ? kLdsBytesStats : (KIND == kKindHisto && SRC == kSrcRgb24) ? kHistoLdsBytes : L::kLdsBytes];
  static_assert(kHistoLdsBytes >= kSamplesBytes && kHistoLdsBytes <= L::kLdsBytes && 4 * kHistoLdsBytes <= 160 * 1024, "histogram kind: slots, then staging + bins");
  uint32_t* const win = reinterpret_cast<uint32_t*>(smem + L::kOffWin);
  uint4* const lq = reinterpret_cast<uint4*>(smem + L::kOffQ);
  uint32_t* const ldc = reinterpret_cast<uint32_t*>(smem + L::kOffDc);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + L::kOffMisc);
  uint32_t* const dcw = reinterpret_cast<uint32_t*>(smem + L::kOffDcw);      // DC code words, by block
  // (the four spare threads of the compact layout have no slot: theirs would be the tables)
  const bool has_slot = !COMPACT || threadIdx.x < L::kSlots;
  typedef uint16_t __attribute__((may_alias)) u16_may_alias;

  const int tid = threadIdx.x;
  const int frame = blockIdx.y;
  // the histogram kind's 16-bit counters, two to a register: the 8-bit counters of words 16 * tid .. 16 * tid + 15 of
  // the LDS histogram (overflow words not counted), bytes 0 / 2 in the even and bytes 1 / 3 in the odd register (no
  // other kind has them)
  uint32_t hacc[KIND == kKindHisto ? 32 : 1];
  if (KIND == kKindHisto) {
#pragma unroll
    for (int i = 0; i < 32; ++i) hacc[i] = 0;
  }
  // every kind but the histogram takes ONE trip (gridDim.x = the frame's segments); the body ends with a return
  for (int seg = blockIdx.x;; seg += gridDim.x) {
  auto stamp = [&](int k) {
    if (a.stamps != nullptr && tid == 0) {
      a.stamps[(static_cast<size_t>(frame) * a.nseg + seg) * 8 + k] =
          a.stamp_real ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();
    }
  };
  stamp(0);
  PHASE_PRIO(0);
  RACE_POINT(0);
  if (KIND == kKindEncode && a.clear_ff != nullptr) {
    for (uint32_t i = tid; i < a.clear_per; i += kScanThreads) {
      const uint32_t idx = static_cast<uint32_t>(seg) * a.clear_per + i;
      if (idx < a.clear_n) a.clear_ff[static_cast<size_t>(frame) * a.clear_n + idx] = 0u;
    }
  }
  const int m_first = (seg + a.seg_first) * G::kSegMcus;       // first coded MCU of the segment
  const int n_coded = min(G::kSegMcus, a.n_mcus - m_first);
  // (restart mode: every segment is a restart interval, its DC predictors start at zero)
  const int halo = (m_first > 0 && !a.rst) ? 1 : 0;            // previous MCU: DC predictors only
  const uint8_t* const frame_px = a.plane[0] + frame * a.frame_stride[0];

  // tables -> LDS; issued once the first pixel loads are in flight (see P1)
#ifdef SJPEG_HIP_PRIO_STRESS
  // race stress (make STRESS=1|2): the waves of a workgroup run at different priorities, flipped at
  // the entropy phase -- a missing barrier shows up as a parity failure (one did, profiles/HISTORY_r01.md)
  prio_stress<SJPEG_HIP_PRIO_STRESS>(0);
#endif
  auto stage_tables = [&]() {
    if (KIND == kKindHisto) return;                // (no table is read, and there may be no room for one)
    const DevTables* t = a.tables + frame * a.tables_stride;
    // two contiguous groups, 16 bytes per thread: quantizer + DC codes + level bounds into the idle
    // window, AC codes + merged code words + ZRL patterns behind it
    const uint4* const t16 = reinterpret_cast<const uint4*>(t);
    if (COMPACT) {
      // quantizer | DC codes, level bounds, EOB / ZRL words | merged code words | ZRL patterns: four places,
      // still one 16-byte load per thread and group; the raw AC table is not staged at all
      constexpr int kAcm16 = kTablesA16 + 128;       // group B: 128 uint4 of raw AC codes in front of the merged ones
      if (tid < 64) reinterpret_cast<uint4*>(smem + L::kOffQ)[tid] = t16[tid];
      else if (tid < kTablesA16) reinterpret_cast<uint4*>(smem + L::kOffDc)[tid - 64] = t16[tid];
      if (KIND != kKindStats) {                    // (the statistics kind codes nothing: its counters lie where the merged code words would)
        if (tid >= 128 && tid < 128 + 80) reinterpret_cast<uint4*>(smem + L::kOffAcm)[tid - 128] = t16[kAcm16 + tid - 128];
        else if (tid >= 128 + 80 && tid < 128 + 88) reinterpret_cast<uint4*>(smem + L::kOffZrl)[tid - 208] = t16[kAcm16 + tid - 128];
      }
    } else {
      if (tid < kTablesB16) reinterpret_cast<uint4*>(smem + L::kOffAc)[tid] = t16[kTablesA16 + tid];
      if (tid < kTablesA16) reinterpret_cast<uint4*>(smem + L::kOffQ)[tid] = t16[tid];
      if (TRELLIS && tid < 128) {
        reinterpret_cast<uint32_t*>(smem + L::kOffTlen)[tid] = reinterpret_cast<const uint32_t*>(&t->tlen[0][0])[tid];
        // the trellis walks the positions of a block in a run-time loop: its quantizer entry -- reciprocal | step << 16,
        // bias -- in zig-zag order (looked up through kZigTab per position, a global load sat in front of every one)
        const int zj = kZigTab[tid & 63];
        const uint4 q4 = t16[(tid >> 6) * 32 + (zj >> 1)];
        reinterpret_cast<uint2*>(smem + L::kOffTq)[tid] = (zj & 1) ? make_uint2((q4.x >> 16) | (q4.w & 0xffff0000u), q4.z)
                                                                  : make_uint2((q4.x & 0xffffu) | (q4.w << 16), q4.y);
      }
    }
    // bookkeeping of the entropy phase that nothing touches until then: the sort's bins, the group queue
    if (KIND == kKindEncode || KIND == kKindStats) {
      if (tid < 20) reinterpret_cast<uint32_t*>(smem + L::kOffHist)[tid] = 0;
      if (tid == 32) misc[10] = 0;
    }
    if (KIND == kKindStats) {                      // the symbol counters (their own LDS behind everything else)
      uint32_t* const lf0 = reinterpret_cast<uint32_t*>(smem + L::kOffStats);
      for (int i = tid; i < L::kStatsCopies * kStatsWords + (COMPACT ? 60 : 0); i += kScanThreads) lf0[i] = 0;   // (+ the hot symbols' sets, compact carve)
    }
  };

  // Does the segment (its halo MCU included) touch a clipped MCU?  Uniform.  Interior segments run
  // a clamp-free copy of P1 and skip the AverageExtraLuma fix-up of P2 with its two barriers.
  bool interior = true;
  if (a.has_clip) {
    const int mcu0c = m_first - halo, mcu1 = m_first + n_coded - 1;   // first and last processed MCU
    const int my0c = mcu0c / a.mb_w;
    const int my1 = mcu1 / a.mb_w, mx1 = mcu1 - my1 * a.mb_w;
    const int mb_h = a.n_mcus / a.mb_w;
    const bool clip_row = (a.H % PX) != 0 && my1 == mb_h - 1;
    const bool clip_col = (a.W % PX) != 0 && (my1 > my0c || mx1 == a.mb_w - 1);
    interior = !(clip_row || clip_col);
  }

  // ---- P1: colour conversion, strips of 8 pixels (x2 rows for 4:2:0) --------------------
  // local MCU index ml: 0 = halo, 1..n_coded = coded MCUs; block slot = ml*BPM + k
  if (REPLAY || COEF) stage_tables();
  if (!REPLAY && !COEF) {
    const int ml_lo = 1 - halo;
    const int n_proc = n_coded + halo;
    constexpr int kRowsPerStrip = (MODE == SJPEG_HIP_YUV420) ? 2 : 1;
    constexpr int kStripsX = PX / 8;                           // strips per MCU row
    const int per_row = kStripsX * n_proc;
    // A thread keeps ONE strip column (one MCU, one x-half) and walks down its row pairs
    // yp0, yp0 + ngroups, ...: the index arithmetic (two integer divisions by run-time
    // values) is done once per thread instead of once per strip, and consecutive lanes still
    // read consecutive 24-byte pieces of a picture row.
    const int ngroups = kScanThreads / per_row;                 // 3 for a full 4:2:0 segment
    // tid / per_row without a per-lane division: both are below 256, so
    // (tid * ceil(2^16 / per_row)) >> 16 is exact
    const uint32_t magic = (65535u + static_cast<uint32_t>(per_row)) / static_cast<uint32_t>(per_row);   // uniform
    const int yp0 = static_cast<int>((static_cast<uint32_t>(tid) * magic) >> 16);
    const int rem = tid - yp0 * per_row;
    const int dl = rem / kStripsX;                              // MCU of the strip, counted from the first processed one
    const int ml = ml_lo + dl;
    const int xs = rem % kStripsX;
    // MCU coordinates: the first processed MCU is uniform (scalar division); a thread's own is at
    // most 41 further on, i.e. at most one row wrap when the picture is 42 MCUs wide or more
    const int mcu0 = m_first - halo;
    const int my0 = mcu0 / a.mb_w, mx0 = mcu0 - my0 * a.mb_w;
    int mb_x = mx0 + dl, mb_y = my0;
    if (a.mb_w > G::kSegMcus) {
      if (mb_x >= a.mb_w) { mb_x -= a.mb_w; ++mb_y; }
    } else {
      const int mcu = mcu0 + dl;
      mb_y = mcu / a.mb_w;
      mb_x = mcu - mb_y * a.mb_w;
    }
    const int x0 = mb_x * PX + xs * 8;
    const uint32_t k7471 = 7471u, k32768 = 32768u;             // multiplier operands (low halves)
    constexpr int kNW = (SRC == kSrcRgb24) ? 6 : 8;             // dwords per 8 pixels
#ifndef SJPEG_HISTO_BATCH
#define SJPEG_HISTO_BATCH 2
#endif
    // (the histogram kind holds 32 registers of counters across its segments: two row pairs in flight keep it
    // inside the 128 registers of four workgroups per CU)
    constexpr int kBatch = (KIND == kKindHisto) ? SJPEG_HISTO_BATCH : 3;   // row pairs in flight per thread
    // A segment without clipped MCUs (every segment of a picture whose sides are multiples of the MCU,
    // most segments otherwise) runs a copy of the loop that has no clamped-coordinate path at all:
    // that path's address arithmetic is hoisted in front of the loop by the compiler, and its mere
    // presence turns the pixel loads into branches.
    auto convert = [&](auto interior_tag) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
    bool tables_staged = false;
    // interior copy: the strip's row pointer advances by a uniform step (no 64-bit multiply per row)
    constexpr int kBpp = (SRC == kSrcRgb24) ? 3 : 4;
    const long long rs = a.row_stride[0];
    const uint8_t* prow = frame_px + static_cast<long long>(mb_y * PX + yp0 * kRowsPerStrip) * rs + kBpp * x0;
    const long long pstep = static_cast<long long>(ngroups * kRowsPerStrip) * rs;      // uniform
    for (int ypb = yp0; ypb < 8 && yp0 < ngroups; ypb += kBatch * ngroups) {
    // all global loads of the batch are issued before the first one is consumed
    uint32_t raw[kBatch][kRowsPerStrip][kNW];
    if (SRC != kSrcPlanes) {
#pragma unroll
      for (int it = 0; it < kBatch; ++it) {
        const int yp = ypb + it * ngroups;
        if (yp < 8) {
          const int y0 = mb_y * PX + yp * kRowsPerStrip;
          if (INTERIOR) {
#pragma unroll
            for (int r = 0; r < kRowsPerStrip; ++r) __builtin_memcpy(raw[it][r], prow + r * rs, kNW * 4);
          } else {
            const bool inside = (x0 + 8 <= a.W) && (y0 + kRowsPerStrip <= a.H);
#pragma unroll
            for (int r = 0; r < kRowsPerStrip; ++r) load_px8<SRC>(a, frame_px, x0, y0 + r, inside, raw[it][r]);
          }
        }
        prow += pstep;
      }
    }
    if (!tables_staged) { stage_tables(); tables_staged = true; }
#pragma unroll
    for (int it = 0; it < kBatch; ++it) {
      const int yp = ypb + it * ngroups;
      if (yp >= 8) break;
      const int y0 = mb_y * PX + yp * kRowsPerStrip;
      if (SRC == kSrcPlanes) {
        // 8-bit planes are used as they are, minus 128 (src/encoders.cc:256-490)
        int ya[8];
        fetch_plane(frame_px, a.row_stride[0], 1, a.W, a.H, x0, y0, 8, ya);
        if (MODE == SJPEG_HIP_YUV420) {
          int yb[8], U[8], V[8];
          fetch_plane(frame_px, a.row_stride[0], 1, a.W, a.H, x0, y0 + 1, 8, yb);
          const int cw = (a.W + 1) >> 1, ch = (a.H + 1) >> 1;
          const uint8_t* pu = a.plane[1] + frame * a.frame_stride[1] + a.uoff;
          const uint8_t* pv = a.plane[2] + frame * a.frame_stride[2] + a.voff;
          fetch_plane(pu, a.row_stride[1], a.cstep, cw, ch, mb_x * 8 + xs * 4, mb_y * 8 + yp, 4, U);
          fetch_plane(pv, a.row_stride[2], a.cstep, cw, ch, mb_x * 8 + xs * 4, mb_y * 8 + yp, 4, V);
          const int k = (yp >> 2) * 2 + xs;
          const int row = (yp & 3) * 2;
          unsigned char* ys = smem + (ml * BPM + k) * kSlotBytes + row * 16;
          *reinterpret_cast<uint4*>(ys) =
              make_uint4(pack16(ya[0], ya[1]), pack16(ya[3], ya[2]), pack16(ya[4], ya[5]), pack16(ya[7], ya[6]));
          *reinterpret_cast<uint4*>(ys + 16) =
              make_uint4(pack16(yb[0], yb[1]), pack16(yb[3], yb[2]), pack16(yb[4], yb[5]), pack16(yb[7], yb[6]));
          unsigned char* us = smem + (ml * BPM + 4) * kSlotBytes + yp * 16 + xs * 8;
          *reinterpret_cast<uint2*>(us) = make_uint2(pack16(U[0], U[1]), pack16(U[3], U[2]));
          *reinterpret_cast<uint2*>(us + kSlotBytes) = make_uint2(pack16(V[0], V[1]), pack16(V[3], V[2]));
        } else {
          unsigned char* ys = smem + (ml * BPM) * kSlotBytes + yp * 16;
          *reinterpret_cast<uint4*>(ys) =
              make_uint4(pack16(ya[0], ya[1]), pack16(ya[3], ya[2]), pack16(ya[4], ya[5]), pack16(ya[7], ya[6]));
          if (MODE == SJPEG_HIP_YUV444) {
            int uv[8], vv[8];
            fetch_plane(a.plane[1] + frame * a.frame_stride[1], a.row_stride[1], 1, a.W, a.H, x0, y0, 8, uv);
            fetch_plane(a.plane[2] + frame * a.frame_stride[2], a.row_stride[2], 1, a.W, a.H, x0, y0, 8, vv);
            *reinterpret_cast<uint4*>(ys + kSlotBytes) =
                make_uint4(pack16(uv[0], uv[1]), pack16(uv[3], uv[2]), pack16(uv[4], uv[5]), pack16(uv[7], uv[6]));
            *reinterpret_cast<uint4*>(ys + 2 * kSlotBytes) =
                make_uint4(pack16(vv[0], vv[1]), pack16(vv[3], vv[2]), pack16(vv[4], vv[5]), pack16(vv[7], vv[6]));
          }
        }
        continue;
      }
      uint32_t rg0[8], bb0[4];
      unpack_px8<SRC>(a, raw[it][0], rg0, bb0);
      if (MODE == SJPEG_HIP_YUV420) {
        uint32_t rg1[8], bb1[4];
        unpack_px8<SRC>(a, raw[it][kRowsPerStrip - 1], rg1, bb1);
        uint32_t ya[4], yb[4], us32[4], vs32[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // (slot order: the second and fourth pair of a row are stored with their halves exchanged)
          ya[c] = luma_pair(rg0[2 * c], rg0[2 * c + 1], bb0[c], k7471, kLumaRound, c & 1);
          yb[c] = luma_pair(rg1[2 * c], rg1[2 * c + 1], bb1[c], k7471, kLumaRound, c & 1);
          // 2x2 sums: halves stay below 1021, plain 32-bit adds never carry across; the two blue
          // halves of BB are added by the dot products themselves
          const uint32_t RG = (rg0[2 * c] + rg0[2 * c + 1]) + (rg1[2 * c] + rg1[2 * c + 1]);
          const uint32_t BB = bb0[c] + bb1[c];
          us32[c] = cb_sum<1, 1>(RG, BB, 32768u << 2);
          vs32[c] = cr_sum<1, 1>(RG, BB, k32768, 32768u << 2);
        }
        // (sum >> 16) >> 2 == sum >> 18 (floor of floor)
        const s16x2 two = pk_const(2, 2);
        const uint32_t u01 = as_u32(as_pk(pk_top(us32[0], us32[1])) >> two);
        const uint32_t u23 = as_u32(as_pk(pk_top(us32[3], us32[2])) >> two);
        const uint32_t v01 = as_u32(as_pk(pk_top(vs32[0], vs32[1])) >> two);
        const uint32_t v23 = as_u32(as_pk(pk_top(vs32[3], vs32[2])) >> two);
        const int k = (yp >> 2) * 2 + xs;
        const int row = (yp & 3) * 2;
        unsigned char* ys = smem + (ml * BPM + k) * kSlotBytes + row * 16;
        *reinterpret_cast<uint4*>(ys) = make_uint4(ya[0], ya[1], ya[2], ya[3]);
        *reinterpret_cast<uint4*>(ys + 16) = make_uint4(yb[0], yb[1], yb[2], yb[3]);
        unsigned char* us = smem + (ml * BPM + 4) * kSlotBytes + yp * 16 + xs * 8;
        *reinterpret_cast<uint2*>(us) = make_uint2(u01, u23);
        *reinterpret_cast<uint2*>(us + kSlotBytes) = make_uint2(v01, v23);
      } else {
        uint32_t yv[4], uv[4], vv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          yv[c] = luma_pair(rg0[2 * c], rg0[2 * c + 1], bb0[c], k7471, kLumaRound, c & 1);
          if (MODE == SJPEG_HIP_YUV444) {
            const uint32_t u0 = cb_sum<1, 0>(rg0[2 * c], bb0[c], 32768u), u1 = cb_sum<0, 1>(rg0[2 * c + 1], bb0[c], 32768u);
            const uint32_t v0 = cr_sum<1, 0>(rg0[2 * c], bb0[c], k32768, 32768u);
            const uint32_t v1 = cr_sum<0, 1>(rg0[2 * c + 1], bb0[c], k32768, 32768u);
            uv[c] = (c & 1) ? pk_top(u1, u0) : pk_top(u0, u1);
            vv[c] = (c & 1) ? pk_top(v1, v0) : pk_top(v0, v1);
          }
        }
        unsigned char* ys = smem + (ml * BPM) * kSlotBytes + yp * 16;
        *reinterpret_cast<uint4*>(ys) = make_uint4(yv[0], yv[1], yv[2], yv[3]);
        if (MODE == SJPEG_HIP_YUV444) {
          *reinterpret_cast<uint4*>(ys + kSlotBytes) = make_uint4(uv[0], uv[1], uv[2], uv[3]);
          *reinterpret_cast<uint4*>(ys + 2 * kSlotBytes) = make_uint4(vv[0], vv[1], vv[2], vv[3]);
        }
      }
    }
    }
    if (!tables_staged) stage_tables();
    };
    if (interior) convert(std::integral_constant<bool, true>()); else convert(std::integral_constant<bool, false>());
  }
  __syncthreads();
  stamp(1);
  PHASE_PRIO(1);
  RACE_POINT(1);
  if (a.ablate == 1) return;

  // ---- P2: one thread per block: fix-up, fDCT, quantize ---------------------------------
  const int ml = tid / BPM;                    // local MCU (0 = halo)
  const int k = tid - ml * BPM;                // block inside the MCU
  const bool has_block = (ml <= n_coded) && (ml >= 1 || halo);
  const bool emits = (ml >= 1) && (ml <= n_coded);
  const int tbl = (MODE == SJPEG_HIP_YUV420) ? (k >= 4) : (MODE == SJPEG_HIP_YUV444 ? (k >= 1) : 0);
  unsigned char* const slot = smem + tid * kSlotBytes;
  uint32_t nzq[4] = {0, 0, 0, 0};                   // non-zero masks of the four zig-zag quarters
  int dc_val = 0;
  // some AC level of the block has more bits than the lean walk (P3) is provably in place for
  uint32_t unsafe = 0;
  // the OR of the block's AC entries: what the test is made on -- kept with a replayed block, because
  // the bound depends on the AC table of the pass that CODES it, not of the pass that quantized it
  uint32_t any_ac = 0;
  // what a statistics pass keeps for the replay kind: the slot as P2 leaves it + masks + DC value
  // (row-major over the workgroup -- row r of thread t at [r * 256 + t] --: one store or load instruction of a
  // wave covers 1 KiB in one piece.  Thread-major (144 B apart) made every one of them touch 64 cache lines.)
  uint4* const keep = (a.replay == nullptr) ? nullptr
      : reinterpret_cast<uint4*>(a.replay) + (static_cast<size_t>(frame) * a.nseg + seg) * kScanThreads * 9 + tid;
  constexpr int kKeepRow = kScanThreads;
  // (SJPEG_KEEP_NT: the kept blocks / coefficients are streamed -- written once, read once by a later launch)
#ifndef SJPEG_KEEP_NT
#define SJPEG_KEEP_NT 3
#endif
  typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
  auto keep_store = [&](int idx, uint4 v) {
    if (SJPEG_KEEP_NT & 1) { const u32x4_nt x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<u32x4_nt*>(keep + idx)); }
    else keep[idx] = v;
  };
  auto keep_load = [&](int idx) -> uint4 {
    if (SJPEG_KEEP_NT & 2) { const u32x4_nt x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(keep + idx)); return make_uint4(x.x, x.y, x.z, x.w); }
    return keep[idx];
  };
  // A kept block is 64 + 16 bytes when no AC level of it exceeds 127 (every ordinary block): its entries as BYTES --
  // sign in bit 7, level in bits 0..6 --, rows 0..3, and the tail (masks, DC value, OR of the AC entries) in row 4.
  // A block with a larger level keeps the low bytes there and the high bytes (sign, level bits 8..14) in rows 5..8.
  // Entry 0, the quantized DC, is not kept in either: the tail has the value.  (16-bit entries, 144 bytes, until
  // round 4: the statistics pass wrote more than the pixels it read, and the replay pass read it all back.)
  constexpr uint32_t kWideLevels = 0x7f807f80u;     // (of the OR of the entries: some level is 128 or more)
  if (REPLAY) {
    uint4 lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] = keep_load(j * kKeepRow);
    const uint4 t = keep_load(4 * kKeepRow);
    nzq[0] = t.x & 0xffffu; nzq[1] = t.x >> 16; nzq[2] = t.y & 0xffffu; nzq[3] = t.y >> 16;
    dc_val = static_cast<int>(t.z);
    any_ac = t.w;
    const bool wide = (any_ac & kWideLevels) != 0u;
    if (wide) {
#pragma unroll
      for (int j = 0; j < 4; ++j) hi[j] = keep_load((5 + j) * kKeepRow);
    }
    if (has_slot) {
      const uint32_t dm = static_cast<uint32_t>(dc_val < 0 ? -dc_val : dc_val);
      const uint32_t dc_entry = dm | (dc_val < 0 ? 0x8000u : 0u);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t b[4] = {lo[j].x, lo[j].y, lo[j].z, lo[j].w};
        uint32_t e[8];
        if (!wide) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // bytes -> 16-bit entries: (b0, 0, b1, 0), then bit 7 of each half moves to bit 15: x + (x & 0x80) * 255
            const uint32_t x0 = __builtin_amdgcn_perm(0u, b[k], 0x0c010c00u), x1 = __builtin_amdgcn_perm(0u, b[k], 0x0c030c02u);
            e[2 * k] = __umul24(x0 & 0x00800080u, 255u) + x0;
            e[2 * k + 1] = __umul24(x1 & 0x00800080u, 255u) + x1;
          }
        } else {
          const uint32_t h[4] = {hi[j].x, hi[j].y, hi[j].z, hi[j].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            e[2 * k] = __builtin_amdgcn_perm(h[k], b[k], 0x05010400u);
            e[2 * k + 1] = __builtin_amdgcn_perm(h[k], b[k], 0x07030602u);
          }
        }
        if (j == 0) e[0] = (e[0] & 0xffff0000u) | dc_entry;
        *reinterpret_cast<uint4*>(slot + 32 * j) = make_uint4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<uint4*>(slot + 32 * j + 16) = make_uint4(e[4], e[5], e[6], e[7]);
      }
    }
  }
  if (!REPLAY) {
  // rows as packed int16 pairs, straight from the slot, in slot order: (s0,s1) (s3,s2) (s4,s5) (s7,s6)
  // (COEF: the coefficients themselves, natural order, as the histogram kind below leaves them)
  uint32_t p[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint4 q = make_uint4(0, 0, 0, 0);
    if (has_block) q = COEF ? keep_load(r * kKeepRow) : *reinterpret_cast<const uint4*>(slot + 16 * r);
    p[r][0] = q.x; p[r][1] = q.y; p[r][2] = q.z; p[r][3] = q.w;
  }

  if (MODE == SJPEG_HIP_YUV420 && !interior && !COEF) {
    // AverageExtraLuma (src/encoders.cc:107-125): luma blocks wholly outside the picture
    // become flat at (sum + 32) >> 6 of a neighbouring real block.
    int sum = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) sum = dot2(as_pk(p[r][c]), 1, 1, sum);
    }
    if (has_slot) reinterpret_cast<int*>(slot + 128)[3] = sum;
    __syncthreads();
    RACE_POINT(12);
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
        const int flat = (reinterpret_cast<const int*>(smem + (ml * BPM + src) * kSlotBytes + 128)[3] + 32) >> 6;
        const uint32_t ff = pack16(flat, flat);
#pragma unroll
        for (int r = 0; r < 8; ++r) { p[r][0] = ff; p[r][1] = ff; p[r][2] = ff; p[r][3] = ff; }
      }
    }
    __syncthreads();
    RACE_POINT(13);
  }

  // forward DCT: two columns per op, then row by row fused with quantization
  if (!COEF) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fdct_col8_pk(p[0][c], p[1][c], p[2][c], p[3][c], p[4][c], p[5][c], p[6][c], p[7][c]);
    }
  }
  if (KIND == kKindError) {
    // Quantization error of the picture (reference QuantizeError, src/quantize.cc:553-565):
    // sum of ((|c| >> 4) - quant * level)^2, per block in 32-bit wrap-around, 64 bits overall.
    const uint4* qt = lq + tbl * 32;
    uint32_t err = 0;
    int acc[8];
    auto add_row = [&](int row, const int* ac8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = ac8[i] >> 16;
        const uint32_t mag = static_cast<uint32_t>(c < 0 ? -c : c);
        const uint4 t = qt[row * 4 + (i >> 1)];
        const uint32_t iq = (i & 1) ? (t.x >> 16) : (t.x & 0xffffu);
        const uint32_t biq = (i & 1) ? t.z : t.y;
        const uint32_t qv = (i & 1) ? (t.w >> 16) : (t.w & 0xffffu);
        const uint32_t v = qv * ((mag * iq + biq) >> 20);
        const uint32_t d = (mag >> 4) - v;
        err += d * d;
      }
    };
    fdct_row8_pk<22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[0], acc); add_row(0, acc);
    fdct_row8_pk<31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[1], acc); add_row(1, acc);
    fdct_row8_pk<29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[2], acc); add_row(2, acc);
    fdct_row8_pk<26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[3], acc); add_row(3, acc);
    fdct_row8_pk<22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[4], acc); add_row(4, acc);
    fdct_row8_pk<26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[5], acc); add_row(5, acc);
    fdct_row8_pk<29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[6], acc); add_row(6, acc);
    fdct_row8_pk<31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[7], acc); add_row(7, acc);
    unsigned long long e64 = emits ? err : 0u;
    for (int d = 32; d > 0; d >>= 1) e64 += __shfl_xor(e64, d, 64);
    unsigned long long* const we = reinterpret_cast<unsigned long long*>(misc);
    if ((tid & 63) == 0) we[tid >> 6] = e64;
    RACE_POINT(19);
    __syncthreads();
    if (tid == 0) {
      unsigned long long sum = 0;
      for (int w = 0; w < kScanThreads / 64; ++w) sum += we[w];
      reinterpret_cast<unsigned long long*>(a.partial)[static_cast<size_t>(frame) * a.nseg + seg] = sum;
    }
    return;
  }
  if (KIND == kKindHisto) {
    // Adaptive-quantization statistics (reference StoreHisto, src/histogram.cc:56-108): for every natural
    // position, histogram of |coefficient| >> 2 (bins < 128), one histogram per quantizer table.
    //
    // Binning is done TRANSPOSED.  With one thread per block every lane of a wave bumps the SAME position at the
    // same time, and the bins of one position crowd into a few words: an LDS atomic serialises lanes that meet in
    // a word (round 4: 47 % of the kernel LDS-busy, two thirds of it conflicts, eight replicas of the lowest word
    // notwithstanding).  Instead a thread stages its coefficients in LDS, half a block at a time (64 bytes + a
    // tail word: the byte offset of its table's histogram), and the wave reads them back with 16 lanes to a
    // block -- lane q takes the pair of coefficients 2q, 2q + 1 of the half -- four blocks per step: lanes of a
    // wave bump 32 different positions, only the four that share a position can meet.  129 words per group of
    // four positions: consecutive groups start in consecutive banks.  Blocks that are not coded stage 0x7fff: bin
    // "128 and above", the word nobody reads.  8-bit counters, four to a word (a segment has at most 252 blocks); after the
    // segment every thread adds sixteen words of them to its 16-bit counters in registers, and the workgroup goes on
    // to its next segment: ONE partial per workgroup (32 KB) instead of one per segment (16 KB: 13 MB a 4K frame,
    // written and read back by the summing kernel).
    RACE_POINT(14);
    __syncthreads();                            // every thread holds its samples: the slots are free
    RACE_POINT(15);
    {
      uint4* const hz = reinterpret_cast<uint4*>(smem + kHistoOffBins);
      for (int i = tid; i < 2 * kHistoTblBytes / 16; i += kScanThreads) hz[i] = make_uint4(0, 0, 0, 0);
    }
    unsigned char* const stage = smem + tid * kHistoStageStride;
    *reinterpret_cast<uint32_t*>(stage + 64) = static_cast<uint32_t>(tbl * kHistoTblBytes);
    int acc[8];
    const bool wave_emits = __builtin_amdgcn_ballot_w64(emits) == ~0ull;
    auto stage_row = [&](int row, const int* ac8) {
      uint32_t cq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cq[k] = __builtin_amdgcn_perm(static_cast<uint32_t>(ac8[2 * k + 1]), static_cast<uint32_t>(ac8[2 * k]), 0x07060302u);
      // the coefficients stay behind for the statistics pass of the same call (kKindStatsCoef)
      if (keep != nullptr) keep_store(row * kKeepRow, make_uint4(cq[0], cq[1], cq[2], cq[3]));
      // (a uniform branch: the waves whose blocks are all coded -- two of four in an ordinary segment -- pay no select)
      if (!wave_emits) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cq[k] = emits ? cq[k] : 0x7fff7fffu;
      }
      *reinterpret_cast<uint4*>(stage + (row & 3) * 16) = make_uint4(cq[0], cq[1], cq[2], cq[3]);
    };
    auto bin_half = [&](auto half_tag) {
      constexpr int HALF = decltype(half_tag)::value;
      const int q = tid & 15;
      const unsigned char* const rd = smem + (tid >> 4) * kHistoStageStride;
      const uint32_t qb = static_cast<uint32_t>(kHistoOffBins + q * kHistoPosBytes);
      // (the bytes of a word: positions 2q, 2q + 1 of the first half, then of the second)
      const uint32_t one0 = 1u << (16 * HALF), one1 = 1u << (16 * HALF + 8);
      // (four steps' reads are in flight before the first is consumed)
#pragma unroll
      for (int s4 = 0; s4 < 16; s4 += 4) {
        uint32_t d[4], tb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[i] = *reinterpret_cast<const uint32_t*>(rd + (s4 + i) * 16 * kHistoStageStride + q * 4);
          tb[i] = *reinterpret_cast<const uint32_t*>(rd + (s4 + i) * 16 * kHistoStageStride + 64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // both coefficients of the pair at once: (|c| >> 2) * 4 = the byte offset of the bin's word, bins from 128 on
          // at 512 (packed 16-bit operations)
          const s16x2 c = as_pk(d[i]);
          const uint32_t mag = as_u32(__builtin_elementwise_max(c, pk_const(0, 0) - c)) & 0xfffcfffcu;
          uint32_t off;
          asm("v_pk_min_u16 %0, %1, %2" : "=v"(off) : "v"(mag), "v"(0x02000200u));
          const uint32_t base = tb[i] + qb;
          atomicAdd(reinterpret_cast<uint32_t*>(smem + base + (off & 0xffffu)), one0);
          atomicAdd(reinterpret_cast<uint32_t*>(smem + base + (off >> 16)), one1);
        }
      }
    };
    fdct_row8_pk<22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[0], acc); stage_row(0, acc);
    fdct_row8_pk<31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[1], acc); stage_row(1, acc);
    fdct_row8_pk<29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[2], acc); stage_row(2, acc);
    fdct_row8_pk<26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[3], acc); stage_row(3, acc);
    RACE_POINT(16);
    __syncthreads();
    stamp(2);
    bin_half(std::integral_constant<int, 0>());
    RACE_POINT(17);
    __syncthreads();                            // the first halves are read: the second may take their place
    stamp(3);
    fdct_row8_pk<22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[4], acc); stage_row(4, acc);
    fdct_row8_pk<26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[5], acc); stage_row(5, acc);
    fdct_row8_pk<29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[6], acc); stage_row(6, acc);
    fdct_row8_pk<31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[7], acc); stage_row(7, acc);
    RACE_POINT(18);
    __syncthreads();
    stamp(4);
    bin_half(std::integral_constant<int, 1>());
    RACE_POINT(22);
    __syncthreads();
    stamp(5);
    {
      // words 16 * tid .. 16 * tid + 15 of the [2][16][128] words: group tid >> 3, bins 16 * (tid & 7) ..
      const uint32_t* const fw = reinterpret_cast<const uint32_t*>(smem + kHistoOffBins) + (tid >> 3) * 129 + (tid & 7) * 16;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const uint32_t w = fw[m];
        hacc[2 * m] += w & 0x00ff00ffu;
        hacc[2 * m + 1] += __builtin_amdgcn_perm(0u, w, 0x0c030c01u);        // (w >> 8) & 0x00ff00ff
      }
    }
    stamp(6);
    if (seg + static_cast<int>(gridDim.x) >= a.nseg) {
      // this workgroup's partial, [8][256] pieces of 16 bytes: a store instruction of a wave covers 1 KiB in one piece
      uint4* const dst = reinterpret_cast<uint4*>(a.partial) + (static_cast<size_t>(frame) * gridDim.x + blockIdx.x) * (kHistoPartialWords / 4) + tid;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j * kScanThreads] = make_uint4(hacc[4 * j], hacc[4 * j + 1], hacc[4 * j + 2], hacc[4 * j + 3]);
      return;
    }
    RACE_POINT(23);
    __syncthreads();                            // the counters are folded: the next segment's samples may take the slots
    stamp(7);
    continue;
  }
  uint32_t ent[32];                             // natural order, 2 entries per dword
  {
    const uint4* qt = lq + tbl * 32;
    // cos(k*pi/16)/sqrt(2) tables, rows 1/7, 2/6, 3/5 pre-scaled (src/fdct.cc:28-35,599-606)
    if (COEF) {
      quant_row<0>(p[0], qt, ent + 0, nzq); quant_row<1>(p[1], qt, ent + 4, nzq);
      quant_row<2>(p[2], qt, ent + 8, nzq); quant_row<3>(p[3], qt, ent + 12, nzq);
      quant_row<4>(p[4], qt, ent + 16, nzq); quant_row<5>(p[5], qt, ent + 20, nzq);
      quant_row<6>(p[6], qt, ent + 24, nzq); quant_row<7>(p[7], qt, ent + 28, nzq);
    } else if (!TRELLIS) {
      row_quant<0, 22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[0], qt, ent + 0, nzq);
      row_quant<1, 31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[1], qt, ent + 4, nzq);
      row_quant<2, 29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[2], qt, ent + 8, nzq);
      row_quant<3, 26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[3], qt, ent + 12, nzq);
      row_quant<4, 22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[4], qt, ent + 16, nzq);
      row_quant<5, 26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[5], qt, ent + 20, nzq);
      row_quant<6, 29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[6], qt, ent + 24, nzq);
      row_quant<7, 31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[7], qt, ent + 28, nzq);
    } else {
      row_raw<22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[0], ent + 0);
      row_raw<31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[1], ent + 4);
      row_raw<29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[2], ent + 8);
      row_raw<26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[3], ent + 12);
      row_raw<22725, 21407, 19266, 16384, 12873, 8867, 4520>(p[4], ent + 16);
      row_raw<26722, 25172, 22654, 19266, 15137, 10426, 5315>(p[5], ent + 20);
      row_raw<29692, 27969, 25172, 21407, 16819, 11585, 5906>(p[6], ent + 24);
      row_raw<31521, 29692, 26722, 22725, 17855, 12299, 6270>(p[7], ent + 28);
    }
  }
  // zig-zag reorder with byte permutes, 4 entries per ds_write_b64
  uint32_t zz[DIRECT ? 32 : 1];                    // DIRECT: the zig-zag order in registers, for the kept block only
  if (DIRECT) {                                    // (only the DC entry goes to the slot: the next block's predictor)
    if (has_slot) *reinterpret_cast<u16_may_alias*>(slot) = static_cast<uint16_t>(ent[0] & 0xffffu);
    if (keep != nullptr) {
#pragma unroll
      for (int i = 0; i < 64; i += 2) {
        zz[i >> 1] = __builtin_amdgcn_perm(ent[kZig(i + 1) >> 1], ent[kZig(i) >> 1], kPairSel(kZig(i), kZig(i + 1)));
      }
    }
  } else if (has_slot) {
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
      const uint32_t w0 = __builtin_amdgcn_perm(ent[kZig(i + 1) >> 1], ent[kZig(i) >> 1],
                                                kPairSel(kZig(i), kZig(i + 1)));
      const uint32_t w1 = __builtin_amdgcn_perm(ent[kZig(i + 3) >> 1], ent[kZig(i + 2) >> 1],
                                                kPairSel(kZig(i + 2), kZig(i + 3)));
      *reinterpret_cast<uint2*>(slot + 2 * i) = make_uint2(w0, w1);
    }
  }
  if (!TRELLIS) {
    const int dc_mag = static_cast<int>(ent[0] & 0x7fffu);
    dc_val = (ent[0] & 0x8000u) ? -dc_mag : dc_mag;
    if (KIND == kKindEncode || KIND == kKindStats) {
      any_ac = ent[0] & 0xffff0000u;               // (the DC entry is not an AC level)
#pragma unroll
      for (int i = 1; i < 32; ++i) any_ac |= ent[i];
    }
  } else {
    // Trellis quantization (reference Encoder::TrellisQuantizeBlock + SearchBestPrev,
    // src/quantize.cc:325-457): the slot holds the RAW coefficients in zig-zag order.  For every
    // coefficient that does not quantize to zero, two candidate levels become nodes of a graph;
    // an edge costs distortion + lambda * bits (bits priced with the AC code lengths in `tl`).
    typedef int16_t __attribute__((may_alias)) i16_alias;
    typedef uint16_t __attribute__((may_alias)) u16_alias2;
    const i16_alias* const raw = reinterpret_cast<const i16_alias*>(slot);
    const uint4* const qt = lq + tbl * 32;
    const uint8_t* const tl = smem + L::kOffTlen + tbl * 256;
    {
      const int d = raw[0];
      const uint4 t0 = qt[0];
      const uint32_t ad = static_cast<uint32_t>(d < 0 ? -d : d);
      const int lv = static_cast<int>((ad * (t0.x & 0xffffu) + t0.y) >> 20);
      dc_val = d < 0 ? -lv : lv;
    }
    unsigned long long nzm = 0;
    if (emits) {
      // One thread per block.  What the walk towards the sink reads of a node is ONE 16-byte record -- score, info
      // (level | neg << 11 | pos << 12 | prev << 25), disto0 at its position -- asked for one candidate ahead: the three
      // dependent private-memory reads per candidate of rounds 3-5 (info -> disto0[pos] -> score) were the kernel's time.
      // The best entry point is kept as the nodes are made (the reference searches it backwards afterwards with a
      // strict <: a later node wins a tie, and only a score below kMaxScore can win) -- for which disto0[63] is summed
      // up front; the next position's coefficient and quantizer entry are asked for before this one's nodes are walked.
      constexpr int kNodes = 1 + 2 * 63;
      uint4 node[kNodes];
      node[0] = make_uint4(0u, 0u, 0u, 0u);
#ifndef SJPEG_TRELLIS_RING
#define SJPEG_TRELLIS_RING 8
#endif
      constexpr int kRing = SJPEG_TRELLIS_RING;
      uint4 ring[kRing];                           // [0] = the newest node
#pragma unroll
      for (int r = 0; r < kRing; ++r) ring[r] = make_uint4(0u, 0u, 0u, 0u);
      uint32_t total = 0;                          // disto0[63] (32-bit arithmetic that may wrap, like the reference's)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint4 w4 = *reinterpret_cast<const uint4*>(slot + 16 * r);
        const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int lo = static_cast<int>(ws[u] << 16) >> 16, hi = static_cast<int>(ws[u]) >> 16;
          if (r != 0 || u != 0) total += static_cast<uint32_t>(__mul24(lo, lo));
          total += static_cast<uint32_t>(__mul24(hi, hi));
        }
      }
      int count = 1;                               // node 0 = the sink
      int best = 0;
      uint32_t best_sc = total;                    // (the sink's: nothing coded, everything distortion)
      uint32_t dprev = 0;                          // disto0[i - 1]
      const uint32_t zrl_len = tl[0xf0];
      const uint2* const tq = reinterpret_cast<const uint2*>(smem + L::kOffTq) + tbl * 64;
      int rv_n = raw[1];
      uint2 t_n = tq[1];
      for (int i = 1; i < 64; ++i) {
        const int rv = rv_n;
        const uint2 t = t_n;
        if (i < 63) { rv_n = raw[i + 1]; t_n = tq[i + 1]; }
        const uint32_t iq = t.x & 0xffffu, biq = t.y, qq = (t.x >> 16) << 4;
        const uint32_t lambda = __umul24(qq, qq) / 32u;              // (every product here has factors below 2^24: full-rate multiplies)
        const uint32_t neg = rv < 0 ? 1u : 0u;
        const int V = rv < 0 ? -rv : rv;
        const uint32_t dhere = static_cast<uint32_t>(__mul24(V, V)) + dprev;
        int v = static_cast<int>((__umul24(static_cast<uint32_t>(V), iq) + biq) >> 20);
        if (v != 0) {
          int nbits = 32 - __clz(v);
          for (int kk = 0; kk < 2; ++kk) {
            const int err = V - __mul24(v, static_cast<int>(qq));
            uint32_t my_score = 0xffffffffu, my_prev = 0;
            bool found = false;
            const uint32_t base_disto = static_cast<uint32_t>(__mul24(err, err)) + dprev;
            // (the newest nodes are in registers: the walk mostly ends there, and every private-memory read is a trip
            // to HBM -- three workgroups' node arrays are no L2's size -- that a wave waits ~1 000 cycles for)
            // Pricing one node is PREDICATED, not branched: a lane whose walk has stopped (or has no such node) goes
            // through the same instructions and keeps its values -- the nested ifs of the first form cost as many scalar
            // instructions (exec masks saved, combined, restored) as vector ones; one uniform test per node ends the
            // walk when no lane is at it any more.
            bool walking = true;
            auto price = [&](const uint4& cur, int c) {
              const int run = i - 1 - static_cast<int>((cur.y >> 12) & 63u);
              const bool here = walking && c >= 0 && run >= 0;           // (run < 0: the other node of this position -- skipped, the walk goes on)
              const uint32_t bits0 = static_cast<uint32_t>(nbits) + __umul24(static_cast<uint32_t>(run >> 4) & 3u, zrl_len);
              const uint32_t bound = base_disto - cur.z + __umul24(lambda, bits0);
              const uint32_t len = tl[((run & 15) << 4) | nbits];
              const bool stop = here && bound >= my_score;
              const uint32_t score = bound + __umul24(lambda, len) + cur.x;
              const bool better = here && !stop && score < my_score;
              my_score = better ? score : my_score;
              my_prev = better ? static_cast<uint32_t>(c) : my_prev;
              found = found || better;
              walking = walking && !stop && c > 0;
            };
#pragma unroll
            for (int r = 0; r < kRing; ++r) {
              if (__builtin_amdgcn_ballot_w64(walking && count - 1 - r >= 0) == 0ull) break;
              price(ring[r], count - 1 - r);
            }
            walking = walking && count > kRing;
            if (__builtin_amdgcn_ballot_w64(walking) != 0ull) {
              // the older nodes: four records in flight (the trips to private memory are independent of each other)
              int c = count - 1 - kRing;                                // (per lane; a lane that is not walking reads node 0)
              auto at = [&](int k) { const int x = walking ? c - k : 0; return node[x > 0 ? x : 0]; };
              uint4 f0 = at(0), f1 = at(1), f2 = at(2), f3 = at(3);
              for (;;) {
                price(f0, c); if (__builtin_amdgcn_ballot_w64(walking) == 0ull) break;
                --c; f0 = at(3);
                price(f1, c); if (__builtin_amdgcn_ballot_w64(walking) == 0ull) break;
                --c; f1 = at(3);
                price(f2, c); if (__builtin_amdgcn_ballot_w64(walking) == 0ull) break;
                --c; f2 = at(3);
                price(f3, c); if (__builtin_amdgcn_ballot_w64(walking) == 0ull) break;
                --c; f3 = at(3);
              }
            }
            if (found) {
#pragma unroll
              for (int r = kRing - 1; r > 0; --r) ring[r] = ring[r - 1];
              ring[0] = make_uint4(my_score, static_cast<uint32_t>(v) | (neg << 11) | (static_cast<uint32_t>(i) << 12) | (my_prev << 25), dhere, 0u);
              node[count] = ring[0];
              const uint32_t sc = my_score + (total - dhere);
              if (sc <= best_sc && sc != 0xffffffffu) { best = count; best_sc = sc; }
              ++count;
            }
            --nbits;
            if (nbits <= 0) break;
            v = (1 << nbits) - 1;
          }
        }
        dprev = dhere;
      }
      // the slot becomes the usual sign-magnitude entries: zeros but for the chosen chain
#pragma unroll
      for (int r = 0; r < 8; ++r) *reinterpret_cast<uint4*>(slot + 16 * r) = make_uint4(0, 0, 0, 0);
      u16_alias2* const zzw = reinterpret_cast<u16_alias2*>(slot);
      for (int c = best; c > 0;) {
        const uint32_t ci = node[c].y;
        c = static_cast<int>(ci >> 25);
        const uint32_t pos = (ci >> 12) & 63u;
        zzw[pos] = static_cast<uint16_t>((ci & 0x7ffu) | (((ci >> 11) & 1u) << 15));
        nzm |= 1ull << pos;
        any_ac |= ci & 0x7ffu;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) *reinterpret_cast<uint4*>(slot + 16 * r) = make_uint4(0, 0, 0, 0);
    }
    {   // position 0 = the quantized DC, sign-magnitude like every entry: the next block's predictor is read from here
      const int dm = dc_val < 0 ? -dc_val : dc_val;
      *reinterpret_cast<u16_alias2*>(slot) = static_cast<uint16_t>(static_cast<uint32_t>(dm) | (dc_val < 0 ? 0x8000u : 0u));
    }
    nzq[0] = static_cast<uint32_t>(nzm) & 0xffffu; nzq[1] = static_cast<uint32_t>(nzm >> 16) & 0xffffu;
    nzq[2] = static_cast<uint32_t>(nzm >> 32) & 0xffffu; nzq[3] = static_cast<uint32_t>(nzm >> 48);
  }
  nzq[0] &= ~1u;                                // DC is coded separately
  if (KIND == kKindStats && keep != nullptr) {   // leave the quantized block behind for the replay kind
    const bool wide = (any_ac & kWideLevels) != 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 ra, rb;
      if (DIRECT) {
        ra = make_uint4(zz[(8 * j) % (DIRECT ? 32 : 1)], zz[(8 * j + 1) % (DIRECT ? 32 : 1)], zz[(8 * j + 2) % (DIRECT ? 32 : 1)], zz[(8 * j + 3) % (DIRECT ? 32 : 1)]);
        rb = make_uint4(zz[(8 * j + 4) % (DIRECT ? 32 : 1)], zz[(8 * j + 5) % (DIRECT ? 32 : 1)], zz[(8 * j + 6) % (DIRECT ? 32 : 1)], zz[(8 * j + 7) % (DIRECT ? 32 : 1)]);
      } else {
        ra = *reinterpret_cast<const uint4*>(slot + 32 * j); rb = *reinterpret_cast<const uint4*>(slot + 32 * j + 16);
      }
      const uint32_t p[4] = {ra.x, ra.z, rb.x, rb.z}, q[4] = {ra.y, ra.w, rb.y, rb.w};
      uint32_t lb[4], hb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lb[k] = __builtin_amdgcn_perm(q[k], p[k], 0x06040200u);        // low bytes of entries 4k .. 4k + 3 of the row pair
        hb[k] = __builtin_amdgcn_perm(q[k], p[k], 0x07050301u);        // high bytes: sign, level bits 8..14
        if (!wide) lb[k] |= hb[k] & 0x80808080u;
      }
      keep_store(j * kKeepRow, make_uint4(lb[0], lb[1], lb[2], lb[3]));
      if (wide) keep_store((5 + j) * kKeepRow, make_uint4(hb[0], hb[1], hb[2], hb[3]));
    }
    keep_store(4 * kKeepRow, make_uint4(nzq[0] | (nzq[1] << 16), nzq[2] | (nzq[3] << 16), static_cast<uint32_t>(dc_val), any_ac));
  }
  if (DIRECT) {
    // kKindStats, direct: symbol statistics for optimised Huffman tables (reference AddEntropyStats,
    // src/entropy.cc:208-227; the run/size walk of src/entropy.cc:161-198) by the block's own thread, out of the
    // registers its quantizer left the entries in.  The walk of sorted parts (below: the trellis kind still takes it)
    // read every entry back from LDS at a lane's own address and bumped a counter per symbol with all 64 lanes at
    // it: 70 % of the kernel LDS-busy, two thirds of that conflicts.  Here position i of the zig-zag scan is a
    // compile-time constant -- the entry's register and half, whether a run of 16 can end in it (i >= 17), whether
    // the position is dense enough for the hot symbols' own counters (i <= 16) -- and only the lanes with a
    // non-zero level at i execute: a few lanes per atomic instead of a wave, no sort, no list, no loop.
    __syncthreads();                               // the DC entries are in the slots
    RACE_POINT(24);
    if (emits) {
      int pred_dc = 0;
      int prevb;
      if (MODE == SJPEG_HIP_YUV420) prevb = (k == 0) ? tid - 3 : (k <= 3 ? tid - 1 : tid - 6);
      else prevb = tid - BPM;
      if (!(prevb < BPM && !halo)) {
        const uint32_t e = *reinterpret_cast<const u16_may_alias*>(smem + prevb * kSlotBytes);
        const int mag = static_cast<int>(e & 0x7fffu);
        pred_dc = (e & 0x8000u) ? -mag : mag;
      }
      uint32_t* const f = reinterpret_cast<uint32_t*>(smem + L::kOffStats) + ((L::kStatsCopies == 2) ? (tid & 1) * kStatsWords : 0) + tbl * 272;
      uint32_t* const fhot = COMPACT ? reinterpret_cast<uint32_t*>(smem + L::kOffStats + kStatsWords * 4) + (((tid & 63) * 10) >> 6) * 6 + tbl * 3 - 1 : f;
      {
        const int diff = dc_val - pred_dc;
        const int ad = diff < 0 ? -diff : diff;
        atomicAdd(&f[256 + (32 - __clz(ad))], 1u);
      }
      int prev = 1;                                // zig-zag position behind the last non-zero coefficient
      uint32_t zrls = 0;
#pragma unroll
      for (int i = 1; i < 64; ++i) {
        const int j = kZig(i);
        const uint32_t mag = (j & 1) ? ((ent[j >> 1] >> 16) & 0x7fffu) : (ent[j >> 1] & 0x7fffu);
        if (mag != 0u) {
          const int run = i - prev;
          prev = i + 1;
          uint32_t sym;
          if (i >= 17) { zrls += static_cast<uint32_t>(run >> 4); sym = static_cast<uint32_t>((run & 15) << 4); }
          else sym = static_cast<uint32_t>(run << 4);
          sym += 32u - static_cast<uint32_t>(__clz(mag));
          if (COMPACT && i <= 16) atomicAdd(&((sym - 1u < 3u) ? fhot : f)[sym], 1u);
          else atomicAdd(&f[sym], 1u);
        }
      }
      if (zrls != 0u) atomicAdd(&f[0xf0], zrls);
      if (prev <= 63) atomicAdd(&f[0x00], 1u);
    }
    RACE_POINT(25);
    __syncthreads();
    uint32_t* const dst = a.partial + (static_cast<size_t>(frame) * a.nseg + seg) * kStatsWords;
    const uint32_t* const lf_all = reinterpret_cast<const uint32_t*>(smem + L::kOffStats);
    for (int i = tid; i < kStatsWords; i += kScanThreads) {
      uint32_t v = lf_all[i] + (L::kStatsCopies == 2 ? lf_all[kStatsWords + i] : 0u);
      if (COMPACT) {
        const int t = i >= 272 ? 1 : 0, sym = i - t * 272;
        if (sym >= 1 && sym <= 3) {
#pragma unroll
          for (int c = 0; c < 10; ++c) v += lf_all[kStatsWords + c * 6 + t * 3 + sym - 1];
        }
      }
      dst[i] = v;
    }
    return;
  }
  }   // !REPLAY
  const uint32_t nz_lo = nzq[0] | (nzq[1] << 16), nz_hi = nzq[2] | (nzq[3] << 16);
  if (KIND == kKindTap) {
    if (emits) {
      const long long nblk_frame = static_cast<long long>(a.n_mcus) * BPM;
      const long long blk = frame * nblk_frame + static_cast<long long>(m_first - 1 + ml) * BPM + k;
      const uint16_t* src = reinterpret_cast<const uint16_t*>(slot);
      int16_t* dst = a.coeffs + blk * 64;
      for (int i = 0; i < 64; ++i) {
        const int e = src[i], mag = e & 0x7fff;
        dst[i] = static_cast<int16_t>((e & 0x8000) ? -mag : mag);
      }
    }
    return;                                        // the tap ends here: no entropy coding
  }

  if (a.ablate == 2) { if (nz_lo + nz_hi + dc_val == 0x7fffffff) a.seg_nbits[0] = 1; return; }

  stamp(2);
  PHASE_PRIO(2);
#ifdef SJPEG_HIP_PRIO_STRESS
  prio_stress<SJPEG_HIP_PRIO_STRESS>(1);
#endif
  // ---- P3: entropy coding ----------------------------------------------------------------
  // DC prediction (src/entropy.cc:133-150) through an array in the idle bit window; the 16 spare
  // bytes of each slot (its tail) take what the block's parts need to know: masks, DC word, and what
  // the masks say about every quarter (below).
  uint32_t* const tail = reinterpret_cast<uint32_t*>(slot + 128);
  // (a block that is not coded -- the halo MCU, a thread without a block -- makes no part and no length)
  if (!emits) { nzq[0] = 0; nzq[1] = 0; nzq[2] = 0; nzq[3] = 0; }
  RACE_POINT(2);
  // The counting sort of the parts (below) starts here: the bins were cleared when the tables were
  // staged, and the atomics that rank this block's parts are in flight across the DC barrier.
  uint32_t pc[4] = {0, 0, 0, 0}, rank[4] = {0, 0, 0, 0};
  // Two quarters of one half of the scan (0 + 1, 2 + 3) are coded as ONE part when the lean walk can take them in
  // one go: both hold symbols, no more than 16 between them (the sort's bins, the balance of a wave's trip counts),
  // and the run between the last symbol of the first and the first of the second is below 16 -- so that, as in a
  // quarter, only the part's FIRST symbol can need ZRL codes.  The ordinary block of a picture makes one or two
  // parts this way instead of two or three, and a part's start-up, wind-down and placement are what a round costs.
  uint32_t mg01 = 0, mg23 = 0;
  if (KIND == kKindEncode || KIND == kKindStats) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pc[q] = static_cast<uint32_t>(__popc(nzq[q]));
    if (KIND == kKindEncode) {
      // the block takes the checked walk (some AC level has more bits than the lean walk is proven for): no merging
      unsafe = (any_ac & ldc[24 + tbl]) != 0u ? 1u : 0u;
      const uint32_t end0 = 32u - static_cast<uint32_t>(__clz(nzq[0] | 1u));      // position after the last non-zero of quarter 0 (1 = none)
      const uint32_t end2 = 32u - static_cast<uint32_t>(__clz(nzq[2]));           // the same of quarter 2, local (0 = none)
      // (run between the two < 16  <=>  first position of the upper quarter, local, < end of the lower one, local)
#ifdef SJPEG_MUTATE_MERGE                          // (a WRONG rule, to see the directed test fail: tools/build_variant.sh)
      const uint32_t end0m = end0 + 1u;
#else
      const uint32_t end0m = end0;
#endif
      mg01 = (unsafe == 0u && nzq[1] != 0u && static_cast<uint32_t>(__builtin_ctz(nzq[1] | 0x10000u)) < end0m && pc[0] + pc[1] <= 16u) ? 1u : 0u;
      mg23 = (unsafe == 0u && nzq[3] != 0u && static_cast<uint32_t>(__builtin_ctz(nzq[3] | 0x10000u)) < end2 && pc[2] + pc[3] <= 16u) ? 1u : 0u;
      if (SJPEG_NO_MERGE) { mg01 = 0u; mg23 = 0u; }
      if (mg01) { pc[0] += pc[1]; pc[1] = 0u; }
      if (mg23) { pc[2] += pc[3]; pc[3] = 0u; }
    }
    if (emits) {
      uint32_t* const hist0 = reinterpret_cast<uint32_t*>(smem + L::kOffHist);
      rank[0] = atomicAdd(&hist0[pc[0]], 1u);      // quarter 0 always makes a part (DC, EOB)
#pragma unroll
      for (int q = 1; q < 4; ++q) if (pc[q] != 0u) rank[q] = atomicAdd(&hist0[pc[q]], 1u);
    }
  }
  __syncthreads();
  RACE_POINT(3);
  int pred = 0;
  {
    int prev;   // slot holding the previous block of the same component, stream order
    if (MODE == SJPEG_HIP_YUV420) prev = (k == 0) ? tid - 3 : (k <= 3 ? tid - 1 : tid - 6);
    else prev = tid - BPM;
    const bool prev_in_halo = prev < BPM;
    if (emits && !(prev_in_halo && !halo)) {
      // the previous block's quantized DC: entry 0 of its slot (sign-magnitude), untouched until the walks
      const uint32_t e = *reinterpret_cast<const u16_may_alias*>(smem + prev * kSlotBytes);
      const int mag = static_cast<int>(e & 0x7fffu);
      pred = (e & 0x8000u) ? -mag : mag;
    }
  }
  uint32_t dc_word = 0;                            // dc_len << 24 | dc_bits (<= 22); 0 = emits nothing
  if (emits) {
    const int diff = dc_val - pred;
    const int ad = diff < 0 ? -diff : diff;
    const int n = 32 - __clz(ad);                 // 0 for diff == 0 (clz(0) == 32)
    const uint32_t suffix = static_cast<uint32_t>(diff < 0 ? diff - 1 : diff) & ((1u << n) - 1u);
    const uint32_t code = ldc[tbl * 12 + n];
    dc_word = (((code & 0xffu) + n) << 24) | ((code >> 16) << n) | suffix;
  }
  // What a part's walk needs to know of its block, one word per quarter q in the slot's tail: the quarter's
  // non-zero mask (bits 0..15), the run in front of its first symbol, ZRLs included (bits 16..21; the walk splits
  // it into ZRL count and run), bit 22 = "this part carries the EOB" (nothing non-zero above the quarter, and
  // position 63 is zero), bit 24 = checked walk, bit 25 = chroma tables.  Here q is a constant; a part's walk
  // would work the same out of the two 32-bit masks with a dozen selects.  The word is read by the ONE thread
  // that walks the part, which then stores the part's bit length over it: the lengths need no array of their
  // own, and a quarter without a part keeps its mask -- zero -- as its length.  The DC code word
  // (length << 24 | bits) goes to an array by block.
  if (KIND == kKindEncode) {
    const uint32_t p1 = 32u - static_cast<uint32_t>(__clz(nzq[0] | 1u));            // position after the last non-zero below quarter 1 (1 = none)
    const uint32_t p2 = 32u - static_cast<uint32_t>(__clz(nz_lo | 1u));
    const uint32_t p3 = nzq[2] != 0u ? 64u - static_cast<uint32_t>(__clz(nzq[2])) : p2;
    const uint32_t r0 = static_cast<uint32_t>(__builtin_ctz(nzq[0] | 0x10000u)) - 1u;
    const uint32_t r1 = 16u + static_cast<uint32_t>(__builtin_ctz(nzq[1] | 0x10000u)) - p1;
    const uint32_t r2 = 32u + static_cast<uint32_t>(__builtin_ctz(nzq[2] | 0x10000u)) - p2;
    const uint32_t r3 = 48u + static_cast<uint32_t>(__builtin_ctz(nzq[3] | 0x10000u)) - p3;
    const uint32_t e0 = ((nz_lo >> 16) | nz_hi) == 0u ? 0x40u : 0u, e1 = nz_hi == 0u ? 0x40u : 0u;
    const uint32_t e2 = nzq[3] == 0u ? 0x40u : 0u, e3 = (nzq[3] >> 15) == 0u ? 0x40u : 0u;
    const uint32_t fl = (unsafe << 24) | (static_cast<uint32_t>(tbl) << 25);
    if (has_slot) {
      // (a merged part: bit 23 in the word of its first quarter, which takes the EOB flag of the second; the second
      // quarter's word keeps only its mask, in the UPPER half -- read as a length it is zero, like a quarter without
      // a part)
      const uint32_t w0 = nzq[0] | (((r0 & 63u) | (mg01 ? e1 | 0x80u : e0)) << 16) | fl;
      const uint32_t w1 = mg01 ? nzq[1] << 16 : nzq[1] | (((r1 & 63u) | e1) << 16) | fl;
      const uint32_t w2 = nzq[2] | (((r2 & 63u) | (mg23 ? e3 | 0x80u : e2)) << 16) | fl;
      const uint32_t w3 = mg23 ? nzq[3] << 16 : nzq[3] | (((r3 & 63u) | e3) << 16) | fl;
      *reinterpret_cast<uint4*>(tail) = make_uint4(w0, w1, w2, w3);
      dcw[tid] = dc_word;
    }
  } else {
    // (the statistics kind counts a part's symbols out of the two whole masks)
    if (has_slot) *reinterpret_cast<uint4*>(tail) = make_uint4(nz_lo, nz_hi, dc_word | (static_cast<uint32_t>(tbl) << 30), 0u);
  }

  // kKindStats: [2][272] counters, 256 AC then 16 DC, in TWO copies picked by lane parity: the lanes of a
  // wave count the same few symbols most of the time and an LDS atomic serialises the lanes that hit one
  // word; the copies are added up at the flush
  uint32_t* const lf = reinterpret_cast<uint32_t*>(smem + L::kOffStats) + ((KIND == kKindStats && L::kStatsCopies == 2) ? (tid & 1) * kStatsWords : 0);
  // (compact carve only: ten sets of the hot symbols' counters in the 240 idle bytes between the counters and the scan scratch)
  constexpr bool HOT = (KIND == kKindStats) && COMPACT;
  constexpr int kHotCopies = 10;
  static_assert(!COMPACT || (L::kOffStats + kStatsWords * 4 + kHotCopies * 6 * 4 <= L::kOffMisc), "hot counters of the statistics kind");
  uint32_t* const hot = reinterpret_cast<uint32_t*>(smem + L::kOffStats + kStatsWords * 4);
  const int hot_copy = ((tid & 63) * kHotCopies) >> 6;
  if (KIND == kKindStats) {
    // Symbol statistics for optimised Huffman tables (reference AddEntropyStats,
    // src/entropy.cc:208-227): per table, counts of AC symbols (run << 4 | size, ZRL, EOB) and
    // of DC size categories, in LDS counters (cleared with the tables) flushed as this
    // workgroup's partial.  The DC category here, the AC symbols by the same sorted parts the
    // encode kind walks (below).
    if (emits) {
      const int diff = dc_val - pred;
      const int ad = diff < 0 ? -diff : diff;
      atomicAdd(&lf[tbl * 272 + 256 + (32 - __clz(ad))], 1u);
    }
  }

  // The run/size coding of a block (src/entropy.cc:161-198) is a serial walk over its non-zero
  // coefficients, and a structured 4K picture averages 13 non-zeros per block but 45 for the
  // worst block of a segment: one thread per block leaves the whole workgroup waiting for that
  // one walk.  So a block is coded as up to four independent PARTS, one per quarter of the
  // zig-zag scan (positions 1-15 with the DC, 16-31, 32-47, 48-63; empty quarters make no part).
  // A part needs only the block's non-zero mask to know the run in front of its first symbol and
  // whether it carries the EOB, and its bits are stitched at bit granularity like the blocks
  // themselves.  Parts are handed to threads sorted by their number of non-zeros (counting sort,
  // descending) in groups of 64 the waves draw from a queue: walks of at most 16 symbols with
  // similar trip counts per wave.
  // The unit list and the part lengths live in the bit window, idle until the stitch.
  uint32_t* const hist = reinterpret_cast<uint32_t*>(smem + L::kOffHist);   // bins 0..16 (cleared with the tables, filled before the DC barrier)
  uint16_t* const ulist = reinterpret_cast<uint16_t*>(smem + L::kOffList);  // block | quarter << 8
  uint32_t n_units;
  {
    RACE_POINT(4);
    // every wave scans the 17 bins (16, 15, ... 0) for itself: no hand-over through LDS, no barrier
    const int ln = tid & 63;
    const uint32_t hcnt = ln <= 16 ? hist[16 - ln] : 0u;
    const uint32_t incl = wave_inclusive_scan(hcnt);
    const uint32_t excl = incl - hcnt;             // lane l: where bin 16 - l starts
    n_units = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 16));   // number of parts
    uint32_t start[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) start[q] = __shfl(excl, 16 - static_cast<int>(pc[q]), 64);
    if (emits) {
      ulist[start[0] + rank[0]] = static_cast<uint16_t>(tid);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        if (pc[q] != 0u) ulist[start[q] + rank[q]] = static_cast<uint16_t>(tid | (q << 8));
      }
    }
    __syncthreads();
  }
  stamp(3);
  RACE_POINT(5);
  if (KIND == kKindStats) {
    // a part's symbols are counted instead of coded: same masks, same runs, same quarter cut
    const uint32_t n_groups_s = (n_units + 63u) >> 6;
    for (int r = 0; r < 4; ++r) {
      uint32_t grp = 0;
      if ((tid & 63) == 0) grp = atomicAdd(&misc[10], 1u);
      grp = __builtin_amdgcn_readfirstlane(grp);
      if (grp >= n_groups_s) break;
      const uint32_t idx = grp * 64u + (tid & 63u);
      if (idx < n_units) {
        const uint32_t unit = ulist[idx];
        const int blk = static_cast<int>(unit & 255u), q = static_cast<int>(unit >> 8);
        const unsigned char* const bslot = smem + blk * kSlotBytes;
        const uint32_t* const btail = reinterpret_cast<const uint32_t*>(bslot + 128);
        const unsigned long long m_all = (static_cast<unsigned long long>(btail[1]) << 32) | btail[0];
        const int b_k = blk % BPM;
        const int b_tbl = (MODE == SJPEG_HIP_YUV420) ? (b_k >= 4) : (MODE == SJPEG_HIP_YUV444 ? (b_k >= 1) : 0);
        uint32_t* const f = lf + b_tbl * 272;
        // the three symbols that make up half of an ordinary picture's -- a coefficient of 1, 2..3 or 4..7 right
        // behind the previous one -- are counted in kHotCopies sets of counters picked by lane: an LDS atomic
        // serialises the lanes that meet in a word, and with one set a third of the wave met in the word of symbol 0x01
        uint32_t* const fh = HOT ? hot + hot_copy * 6 + b_tbl * 3 - 1 : f;
        const uint16_t* const zz = reinterpret_cast<const uint16_t*>(bslot);
        const int sh = 16 * q;
        uint32_t m = static_cast<uint32_t>(m_all >> sh) & 0xffffu;
        const unsigned long long below = m_all & ((1ull << sh) - 1ull);
        const bool is_last = (q == 3) || ((m_all >> (sh + 16)) == 0ull);
        int prev = below ? 64 - __builtin_clzll(below) : 1;
        // The entry of the NEXT symbol is read before the counter of this one is bumped: LDS operations return in
        // order, and a read issued behind an atomic waits for it -- with the lanes of a wave bumping the same few
        // counters that was most of the time this loop took.  (Past the part's last symbol the read lands on the
        // next quarter or the slot's tail: in the slot, never used.)
        int i = sh + __builtin_ctz(m | 0x10000u);
        uint32_t e = zz[i];
        uint32_t zrls = 0;
        while (m) {
          m &= m - 1;
          const int i_next = sh + __builtin_ctz(m | 0x10000u);
          const uint32_t e_next = zz[i_next];
          const uint32_t mag = e & 0x7fffu;
          const int run = i - prev;
          prev = i + 1;
          zrls += static_cast<uint32_t>(run >> 4);           // (counted once behind the loop: no branch in it)
          const uint32_t sym = static_cast<uint32_t>(((run & 15) << 4) | (32 - __clz(mag)));
          atomicAdd(&((HOT && sym - 1u < 3u) ? fh : f)[sym], 1u);
          i = i_next; e = e_next;
        }
        if (zrls != 0u) atomicAdd(&f[0xf0], zrls);
        if (is_last && prev <= 63) atomicAdd(&f[0x00], 1u);
      }
    }
    RACE_POINT(20);
    __syncthreads();
    uint32_t* const dst = a.partial + (static_cast<size_t>(frame) * a.nseg + seg) * kStatsWords;
    const uint32_t* const lf_all = reinterpret_cast<const uint32_t*>(smem + L::kOffStats);
    for (int i = tid; i < kStatsWords; i += kScanThreads) {
      uint32_t v = lf_all[i] + (L::kStatsCopies == 2 ? lf_all[kStatsWords + i] : 0u);
      if (HOT) {
        const int t = i >= 272 ? 1 : 0, sym = i - t * 272;
        if (sym >= 1 && sym <= 3) {
#pragma unroll
          for (int c = 0; c < kHotCopies; ++c) v += hot[c * 6 + t * 3 + sym - 1];
        }
      }
      dst[i] = v;
    }
    return;
  }

  // the walk reads 16-bit entries and writes 32-bit words in the same slot: no type-based reordering
  typedef uint32_t __attribute__((may_alias)) u32_alias;
  constexpr uint32_t kNoRow = 0xffffffffu;
  // this frame's pool (scan_device.h): the part of a segment that does not fit its slot, and the rows
  // of the checked walk; one bump counter per frame, and a flag that says the frame overran it
  uint32_t* const pool = a.pool + static_cast<size_t>(frame) * a.pool_words;
  auto pool_take = [&](uint32_t nwords) -> uint32_t {          // word offset in the pool, kNoRow if it is full
    const uint32_t at = atomicAdd(&a.pool_ctr[2 * frame], nwords);
    if (at + nwords > a.pool_words) { atomicOr(&a.pool_ctr[2 * frame + 1], 1u); return kNoRow; }
    return at;
  };

  // The CHECKED walk codes the parts of a block with an AC level of more than n_safe bits (none in
  // ordinary pictures; the q >= 97 noise tests).  Nothing bounds the bits such a part makes per
  // entry, so it first reads its 16 entries into registers -- then every word up to the eighth can
  // be stored over them in place -- and takes a row of 8 words from the frame's pool for words
  // 8 .. 15 when it gets that far (a part is shorter than 496 bits).  A row is only taken by a part
  // that has already produced 32 bytes, so the rows of a frame never add up to more than its
  // output.  Plain and slow: picks an entry out of the registers by a chain of selects.
  auto walk_checked = [&](uint32_t unit, uint32_t pw, uint32_t dcword, uint32_t& rec_out, uint32_t& row_out) {
    const uint32_t blk = unit & 255u, q = unit >> 8;
    const uint32_t b_tbl = (pw >> 25) & 1u;
    // (the compact layout keeps the raw AC table in global memory: this walk is the q >= 97 noise path)
    const uint32_t* const ac = COMPACT ? &(a.tables + frame * a.tables_stride)->ac[b_tbl][0]
                                       : reinterpret_cast<const uint32_t*>(smem + (COMPACT ? 0 : L::kOffAc)) + b_tbl * 256;
    const uint32_t wp0 = blk * kSlotBytes + 32u * q;
    const uint4 e0 = *reinterpret_cast<const uint4*>(smem + wp0), e1 = *reinterpret_cast<const uint4*>(smem + wp0 + 16);
    const uint32_t ent[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    uint32_t m = pw & 0xffffu;                     // the part's own 16 positions
    // local position after the previous non-zero, from the run in front of the first symbol (ZRLs included:
    // this walk codes them itself); the EOB flag says whether anything follows the part
    int prevl = __builtin_ctz(m | 0x10000u) - static_cast<int>((pw >> 16) & 63u);
    uint32_t acc = 0, fill = 0, wr = 0, row = kNoRow;
    bool lost = false;                             // the pool is full: the frame reports size 0 anyway
    auto put_word = [&](uint32_t word) {
      if (wr < 8u) {
        *reinterpret_cast<u32_alias*>(smem + wp0 + 4u * wr) = word;
      } else {
        if (wr == 8u) { row = pool_take(8u); lost = (row == kNoRow); }
        if (!lost) pool[row + wr - 8u] = word;
      }
      ++wr;
    };
    auto append = [&](uint32_t bits, uint32_t nb) {                  // 1 <= nb <= 31
      const uint32_t t = fill + nb;
      const uint32_t s5 = t & 31u;
      const uint32_t P = __builtin_amdgcn_alignbit(bits, 0u, s5);
      if (t >= 32u) { put_word(acc | (bits >> s5)); acc = P; } else { acc |= P; }
      fill = s5;
    };
    if (q == 0u) append(dcword & 0xffffffu, (dcword >> 24) & 31u);
    const uint32_t zrl = ac[0xf0];
    while (m) {
      const int i = __builtin_ctz(m);
      m &= m - 1u;
      uint32_t d = ent[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) d = (i >> 1) == k ? ent[k] : d;
      const uint32_t e = (i & 1) ? (d >> 16) : (d & 0xffffu);
      int run = i - prevl;
      prevl = i + 1;
      for (; run >= 16; run -= 16) append(zrl >> 16, zrl & 0xffu);
      const uint32_t mag = e & 0x7fffu;
      const uint32_t n = 32u - static_cast<uint32_t>(__clz(mag));
      const uint32_t suffix = (e & 0x8000u) ? (mag ^ ((1u << n) - 1u)) : mag;
      const uint32_t cw = ac[(static_cast<uint32_t>(run) << 4) | n];
      append(((cw >> 16) << n) | suffix, (cw & 0xffu) + n);
    }
    if ((pw >> 22) & 1u) { const uint32_t eob = ac[0x00]; append(eob >> 16, eob & 0xffu); }
    const uint32_t len = 32u * wr + fill;
    if (fill != 0u) put_word(acc);                 // the last word, left-aligned, goes where the others are
    *reinterpret_cast<u32_alias*>(smem + blk * kSlotBytes + 128u + 4u * q) = len;   // the part's length, over its description
    // (spill field: 8 = words 8.. are in the pool row, 31 = all in place)
    rec_out = unit | ((wr > 8u ? 8u : 31u) << 10) | (len << 17) | (b_tbl << 28);
    row_out = row;
  };

  // The LEAN walk codes every part of a block whose AC levels have at most n_safe bits (all of them in
  // ordinary pictures; DevTables::safe_mask).  It reads no entry ahead, takes no pool row and needs no
  // frontier test: a finished word is stored over the part's own entries unconditionally, because
  // under the level bound it can never reach an entry that is still to be read.  Proof: let T(i) be
  // the bits produced once the entry at local position i (0..15) is coded.  A symbol of run r makes
  // at most 16 + n_safe <= 16 (r + 1) bits when r >= 1, and at most 16 when r = 0 (that IS the
  // definition of n_safe); the DC symbol in front of quarter 0 has at most 27 <= 16 + 15 bits; the run
  // of a part's FIRST symbol is taken modulo 16 here, its ZRL codes travel in the part's record and
  // are placed by the stitch.  Hence T(i) <= 16 (i + 1) + 15 by induction, the words complete at that
  // point are 0 .. floor(T(i) / 32) - 1, and the last of them covers the entries up to
  // 2 floor(T(i) / 32) - 1 <= i: all read.  With the EOB (<= 16 bits) T <= 287, so at most 8 words are
  // stored (the quarter has room for exactly 8); the last, partial word stays in a register.
  // Code words come from the merged table (code << n | total length << 27), indexed by clz(level)
  // and run, so a symbol costs two LDS reads and about thirty simple instructions.
  const uint32_t acm_base = static_cast<uint32_t>(L::kOffAcm) - 22u * 4u;    // word [run][clz - 22]: 40 bytes per run
  typedef uint16_t __attribute__((may_alias)) u16_alias2;
  auto walk_lean = [&](uint32_t unit, uint32_t pw, uint32_t pw2, uint32_t dcword, uint32_t& rec_out, uint32_t& tail_out) {
    const uint32_t blk = unit & 255u, q = unit >> 8;
    const uint32_t slot_off = blk * kSlotBytes;
    const uint32_t b_tbl = (pw >> 25) & 1u;
    const uint32_t tb = acm_base + b_tbl * 640u;
    // the part's own 16 positions -- 32 for a merged part (bit 23), whose second quarter's mask is the upper half
    // of the word behind.  Everything below holds for 32 positions as it does for 16: T(i) <= 16 (i + 1) + 15 for
    // i = 0 .. 31 (no run inside the part reaches 16: that is what made it one part), with the EOB T <= 543, so
    // at most 16 words are stored where the 32 entries were, the last, partial word stays in a register.
    const bool merged = ((pw >> 23) & 1u) != 0u;
    const uint32_t m = (pw & 0xffffu) | (merged ? pw2 & 0xffff0000u : 0u);
    // the block's thread has read the masks for this quarter already (P3 start)
    const uint32_t inf = (pw >> 16) & 0x7fu;
    uint32_t acc = 0, fill = 0;                    // bits of the word in the making, left-aligned; their number
    const uint32_t wp0 = slot_off + 32u * q;       // the quarter: 16 entries, then up to 8 words
    uint32_t wp = wp0;                             // byte offset of the next word
    if (q == 0u) {
      fill = (dcword >> 24) & 31u;
      acc = __builtin_amdgcn_alignbit(dcword & 0xffffffu, 0u, fill);   // DC bits << (32 - fill)
    }
    // positions are local to the quarter from here on; the ZRLs of the first run are taken out of it
    // (only the first symbol of a part can have a run of 16 or more, and never in quarter 0)
    const uint32_t nzrl = (inf >> 4) & 3u;
    // bit 16 set for good: a guard behind the part's 16 positions, so that "the next non-zero" is a bare v_ffbl; a
    // merged part has no room for one: its mask runs out at zero, of which v_ffbl makes -1 -- the fetch behind the
    // last symbol then goes to the two bytes in front of the part (in the slot of a coded block, never the first
    // of LDS), and is not used either
    const uint32_t guard = merged ? 0u : 0x10000u;
    uint32_t ms = m | guard;
    auto first_bit = [](uint32_t x) { int r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; };
    // local position OF the previous non-zero (may be negative); a symbol's row of merged code words is picked by
    // pos - prevp = run + 1, with the table base one row down: one subtraction instead of a three-operand form
    int prevp = first_bit(ms) - static_cast<int>(inf & 15u) - 1;
    const uint32_t tbm = tb - 40u;
    // (the two code words the end of the part may need: fetched here, under the symbols' round trips)
    const uint2 ez = *reinterpret_cast<const uint2*>(ldc + 26 + 2 * b_tbl);
    const uint32_t eob = ez.x, zrl = ez.y;
    auto append = [&](uint32_t bits, uint32_t nb) {                  // 1 <= nb <= 27
      const uint32_t t = fill + nb;
      const uint32_t s5 = t & 31u;
      const uint32_t P = __builtin_amdgcn_alignbit(bits, 0u, s5);   // bits << (32 - s5); 0 for s5 == 0
      if (t >= 32u) {
        *reinterpret_cast<u32_alias*>(smem + wp) = acc | (bits >> s5);
        wp += 4u;
        acc = P;
      } else {
        acc |= P;
      }
      fill = s5;
    };
    // Software pipeline over the symbols, so that no LDS round trip is left in the dependent chain of
    // a symbol: while symbol k is appended, the code word of symbol k + 1 and the entry of symbol
    // k + 2 are in flight.  (Reading an entry earlier than the in-place argument assumes is always
    // safe; behind the last symbol the fetch goes to position 16, the first bytes of the next
    // quarter, and is not used.)
    auto entry_at = [&](int pos) {
      return static_cast<uint32_t>(*reinterpret_cast<const u16_alias2*>(smem + wp0 + 2u * static_cast<uint32_t>(pos)));
    };
    // first half of a symbol: its level bits and the address of its merged code word
    auto stage = [&](int pos, uint32_t e, uint32_t& lv, uint32_t& cw_at) {
      const uint32_t run1 = static_cast<uint32_t>(pos - prevp);     // run + 1: 1 .. 16
      prevp = pos;
      const uint32_t mag = e & 0x7fffu;                            // 1 .. 1023
      const uint32_t nl = static_cast<uint32_t>(__builtin_clz(mag));   // 32 - n (mag != 0: a bare v_ffbh_u32)
      const uint32_t ones = 0xffffffffu >> nl;
      uint32_t sgn;                                // bit 15 over the whole word (asm: the builtin is turned into compare + select)
      asm("v_bfe_i32 %0, %1, 15, 1" : "=v"(sgn) : "v"(e));
      lv = mag ^ (ones & sgn);
      uint32_t row;                                // tb + 40 run in one operation (the compiler makes it mul + shift + add3)
      asm("v_mad_u32_u24 %0, %1, 40, %2" : "=v"(row) : "v"(run1), "v"(tbm));
      cw_at = row + nl * 4u;
    };
#if SJPEG_WALK_PIPE == 2
    if (m) {
      int i = first_bit(ms);
      uint32_t e = entry_at(i);
      ms &= ms - 1u;
      int i_next = first_bit(ms);
      uint32_t e_next = entry_at(i_next);
      uint32_t lv, cw_at;
      stage(i, e, lv, cw_at);
      uint32_t cw = *reinterpret_cast<const uint32_t*>(smem + cw_at);
      while (ms != guard) {                        // (i_next, e_next) is a symbol
        ms &= ms - 1u;
        const int i_after = first_bit(ms);
        const uint32_t e_after = entry_at(i_after);
        uint32_t lv_next, cw_at_next;
        stage(i_next, e_next, lv_next, cw_at_next);
        const uint32_t cw_next = *reinterpret_cast<const uint32_t*>(smem + cw_at_next);
        append((cw & 0x07ffffffu) | lv, cw >> 27);
        cw = cw_next; lv = lv_next;
        i_next = i_after; e_next = e_after;
      }
      append((cw & 0x07ffffffu) | lv, cw >> 27);
    }
#elif SJPEG_WALK_PIPE == 0
    // no read ahead at all: the other waves of the SIMD cover the two round trips of a symbol
    while (ms != guard) {
      const int i = first_bit(ms);
      ms &= ms - 1u;
      const uint32_t e = entry_at(i);
      uint32_t lv, cw_at;
      stage(i, e, lv, cw_at);
      const uint32_t cw = *reinterpret_cast<const uint32_t*>(smem + cw_at);
      append((cw & 0x07ffffffu) | lv, cw >> 27);
    }
#elif SJPEG_WALK_PIPE == 1
    // the next symbol's entry is fetched one symbol ahead; its code word inside the iteration
    if (m) {
      int i = first_bit(ms);
      uint32_t e = entry_at(i);
      do {
        ms &= ms - 1u;
        const int i_next = first_bit(ms);
        const uint32_t e_next = entry_at(i_next);
        uint32_t lv, cw_at;
        stage(i, e, lv, cw_at);
        const uint32_t cw = *reinterpret_cast<const uint32_t*>(smem + cw_at);
        append((cw & 0x07ffffffu) | lv, cw >> 27);
        i = i_next; e = e_next;
      } while (ms != guard);
    }
#elif SJPEG_WALK_PIPE == 3
    // the depth of variant 2 (entries two symbols ahead, code words one), two symbols per trip of the loop: the
    // registers change roles by name, not by moves
    if (m) {
      int i0 = first_bit(ms);
      uint32_t e0 = entry_at(i0);
      ms &= ms - 1u;
      int i1 = first_bit(ms);
      uint32_t e1 = entry_at(i1);
      uint32_t lv0, at0, lv1, at1;
      stage(i0, e0, lv0, at0);
      uint32_t cw0 = *reinterpret_cast<const uint32_t*>(smem + at0), cw1;
      for (;;) {
        if (ms == guard) { append((cw0 & 0x07ffffffu) | lv0, cw0 >> 27); break; }
        ms &= ms - 1u;
        i0 = first_bit(ms);
        e0 = entry_at(i0);
        stage(i1, e1, lv1, at1);
        cw1 = *reinterpret_cast<const uint32_t*>(smem + at1);
        append((cw0 & 0x07ffffffu) | lv0, cw0 >> 27);
        if (ms == guard) { append((cw1 & 0x07ffffffu) | lv1, cw1 >> 27); break; }
        ms &= ms - 1u;
        i1 = first_bit(ms);
        e1 = entry_at(i1);
        stage(i0, e0, lv0, at0);
        cw0 = *reinterpret_cast<const uint32_t*>(smem + at0);
        append((cw1 & 0x07ffffffu) | lv1, cw1 >> 27);
      }
    }
#endif
    if (inf & 0x40u) append(eob >> 16, eob & 0xffu);
    const uint32_t len = ((wp - wp0) << 3) + fill;
    const uint32_t zl = zrl & 0xffu;
    *reinterpret_cast<u32_alias*>(smem + slot_off + 128u + 4u * q) = len + nzrl * zl;   // the part's length, over its description
    rec_out = unit | (31u << 10) | (nzrl << 15) | (len << 17) | (1u << 27) | (b_tbl << 28);
    tail_out = acc;
  };

  // what this thread coded in each round: unit | spill << 10 | ZRLs in front << 15 | len << 17 |
  // lean << 27 | chroma tables << 28, 0xffffffff = nothing; and the part's last, partial word (lean walk)
  // (four registers picked by the round counter: one copy of the walks for all rounds)
  uint32_t ur0 = 0xffffffffu, ur1 = 0xffffffffu, ur2 = 0xffffffffu, ur3 = 0xffffffffu;
  uint32_t tw0 = 0, tw1 = 0, tw2 = 0, tw3 = 0;
  auto ur_get = [&](int r) { return r == 0 ? ur0 : r == 1 ? ur1 : r == 2 ? ur2 : ur3; };
  auto tw_get = [&](int r) { return r == 0 ? tw0 : r == 1 ? tw1 : r == 2 ? tw2 : tw3; };
  // The list is sorted: handed out in order, wave 0's 64 parts would be the heaviest of every
  // round and the other waves would wait for it at the barrier below.  Groups of 64 parts go to
  // the waves in boustrophedon order instead (0 1 2 3 / 7 6 5 4 / ...), which is static: a thread
  // knows its (up to) four parts at once and fetches their list entries and block tails together,
  // instead of one dependent chain of LDS round trips in front of every walk.
  // (rounds: 256 parts each; the ordinary segment has two -- the rounds behind the last are skipped by everybody)
  const int nrounds = static_cast<int>((n_units + 255u) >> 8);           // uniform
  uint32_t un[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, pws[4] = {0, 0, 0, 0}, pw2s[4] = {0, 0, 0, 0}, dws[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r < nrounds) {
      const uint32_t grp = static_cast<uint32_t>(4 * r) + ((r & 1) ? 3u - (tid >> 6) : (tid >> 6));
      const uint32_t idx = grp * 64u + (tid & 63u);
      un[r] = idx < n_units ? ulist[idx] : 0xffffffffu;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r < nrounds) {
      // (no part: some word inside the kernel's LDS is read and not used)
      const uint32_t blk = un[r] & 255u, q = (un[r] >> 8) & 3u;
      pws[r] = *reinterpret_cast<const u32_alias*>(smem + blk * kSlotBytes + 128u + 4u * q);
      // (the word behind: the second quarter's mask of a merged part; something in LDS otherwise)
      pw2s[r] = *reinterpret_cast<const u32_alias*>(smem + blk * kSlotBytes + 132u + 4u * q);
      dws[r] = dcw[blk];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    RACE_POINT(6);
    if (un[r] != 0xffffffffu) {
      uint32_t rec, tw = 0;
      if ((pws[r] >> 24) & 1u) {
        walk_checked(un[r], pws[r], dws[r], rec, tw);      // (tw: the part's pool row)
      } else {
        walk_lean(un[r], pws[r], pw2s[r], dws[r], rec, tw);
      }
      if (r == 0) { ur0 = rec; tw0 = tw; } else if (r == 1) { ur1 = rec; tw1 = tw; }
      else if (r == 2) { ur2 = rec; tw2 = tw; } else { ur3 = rec; tw3 = tw; }
    }
  }
  __syncthreads();
  stamp(4);
  PHASE_PRIO(3);
  RACE_POINT(7);
  // offsets are a prefix sum in STREAM order (thread tid owns block tid here); the block's tail
  // (masks and DC word: consumed) becomes the bit offsets of its four parts
  uint32_t total;
  {
    // the lengths the walks left in the tail (a quarter without a part still holds its mask: zero)
    uint4 Lw = make_uint4(0, 0, 0, 0);
    if (has_slot) Lw = *reinterpret_cast<const uint4*>(tail);
    const uint32_t l0 = Lw.x & 0xffffu, l1 = Lw.y & 0xffffu, l2 = Lw.z & 0xffffu, l3 = Lw.w & 0xffffu;
    // (the scratch words are not used again: the barrier after the window is cleared, below, closes the scan)
    const uint32_t s0 = wg_exclusive_scan<kScanThreads, false>(l0 + l1 + l2 + l3, misc, &total);
    if (has_slot) *reinterpret_cast<uint4*>(tail) = make_uint4(s0, s0 + l0, s0 + l0 + l1, s0 + l0 + l1 + l2);
  }
  // Restart mode (optional, never the reference's bytes): the interval ends on a byte boundary,
  // padded with 1-bits, and 16 zero bits hold the place of its RSTn marker -- zero bytes pass the
  // byte stuffing untouched, a small kernel writes FF D0+n over them afterwards (stitch_kernels.h).
  const uint32_t data_bits = total;
  if (a.rst) {
    const bool last_of_frame = (m_first + n_coded == a.n_mcus);
    total = ((total + 7u) & ~7u) + (last_of_frame ? 0u : 16u);
  }
  RACE_POINT(8);
  // every thread has read its part lengths (barrier inside the scan): the window can be cleared
  // for the stitch under the same barrier that publishes the offsets
  if (a.ablate != 3) {
    for (int i = tid; i < kWinWords / 4; i += kScanThreads) reinterpret_cast<uint4*>(win)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) win[kWinWords] = 0;              // (the spare word behind the window: a part's last store may land there)
  }
  if (a.ablate == 3) { if (tid == 0) a.seg_nbits[static_cast<size_t>(frame) * a.nseg + seg] = total; return; }
  __syncthreads();
  RACE_POINT(9);
  // a part's bit offset in the segment: word q of its block's tail
  auto part_start = [&](uint32_t rec) {
    return *reinterpret_cast<const uint32_t*>(smem + (rec & 255u) * kSlotBytes + 128 + 4u * ((rec >> 8) & 3u));
  };

  stamp(5);
  // Stitch: every part's words are shifted to its bit offset and ORed into the LDS window,
  // round by round (one round unless the segment overflows the window); the window is flushed
  // coalesced to the segment's slot.
  // Where the segment's words go: the first slot_words of them into its own slot, the rest into
  // the frame's pool (one bump allocation, now that the length is known).  Slots are sized from the
  // caller's out_stride, not for the worst case: see scan_engine.hip.
  uint32_t* const out_words = a.seg_words + (static_cast<size_t>(frame) * a.nseg + seg) * a.slot_words;
  {
    const uint32_t nw_seg = (total + 31u) >> 5;
    if (tid == 0) {
      const uint32_t xb = nw_seg > a.slot_words ? pool_take(nw_seg - a.slot_words) : kNoRow;
      misc[9] = xb;
      a.seg_xbase[static_cast<size_t>(frame) * a.nseg + seg] = xb;
    }
  }
  // `pos`: the part's bit position in the segment; `base_w`: the segment word the window starts at.  CLIP (a
  // segment longer than the window, coded window by window): only the words that fall inside the window are
  // placed -- a part that straddles a window's end is placed twice, each time with the words of that window.
  auto place = [&](auto clip_tag, uint32_t rec, uint32_t pos, uint32_t tailw, uint32_t base_w) {
    constexpr bool CLIP = decltype(clip_tag)::value;
    const uint32_t blk = rec & 255u, q = (rec >> 8) & 3u, len = (rec >> 17) & 1023u;
    const uint32_t nzrl = (rec >> 15) & 3u;
    // window word d (may be "negative" = huge when the word lies in front of the window)
    auto put = [&](uint32_t d, uint32_t v) {
      if (!CLIP || d < static_cast<uint32_t>(kWinWords)) atomicOr(win + d, v);
    };
    if (nzrl) {
      // the ZRL codes in front of the part's first symbol (lean walk): up to 3 x 16 bits
      const uint4 zp = reinterpret_cast<const uint4*>(smem + L::kOffZrl)[((rec >> 28) & 1u) * 4u + nzrl];
      const uint32_t o = pos & 31u, d = (pos >> 5) - base_w;
      put(d, zp.x >> o);
      put(d + 1u, __builtin_amdgcn_alignbit(zp.x, zp.y, o));
      if (o + zp.z > 64u) put(d + 2u, zp.y << (32u - o));
      pos += zp.z;
    }
    const uint32_t o = pos & 31u;
    if (!CLIP && ((rec >> 27) & 1u)) {
      // lean walk: len >> 5 full words in the part's quarter, the rest (left-aligned) in tailw
      uint32_t src = blk * kSlotBytes + 32u * q;
      const uint32_t src_end = src + ((len >> 5) << 2);
      uint32_t dst = static_cast<uint32_t>(L::kOffWin) + ((pos >> 5) << 2);
      uint32_t before = 0;                         // source word j - 1
      while (src != src_end) {
        const uint32_t v = *reinterpret_cast<const u32_alias*>(smem + src);
        atomicOr(reinterpret_cast<uint32_t*>(smem + dst), __builtin_amdgcn_alignbit(before, v, o));   // (before:v) >> o
        before = v;
        src += 4u; dst += 4u;
      }
      atomicOr(reinterpret_cast<uint32_t*>(smem + dst), __builtin_amdgcn_alignbit(before, tailw, o));
      if (o + (len & 31u) > 32u) atomicOr(reinterpret_cast<uint32_t*>(smem + dst + 4u), tailw << (32u - o));
      return;
    }
    // all words from memory: lean walk -- the full words in the part's quarter, the last (left-aligned) one in
    // tailw; checked walk -- the first eight in the part's quarter, the rest in its pool row
    const u32_alias* const bw = reinterpret_cast<const u32_alias*>(smem + blk * kSlotBytes) + 8 * q;
    const bool lean = ((rec >> 27) & 1u) != 0u;
    const bool rowed = ((rec >> 10) & 31u) == 8u;
    const uint32_t nw = (len + 31u) >> 5;          // words of the part, the last one left-aligned
    const uint32_t d0 = (pos >> 5) - base_w;
    uint32_t before = 0;
    for (uint32_t j = 0; j <= nw; ++j) {
      uint32_t v = 0;
      if (j < nw) {
        if (lean) v = (j < (len >> 5)) ? bw[j] : tailw;
        else v = (j < 8u) ? bw[j] : ((rowed && tailw != kNoRow) ? pool[tailw + j - 8u] : 0u);
      }
      if (j < nw || o != 0u) put(d0 + j, __builtin_amdgcn_alignbit(before, v, o));
      before = v;
    }
  };
  // bits of a part in the stream, ZRL codes in front included (an upper bound: 16 each)
  auto part_bits = [&](uint32_t rec) { return ((rec >> 17) & 1023u) + 16u * ((rec >> 15) & 3u); };
  const uint32_t nw_seg = (total + 31u) >> 5;
  auto flush = [&](uint32_t base_w, uint32_t nwords) {
    const uint32_t xbase = misc[9];                // (written before the barriers in front of every flush)
    for (uint32_t i = tid; i < nwords; i += kScanThreads) {
      const uint32_t j = base_w + i;
      if (j < a.slot_words) out_words[j] = win[i];
      else if (xbase != kNoRow) pool[xbase + (j - a.slot_words)] = win[i];
    }
  };
  if (nw_seg <= static_cast<uint32_t>(kWinWords)) {
    // the usual case: the segment fits the window (cleared above)
#pragma unroll 1
    for (int r = 0; r < nrounds; ++r) {            // (NOT unrolled: four inlined copies of place() had the compiler
      const uint32_t rec = ur_get(r);              // hoist ~250 instructions of address arithmetic in front of them)
      if (rec != 0xffffffffu) place(std::false_type(), rec, part_start(rec), tw_get(r), 0u);
    }
    if (a.rst && tid == 0 && (data_bits & 7u) != 0u) {
      // 1-bits from the end of the data to the byte boundary (same word)
      const uint32_t pad = 8u - (data_bits & 7u);
      atomicOr(&win[data_bits >> 5], ((1u << pad) - 1u) << (32u - (data_bits & 31u) - pad));
    }
    __syncthreads();
    stamp(6);
    RACE_POINT(10);
    flush(0u, nw_seg);
  } else {
    // A segment longer than the window is coded window by window: every part places the words of it that fall
    // into the window at hand (at most two windows see a part of less than 1100 bits), the window is flushed,
    // cleared, and stands for the next kWinWords words of the segment.  No vote on what fits, no carried word:
    // two barriers a window.
    for (uint32_t base_w = 0; base_w < nw_seg; base_w += static_cast<uint32_t>(kWinWords)) {
      if (base_w != 0u) {
        for (int i = tid; i < kWinWords / 4; i += kScanThreads) reinterpret_cast<uint4*>(win)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
      }
      const uint32_t lo = base_w << 5, hi = (base_w + static_cast<uint32_t>(kWinWords)) << 5;   // the window's bits
#pragma unroll 1
      for (int r = 0; r < nrounds; ++r) {
        const uint32_t rec = ur_get(r);
        if (rec != 0xffffffffu) {
          const uint32_t st = part_start(rec);
          // (a part's last store may go one word past its bits: + 32).  Nearly every part lies inside ONE window
          // and takes the plain path with its position relative to that window; the few that straddle a window's
          // end are placed word by word, clipped, once in each of the two windows
          const uint32_t end = st + part_bits(rec) + 32u;
          if (st >= lo && end <= hi) place(std::false_type(), rec, st - lo, tw_get(r), 0u);
          else if (st < hi && end > lo) place(std::true_type(), rec, st, tw_get(r), base_w);
        }
      }
      if (a.rst && tid == 0 && (data_bits & 7u) != 0u && data_bits >= lo && data_bits < hi) {
        const uint32_t pad = 8u - (data_bits & 7u);
        atomicOr(&win[(data_bits >> 5) - base_w], ((1u << pad) - 1u) << (32u - (data_bits & 31u) - pad));
      }
      __syncthreads();
      RACE_POINT(10);
      const uint32_t left = nw_seg - base_w;
      flush(base_w, left < static_cast<uint32_t>(kWinWords) ? left : static_cast<uint32_t>(kWinWords));
      __syncthreads();                             // (the next window is cleared behind this)
    }
    stamp(6);
  }
  RACE_POINT(11);
  if (tid == 0) a.seg_nbits[static_cast<size_t>(frame) * a.nseg + seg] = total;
  stamp(7);
  // (SJPEG_HIP_STAMPS=3: the last stamp is the segment's bit count instead -- which segments stitch slowly?)
  if (a.stamps != nullptr && a.stamp_real == 2 && tid == 0) a.stamps[(static_cast<size_t>(frame) * a.nseg + seg) * 8 + 7] = total;
  return;
  }   // (the histogram kind's loop over its segments)
}

