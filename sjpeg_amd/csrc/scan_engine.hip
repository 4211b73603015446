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
#include <type_traits>
#include <string>
#include <chrono>
#include <vector>

#include "sjpeg_hip.h"

namespace {

#include "scan_device.h"
#include "scan_segments.h"
#include "scan_reduce.h"
#include "stitch_kernels.h"

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
  g->slot_words = ((static_cast<uint32_t>(g->seg_mcus) * g->bpm * kMaxBlockBits + 31) / 32 + 2 + 3) & ~3u;   // (whole 16-byte units)
  return true;
}

// measurement aid (SJPEG_HIP_BATCH_DEBUG=1): microseconds since this thread's previous mark, on stderr -- which runtime
// call of a launch sequence the host spent its time in (tools/slow_call_timeline.py)
inline void dbg_mark(const char* what) {
  static const int level = getenv("SJPEG_HIP_BATCH_DEBUG") != nullptr ? atoi(getenv("SJPEG_HIP_BATCH_DEBUG")) : 0;
  if (level < 2) return;
  static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  const auto now = std::chrono::steady_clock::now();
  const double us = std::chrono::duration<double, std::micro>(now - last).count();
  if (us > (level >= 3 ? 6.0 : 200.0)) fprintf(stderr, "    step %-28s %9.1f us\n", what, us);   // (3: every step over 6 us)
  last = now;
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
  int cu_count = 256;                  // compute units of the device (the persistent histogram kind sizes its grid by it)
  int histo_slots = 0;                 // SJPEG_HIP_HISTO_SLOTS, read when the engine is made: workgroups of a histogram launch (tests: many trips)
  DevBuf<DevTables> tables;
  DevBuf<uint8_t> header;
  // what the two buffers hold, and the stream that put it there: a call with the same tables /
  // header on the same stream skips the upload (two of the ~6 runtime calls of a small encode)
  std::vector<DevTables> tables_held;
  // Host-to-device uploads of more than a few KB (the tables and headers of a batch) go through pinned blocks the
  // engine owns: hipMemcpyAsync from pageable memory pins the caller's pages for the length of the copy -- tens of
  // microseconds of host time per upload and, now and then, several milliseconds (seen as one call in twenty of a
  // 32-frame default-parameter batch taking 8 ms instead of 1.6).
  struct Stage { void* p = nullptr; void* dp = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; } stage[4];
  int stage_next = 0;
  std::vector<uint8_t> header_held;
  const void* tables_held_at = nullptr; const void* header_held_at = nullptr;
  hipStream_t tables_stream = nullptr, header_stream = nullptr;
  DevBuf<uint32_t> seg_words, seg_nbits, pool, pool_ctr, seg_xbase, ubuf, chunk_ff, partial, replay;
  DevBuf<uint32_t> frame_flags;        // K2's copy of every frame's "overran its pool" flag (stitch_kernels.h)
  int replay_w = 0, replay_h = 0, replay_mode = 0, replay_nframes = 0;   // what `replay` holds (0 = nothing)
  // A batch coded in parts (sjpeg_hip_encode_batch_src): the statistics / replay calls of a part address the
  // kept blocks of frames [replay_first, replay_first + nframes) of a buffer for replay_total frames.
  int replay_first = 0, replay_total = 0;
  // ... and when the call runs a histogram pass in front of its statistics pass (the adaptive methods with
  // optimized Huffman tables), the histogram pass leaves every block's DCT coefficients in `replay` and the
  // statistics pass starts from them (kKindStatsCoef): set by sjpeg_hip_encode_batch_src for the length of the call
  bool coefs_keep = false, coefs_use = false;
  // ... and their per-workgroup partial statistics likewise live in a buffer for replay_total frames, summed on
  // `reduce_stream` (behind an event) instead of the call's stream: the sums of one part run under the device
  // pass of the next
  hipStream_t reduce_stream = nullptr;
  hipEvent_t reduce_ev = nullptr;
  // ... and its uploads (a part's tables, headers, header offsets) leave EARLY, on `up_stream`, the moment the host has
  // them -- the call's stream only waits for the event behind the last one (sync_uploads) in front of the kernel that
  // reads them.  In the call's own stream every upload was a copy kernel between two dependent launches plus the marker
  // of its pinned block: ~80 us of an idle device per 32-frame call (rocprofv3 trace, round 5).  What makes it safe:
  // the parts' tables / headers lie side by side (part_first_frame()), and the host only has a part's data once the pass
  // in front -- hence everything older on the call's stream -- is done (the sums / counts it waited for say so).
  hipStream_t up_stream = nullptr;
  hipEvent_t up_wait = nullptr;          // the event behind the last early upload nobody waited for yet (one of stage[].ev)
  // The two streams of a batch in parts, made WITH the engine: a stream made late in the life of a process -- behind
  // the first asynchronous copy, it seems; bench.py's batch lines came a minute in -- shares the hardware queue of
  // an older one, the caller's as it turned out: the side stream's sums then queue behind the next part's pass, the
  // early uploads behind the kernel they should run under, and the call takes what it takes with everything in one
  // stream (1.33 ms against 1.19 for 32 4K frames; not the number of hardware queues -- GPU_MAX_HW_QUEUES 8 / 16 the
  // same --, not their priority -- the greatest made it 1.46).  Streams made when the engine is, early, keep queues of
  // their own (bisected in bench.py, round 5; profiles/HISTORY.md).
  hipStream_t batch_side = nullptr, batch_up = nullptr;
  // LANES of the batch path (round 6): a large default-parameter batch is cut into jobs of about eight 4K frames, every
  // job a complete histogram -> statistics -> encode sequence of its own, and the jobs run on up to four streams at once
  // -- lane 0 is this engine on the caller's stream, lanes 1..3 are child engines (their own scratch) on streams the
  // engine owns (batch_side and batch_up, made WITH the engine, see above; the fourth on demand: two lanes are the default).  The three passes have different
  // bottlenecks -- the histogram kind is latency-bound with the VALU 60 % busy, the statistics kind is the memory's, the
  // replay kind the VALU's --: side by side they fill each other's gaps, one after the other (the two parts of round 5)
  // they cannot.  Measured with independent calls from several host threads first (tools/two_stream_batch.py).
  static constexpr int kLanes = 4;
  sjpeg_hip_engine* lane[kLanes] = {nullptr, nullptr, nullptr, nullptr};   // [0] unused (this engine)
  hipStream_t batch_lane3 = nullptr;
  hipEvent_t lane_in = nullptr, lane_done[kLanes] = {nullptr, nullptr, nullptr, nullptr};
  bool is_lane = false;                // a child engine: no streams or lanes of its own
  DevBuf<unsigned long long> seg_off, chunk_off, stamps;
  DevBuf<uint32_t> hdr_off;
  bool want_stamps = false;
  int stamp_mode = 0;              // SJPEG_HIP_STAMPS, parsed ONCE when the engine is made (1 cycle counter, 2 device clock, 3 + segment bits)
  int last_nseg = 0, last_nframes = 0;   // geometry of the last encode call (entropy_bits)
  size_t stamps_n = 0;
  bool timing = false;
  int ablate = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ev_valid = false;
  // Pipelined mode (sjpeg_hip_engine_set_pipelined): K1 on the caller's stream, K2..K5 on the
  // engine's own stream, two sets of segment buffers -- the stitch of call i (HBM-bound, little
  // ALU) runs under the K1 of call i + 1 (ALU-bound, little HBM).
  // the scratch buffers are shared by all calls: a call on another stream than the previous one
  // waits for that one's work first
  hipStream_t last_stream = nullptr;
  bool last_stream_valid = false;
  hipEvent_t cross_ev = nullptr;
  bool pipelined = false;
  size_t scratch_limit = static_cast<size_t>(16) << 30;   // segment scratch of ONE launch (SJPEG_HIP_SCRATCH_LIMIT_BYTES): larger batches go in several
  hipStream_t side = nullptr;
  DevBuf<uint32_t> seg_words2, seg_nbits2, pool2, pool_ctr2, seg_xbase2;
  // K4 leaves the pool counters of the frames it saw at zero, so an encode call only clears them
  // itself when the previous user of the set did not get that far (or was larger / another buffer)
  const void* ctr_clean_at[2] = {nullptr, nullptr};
  size_t ctr_clean_n[2] = {0, 0};
  int set = 0;                                   // buffer set of the NEXT call
  hipEvent_t k1_done = nullptr, side_done = nullptr, k3_done[2] = {nullptr, nullptr};
  bool k3_pending[2] = {false, false}, side_pending = false;
  // side_done is recorded LAZILY, by whoever is about to wait on it (side_mark): an event record is a packet in the
  // queue and about 5 us of host time, and a loop of pipelined calls needs none -- one frame per call was bound by the
  // HOST at five event calls per call (36-46 us against 35 of device time, `tools/one_frame_piped.py`)
  bool side_recorded = false;
};

namespace {

// the quantizer part of DevTables (its first KiB)
void digest_quant(const sjpeg_hip_scan_tables* t, uint4 (*q)[32]) {
  for (int c = 0; c < 2; ++c) {
    for (int j = 0; j < 64; ++j) {
      uint4& e = q[c][j >> 1];
      const uint32_t iq = t->iquant[c][j], biq = static_cast<uint32_t>(t->bias[c][j]) * iq;
      const uint32_t qv = t->quant[c][j];
      if ((j & 1) == 0) { e.x = iq; e.y = biq; e.w = qv; } else { e.x |= iq << 16; e.z = biq; e.w |= qv << 16; }
    }
  }
}

void digest_tables(const sjpeg_hip_scan_tables* t, DevTables* d) {
  digest_quant(t, d->q);
  memset(d->pad_a, 0, sizeof(d->pad_a));
  memcpy(d->dc, t->dc_codes, sizeof(d->dc));
  memcpy(d->ac, t->ac_codes, sizeof(d->ac));
  memcpy(d->tlen, t->trellis_len, sizeof(d->tlen));
  // the lean walk's merged code words and the level bound under which it is provably in place
  // (scan_segments.h, P3): a symbol of run 0 must not emit more than 16 bits
  for (int c = 0; c < 2; ++c) {
    int n_safe = 0;
    for (int n = 1; n <= 10; ++n) {
      const uint32_t len0 = t->ac_codes[c][n] & 0xffu;            // symbol (run 0, size n); 0 = not in the table
      if (len0 + n > 16u) break;
      n_safe = n;
    }
    {
      const uint32_t zrl = t->ac_codes[c][0xf0];
      const uint32_t zl = zrl & 0xffu;
      unsigned long long v = 0;
      d->zrlpat[c][0] = make_uint4(0, 0, 0, 0);
      for (uint32_t k = 1; k <= 3; ++k) {
        v = (v << zl) | (zrl >> 16);
        const unsigned long long left = (zl == 0u) ? 0ull : v << (64u - k * zl);
        d->zrlpat[c][k] = make_uint4(static_cast<uint32_t>(left >> 32), static_cast<uint32_t>(left), k * zl, 0u);
      }
    }
    const uint32_t bad = ~((1u << n_safe) - 1u) & 0x7fffu;
    d->safe_mask[c] = bad | (bad << 16);
    d->eob_zrl[c][0] = t->ac_codes[c][0x00];
    d->eob_zrl[c][1] = t->ac_codes[c][0xf0];
    for (int n = 1; n <= 10; ++n) {
      for (int run = 0; run < 16; ++run) {
        const uint32_t cw = t->ac_codes[c][(run << 4) | n];
        const uint32_t len = cw & 0xffu, code = cw >> 16;
        d->acm[c][run][10 - n] = len == 0u ? 0u : ((code << n) | ((len + n) << 27));
      }
    }
  }
}

template <int KIND, int SRC>
int launch_scan_src(int mode, dim3 grid, hipStream_t st, const ScanArgs& a) {
  static const int kLdsPad = getenv("SJPEG_HIP_LDS_PAD") ? atoi(getenv("SJPEG_HIP_LDS_PAD")) : 0;  // occupancy experiments
  const int lds = kLdsPad;                        // (the kernel's own LDS is a static array)
  switch (mode) {
    case SJPEG_HIP_YUV420:
      hipLaunchKernelGGL((scan_segments<SJPEG_HIP_YUV420, KIND, SRC>), grid, dim3(kScanThreads), lds, st, a);
      break;
    case SJPEG_HIP_YUV444:
      hipLaunchKernelGGL((scan_segments<SJPEG_HIP_YUV444, KIND, SRC>), grid, dim3(kScanThreads), lds, st, a);
      break;
    default:
      hipLaunchKernelGGL((scan_segments<SJPEG_HIP_YUV400, KIND, SRC>), grid, dim3(kScanThreads), lds, st, a);
      break;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int KIND>
int launch_scan(int mode, int src_class, dim3 grid, hipStream_t st, const ScanArgs& a) {
  switch (src_class) {
    case kSrcRgb24: return launch_scan_src<KIND, kSrcRgb24>(mode, grid, st, a);
    case kSrcRgbx32: return launch_scan_src<KIND, kSrcRgbx32>(mode, grid, st, a);
    default: return launch_scan_src<KIND, kSrcPlanes>(mode, grid, st, a);
  }
}

sjpeg_hip_source rgb_source(const void* d_rgb, int64_t row_stride, int64_t frame_stride) {
  sjpeg_hip_source s;
  memset(&s, 0, sizeof(s));
  s.format = SJPEG_HIP_SRC_RGB;
  s.plane[0] = d_rgb; s.row_stride[0] = row_stride; s.frame_stride[0] = frame_stride;
  return s;
}

// the event behind everything the engine has put on its own (stitch) stream so far
int side_mark(sjpeg_hip_engine* e) {
  if (e->side_pending && !e->side_recorded) {
    HIP_TRY(hipEventRecord(e->side_done, e->side));
    e->side_recorded = true;
  }
  return 0;
}

// Orders this call after everything the engine was asked to do on another stream.
int order_on_stream(sjpeg_hip_engine* e, hipStream_t st) {
  if (e->last_stream_valid && e->last_stream != st) {
    // The previous call's stream belongs to the caller and may be gone by now (sjpeg_hip.h asks for
    // it to outlive the engine's next call, but a destroyed handle must not wedge the engine): if
    // the event hand-over fails, wait for the whole device instead and carry on.
    bool ordered = (e->cross_ev != nullptr) || hipEventCreateWithFlags(&e->cross_ev, hipEventDisableTiming) == hipSuccess;
    ordered = ordered && hipEventRecord(e->cross_ev, e->last_stream) == hipSuccess &&
              hipStreamWaitEvent(st, e->cross_ev, 0) == hipSuccess;
    if (ordered && e->side_pending) ordered = side_mark(e) == 0 && hipStreamWaitEvent(st, e->side_done, 0) == hipSuccess;
    if (!ordered) {
      (void)hipGetLastError();                     // clear the sticky error of the stale handle
      e->last_stream_valid = false;
      e->tables_held_at = nullptr; e->header_held_at = nullptr;   // (their streams may be the stale one)
      HIP_TRY(hipDeviceSynchronize());
    }
  }
  e->last_stream = st;
  e->last_stream_valid = true;
  return 0;
}

// The copy of a staged upload: a kernel reads the engine's pinned block over the bus and writes device memory.  Not
// hipMemcpyAsync: that call took 6-9 ms ONCE per process -- the first copy it is asked for while kernels of the same
// stream are still running (the second default-parameter batch call of a process; now and then a later one) --, the
// "6-8 ms call some tens of calls in" of round 4, named in round 5 by the marks of SJPEG_HIP_BATCH_DEBUG=2
// (profiles/r05/slow_call.txt: `upload: hipMemcpyAsync 6920 us`).  A kernel stays on the stream's own queue: no copy
// engine, no hand-over between queues, and a launch costs the host less than the copy call did.
__global__ __launch_bounds__(256) void stage_copy_kernel(uint8_t* dst, const uint8_t* src, size_t bytes) {
  const size_t n16 = bytes >> 4;
  const uint4* const s4 = reinterpret_cast<const uint4*>(src);
  uint4* const d4 = reinterpret_cast<uint4*>(dst);
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * 256) d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15u)) dst[(n16 << 4) + threadIdx.x] = src[(n16 << 4) + threadIdx.x];
}

// bytes between two places the DEVICE can address (device memory, mapped pinned host memory), by that kernel; the
// runtime's copy where the addresses do not allow 16-byte accesses
int copy_by_kernel(void* dst, const void* src, size_t bytes, hipStream_t st, hipMemcpyKind fallback_kind, const void* fallback_src, void* fallback_dst) {
  if (bytes == 0) return 0;
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) != 0 || dst == nullptr || src == nullptr) {
    HIP_TRY(hipMemcpyAsync(fallback_dst, fallback_src, bytes, fallback_kind, st));
    return 0;
  }
  const size_t n16 = bytes >> 4;
  const unsigned blocks = static_cast<unsigned>(n16 >= 256 * 64 ? 64 : (n16 + 255) / 256 + (n16 == 0 ? 1 : 0));
  hipLaunchKernelGGL(stage_copy_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), bytes);
  HIP_TRY(hipGetLastError());
  return 0;
}

// dst <- src on the stream, through one of the engine's pinned blocks when the copy is not tiny (see sjpeg_hip_engine::stage)
int upload(sjpeg_hip_engine* e, void* dst, const void* src, size_t bytes, hipStream_t st) {
  // (for a very large one -- the per-frame tables of a batch of tens of thousands of frames -- the runtime's pinning is
  // small beside the copy, and four pinned blocks of that size would not be)
  if (bytes == 0) return 0;
  if (bytes > (static_cast<size_t>(8) << 20)) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return 0;
  }
  sjpeg_hip_engine::Stage& s = e->stage[e->stage_next];
  e->stage_next = (e->stage_next + 1) & 3;
  dbg_mark("upload: begin");
  if (s.busy) { HIP_TRY(hipEventSynchronize(s.ev)); s.busy = false; }       // (four uploads ago: long done)
  dbg_mark("upload: block free");
  if (s.cap < bytes) {
    // ALL four blocks grow together, to a power of two: hipHostFree + hipHostMalloc of one block takes milliseconds,
    // and blocks that grew one by one -- whenever the ring brought a small block to a large upload -- were the
    // "6-8 ms call some tens of calls in" of the default-parameter batch path (round 4; named in round 5 by the
    // library's own host timeline: the second call's statistics launch, profiles/r05/slow_call.txt).  Now the first
    // upload of a size class pays for the four, once.
    size_t want = static_cast<size_t>(256) << 10;
    while (want < bytes) want <<= 1;
    for (auto& g : e->stage) {
      if (g.cap >= want) continue;
      if (g.busy) { HIP_TRY(hipEventSynchronize(g.ev)); g.busy = false; }
      if (g.p) (void)hipHostFree(g.p);
      g.p = nullptr; g.dp = nullptr; g.cap = 0;
      if (hipHostMalloc(&g.p, want, hipHostMallocMapped) != hipSuccess) { g.p = nullptr; return fail(SJPEG_HIP_ENOMEM, "hipHostMalloc(upload block) failed"); }
      g.cap = want;
      // (the address the device reads the block at; without one the copy falls back to the runtime's)
      if (hipHostGetDevicePointer(&g.dp, g.p, 0) != hipSuccess) { (void)hipGetLastError(); g.dp = nullptr; }
    }
  }
  if (s.ev == nullptr && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { s.ev = nullptr; return fail(SJPEG_HIP_ERUNTIME, "hipEventCreate failed"); }
  dbg_mark("upload: block sized");
  memcpy(s.p, src, bytes);
  dbg_mark("upload: memcpy");
  hipStream_t const us = e->up_stream != nullptr ? e->up_stream : st;
  if (int rc = copy_by_kernel(dst, s.dp, bytes, us, hipMemcpyHostToDevice, s.p, dst)) return rc;
  dbg_mark("upload: copy launched");
  HIP_TRY(hipEventRecord(s.ev, us));
  dbg_mark("upload: hipEventRecord");
  s.busy = true;
  if (e->up_stream != nullptr) e->up_wait = s.ev;
  return 0;
}

// the call's stream waits for the early uploads (sjpeg_hip_engine::up_stream) in front of the kernel that reads them
int sync_uploads(sjpeg_hip_engine* e, hipStream_t st) {
  if (e->up_wait != nullptr) {
    HIP_TRY(hipStreamWaitEvent(st, e->up_wait, 0));
    e->up_wait = nullptr;
  }
  return 0;
}

// A batch coded in parts: where a part's per-frame tables / headers start in the engine's buffers (frames), and how many
// frames those buffers are for -- the parts' uploads must not land on each other.
inline int part_first_frame(const sjpeg_hip_engine* e) { return e->replay_total > 0 ? e->replay_first : 0; }
inline int part_total_frames(const sjpeg_hip_engine* e, int nframes) { return e->replay_total > 0 ? e->replay_total : nframes; }

// Segment scratch of an encode call, sized from the bytes the caller gives every frame (out_stride)
// instead of for the worst case: a frame's un-stuffed stream is never longer than its stuffed one, so
// `budget` words hold it whenever the frame fits its output slot -- and a frame that does not fit
// reports size 0 anyway.  Every segment has a slot of about three quarters of its share of the budget (the
// ordinary segment fits; K3 reads it without indirection); what a longer segment has beyond that, and
// the rows of the checked walk (scan_segments.h), come out of a per-frame pool.  Both are at most as
// long as the stream, hence 2 x budget.  budget_bytes == SIZE_MAX: worst case (bands, which have no
// output slot to go by).
struct SegPlan { uint32_t slot_words, pool_words; size_t ubuf_words; };
SegPlan seg_plan(const FrameGeo& g, size_t budget_bytes) {
  const size_t worst_total = static_cast<size_t>(g.nseg) * g.slot_words;
  size_t budget = budget_bytes == SIZE_MAX ? worst_total : budget_bytes / 4 + 16;
  if (budget > worst_total) budget = worst_total;
  SegPlan p;
  // (three quarters of a segment's share: at half of it the segments of an 8K 4:4:4 q90 frame -- 0.77 B per
  // pixel against slots of 0.75 -- went through the pool, K3's slow path, every other time)
  size_t sw = (budget * 3 / static_cast<size_t>(g.nseg) / 4 + 63) & ~size_t(63);
  const size_t floor_words = g.slot_words < 1024u ? g.slot_words : 1024u;
  if (sw < floor_words) sw = floor_words;
  if (sw > g.slot_words || budget >= worst_total) sw = g.slot_words;   // (worst-case budget: no segment is ever longer than its slot)
  p.slot_words = static_cast<uint32_t>(sw);
  size_t pool = (sw == g.slot_words ? budget : 2 * budget) + 64;
  if (pool > 0xfffffff0u) pool = 0xfffffff0u;      // (word offsets are 32 bit: 16 GiB per frame)
  p.pool_words = static_cast<uint32_t>(pool);
  p.ubuf_words = (budget + kChunkWords + 3) & ~size_t(3);
  return p;
}

int prepare_scan(sjpeg_hip_engine* e, const sjpeg_hip_source* src,
                 int W, int H, int mode, int nframes, const sjpeg_hip_scan_tables* tables,
                 hipStream_t st, FrameGeo* g, ScanArgs* a, int* src_class, bool per_frame_tables = false,
                 bool piped_encode = false, size_t seg_budget_bytes = 0 /* 0: not an encode, no segment scratch */,
                 SegPlan* plan_out = nullptr, bool no_tables = false /* the histogram kind reads none */) {
  if (e == nullptr || src == nullptr || src->plane[0] == nullptr || tables == nullptr || nframes <= 0) {
    return fail(SJPEG_HIP_EINVAL, "null argument or nframes <= 0");
  }
  if (!frame_geo(W, H, mode, g)) return fail(SJPEG_HIP_EINVAL, "bad dimensions or yuv_mode");
  // per-plane minimum row size and the colour mode each layout implies
  // (reference argument checks: src/api.cc:35-36,205-206,260; src/encoders.cc:352-355,427-432)
  const int64_t cw = (W + 1) / 2;
  int64_t need[3] = {0, 0, 0};
  int nplanes = 1, implied = 0;
  memset(a, 0, sizeof(*a));
  switch (src->format) {
    case SJPEG_HIP_SRC_RGB: need[0] = 3ll * W; *src_class = kSrcRgb24; break;
    case SJPEG_HIP_SRC_BGRA: need[0] = 4ll * W; *src_class = kSrcRgbx32; a->rsh = 16; a->bsh = 0; break;
    case SJPEG_HIP_SRC_RGBA: need[0] = 4ll * W; *src_class = kSrcRgbx32; a->rsh = 0; a->bsh = 16; break;
    case SJPEG_HIP_SRC_GRAY: need[0] = W; *src_class = kSrcPlanes; implied = SJPEG_HIP_YUV400; break;
    case SJPEG_HIP_SRC_YUV444:
      need[0] = need[1] = need[2] = W; nplanes = 3; *src_class = kSrcPlanes; implied = SJPEG_HIP_YUV444;
      a->cstep = 1;
      break;
    case SJPEG_HIP_SRC_YUV420:
      need[0] = W; need[1] = need[2] = cw; nplanes = 3; *src_class = kSrcPlanes; implied = SJPEG_HIP_YUV420;
      a->cstep = 1;
      break;
    case SJPEG_HIP_SRC_NV12:
    case SJPEG_HIP_SRC_NV21:
      need[0] = W; need[1] = 2 * cw; nplanes = 2; *src_class = kSrcPlanes; implied = SJPEG_HIP_YUV420;
      a->cstep = 2;
      a->uoff = (src->format == SJPEG_HIP_SRC_NV12) ? 0 : 1;
      a->voff = 1 - a->uoff;
      break;
    default: return fail(SJPEG_HIP_EINVAL, "unknown source format");
  }
  if (implied != 0 && mode != implied) return fail(SJPEG_HIP_EINVAL, "yuv_mode does not match the source format");
  for (int i = 0; i < nplanes; ++i) {
    if (src->plane[i] == nullptr) return fail(SJPEG_HIP_EINVAL, "null plane pointer");
    const int64_t st_abs = src->row_stride[i] < 0 ? -src->row_stride[i] : src->row_stride[i];
    if (st_abs < need[i]) return fail(SJPEG_HIP_EINVAL, "|row_stride| smaller than a row of the plane");
    a->plane[i] = static_cast<const uint8_t*>(src->plane[i]);
    a->row_stride[i] = src->row_stride[i];
    a->frame_stride[i] = src->frame_stride[i];
  }
  if (nplanes == 2) {          // interleaved chroma: U and V walk the same plane
    a->plane[2] = a->plane[1]; a->row_stride[2] = a->row_stride[1]; a->frame_stride[2] = a->frame_stride[1];
  }
  if (nframes > 65535) return fail(SJPEG_HIP_EINVAL, "nframes > 65535");
  dbg_mark("prepare: begin");
  HIP_TRY(hipSetDevice(e->device));
  if (int rc0 = order_on_stream(e, st)) return rc0;
  dbg_mark("prepare: ordered");
  // anything but a pipelined encode shares buffers with the stitch still running on the engine's stream
  if (e->side_pending && !piped_encode) {
    if (int rcm = side_mark(e)) return rcm;
    HIP_TRY(hipStreamWaitEvent(st, e->side_done, 0));
  }
  int rc;
  const int ntab = per_frame_tables ? nframes : 1;
  // (per-frame tables of a batch's part lie behind those of the parts in front)
  const int tab_first = per_frame_tables ? part_first_frame(e) : 0;
  if (per_frame_tables && tab_first + nframes > part_total_frames(e, nframes)) return fail(SJPEG_HIP_EINVAL, "part outside the batch");
  if ((rc = e->tables.ensure(per_frame_tables ? part_total_frames(e, nframes) : 1))) return rc;
  const size_t total_segs = static_cast<size_t>(nframes) * g->nseg;
  if ((rc = e->seg_nbits.ensure(total_segs))) return rc;
  SegPlan plan = {0, 0, 0};
  if (seg_budget_bytes != 0) {
    plan = seg_plan(*g, seg_budget_bytes);
    if ((rc = e->seg_words.ensure(total_segs * plan.slot_words))) return rc;
    if ((rc = e->pool.ensure(static_cast<size_t>(nframes) * plan.pool_words))) return rc;
    if ((rc = e->pool_ctr.ensure(static_cast<size_t>(nframes) * 2))) return rc;
    if ((rc = e->seg_xbase.ensure(total_segs))) return rc;
  }
  if (plan_out != nullptr) *plan_out = plan;
  dbg_mark("prepare: buffers");
  if (!no_tables) {
    // (pageable source: the copy has left the host buffer when the call returns)
    std::vector<DevTables> host_tables(ntab);
    for (int i = 0; i < ntab; ++i) digest_tables(tables + i, &host_tables[i]);
    // (early uploads are never skipped: what the buffer holds was put there by another stream's order)
    const bool held = e->up_stream == nullptr && tab_first == 0 && e->tables_held_at == e->tables.p && e->tables_stream == st &&
                      e->tables_held.size() == static_cast<size_t>(ntab) &&
                      memcmp(e->tables_held.data(), host_tables.data(), sizeof(DevTables) * ntab) == 0;
    dbg_mark("prepare: tables digested");
    if (!held) {
      e->tables_held_at = nullptr;
      if (int rcu = upload(e, e->tables.p + tab_first, host_tables.data(), sizeof(DevTables) * ntab, st)) return rcu;
      dbg_mark("prepare: tables uploaded");
      if (ntab <= 16 && e->up_stream == nullptr && tab_first == 0) {   // (a big batch of per-frame tables is not worth holding)
        e->tables_held.swap(host_tables);
        e->tables_held_at = e->tables.p; e->tables_stream = st;
      } else {
        e->tables_held.clear();
      }
    }
  }
  a->W = W; a->H = H; a->mb_w = g->mb_w; a->n_mcus = g->n_mcus; a->nseg = g->nseg;
  a->seg_first = 0;
  a->rst = (tables->flags & SJPEG_HIP_RESTART_MARKERS) ? 1 : 0;
  a->has_clip = (W % g->px != 0) || (H % g->px != 0);
  a->tables = e->tables.p + tab_first;
  a->tables_stride = per_frame_tables ? 1 : 0;
  a->seg_words = e->seg_words.p;
  a->pool = e->pool.p; a->pool_words = plan.pool_words; a->pool_ctr = e->pool_ctr.p; a->seg_xbase = e->seg_xbase.p;
  a->replay = nullptr;
  a->slot_words = plan.slot_words;
  a->seg_nbits = e->seg_nbits.p;
  a->coeffs = nullptr;
  a->partial = nullptr;
  a->ablate = e->ablate;
  a->stamps = nullptr;
  if (e->want_stamps) {
    if (int rc2 = e->stamps.ensure(total_segs * 8)) return rc2;
    a->stamps = e->stamps.p;
    // SJPEG_HIP_STAMPS=2: the 100 MHz real-time counter (one clock for the whole device: when workgroups start
    // and end relative to each other) instead of the shader-clock cycle counter (per CU: phase durations)
    a->stamp_real = e->stamp_mode == 2 ? 1 : (e->stamp_mode == 3 ? 2 : 0);
    e->stamps_n = total_segs * 8;
  }
  return 0;
}

}  // namespace

extern "C" {

int sjpeg_hip_abi_version(void) { return SJPEG_HIP_ABI_VERSION; }

const char* sjpeg_hip_last_error(void) { return g_last_error.c_str(); }
// (the other translation units of the C-ABI report through the same thread-local text)
__attribute__((visibility("hidden"))) void sjpeg_hip_internal_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }

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
  if (prop.multiProcessorCount > 0) e->cu_count = prop.multiProcessorCount;
  // (two streams, as in round 5, and not one more: the device has four hardware queues by default, and with a third
  // engine-owned stream made here the stitch stream of the pipelined mode -- made later, by set_pipelined -- shared the
  // caller's queue: the headline step went from 0.889 to 1.096 ms, K1 and the stitch one after the other.  The fourth
  // lane of the batch path, an experiment knob, makes its stream when it is asked for.)
  if (hipStreamCreateWithFlags(&e->batch_side, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&e->batch_up, hipStreamNonBlocking) != hipSuccess) {
    if (e->batch_side) (void)hipStreamDestroy(e->batch_side);
    delete e;
    return fail(SJPEG_HIP_ERUNTIME, "hipStreamCreate failed");
  }
  if (const char* ab = getenv("SJPEG_HIP_ABLATE")) {                       // profiling / race-stress aid only
    e->ablate = atoi(ab);
    if (e->ablate != 0) {
      fprintf(stderr, "sjpeg_amd: SJPEG_HIP_ABLATE=%d is set: kernels skip phases or stall waves on purpose, "
                      "the output of this engine is NOT valid JPEG data\n", e->ablate);
    }
  }
  if (const char* hs = getenv("SJPEG_HIP_HISTO_SLOTS")) e->histo_slots = atoi(hs);
  if (const char* sm = getenv("SJPEG_HIP_STAMPS")) { e->want_stamps = true; e->stamp_mode = atoi(sm); }
  if (const char* sl = getenv("SJPEG_HIP_SCRATCH_LIMIT_BYTES")) { const long long v = atoll(sl); if (v > 0) e->scratch_limit = static_cast<size_t>(v); }
  *engine = e;
  return 0;
}

void sjpeg_hip_engine_destroy(sjpeg_hip_engine* e) {
  if (e == nullptr) return;
  (void)hipSetDevice(e->device);
  if (!e->is_lane) {
    for (hipStream_t bs : {e->batch_side, e->batch_up, e->batch_lane3}) if (bs) (void)hipStreamSynchronize(bs);   // (the lanes' work)
    for (auto& l : e->lane) { if (l) sjpeg_hip_engine_destroy(l); l = nullptr; }
    for (auto& ev : e->lane_done) if (ev) (void)hipEventDestroy(ev);
    if (e->lane_in) (void)hipEventDestroy(e->lane_in);
  }
  e->tables.release(); e->header.release(); e->seg_words.release(); e->seg_nbits.release(); e->pool.release(); e->pool_ctr.release(); e->seg_xbase.release(); e->replay.release();
  e->ubuf.release(); e->chunk_ff.release(); e->partial.release(); e->seg_off.release(); e->chunk_off.release(); e->hdr_off.release(); e->stamps.release();
  e->frame_flags.release();
  for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& sg : e->stage) {
    if (sg.busy) (void)hipEventSynchronize(sg.ev);
    if (sg.ev) (void)hipEventDestroy(sg.ev);
    if (sg.p) (void)hipHostFree(sg.p);
  }
  e->seg_words2.release(); e->seg_nbits2.release(); e->pool2.release(); e->pool_ctr2.release(); e->seg_xbase2.release();
  if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
  for (hipStream_t bs : {e->batch_side, e->batch_up, e->batch_lane3}) if (bs) { (void)hipStreamSynchronize(bs); (void)hipStreamDestroy(bs); }
  for (hipEvent_t ev : {e->k1_done, e->side_done, e->k3_done[0], e->k3_done[1], e->cross_ev}) if (ev) (void)hipEventDestroy(ev);
  delete e;
}

int sjpeg_hip_engine_trim(sjpeg_hip_engine* e) {
  if (e == nullptr) return fail(SJPEG_HIP_EINVAL, "engine == NULL");
  HIP_TRY(hipSetDevice(e->device));
  // the engine does not own the streams its calls ran on: wait for the whole device
  HIP_TRY(hipDeviceSynchronize());
  for (auto& l : e->lane) { if (l != nullptr) { if (int rcl = sjpeg_hip_engine_trim(l)) return rcl; } }
  e->seg_words.release(); e->seg_nbits.release(); e->pool.release(); e->pool_ctr.release(); e->seg_xbase.release();
  e->seg_words2.release(); e->seg_nbits2.release(); e->pool2.release(); e->pool_ctr2.release(); e->seg_xbase2.release();
  e->ubuf.release(); e->chunk_ff.release(); e->partial.release(); e->replay.release();
  e->seg_off.release(); e->chunk_off.release(); e->hdr_off.release(); e->stamps.release();
  e->frame_flags.release();
  e->tables.release(); e->header.release();        // (per-frame tables of a large batch are scratch like the rest)
  for (auto& sg : e->stage) {                      // ... and so are the pinned blocks they were uploaded through
    // (their copies are done: the device was waited for above; the event is waited for all the same, so that the
    // block's life does not hang on that one line)
    if (sg.busy && sg.ev) (void)hipEventSynchronize(sg.ev);
    if (sg.p) (void)hipHostFree(sg.p);
    sg.p = nullptr; sg.dp = nullptr; sg.cap = 0; sg.busy = false;
  }
  e->tables_held_at = nullptr; e->header_held_at = nullptr;
  e->replay_w = e->replay_h = e->replay_mode = e->replay_nframes = 0;
  e->stamps_n = 0;
  e->last_nseg = e->last_nframes = 0;
  e->ctr_clean_at[0] = e->ctr_clean_at[1] = nullptr;
  e->ctr_clean_n[0] = e->ctr_clean_n[1] = 0;
  e->k3_pending[0] = e->k3_pending[1] = e->side_pending = e->side_recorded = false;
  e->last_stream_valid = false;
  e->ev_valid = false;
  return 0;
}

int sjpeg_hip_engine_set_pipelined(sjpeg_hip_engine* e, int on) {
  if (e == nullptr) return fail(SJPEG_HIP_EINVAL, "engine == NULL");
  HIP_TRY(hipSetDevice(e->device));
  if (on && e->side == nullptr) {
    {
      // the stitch stream gets the device's least priority: its kernels are meant to take the issue
      // slots K1 leaves idle, not K1's (measured: step -0.3 % at best, profiles/HISTORY.md)
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
          hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, least) != hipSuccess) {
        (void)hipGetLastError();
        e->side = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
      }
    }
    for (hipEvent_t* ev : {&e->k1_done, &e->side_done, &e->k3_done[0], &e->k3_done[1]}) {
      HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
  }
  if (!on && e->pipelined) {                       // drain: from here on the caller's stream orders everything again
    HIP_TRY(hipStreamSynchronize(e->side));
    e->k3_pending[0] = e->k3_pending[1] = e->side_pending = e->side_recorded = false;
  }
  e->pipelined = on != 0;
  e->header_held_at = nullptr;                    // the header buffer changes streams
  return 0;
}

int sjpeg_hip_engine_wait(sjpeg_hip_engine* e, void* stream) {
  if (e == nullptr) return fail(SJPEG_HIP_EINVAL, "engine == NULL");
  if (e->side_pending) {
    HIP_TRY(hipSetDevice(e->device));
    if (int rcm = side_mark(e)) return rcm;
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), e->side_done, 0));
  }
  return 0;
}

size_t sjpeg_hip_frame_bound(int width, int height, int yuv_mode, size_t header_size) {
  FrameGeo g;
  if (!frame_geo(width, height, yuv_mode, &g)) return 0;
  // un-stuffed worst case = slots; stuffing at most doubles it
  const size_t unstuffed = static_cast<size_t>(g.nseg) * g.slot_words * 4;
  // (a multiple of 16: the default out_stride of the bindings is what sjpeg_hip_compact_streams accepts)
  return (header_size + 2 * unstuffed + 2 + 64 + 15) & ~size_t(15);
}

// profiling only (not in the public header): copies the per-workgroup cycle stamps of the last scan
size_t sjpeg_hip_debug_stamps(sjpeg_hip_engine* e, unsigned long long* out, size_t cap) {
  if (e == nullptr || !e->want_stamps || e->stamps.p == nullptr) return 0;
  const size_t n = e->stamps_n < cap ? e->stamps_n : cap;
  if (hipMemcpy(out, e->stamps.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
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

size_t sjpeg_hip_engine_scratch_bytes(sjpeg_hip_engine* e) {
  if (e == nullptr) return 0;
  auto b = [](const auto& buf) { return buf.cap * sizeof(*buf.p); };
  size_t lanes = 0;
  for (auto* l : e->lane) if (l != nullptr) lanes += sjpeg_hip_engine_scratch_bytes(l);
  return lanes + b(e->tables) + b(e->header) + b(e->seg_words) + b(e->seg_nbits) + b(e->pool) + b(e->pool_ctr) + b(e->seg_xbase) +
         b(e->ubuf) + b(e->chunk_ff) + b(e->partial) + b(e->replay) + b(e->seg_off) + b(e->chunk_off) + b(e->stamps) +
         b(e->hdr_off) + b(e->seg_words2) + b(e->seg_nbits2) + b(e->pool2) + b(e->pool_ctr2) + b(e->seg_xbase2);
}

int sjpeg_hip_scan_coeffs_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                              int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                              int16_t* d_coeffs, void* stream) {
  if (d_coeffs == nullptr) return fail(SJPEG_HIP_EINVAL, "d_coeffs == NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameGeo g;
  ScanArgs a;
  int cls = 0;
  const int rc = prepare_scan(e, src, width, height, yuv_mode, nframes, tables, st, &g, &a, &cls);
  if (rc) return rc;
  a.coeffs = d_coeffs;
  return launch_scan<kKindTap>(yuv_mode, cls, dim3(g.nseg, nframes), st, a);
}

int sjpeg_hip_scan_coeffs(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride,
                          int64_t frame_stride, int width, int height, int yuv_mode, int nframes,
                          const sjpeg_hip_scan_tables* tables, int16_t* d_coeffs, void* stream) {
  const sjpeg_hip_source s = rgb_source(d_rgb, row_stride, frame_stride);
  return sjpeg_hip_scan_coeffs_src(e, &s, width, height, yuv_mode, nframes, tables, d_coeffs, stream);
}

static int scan_statistics(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                           int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                           bool histogram, uint32_t* d_out, void* stream, bool per_frame_tables = false) {
  if (d_out == nullptr) return fail(SJPEG_HIP_EINVAL, "output pointer == NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  sjpeg_hip_scan_tables dummy;
  if (tables == nullptr) {                        // the histogram does not depend on any table
    memset(&dummy, 0, sizeof(dummy));
    tables = &dummy;
  }
  FrameGeo g;
  ScanArgs a;
  int cls = 0;
  int rc = prepare_scan(e, src, width, height, yuv_mode, nframes, tables, st, &g, &a, &cls, per_frame_tables, false, 0, nullptr, histogram);
  if (rc) return rc;
  const int words = histogram ? kHistoWords : kStatsWords;
  const int part_total = e->replay_total > 0 ? e->replay_total : nframes;
  const int part_first = e->replay_total > 0 ? e->replay_first : 0;
  if (part_first < 0 || part_first + nframes > part_total) return fail(SJPEG_HIP_EINVAL, "part outside the batch");
  // The histogram kind is persistent: `groups` workgroups per frame walk the frame's segments and leave one partial
  // each (kHistoPartialWords).  As many as run at once, in whole trips -- 16 4K frames of 791 segments on 768 slots: 17
  // trips, 47 groups a frame, 752 workgroups --; at least nseg / 256 (16-bit counters), at most what the partial
  // buffer's frame stride has room for.
  size_t frame_words = static_cast<size_t>(g.nseg) * words;
  int groups = g.nseg;
  if (histogram) {
    // (THREE per CU although four would fit: the kernel runs no faster with four -- tools/histogram_ablate.py with
    // SJPEG_HIP_HISTO_SLOTS --, and four persistent workgroups of 128 registers hold a SIMD's whole register file until the
    // launch ends: the small kernels of the side stream -- the sums of the previous part, their read-back -- would wait)
    const long long slots = e->histo_slots > 0 ? e->histo_slots : 3ll * e->cu_count;
    const long long trips = std::max(1ll, (static_cast<long long>(g.nseg) * nframes + slots - 1) / slots);
    groups = static_cast<int>((g.nseg + trips - 1) / trips);
    const int min_groups = (g.nseg + kHistoMaxSegsPerGroup - 1) / kHistoMaxSegsPerGroup;
    // (small batches: a partial per segment if need be -- twice the old stride; large ones keep the old stride)
    if (static_cast<size_t>(part_total) * g.nseg * kHistoPartialWords * sizeof(uint32_t) <= (256u << 20)) frame_words = static_cast<size_t>(g.nseg) * kHistoPartialWords;
    frame_words = std::max(frame_words, static_cast<size_t>(min_groups) * kHistoPartialWords);
    groups = std::min(groups, static_cast<int>(frame_words / kHistoPartialWords));
    groups = std::max(groups, min_groups);
  }
  if ((rc = e->partial.ensure(static_cast<size_t>(part_total) * frame_words))) return rc;
  uint32_t* const partial = e->partial.p + static_cast<size_t>(part_first) * frame_words;
  a.partial = partial;
  const bool coefs_in = !histogram && e->coefs_use && (tables->flags & SJPEG_HIP_QUANT_KEEP) && !(tables->flags & SJPEG_HIP_QUANT_TRELLIS);
  if (coefs_in) {
    const int total = e->replay_total > 0 ? e->replay_total : nframes;
    if (e->replay.p == nullptr || e->replay_w != width || e->replay_h != height || e->replay_mode != yuv_mode || e->replay_nframes != total) {
      return fail(SJPEG_HIP_EINVAL, "no histogram pass of these frames left its coefficients behind");
    }
  }
  static const bool force_keep = getenv("SJPEG_HIP_FORCE_COEF_KEEP") != nullptr;   // (measurement aid: tools/histogram_ablate.py)
  if ((!histogram && (tables->flags & SJPEG_HIP_QUANT_KEEP)) || (histogram && (e->coefs_keep || force_keep))) {
    const size_t per_frame = static_cast<size_t>(g.nseg) * kScanThreads * 36;
    const int total = e->replay_total > 0 ? e->replay_total : nframes;
    const int first = e->replay_total > 0 ? e->replay_first : 0;
    if (first < 0 || first + nframes > total) return fail(SJPEG_HIP_EINVAL, "kept-block range outside the batch");
    if ((rc = e->replay.ensure(static_cast<size_t>(total) * per_frame))) return rc;
    a.replay = e->replay.p + static_cast<size_t>(first) * per_frame;
    e->replay_w = width; e->replay_h = height; e->replay_mode = yuv_mode; e->replay_nframes = total;
  }
  dbg_mark("statistics: buffers");
  if ((rc = sync_uploads(e, st))) return rc;
  if (histogram) rc = launch_scan<kKindHisto>(yuv_mode, cls, dim3(groups, nframes), st, a);
  else if (tables != nullptr && (tables->flags & SJPEG_HIP_QUANT_TRELLIS)) rc = launch_scan<kKindStatsTrellis>(yuv_mode, cls, dim3(g.nseg, nframes), st, a);
  else if (coefs_in) rc = launch_scan_src<kKindStatsCoef, kSrcRgb24>(yuv_mode, dim3(g.nseg, nframes), st, a);   // (reads no pixel)
  else rc = launch_scan<kKindStats>(yuv_mode, cls, dim3(g.nseg, nframes), st, a);
  if (rc) return rc;
  dbg_mark("statistics: pass launched");
  // The symbol counts (544 words a frame): slices of the segments meet in the output with device-scope atomics -- small
  // enough for the atomics not to matter, they take the slices they can get (16 frames: 7 us with 32, 25 with 2).
  const int xblocks = histogram ? 32 : (words + kThreads - 1) / kThreads;
  int slices = 4096 / (xblocks * nframes);
  if (slices > 32) slices = 32;
  // (the histogram's partials: a workgroup's four waves share the groups; slices -- and their atomics -- only where one
  // wave would walk more than 64 partials: single large frames)
  if (histogram) slices = std::min(slices, (groups + 255) / 256);
  static const int slices_env = getenv("SJPEG_HIP_REDUCE_SLICES") ? atoi(getenv("SJPEG_HIP_REDUCE_SLICES")) : 0;   // (experiments)
  if (slices_env > 0) slices = slices_env;
  if (slices < 1 || g.nseg < 64) slices = 1;
  if (histogram && groups < slices) slices = groups;
  const dim3 grid(xblocks, nframes, slices);
  hipStream_t rs = st;
  if (e->reduce_stream != nullptr && e->reduce_ev != nullptr) {   // (a batch coded in parts: see sjpeg_hip_encode_batch_src)
    rs = e->reduce_stream;
    HIP_TRY(hipEventRecord(e->reduce_ev, st));
    HIP_TRY(hipStreamWaitEvent(rs, e->reduce_ev, 0));
    dbg_mark("statistics: side stream waits");
  }
  if (!(histogram && slices == 1)) {               // (one slice stores, several add)
    HIP_TRY(hipMemsetAsync(d_out, 0, static_cast<size_t>(nframes) * words * (histogram ? 4 : 1) * sizeof(uint32_t), rs));
    dbg_mark("statistics: memset");
  }
  if (histogram) {
    static_assert(kThreads == kScanThreads, "reduce_partials16: a thread per thread of the scan kernel");
    // (the partials of a launch lie back to back: frame f of this launch at f * groups * kHistoPartialWords)
    hipLaunchKernelGGL(reduce_partials16, grid, dim3(kThreads), 0, rs, reinterpret_cast<const uint4*>(partial), groups, d_out);
  } else {
    hipLaunchKernelGGL(reduce_partials, grid, dim3(kThreads), 0, rs, partial, g.nseg, words, d_out);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int sjpeg_hip_scan_quant_error_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width,
                                   int height, int yuv_mode, int nframes,
                                   const sjpeg_hip_scan_tables* tables, uint64_t* d_err, void* stream) {
  if (d_err == nullptr || tables == nullptr) return fail(SJPEG_HIP_EINVAL, "null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameGeo g;
  ScanArgs a;
  int cls = 0;
  int rc = prepare_scan(e, src, width, height, yuv_mode, nframes, tables, st, &g, &a, &cls);
  if (rc) return rc;
  if ((rc = e->partial.ensure(static_cast<size_t>(nframes) * g.nseg * 2))) return rc;
  a.partial = e->partial.p;
  if ((rc = launch_scan<kKindError>(yuv_mode, cls, dim3(g.nseg, nframes), st, a))) return rc;
  hipLaunchKernelGGL(reduce_error, dim3(nframes), dim3(kThreads), 0, st,
                     reinterpret_cast<const unsigned long long*>(e->partial.p), g.nseg,
                     reinterpret_cast<unsigned long long*>(d_err));
  HIP_TRY(hipGetLastError());
  return 0;
}

int sjpeg_hip_engine_entropy_bits(sjpeg_hip_engine* e, uint64_t* bits, int nframes) {
  if (e == nullptr || bits == nullptr || nframes <= 0) return fail(SJPEG_HIP_EINVAL, "null argument");
  if (e->last_nseg <= 0 || nframes > e->last_nframes) return fail(SJPEG_HIP_EINVAL, "no matching encode call");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  for (int f = 0; f < nframes; ++f) {
    HIP_TRY(hipMemcpy(&bits[f], e->seg_off.p + static_cast<size_t>(f) * (e->last_nseg + 1) + e->last_nseg,
                      sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return 0;
}

int sjpeg_hip_scan_histogram_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                                 int yuv_mode, int nframes, uint32_t* d_hist, void* stream) {
  return scan_statistics(e, src, width, height, yuv_mode, nframes, nullptr, true, d_hist, stream);
}

int sjpeg_hip_scan_histogram(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride,
                             int64_t frame_stride, int width, int height, int yuv_mode, int nframes,
                             uint32_t* d_hist, void* stream) {
  const sjpeg_hip_source s = rgb_source(d_rgb, row_stride, frame_stride);
  return scan_statistics(e, &s, width, height, yuv_mode, nframes, nullptr, true, d_hist, stream);
}

int sjpeg_hip_scan_symbol_stats_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width,
                                    int height, int yuv_mode, int nframes,
                                    const sjpeg_hip_scan_tables* tables, uint32_t* d_freq, void* stream) {
  if (tables == nullptr) return fail(SJPEG_HIP_EINVAL, "tables == NULL");
  return scan_statistics(e, src, width, height, yuv_mode, nframes, tables, false, d_freq, stream);
}

int sjpeg_hip_scan_symbol_stats(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride,
                                int64_t frame_stride, int width, int height, int yuv_mode,
                                int nframes, const sjpeg_hip_scan_tables* tables, uint32_t* d_freq,
                                void* stream) {
  const sjpeg_hip_source s = rgb_source(d_rgb, row_stride, frame_stride);
  return sjpeg_hip_scan_symbol_stats_src(e, &s, width, height, yuv_mode, nframes, tables, d_freq, stream);
}

int sjpeg_hip_encode_scan(sjpeg_hip_engine* e, const void* d_rgb, int64_t row_stride,
                          int64_t frame_stride, int width, int height, int yuv_mode, int nframes,
                          const sjpeg_hip_scan_tables* tables, const void* header,
                          size_t header_size, int append_eoi, void* d_out, size_t out_stride,
                          uint64_t* d_sizes, void* stream) {
  const sjpeg_hip_source s = rgb_source(d_rgb, row_stride, frame_stride);
  return sjpeg_hip_encode_scan_src(e, &s, width, height, yuv_mode, nframes, tables, header, header_size,
                                   append_eoi, d_out, out_stride, d_sizes, stream);
}

// header_offsets == NULL: one header (and one set of tables) for every frame; else per frame
static int encode_scan_one(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                            int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                            const void* header, size_t header_size, const size_t* header_offsets,
                            int append_eoi, void* d_out, size_t out_stride, uint64_t* d_sizes,
                            void* stream, const int* seg_range = nullptr /* restart mode: code only these segments */,
                            uint64_t* d_pack_off = nullptr /* packed output: [nframes + 1] frame starts */) {
  if (e == nullptr) return fail(SJPEG_HIP_EINVAL, "engine == NULL");
  if (d_out == nullptr || d_sizes == nullptr) return fail(SJPEG_HIP_EINVAL, "d_out/d_sizes == NULL");
  if (d_pack_off != nullptr && ((out_stride & 15u) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 15u) != 0 || out_stride >= (1ull << 32))) {
    return fail(SJPEG_HIP_EINVAL, "packed output: d_out and out_stride must be multiples of 16, out_stride below 4 GiB");
  }
  if (header == nullptr) header_size = 0;
  const bool multi = header_offsets != nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameGeo g;
  ScanArgs a;
  int cls = 0;
  const bool piped = e->pipelined;
  SegPlan plan;
  int rc = prepare_scan(e, src, width, height, yuv_mode, nframes, tables, st, &g, &a, &cls, multi, piped,
                        out_stride > 0 ? out_stride : 1, &plan);
  if (rc) return rc;
  int rst_tail = 0;
  if (seg_range != nullptr) {                      // a band of restart intervals: segments [begin, end) of the one frame
    if (nframes != 1 || multi || !a.rst) return fail(SJPEG_HIP_EINVAL, "a segment range needs one frame in restart mode");
    if (seg_range[0] < 0 || seg_range[1] > g.nseg || seg_range[0] >= seg_range[1]) return fail(SJPEG_HIP_EINVAL, "bad segment range");
    rst_tail = seg_range[1] < g.nseg ? 1 : 0;
    a.seg_first = seg_range[0];
    a.nseg = seg_range[1] - seg_range[0];
    g.nseg = a.nseg;                               // (everything below works on the band)
  }
  hipStream_t hs = piped ? e->side : st;           // the stream of the stitch kernels and of what only they read
  size_t largest_header = header_size;
  if (multi) {
    if (tables->flags & SJPEG_HIP_QUANT_REPLAY) {
      for (int f = 1; f < nframes; ++f) if (!(tables[f].flags & SJPEG_HIP_QUANT_REPLAY)) return fail(SJPEG_HIP_EINVAL, "flags must agree between the frames' tables");
    }
    for (int f = 1; f < nframes; ++f) {
      if ((tables[f].flags ^ tables->flags) & SJPEG_HIP_RESTART_MARKERS) return fail(SJPEG_HIP_EINVAL, "flags must agree between the frames' tables");
    }
    largest_header = 0;
    std::vector<uint32_t> offs(static_cast<size_t>(nframes) + 1);
    for (int f = 0; f <= nframes; ++f) {
      if (header_offsets[f] > header_size || (f > 0 && header_offsets[f] < header_offsets[f - 1])) {
        return fail(SJPEG_HIP_EINVAL, "header_offsets must ascend inside the header blob");
      }
      offs[f] = static_cast<uint32_t>(header_offsets[f]);
      if (f > 0 && header_offsets[f] - header_offsets[f - 1] > largest_header) largest_header = header_offsets[f] - header_offsets[f - 1];
    }
    // (a batch's part: its nframes + 1 offsets behind those of the parts in front -- two slots a frame)
    if ((rc = e->hdr_off.ensure(2 * static_cast<size_t>(part_total_frames(e, nframes)) + 2))) return rc;
    if (int rcu = upload(e, e->hdr_off.p + 2 * static_cast<size_t>(part_first_frame(e)), offs.data(), offs.size() * sizeof(uint32_t), hs)) return rcu;
  }
  header_size = header == nullptr ? 0 : header_size;
  if (out_stride < largest_header + 2 + 64) {
    return fail(SJPEG_HIP_ECAPACITY, "out_stride " + std::to_string(out_stride) + " too small");
  }
  const size_t ubuf_words = plan.ubuf_words;
  const uint32_t max_chunks = static_cast<uint32_t>((ubuf_words + kChunkWords - 1) / kChunkWords);
  if ((rc = e->seg_off.ensure(static_cast<size_t>(nframes) * (g.nseg + 1)))) return rc;
  if ((rc = e->ubuf.ensure(static_cast<size_t>(nframes) * ubuf_words))) return rc;
  if ((rc = e->chunk_ff.ensure(static_cast<size_t>(nframes) * max_chunks))) return rc;
  if ((rc = e->chunk_off.ensure(static_cast<size_t>(nframes) * max_chunks))) return rc;
  if ((rc = e->frame_flags.ensure(static_cast<size_t>(nframes)))) return rc;
  // (a batch's part: its headers behind those of the parts in front, kPartHeaderBytes a frame -- what
  // sjpeg_hip_encode_batch_src builds its headers in)
  constexpr size_t kPartHeaderBytes = 2048;
  const size_t hdr_first = (multi && e->replay_total > 0) ? static_cast<size_t>(part_first_frame(e)) * kPartHeaderBytes : 0;
  const size_t hdr_room = (multi && e->replay_total > 0) ? static_cast<size_t>(e->replay_total) * kPartHeaderBytes : 0;
  if (e->up_stream != nullptr && hdr_room != 0 && header_size > static_cast<size_t>(nframes) * kPartHeaderBytes) {
    return fail(SJPEG_HIP_EINVAL, "headers of a batch's part do not fit its share of the header buffer");
  }
  if ((rc = e->header.ensure(std::max(hdr_room, hdr_first + (header_size > 0 ? header_size : static_cast<size_t>(1)))))) return rc;
  if (header_size > 0) {
    const uint8_t* const hb = static_cast<const uint8_t*>(header);
    const bool held = e->up_stream == nullptr && hdr_first == 0 && e->header_held_at == e->header.p && e->header_stream == hs &&
                      e->header_held.size() == header_size && memcmp(e->header_held.data(), hb, header_size) == 0;
    if (!held) {
      e->header_held_at = nullptr;
      if (int rcu = upload(e, e->header.p + hdr_first, header, header_size, hs)) return rcu;
      if (e->up_stream == nullptr && hdr_first == 0) {
        e->header_held.assign(hb, hb + header_size);
        e->header_held_at = e->header.p; e->header_stream = hs;
      }
    }
  }

  e->last_nseg = g.nseg; e->last_nframes = nframes;
  const int set = piped ? e->set : 0;
  if (piped) {
    if (set == 1) {                                // the second set of what K1 writes and K2 / K3 read
      const size_t total_segs = static_cast<size_t>(nframes) * g.nseg;
      if ((rc = e->seg_words2.ensure(total_segs * plan.slot_words))) return rc;
      if ((rc = e->seg_nbits2.ensure(total_segs))) return rc;
      if ((rc = e->pool2.ensure(static_cast<size_t>(nframes) * plan.pool_words))) return rc;
      if ((rc = e->pool_ctr2.ensure(static_cast<size_t>(nframes) * 2))) return rc;
      if ((rc = e->seg_xbase2.ensure(total_segs))) return rc;
      a.seg_words = e->seg_words2.p; a.seg_nbits = e->seg_nbits2.p;
      a.pool = e->pool2.p; a.pool_ctr = e->pool_ctr2.p; a.seg_xbase = e->seg_xbase2.p;
    }
    // this set was last read by the K3 of the call before the previous one
    // (two calls back: as a rule long done, and a query costs the host a tenth of what a wait in the queue does)
    if (e->k3_pending[set]) {
      if (hipEventQuery(e->k3_done[set]) == hipSuccess) e->k3_pending[set] = false;
      else { (void)hipGetLastError(); HIP_TRY(hipStreamWaitEvent(st, e->k3_done[set], 0)); }
    }
  }
  if (e->ctr_clean_at[set] != a.pool_ctr || e->ctr_clean_n[set] < static_cast<size_t>(nframes)) {
    HIP_TRY(hipMemsetAsync(a.pool_ctr, 0, static_cast<size_t>(nframes) * 2 * sizeof(uint32_t), st));
  }
  e->ctr_clean_at[set] = nullptr;                  // dirty from K1 on, until this call's K4 is in the queue
  StitchArgs s;
  s.nseg = g.nseg; s.nframes = nframes;
  s.seg_nbits = a.seg_nbits; s.seg_off = e->seg_off.p;
  s.seg_words = a.seg_words; s.slot_words = plan.slot_words;
  s.pool = a.pool; s.pool_words = plan.pool_words; s.seg_xbase = a.seg_xbase; s.pool_ctr = a.pool_ctr;
  s.ubuf = e->ubuf.p; s.ubuf_words = ubuf_words;
  s.chunk_ff = e->chunk_ff.p; s.chunk_off = e->chunk_off.p; s.max_chunks = max_chunks;
  s.header = e->header.p + hdr_first; s.header_size = static_cast<uint32_t>(header_size);
  s.append_eoi = append_eoi;
  s.out = static_cast<uint8_t*>(d_out); s.out_stride = out_stride;
  s.sizes = reinterpret_cast<unsigned long long*>(d_sizes);
  s.pack_off = reinterpret_cast<unsigned long long*>(d_pack_off);
  s.seg_nbits64 = nullptr; s.total_bits_out = nullptr; s.subs = 1; s.wide_subs = 0;
  // few segments (one 4K frame: 791, one 8K 4:4:4 frame: 6172): one wave per 768 words of a slot instead of
  // one per segment -- a 1030-word segment was two dependent round trips of one wave (8K 4:4:4: K3 38 us)
  if (static_cast<size_t>(nframes) * g.nseg <= 8192 && (plan.slot_words & 3u) == 0u && plan.slot_words >= 776u) {
    s.subs = (plan.slot_words + kWideSpec * 256u - 1u) / (kWideSpec * 256u);
    s.wide_subs = s.subs > 1u ? 1u : 0u;
    if (!s.wide_subs) s.subs = 1;
  }
  s.hdr_off = multi ? e->hdr_off.p + 2 * static_cast<size_t>(part_first_frame(e)) : nullptr;
  s.seg_first = a.seg_first; s.rst_tail = rst_tail;
  s.frame_flags = e->frame_flags.p;
  // few, small frames (the launches whose time is launch latency): K4 inside K5 -- one dependent launch less.  (Not
  // in pipelined mode, whose hand-over hangs on K4; not with packed output or restart markers, which read what K4 writes.)
  static const bool no_fuse_k4 = getenv("SJPEG_HIP_NO_FUSED_K4") != nullptr;        // (A/B)
  // (2 = the large form: ONE frame of up to 8192 segments whatever its output slot -- an 8K 4:4:4 q90 frame is 6172 segments
  // and 25 MB; its K2 + K4 were 10 + 9.6 us as launches, VERDICT r05 #2 iii -- every workgroup adds up what it needs itself)
  static const bool no_fuse_big = getenv("SJPEG_HIP_NO_FUSED_BIG") != nullptr;      // (A/B)
  const bool fuse_ok = !no_fuse_k4 && !piped && s.pack_off == nullptr && !a.rst && static_cast<size_t>(nframes) * g.nseg <= 8192;
  s.fused_k4 = !fuse_ok ? 0 : max_chunks <= kFusedChunks ? 1 : (nframes == 1 && !no_fuse_big && max_chunks <= kFusedChunksBig) ? 2 : 0;
  static const bool no_fuse_k2 = getenv("SJPEG_HIP_NO_FUSED_K2") != nullptr;        // (A/B)
  // (the fused K3 keeps bit offsets in 32 bits: max_chunks <= 2^17 chunks of 4 KiB = 2^32 bits)
  s.fused_k2 = (!s.fused_k4 || no_fuse_k2 || e->ablate != 0) ? 0 : (s.fused_k4 == 1 && g.nseg <= kFusedSegs) ? 1 : (g.nseg <= kFusedSegsBig && nframes == 1 && !no_fuse_big) ? 2 : 0;
  if (s.fused_k2) {                                // K1 clears the 0xFF counters K2 would have
    a.clear_ff = e->chunk_ff.p; a.clear_n = max_chunks;
    a.clear_per = (max_chunks + static_cast<uint32_t>(g.nseg) - 1u) / static_cast<uint32_t>(g.nseg);
  }

  if ((rc = sync_uploads(e, st))) return rc;
  if (hs != st && (rc = sync_uploads(e, hs))) return rc;
  if (e->timing) HIP_TRY(hipEventRecord(e->ev[0], st));
  if (tables->flags & SJPEG_HIP_QUANT_REPLAY) {
    const int first = e->replay_total > 0 ? e->replay_first : 0;
    const int total = e->replay_total > 0 ? e->replay_total : nframes;
    if (e->replay.p == nullptr || e->replay_w != width || e->replay_h != height || e->replay_mode != yuv_mode ||
        e->replay_nframes != total || first < 0 || first + nframes > total) {
      return fail(SJPEG_HIP_EINVAL, "SJPEG_HIP_QUANT_REPLAY: no kept coefficients of this geometry "
                                    "(run the statistics pass with SJPEG_HIP_QUANT_KEEP first)");
    }
    a.replay = e->replay.p + static_cast<size_t>(first) * g.nseg * kScanThreads * 36;
    rc = launch_scan<kKindEncodeReplay>(yuv_mode, cls, dim3(g.nseg, nframes), st, a);
  } else if (tables->flags & SJPEG_HIP_QUANT_TRELLIS) {
    rc = launch_scan<kKindEncodeTrellis>(yuv_mode, cls, dim3(g.nseg, nframes), st, a);
  } else {
    rc = launch_scan<kKindEncode>(yuv_mode, cls, dim3(g.nseg, nframes), st, a);
  }
  if (rc) return rc;
  dbg_mark("encode: K1 launched");
  if (e->timing) HIP_TRY(hipEventRecord(e->ev[1], st));
  if (piped) {
    HIP_TRY(hipEventRecord(e->k1_done, st));
    dbg_mark("encode: k1_done recorded");
    HIP_TRY(hipStreamWaitEvent(hs, e->k1_done, 0));
    dbg_mark("encode: side waits k1_done");
  }

  if (!s.fused_k2) {
    hipLaunchKernelGGL(scan_seg_offsets, dim3(nframes), dim3(kThreads), 0, hs, s);
    HIP_TRY(hipGetLastError());
  }
  dbg_mark("encode: K2 launched");
  // the chunk count is only known on the device: a fixed grid strides over the chunks
  uint32_t gx = 4096u / static_cast<uint32_t>(nframes);
  if (gx < 64) gx = 64;
  if (gx > max_chunks) gx = max_chunks;
  if (s.fused_k2 == 2) hipLaunchKernelGGL(place_segments<2>, dim3((g.nseg * s.subs + 3) / 4, nframes), dim3(kThreads), 0, hs, s);
  else if (s.fused_k2) hipLaunchKernelGGL(place_segments<1>, dim3((g.nseg * s.subs + 3) / 4, nframes), dim3(kThreads), 0, hs, s);
  else hipLaunchKernelGGL(place_segments<0>, dim3((g.nseg * s.subs + 3) / 4, nframes), dim3(kThreads), 0, hs, s);
  HIP_TRY(hipGetLastError());
  if (!s.fused_k4) {
    hipLaunchKernelGGL(scan_chunk_offsets, dim3(nframes), dim3(kThreads), 0, hs, s);
    HIP_TRY(hipGetLastError());
  }
  e->ctr_clean_at[set] = a.pool_ctr; e->ctr_clean_n[set] = static_cast<size_t>(nframes);
  dbg_mark("encode: K3 K4 launched");
  if (piped) {                                     // (K4 is the last reader of this set: it looks at the pool's overrun flag)
    HIP_TRY(hipEventRecord(e->k3_done[set], hs));
    dbg_mark("encode: k3_done recorded");
    e->k3_pending[set] = true;
    e->set = set ^ 1;
  }
  if (s.pack_off != nullptr) {                     // packed output: the frames' places, then header / EOI / padding
    hipLaunchKernelGGL(pack_frame_offsets, dim3(1), dim3(kThreads), 0, hs, s);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(pack_frame_edges, dim3(nframes), dim3(kThreads), 0, hs, s);
    HIP_TRY(hipGetLastError());
  }
  if (s.fused_k4 == 2) hipLaunchKernelGGL(stuff_chunks<2>, dim3(gx, nframes), dim3(kThreads), 0, hs, s);
  else if (s.fused_k4) hipLaunchKernelGGL(stuff_chunks<1>, dim3(gx, nframes), dim3(kThreads), 0, hs, s);
  else hipLaunchKernelGGL(stuff_chunks<0>, dim3(gx, nframes), dim3(kThreads), 0, hs, s);
  HIP_TRY(hipGetLastError());
  dbg_mark("encode: K5 launched");
  if (a.rst && g.nseg - 1 + rst_tail > 0) {
    hipLaunchKernelGGL(patch_restart_markers, dim3((g.nseg - 1 + rst_tail + 3) / 4, nframes), dim3(kThreads), 0, hs, s);
    HIP_TRY(hipGetLastError());
  }
  if (piped) {
    e->side_pending = true;                        // (side_done itself: side_mark(), when somebody waits for it)
    e->side_recorded = false;
  }
  if (e->timing) {
    HIP_TRY(hipEventRecord(e->ev[2], hs));
    e->ev_valid = true;
  }
  return 0;
}

// A batch whose segment scratch would pass the engine's limit (SJPEG_HIP_SCRATCH_LIMIT_BYTES, 16 GiB by default: 64 8K
// 4:4:4 frames would take 21 GB in pipelined mode, sized as it is from out_stride alone -- VERDICT r04 #10) is coded
// as several launches of as many frames as the limit holds; the calls queue behind each other like any others (and
// pipeline with each other in pipelined mode).  Packed output and band calls are one launch by construction.
static int encode_scan_impl(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                            int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                            const void* header, size_t header_size, const size_t* header_offsets,
                            int append_eoi, void* d_out, size_t out_stride, uint64_t* d_sizes,
                            void* stream, const int* seg_range = nullptr, uint64_t* d_pack_off = nullptr) {
  FrameGeo g;
  if (e != nullptr && src != nullptr && nframes > 1 && seg_range == nullptr && d_pack_off == nullptr && d_out != nullptr && d_sizes != nullptr &&
      frame_geo(width, height, yuv_mode, &g)) {
    const SegPlan plan = seg_plan(g, out_stride > 0 ? out_stride : 1);
    const size_t per_frame = ((static_cast<size_t>(g.nseg) * plan.slot_words + plan.pool_words) * (e->pipelined ? 2 : 1) + plan.ubuf_words) * sizeof(uint32_t);
    size_t fit = per_frame == 0 ? static_cast<size_t>(nframes) : e->scratch_limit / per_frame;
    if (fit < 1) fit = 1;
    if (fit < static_cast<size_t>(nframes)) {
      // (a replay call addresses the kept blocks of frames [replay_first, replay_first + nframes) of a buffer for
      // replay_total frames: every launch its own range of it)
      const bool replay = tables != nullptr && (tables->flags & SJPEG_HIP_QUANT_REPLAY) != 0;
      // (the launches share what is uploaded once for all of them -- the header blob -- and follow each other: their
      // uploads stay in the call's stream)
      struct Restore { sjpeg_hip_engine* e; int first, total; hipStream_t up; ~Restore() { e->replay_first = first; e->replay_total = total; e->up_stream = up; } } restore{e, e->replay_first, e->replay_total, e->up_stream};
      e->up_stream = nullptr;
      if (replay && e->replay_total <= 0) { e->replay_total = nframes; e->replay_first = 0; }
      const int base_first = e->replay_first;
      for (int f0 = 0; f0 < nframes; f0 += static_cast<int>(fit)) {
        if (replay) e->replay_first = base_first + f0;
        const int nf = nframes - f0 < static_cast<int>(fit) ? nframes - f0 : static_cast<int>(fit);
        sjpeg_hip_source part = *src;
        for (int i = 0; i < 3; ++i) {
          if (part.plane[i] != nullptr) part.plane[i] = static_cast<const uint8_t*>(part.plane[i]) + static_cast<int64_t>(f0) * part.frame_stride[i];
        }
        const bool multi = header_offsets != nullptr;
        const int rc = encode_scan_one(e, &part, width, height, yuv_mode, nf, multi ? tables + f0 : tables, header, header_size,
                                       multi ? header_offsets + f0 : nullptr, append_eoi, static_cast<uint8_t*>(d_out) + static_cast<size_t>(f0) * out_stride,
                                       out_stride, d_sizes + f0, stream);
        if (rc) return rc;
      }
      return 0;
    }
  }
  return encode_scan_one(e, src, width, height, yuv_mode, nframes, tables, header, header_size, header_offsets, append_eoi,
                         d_out, out_stride, d_sizes, stream, seg_range, d_pack_off);
}

int sjpeg_hip_encode_scan_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                              int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                              const void* header, size_t header_size, int append_eoi, void* d_out,
                              size_t out_stride, uint64_t* d_sizes, void* stream) {
  return encode_scan_impl(e, src, width, height, yuv_mode, nframes, tables, header, header_size, nullptr,
                          append_eoi, d_out, out_stride, d_sizes, stream);
}

int sjpeg_hip_encode_scan_packed_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                                     int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                                     const void* header, size_t header_size, int append_eoi, void* d_out,
                                     size_t out_stride, uint64_t* d_sizes, uint64_t* d_offsets, void* stream) {
  if (d_offsets == nullptr) return fail(SJPEG_HIP_EINVAL, "sjpeg_hip_encode_scan_packed_src: d_offsets == NULL");
  return encode_scan_impl(e, src, width, height, yuv_mode, nframes, tables, header, header_size, nullptr,
                          append_eoi, d_out, out_stride, d_sizes, stream, nullptr, d_offsets);
}

int sjpeg_hip_encode_intervals_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                                   int yuv_mode, const sjpeg_hip_scan_tables* tables, int seg_begin, int seg_end,
                                   void* d_out, size_t out_cap, uint64_t* d_size, void* stream) {
  if (tables == nullptr || !(tables->flags & SJPEG_HIP_RESTART_MARKERS)) {
    return fail(SJPEG_HIP_EINVAL, "sjpeg_hip_encode_intervals_src needs tables with SJPEG_HIP_RESTART_MARKERS");
  }
  const int range[2] = {seg_begin, seg_end};
  return encode_scan_impl(e, src, width, height, yuv_mode, 1, tables, nullptr, 0, nullptr, /*append_eoi=*/0,
                          d_out, out_cap, d_size, stream, range);
}

int sjpeg_hip_encode_scan_multi(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                                int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                                const void* headers, const size_t* header_offsets, int append_eoi,
                                void* d_out, size_t out_stride, uint64_t* d_sizes, void* stream) {
  if (tables == nullptr || headers == nullptr || header_offsets == nullptr || nframes <= 0) {
    return fail(SJPEG_HIP_EINVAL, "null argument or nframes <= 0");
  }
  return encode_scan_impl(e, src, width, height, yuv_mode, nframes, tables, headers, header_offsets[nframes],
                          header_offsets, append_eoi, d_out, out_stride, d_sizes, stream);
}

int sjpeg_hip_scan_symbol_stats_multi(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                                      int yuv_mode, int nframes, const sjpeg_hip_scan_tables* tables,
                                      uint32_t* d_freq, void* stream) {
  if (tables == nullptr) return fail(SJPEG_HIP_EINVAL, "tables == NULL");
  return scan_statistics(e, src, width, height, yuv_mode, nframes, tables, false, d_freq, stream, true);
}

// ---- one frame over several GPUs: bands of consecutive segments ----------

int sjpeg_hip_segment_count(int width, int height, int yuv_mode) {
  FrameGeo g;
  if (!frame_geo(width, height, yuv_mode, &g)) return fail(SJPEG_HIP_EINVAL, "bad geometry / yuv_mode");
  return g.nseg;
}

int sjpeg_hip_restart_interval(int yuv_mode) {
  switch (yuv_mode) {
    case SJPEG_HIP_YUV420: return Geo<SJPEG_HIP_YUV420>::kSegMcus;
    case SJPEG_HIP_YUV444: return Geo<SJPEG_HIP_YUV444>::kSegMcus;
    case SJPEG_HIP_YUV400: return Geo<SJPEG_HIP_YUV400>::kSegMcus;
    default: return 0;
  }
}

size_t sjpeg_hip_header_add_restart(uint8_t* header, size_t size, size_t cap, int yuv_mode) {
  const int ri = sjpeg_hip_restart_interval(yuv_mode);
  if (header == nullptr || ri == 0 || size < 4 || cap < size + 6) return 0;
  size_t sos = size;                               // the SOS segment: the last FF DA of the header
  while (sos >= 2 && !(header[sos - 2] == 0xff && header[sos - 1] == 0xda)) --sos;
  if (sos < 2) return 0;
  sos -= 2;
  memmove(header + sos + 6, header + sos, size - sos);
  const uint8_t dri[6] = {0xff, 0xdd, 0x00, 0x04, static_cast<uint8_t>(ri >> 8), static_cast<uint8_t>(ri)};
  memcpy(header + sos, dri, 6);
  return size + 6;
}

size_t sjpeg_hip_band_bound(int width, int height, int yuv_mode, int seg_begin, int seg_end) {
  FrameGeo g;
  if (!frame_geo(width, height, yuv_mode, &g)) return 0;
  if (seg_begin < 0 || seg_end > g.nseg || seg_begin >= seg_end) return 0;
  return (static_cast<size_t>(seg_end - seg_begin) * g.slot_words + kChunkWords + 3) & ~size_t(3);
}

int sjpeg_hip_encode_band_src(sjpeg_hip_engine* e, const sjpeg_hip_source* src, int width, int height,
                              int yuv_mode, const sjpeg_hip_scan_tables* tables, int seg_begin,
                              int seg_end, uint32_t* d_words, size_t cap_words, uint64_t* d_nbits,
                              void* stream) {
  if (d_words == nullptr || d_nbits == nullptr) return fail(SJPEG_HIP_EINVAL, "d_words/d_nbits == NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameGeo g;
  ScanArgs a;
  int cls = 0;
  SegPlan plan;
  int rc = prepare_scan(e, src, width, height, yuv_mode, 1, tables, st, &g, &a, &cls, false, false, SIZE_MAX, &plan);
  if (rc) return rc;
  if (seg_begin < 0 || seg_end > g.nseg || seg_begin >= seg_end) return fail(SJPEG_HIP_EINVAL, "bad segment range");
  a.rst = 0;                                       // (bands are stitched at bit granularity: the exact mode)
  HIP_TRY(hipMemsetAsync(a.pool_ctr, 0, 2 * sizeof(uint32_t), st));
  e->ctr_clean_at[0] = nullptr;                    // (no K4 in this path)
  const int nloc = seg_end - seg_begin;
  const size_t need = sjpeg_hip_band_bound(width, height, yuv_mode, seg_begin, seg_end);
  if (cap_words < need) {
    return fail(SJPEG_HIP_ECAPACITY, "band buffer of " + std::to_string(cap_words) + " words, need " + std::to_string(need));
  }
  a.nseg = nloc;
  a.seg_first = seg_begin;
  const uint32_t max_chunks = static_cast<uint32_t>((cap_words + kChunkWords - 1) / kChunkWords);
  if ((rc = e->seg_off.ensure(static_cast<size_t>(nloc) + 1))) return rc;
  if ((rc = e->chunk_ff.ensure(max_chunks))) return rc;
  StitchArgs s;
  memset(&s, 0, sizeof(s));
  s.nseg = nloc; s.nframes = 1;
  s.seg_nbits = e->seg_nbits.p; s.seg_off = e->seg_off.p;
  s.seg_words = e->seg_words.p; s.slot_words = plan.slot_words;
  s.pool = a.pool; s.pool_words = plan.pool_words; s.seg_xbase = a.seg_xbase; s.pool_ctr = a.pool_ctr;
  s.ubuf = d_words; s.ubuf_words = cap_words;
  s.chunk_ff = e->chunk_ff.p; s.max_chunks = max_chunks;
  s.total_bits_out = reinterpret_cast<unsigned long long*>(d_nbits);
  s.subs = 1;
  if (tables->flags & SJPEG_HIP_QUANT_TRELLIS) rc = launch_scan<kKindEncodeTrellis>(yuv_mode, cls, dim3(nloc, 1), st, a);
  else rc = launch_scan<kKindEncode>(yuv_mode, cls, dim3(nloc, 1), st, a);
  if (rc) return rc;
  hipLaunchKernelGGL(scan_seg_offsets, dim3(1), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(place_segments<false>, dim3((nloc + 3) / 4, 1), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

int sjpeg_hip_stitch_bands(sjpeg_hip_engine* e, int nbands, const uint32_t* d_words,
                           size_t band_stride_words, const uint64_t* d_nbits, const void* header,
                           size_t header_size, int append_eoi, void* d_out, size_t out_cap,
                           uint64_t* d_size, void* stream) {
  if (e == nullptr || d_words == nullptr || d_nbits == nullptr || d_out == nullptr || d_size == nullptr) {
    return fail(SJPEG_HIP_EINVAL, "null argument");
  }
  if (nbands <= 0 || nbands > 65535) return fail(SJPEG_HIP_EINVAL, "nbands out of range");
  if (band_stride_words < 4 || band_stride_words >= (size_t(1) << 27)) {
    return fail(SJPEG_HIP_EINVAL, "band_stride_words out of range (a band is shorter than 2^32 bits)");
  }
  if (header == nullptr) header_size = 0;
  if (out_cap < header_size + 2 + 64) return fail(SJPEG_HIP_ECAPACITY, "out_cap too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipSetDevice(e->device));
  if (int rc0 = order_on_stream(e, st)) return rc0;
  int rc;
  const size_t ubuf_words = (static_cast<size_t>(nbands) * band_stride_words + kChunkWords + 3) & ~size_t(3);
  const uint32_t max_chunks = static_cast<uint32_t>((ubuf_words + kChunkWords - 1) / kChunkWords);
  if ((rc = e->seg_off.ensure(static_cast<size_t>(nbands) + 1))) return rc;
  if ((rc = e->ubuf.ensure(ubuf_words))) return rc;
  if ((rc = e->chunk_ff.ensure(max_chunks))) return rc;
  if ((rc = e->chunk_off.ensure(max_chunks))) return rc;
  if ((rc = e->header.ensure(header_size > 0 ? header_size : 1))) return rc;
  e->header_held_at = nullptr;                     // (this path does not track what it uploads)
  if (header_size > 0) { if (int rcu = upload(e, e->header.p, header, header_size, st)) return rcu; }
  StitchArgs s;
  memset(&s, 0, sizeof(s));
  s.nseg = nbands; s.nframes = 1;
  s.seg_nbits64 = reinterpret_cast<const unsigned long long*>(d_nbits);
  s.seg_off = e->seg_off.p;
  s.seg_words = d_words; s.slot_words = static_cast<uint32_t>(band_stride_words);
  s.ubuf = e->ubuf.p; s.ubuf_words = ubuf_words;
  s.chunk_ff = e->chunk_ff.p; s.chunk_off = e->chunk_off.p; s.max_chunks = max_chunks;
  s.header = e->header.p; s.header_size = static_cast<uint32_t>(header_size);
  s.append_eoi = append_eoi;
  s.out = static_cast<uint8_t*>(d_out); s.out_stride = out_cap;
  s.sizes = reinterpret_cast<unsigned long long*>(d_size);
  const uint32_t per_wave = kSpec * kPlaceLanes;
  s.subs = static_cast<uint32_t>((band_stride_words + per_wave - 1) / per_wave);
  e->last_nseg = nbands; e->last_nframes = 1;
  hipLaunchKernelGGL(scan_seg_offsets, dim3(1), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  const uint32_t units = static_cast<uint32_t>(nbands) * s.subs;
  hipLaunchKernelGGL(place_segments<false>, dim3((units + 3) / 4, 1), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(scan_chunk_offsets, dim3(1), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  uint32_t gx = 4096u;
  if (gx > max_chunks) gx = max_chunks;
  hipLaunchKernelGGL(stuff_chunks<false>, dim3(gx, 1), dim3(kThreads), 0, st, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

namespace {
// adapt_decide_kernel behind sjpeg_hip_adapt_sums on the same stream: d_quant_out[nframes][2][64]
int adapt_decide(const int64_t* d_sums, const int32_t* d_totlast, int nframes, const uint8_t quant[2][64], int ntables,
                 int qdelta_max_luma, int qdelta_max_chroma, uint8_t* d_quant_out, hipStream_t st) {
  DecideArgs a;
  a.sums = reinterpret_cast<const long long*>(d_sums);
  a.totlast = d_totlast;
  a.quant_out = d_quant_out;
  memcpy(a.quant_in, quant, sizeof(a.quant_in));
  a.last_step[0] = qdelta_max_luma + 12;
  a.last_step[1] = qdelta_max_chroma + 12;
  hipLaunchKernelGGL(adapt_decide_kernel, dim3(ntables, nframes), dim3(64), 0, st, a);
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

int sjpeg_hip_adapt_decide(const int64_t* d_sums, const int32_t* d_totlast, int nframes, const uint8_t quant[2][64],
                           int yuv_mode, int qdelta_max_luma, int qdelta_max_chroma, uint8_t* d_quant_out, void* stream) {
  if (d_sums == nullptr || d_totlast == nullptr || quant == nullptr || d_quant_out == nullptr || nframes <= 0 || nframes > 65535) {
    return fail(SJPEG_HIP_EINVAL, "null argument or bad nframes");
  }
  if (qdelta_max_luma < -12 || qdelta_max_luma > 12 || qdelta_max_chroma < -12 || qdelta_max_chroma > 12) {
    return fail(SJPEG_HIP_EINVAL, "qdelta_max outside -12 .. 12");
  }
  return adapt_decide(d_sums, d_totlast, nframes, quant, yuv_mode == SJPEG_HIP_YUV400 ? 1 : 2, qdelta_max_luma, qdelta_max_chroma,
                      d_quant_out, static_cast<hipStream_t>(stream));
}

int sjpeg_hip_adapt_sums(const uint32_t* d_hist, int nframes, const uint8_t quant[2][64],
                         const uint8_t* min_quant, int64_t* d_sums, int32_t* d_totlast, void* stream) {
  if (d_hist == nullptr || quant == nullptr || d_sums == nullptr || d_totlast == nullptr || nframes <= 0 || nframes > 65535) {
    return fail(SJPEG_HIP_EINVAL, "null argument or bad nframes");
  }
  AdaptArgs a;
  a.hist = d_hist;
  a.sums = reinterpret_cast<long long*>(d_sums);
  a.totlast = d_totlast;
  memcpy(a.quant, quant, sizeof(a.quant));
  if (min_quant != nullptr) memcpy(a.min_quant, min_quant, sizeof(a.min_quant)); else memset(a.min_quant, 1, sizeof(a.min_quant));
  hipLaunchKernelGGL(adapt_sums_kernel, dim3(64, 2, nframes), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- a whole batch with the reference's per-picture analysis (methods 0..6), device-resident ----
// What Encoder::Encode does for ONE picture with adaptive quantization and optimised Huffman
// codes (src/enc.cc:391-448: CollectHistograms + AnalyseHisto, the statistics half of
// SinglePassScanOptimized, headers, the scan), done for nframes pictures with one launch per
// device pass and the per-picture float / Huffman work on the host in between.
namespace {
constexpr int kAdaptDeltas = 25;      // candidate steps -12 .. +12 (jpeg_host.h)
struct BatchScratch {                 // per host thread: device scratch of sjpeg_hip_encode_batch_src
  void* d_hist = nullptr; size_t hist_cap = 0;
  void* d_sums = nullptr; size_t sums_cap = 0;
  void* d_freq = nullptr; size_t freq_cap = 0;
  void* h_pinned = nullptr; size_t pinned_cap = 0;   // where the device's sums / counts land on the host
  void* d_pinned = nullptr;                          // ... as the device addresses it
  int device = -1;
  void Drop() {
    if (d_hist) (void)hipFree(d_hist);
    if (d_sums) (void)hipFree(d_sums);
    if (d_freq) (void)hipFree(d_freq);
    if (h_pinned) (void)hipHostFree(h_pinned);
    for (auto& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (pass_done) (void)hipEventDestroy(pass_done);
    if (call_begin) (void)hipEventDestroy(call_begin);
    pass_done = nullptr; call_begin = nullptr;
    for (auto& e : job_ev) if (e) (void)hipEventDestroy(e);
    job_ev.clear();
    d_hist = d_sums = d_freq = h_pinned = d_pinned = nullptr; hist_cap = sums_cap = freq_cap = pinned_cap = 0;
  }
  hipEvent_t ev[12] = {};                        // behind the read-backs of a part: sums [0..7] (two halves a part), counts [8..11]
  hipEvent_t pass_done = nullptr;
  hipEvent_t call_begin = nullptr;                 // on the call's stream before anything of the call: the early uploads wait for it
  std::vector<hipEvent_t> job_ev;                  // one per job of a batch coded in lanes
  bool EnsureJobEvents(size_t n) {
    while (job_ev.size() < n) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
      job_ev.push_back(e);
    }
    return true;
  }
  bool EnsureEvents() {
    for (auto& e : ev) {
      if (e == nullptr && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return false; }
    }
    if (pass_done == nullptr && hipEventCreateWithFlags(&pass_done, hipEventDisableTiming) != hipSuccess) { pass_done = nullptr; return false; }
    if (call_begin == nullptr && hipEventCreateWithFlags(&call_begin, hipEventDisableTiming) != hipSuccess) { call_begin = nullptr; return false; }
    return true;
  }
  bool EnsurePinned(size_t need) {
    if (need <= pinned_cap) return true;
    if (h_pinned) (void)hipHostFree(h_pinned);
    h_pinned = nullptr; d_pinned = nullptr; pinned_cap = 0;
    if (hipHostMalloc(&h_pinned, need, hipHostMallocMapped) != hipSuccess) return false;
    pinned_cap = need;
    // (the address the device writes the block at: the read-backs are kernels too, see stage_copy_kernel)
    if (hipHostGetDevicePointer(&d_pinned, h_pinned, 0) != hipSuccess) { (void)hipGetLastError(); d_pinned = nullptr; }
    return true;
  }
  ~BatchScratch() { if (device >= 0) { (void)hipSetDevice(device); Drop(); } }
  bool Ensure(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    if (hipMalloc(p, need) != hipSuccess) return false;
    *cap = need;
    return true;
  }
};
thread_local BatchScratch g_batch;

}  // namespace

int sjpeg_hip_encode_batch_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                                          int width, int height, int yuv_mode, int nframes,
                                          const uint8_t quant_in[2][64], const uint8_t* min_quant, int q_bias,
                                          int method, int qdelta_max_luma, int qdelta_max_chroma,
                                          void* d_out, size_t out_stride, uint64_t* d_sizes, void* stream) {
  if (engine == nullptr || src == nullptr || quant_in == nullptr || nframes <= 0) {
    return fail(SJPEG_HIP_EINVAL, "null argument or nframes <= 0");
  }
  if (d_out == nullptr || d_sizes == nullptr) return fail(SJPEG_HIP_EINVAL, "d_out/d_sizes == NULL");
  if (method < 0) method = 0;
  if (method > 6) return fail(SJPEG_HIP_EINVAL, "sjpeg_hip_encode_batch_src: methods 0..6 (trellis goes through the host API)");
  struct PartsGuard {                              // the engine addresses whole calls again when this returns
    sjpeg_hip_engine* e;
    ~PartsGuard() { e->replay_first = 0; e->replay_total = 0; e->reduce_stream = nullptr; e->reduce_ev = nullptr; e->coefs_keep = e->coefs_use = false; e->up_stream = nullptr; e->up_wait = nullptr; }
  } parts_guard{engine};
  try {
    const bool adaptive = method >= 3, optimize = (method != 0) && (method != 3);
    hipStream_t st = static_cast<hipStream_t>(stream);
    BatchScratch& sc = g_batch;
    const int device = engine->device;
    if (sc.device != device) { if (sc.device >= 0) { (void)hipSetDevice(sc.device); sc.Drop(); } sc.device = device; }
    HIP_TRY(hipSetDevice(device));
    const size_t n = static_cast<size_t>(nframes);
    std::vector<sjpeg_hip_scan_tables> tables(n);
    std::vector<uint8_t> quant(n * 128);
    {
      sjpeg_hip_scan_tables t0;
      memset(&t0, 0, sizeof(t0));
      uint8_t q0[2][64];
      memcpy(q0, quant_in, sizeof(q0));
      sjpeg_hip_finalize_quant(q0, min_quant, q_bias, &t0);
      sjpeg_hip_default_huffman(&t0);
      for (size_t f = 0; f < n; ++f) { tables[f] = t0; memcpy(&quant[f * 128], q0, 128); }
    }
    // The batch is coded in PARTS (two halves from 24 frames on): every device pass of a part is one
    // launch, and the host analysis a part needs between two passes -- the reference's float regression
    // per frame (10 us), its Huffman table builder (10 us) -- runs while the device is busy with the
    // other part's pass instead of leaving it idle (a quarter of the call for 32 4K frames).  Everything
    // goes to the caller's stream in order; the host waits on events behind the read-backs only.
    constexpr int kMaxParts = 4;
    static const int parts_env = getenv("SJPEG_HIP_BATCH_PARTS") ? atoi(getenv("SJPEG_HIP_BATCH_PARTS")) : 0;   // (experiments)
    // (measured: 32 4K frames 1.84 -> 1.61 ms in two parts, 16 frames 1.13 -> 1.21: parts of fewer than a dozen
    // frames lose more to their smaller launches than the overlap gives)
    // (round 5, with the persistent histogram kind, early uploads and the fit on the device: 4K frames -- 8 frames 0.40 ms
    // in one part, 0.44 in two; 12 frames 0.54 / 0.55; 16 frames 0.68 / 0.66; 20 frames 0.80 / 0.79; 24 frames 0.99 / 0.89:
    // two parts from 125 Mpixels on.  With the regression on the host it was 80: 12 frames 0.64 / 0.57)
    const bool big = n >= 2 && static_cast<double>(n) * width * height >= 125e6;
    int nparts = ((n >= 24 || big) && (adaptive || optimize)) ? 2 : 1;
    if (parts_env >= 1 && parts_env <= kMaxParts && (adaptive || optimize) && n >= static_cast<size_t>(parts_env)) nparts = parts_env;
    size_t part_lo[kMaxParts + 1];
    for (int p = 0; p <= kMaxParts; ++p) {               // (the larger parts first: the engine scratch is sized once)
      part_lo[p] = p >= nparts ? n : (n * p + nparts - 1) / nparts;
    }
    constexpr size_t kHist = 2 * 64 * 128 * sizeof(uint32_t);
    constexpr size_t kSums = 2 * 64 * kAdaptDeltas * 2 * sizeof(int64_t), kTot = 2 * 64 * 2 * sizeof(int32_t);
    constexpr size_t kFreq = 2 * 272 * sizeof(uint32_t);
    if (adaptive && (!sc.Ensure(&sc.d_hist, &sc.hist_cap, n * kHist) || !sc.Ensure(&sc.d_sums, &sc.sums_cap, n * (kSums + kTot + 128)))) {
      return fail(SJPEG_HIP_ENOMEM, "hipMalloc(batch scratch) failed");
    }
    if (optimize && !sc.Ensure(&sc.d_freq, &sc.freq_cap, n * kFreq)) return fail(SJPEG_HIP_ENOMEM, "hipMalloc(batch scratch) failed");
    if (!sc.EnsurePinned(n * (kSums + kTot + kFreq)) || !sc.EnsureEvents()) return fail(SJPEG_HIP_ENOMEM, "hipHostMalloc / hipEventCreate(batch scratch) failed");
    // (a part's sums and totals lie back to back -- [nf][kSums] then [nf][kTot] at f0 * (kSums + kTot) --: ONE read-back)
    uint8_t* const h_sums = static_cast<uint8_t*>(sc.h_pinned);
    uint8_t* const h_freq = h_sums + n * (kSums + kTot);                          // [n][kFreq]
    auto part_source = [&](size_t f0) {
      sjpeg_hip_source s = *src;
      for (int i = 0; i < 3; ++i) {
        if (s.plane[i] != nullptr) s.plane[i] = static_cast<const uint8_t*>(s.plane[i]) + static_cast<int64_t>(f0) * s.frame_stride[i];
      }
      return s;
    };
    // ---- LANES (round 6): jobs of about eight 4K frames, each a complete sequence of passes, on up to four streams at once
    // (sjpeg_hip_engine::lane).  One host thread drives them: a job's next step is taken when the event behind its
    // read-back has passed (polled -- the caller would block in hipEventSynchronize otherwise), whichever job that is.
    static const int lanes_env = getenv("SJPEG_HIP_BATCH_LANES") ? atoi(getenv("SJPEG_HIP_BATCH_LANES")) : -1;   // (A/B: 0 = the two parts of round 5)
    static const double job_px = getenv("SJPEG_HIP_BATCH_JOB_MPIX") ? atof(getenv("SJPEG_HIP_BATCH_JOB_MPIX")) * 1e6 : 66.4e6;
    const double total_px = static_cast<double>(n) * width * height;
    size_t njobs = static_cast<size_t>(total_px / job_px + 0.5);
    if (getenv("SJPEG_HIP_BATCH_JOB_MPIX") == nullptr) njobs = total_px >= 90e6 ? 2 : 1;     // (the sweep: DESIGN.md section 4)
    static const int njobs_env = getenv("SJPEG_HIP_BATCH_NJOBS") ? atoi(getenv("SJPEG_HIP_BATCH_NJOBS")) : 0;   // (experiments)
    if (njobs_env > 0) njobs = static_cast<size_t>(njobs_env);
    if (njobs > n) njobs = n;
    if (njobs > 64) njobs = 64;
    // (method 0 in lanes: 0.817 against 0.808 ms for 32 4K frames -- one K1 launch fills the chip by itself, and calls
    // back to back overlap their stitch with the next K1 anyway)
    if (lanes_env != 0 && njobs >= 2 && (adaptive || optimize) && !engine->is_lane && parts_env == 0) {
      const int nlanes = static_cast<int>(std::min<size_t>(njobs, lanes_env > 0 ? std::min(lanes_env, static_cast<int>(sjpeg_hip_engine::kLanes)) : sjpeg_hip_engine::kLanes));
      if (nlanes == 4 && engine->batch_lane3 == nullptr) HIP_TRY(hipStreamCreateWithFlags(&engine->batch_lane3, hipStreamNonBlocking));
      hipStream_t lane_st[sjpeg_hip_engine::kLanes] = {st, engine->batch_side, engine->batch_up, engine->batch_lane3};
      sjpeg_hip_engine* lane_e[sjpeg_hip_engine::kLanes] = {engine, nullptr, nullptr, nullptr};
      for (int l = 1; l < nlanes; ++l) {
        if (engine->lane[l] == nullptr) {
          sjpeg_hip_engine* c = new (std::nothrow) sjpeg_hip_engine;
          if (c == nullptr) return fail(SJPEG_HIP_ENOMEM, "host allocation failed");
          c->device = engine->device; c->cu_count = engine->cu_count; c->histo_slots = engine->histo_slots;
          c->scratch_limit = engine->scratch_limit; c->ablate = engine->ablate; c->is_lane = true;
          engine->lane[l] = c;
        }
        if (engine->lane_done[l] == nullptr) HIP_TRY(hipEventCreateWithFlags(&engine->lane_done[l], hipEventDisableTiming));
        lane_e[l] = engine->lane[l];
      }
      if (engine->lane_in == nullptr) HIP_TRY(hipEventCreateWithFlags(&engine->lane_in, hipEventDisableTiming));
      if (!sc.EnsureJobEvents(njobs)) return fail(SJPEG_HIP_ENOMEM, "hipEventCreate(batch scratch) failed");
      // the lanes' streams start behind everything the caller has put on its own (the pixels, the previous call)
      if (int rco = order_on_stream(engine, st)) return rco;
      HIP_TRY(hipEventRecord(engine->lane_in, st));
      for (int l = 1; l < nlanes; ++l) HIP_TRY(hipStreamWaitEvent(lane_st[l], engine->lane_in, 0));
      struct LanesGuard {                            // the child engines address whole calls again when this returns
        sjpeg_hip_engine** e; int n;
        ~LanesGuard() { for (int l = 1; l < n; ++l) if (e[l]) { e[l]->coefs_keep = e[l]->coefs_use = false; } }
      } lanes_guard{lane_e, nlanes};
      const bool keep_coefs = adaptive && optimize && getenv("SJPEG_HIP_NO_COEF_KEEP") == nullptr;
      for (int l = 0; l < nlanes; ++l) lane_e[l]->coefs_keep = keep_coefs;
      std::vector<sjpeg_hip_huffman_spec> jspecs(optimize ? n * 4 : 0);
      struct Job { size_t f0, nf; int lane; int phase; };   // phase: 0 not started, 1 sums on their way, 2 counts on their way, 3 encode launched
      std::vector<Job> jobs(njobs);
      for (size_t j = 0; j < njobs; ++j) jobs[j] = Job{(n * j) / njobs, (n * (j + 1)) / njobs - (n * j) / njobs, static_cast<int>(j % static_cast<size_t>(nlanes)), 0};
      const int ntab = yuv_mode == SJPEG_HIP_YUV400 ? 1 : 2;
      // (the matrices every frame starts from: quant[] itself is adapted frame by frame while later jobs still start)
      uint8_t q_start[2][64];
      memcpy(q_start, &quant[0], sizeof(q_start));
      auto read_back_on = [&](hipStream_t s2, void* h_dst, const void* d_src, size_t bytes) -> int {
        void* const dv = sc.d_pinned == nullptr ? nullptr : static_cast<uint8_t*>(sc.d_pinned) + (static_cast<uint8_t*>(h_dst) - static_cast<uint8_t*>(sc.h_pinned));
        return copy_by_kernel(dv, d_src, bytes, s2, hipMemcpyDeviceToHost, d_src, h_dst);
      };
      std::vector<uint8_t> jheaders;
      std::vector<size_t> joffs;
      // the three steps of a job; each ends with the launch of a pass (and of the read-back the next step waits for)
      auto step_encode = [&](Job& jb) -> int {
        sjpeg_hip_engine* const e = lane_e[jb.lane];
        jheaders.clear();
        joffs.assign(jb.nf + 1, 0);
        uint8_t one[2048];
        for (size_t f = jb.f0; f < jb.f0 + jb.nf; ++f) {
          const size_t hs = sjpeg_hip_make_header_ex(width, height, yuv_mode, reinterpret_cast<const uint8_t(*)[64]>(&quant[f * 128]),
                                                     optimize ? &jspecs[f * 4] : nullptr, one, sizeof(one));
          if (hs == 0) return fail(SJPEG_HIP_EINVAL, "header generation failed");
          jheaders.insert(jheaders.end(), one, one + hs);
          joffs[f - jb.f0 + 1] = jheaders.size();
        }
        const sjpeg_hip_source ps = part_source(jb.f0);
        const int rc = sjpeg_hip_encode_scan_multi(e, &ps, width, height, yuv_mode, static_cast<int>(jb.nf), &tables[jb.f0], jheaders.data(),
                                                   joffs.data(), /*append_eoi=*/1, static_cast<uint8_t*>(d_out) + jb.f0 * out_stride,
                                                   out_stride, d_sizes + jb.f0, lane_st[jb.lane]);
        jb.phase = 3;
        return rc;
      };
      auto step_stats = [&](Job& jb, size_t j) -> int {
        if (!optimize) return step_encode(jb);
        sjpeg_hip_engine* const e = lane_e[jb.lane];
        for (size_t f = jb.f0; f < jb.f0 + jb.nf; ++f) tables[f].flags |= SJPEG_HIP_QUANT_KEEP;
        e->coefs_use = e->coefs_keep;
        const sjpeg_hip_source ps = part_source(jb.f0);
        uint32_t* const d_freq = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(sc.d_freq) + jb.f0 * kFreq);
        if (int rc = sjpeg_hip_scan_symbol_stats_multi(e, &ps, width, height, yuv_mode, static_cast<int>(jb.nf), &tables[jb.f0], d_freq, lane_st[jb.lane])) return rc;
        if (int rc = read_back_on(lane_st[jb.lane], h_freq + jb.f0 * kFreq, d_freq, jb.nf * kFreq)) return rc;
        HIP_TRY(hipEventRecord(sc.job_ev[j], lane_st[jb.lane]));
        jb.phase = 2;
        return 0;
      };
      auto step_start = [&](Job& jb, size_t j) -> int {
        if (!adaptive) return step_stats(jb, j);
        sjpeg_hip_engine* const e = lane_e[jb.lane];
        hipStream_t const js = lane_st[jb.lane];
        const sjpeg_hip_source ps = part_source(jb.f0);
        uint32_t* const d_hist = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(sc.d_hist) + jb.f0 * kHist);
        if (int rc = sjpeg_hip_scan_histogram_src(e, &ps, width, height, yuv_mode, static_cast<int>(jb.nf), d_hist, js)) return rc;
        uint8_t* const d_grp = static_cast<uint8_t*>(sc.d_sums) + jb.f0 * (kSums + kTot);      // [nf][kSums] then [nf][kTot]
        if (int rc = sjpeg_hip_adapt_sums(d_hist, static_cast<int>(jb.nf), q_start, min_quant,
                                          reinterpret_cast<int64_t*>(d_grp), reinterpret_cast<int32_t*>(d_grp + jb.nf * kSums), js)) return rc;
        uint8_t* const d_q = static_cast<uint8_t*>(sc.d_sums) + n * (kSums + kTot) + jb.f0 * 128;
        if (int rc = adapt_decide(reinterpret_cast<const int64_t*>(d_grp), reinterpret_cast<const int32_t*>(d_grp + jb.nf * kSums), static_cast<int>(jb.nf),
                                  q_start, ntab, qdelta_max_luma, qdelta_max_chroma, d_q, js)) return rc;
        if (int rc = read_back_on(js, h_sums + jb.f0 * 128, d_q, jb.nf * 128)) return rc;
        HIP_TRY(hipEventRecord(sc.job_ev[j], js));
        jb.phase = 1;
        return 0;
      };
      auto step_after = [&](Job& jb, size_t j) -> int {       // the event of phase 1 / 2 has passed
        if (jb.phase == 1) {
          for (size_t f = jb.f0; f < jb.f0 + jb.nf; ++f) {
            memcpy(&quant[f * 128], h_sums + f * 128, static_cast<size_t>(ntab) * 64);
            sjpeg_hip_finalize_quant(reinterpret_cast<uint8_t(*)[64]>(&quant[f * 128]), min_quant, q_bias, &tables[f]);
          }
          return step_stats(jb, j);
        }
        for (size_t f = jb.f0; f < jb.f0 + jb.nf; ++f) {
          tables[f].flags = (tables[f].flags & ~SJPEG_HIP_QUANT_KEEP) | SJPEG_HIP_QUANT_REPLAY;
          sjpeg_hip_optimize_huffman(reinterpret_cast<const uint32_t*>(h_freq + f * kFreq), yuv_mode, &jspecs[f * 4], &tables[f]);
        }
        return step_encode(jb);
      };
      static const bool lanes_debug = getenv("SJPEG_HIP_BATCH_DEBUG") != nullptr;    // (measurement aid: the host's timeline on stderr)
      const auto t_lanes = std::chrono::steady_clock::now();
      auto lmark = [&](const char* what, size_t j) {
        if (lanes_debug) fprintf(stderr, "lanes %-22s job %zu  %8.1f us\n", what, j, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_lanes).count());
      };
      size_t done = 0;
      std::vector<size_t> lane_next(static_cast<size_t>(nlanes));      // the next job of a lane that has not started
      for (int l = 0; l < nlanes; ++l) lane_next[static_cast<size_t>(l)] = static_cast<size_t>(l);
      std::vector<long long> lane_busy(static_cast<size_t>(nlanes), -1);   // the job a lane is working on
      int rc_lanes = 0;
      while (done < njobs && rc_lanes == 0) {
        bool moved = false;
        for (int l = 0; l < nlanes && rc_lanes == 0; ++l) {
          const size_t ls = static_cast<size_t>(l);
          if (lane_busy[ls] < 0) {
            // (SJPEG_HIP_BATCH_STAGGER=1, an experiment that lost: job j starts only when job j - 1 has launched its statistics
            // pass, so that a histogram pass runs beside another KIND of pass instead of beside its twin -- 1.35 against 1.00 ms
            // for 32 4K frames: the persistent histogram launch leaves the other kernel one workgroup per CU.  Default: all lanes
            // start at once.)
            static const int stagger = getenv("SJPEG_HIP_BATCH_STAGGER") ? atoi(getenv("SJPEG_HIP_BATCH_STAGGER")) : 0;
            const bool held = stagger != 0 && lane_next[ls] < njobs && lane_next[ls] > 0 && jobs[lane_next[ls] - 1].phase < (stagger == 2 ? 3 : 2);
            if (lane_next[ls] < njobs && !held) {      // (a lane's jobs follow each other: the engine's scratch is one job's)
              lane_busy[ls] = static_cast<long long>(lane_next[ls]);
              lane_next[ls] += static_cast<size_t>(nlanes);
              rc_lanes = step_start(jobs[static_cast<size_t>(lane_busy[ls])], static_cast<size_t>(lane_busy[ls]));
              lmark("started", static_cast<size_t>(lane_busy[ls]));
              moved = true;
            }
          } else {
            const size_t j = static_cast<size_t>(lane_busy[ls]);
            const hipError_t q = hipEventQuery(sc.job_ev[j]);
            if (q == hipSuccess) {
              lmark(jobs[j].phase == 1 ? "sums here" : "counts here", j);
              rc_lanes = step_after(jobs[j], j);
              lmark(jobs[j].phase == 2 ? "statistics launched" : "encode launched", j);
              moved = true;
            } else if (q != hipErrorNotReady) {
              (void)hipGetLastError();
              rc_lanes = fail(SJPEG_HIP_ERUNTIME, std::string("hipEventQuery: ") + hipGetErrorString(q));
            } else {
              (void)hipGetLastError();
            }
          }
          if (rc_lanes == 0 && lane_busy[ls] >= 0 && jobs[static_cast<size_t>(lane_busy[ls])].phase == 3) { lane_busy[ls] = -1; ++done; moved = true; }
        }
        if (!moved) __builtin_ia32_pause();
      }
      // the caller's stream ends behind every lane, failed call or not (what was launched reads the caller's buffers)
      for (int l = 1; l < nlanes; ++l) {
        if (hipEventRecord(engine->lane_done[l], lane_st[l]) == hipSuccess) (void)hipStreamWaitEvent(st, engine->lane_done[l], 0);
        else (void)hipStreamSynchronize(lane_st[l]);
      }
      return rc_lanes;
    }
    // (in parts: the sums of a part and their read-back go to the side stream, behind the pass that made
    // the partials, so that the next part's pass starts right behind this one's)
    hipStream_t rs = st;
    if (nparts > 1) {
      engine->reduce_stream = engine->batch_side; engine->reduce_ev = sc.pass_done;
      engine->replay_total = nframes;
      rs = engine->batch_side;
      static const bool late_uploads = getenv("SJPEG_HIP_LATE_UPLOADS") != nullptr;       // (A/B: uploads in the call's own stream)
      if (!late_uploads) {
        // The early uploads overwrite what the PREVIOUS call's kernels may still be reading -- its per-frame tables (K1 on
        // its stream), its headers and header offsets (the stitch kernels; the engine's side stream in pipelined mode).  An
        // adaptive call's first upload follows a host wait on its own histogram pass, which sits behind that work on the
        // call's stream; a call with optimised tables alone (methods 1, 2) uploads at once, and the side stream is behind
        // nobody's wait: the upload stream is put behind both before anything goes to it (two asynchronous
        // method-1 batches back to back: test_back_to_back_batches_without_a_host_wait).
        if (int rco = order_on_stream(engine, st)) return rco;
        HIP_TRY(hipEventRecord(sc.call_begin, st));
        HIP_TRY(hipStreamWaitEvent(engine->batch_up, sc.call_begin, 0));
        if (engine->side_pending) {
          if (int rcs = side_mark(engine)) return rcs;
          HIP_TRY(hipStreamWaitEvent(engine->batch_up, engine->side_done, 0));
        }
        engine->up_stream = engine->batch_up;
      }
    }
    // device -> the pinned block, on the side stream: a kernel writes it over the bus (no runtime copy: stage_copy_kernel)
    auto read_back = [&](void* h_dst, const void* d_src, size_t bytes) -> int {
      void* const dv = sc.d_pinned == nullptr ? nullptr : static_cast<uint8_t*>(sc.d_pinned) + (static_cast<uint8_t*>(h_dst) - static_cast<uint8_t*>(sc.h_pinned));
      return copy_by_kernel(dv, d_src, bytes, rs, hipMemcpyDeviceToHost, d_src, h_dst);
    };
    static const int groups_env = getenv("SJPEG_HIP_SUMS_GROUPS") ? atoi(getenv("SJPEG_HIP_SUMS_GROUPS")) : 0;   // (A/B: 1 = one group a part)
    static const bool device_decide = getenv("SJPEG_HIP_HOST_REGRESSION") == nullptr;   // (A/B: the float half of the analysis on the host)
    // (two groups pay when the HOST fits the steps -- it fits the first half while the second is on its way; with the fit on
    // the device the second group's launches only lengthen the chain: 1.150 against 1.141 ms)
    auto sums_groups = [&](size_t nf) -> int { return (groups_env == 1 || nf < 8 || (device_decide && groups_env != 2)) ? 1 : 2; };
    static const bool no_coefs = getenv("SJPEG_HIP_NO_COEF_KEEP") != nullptr;       // (A/B: every pass from the pixels)
    engine->coefs_keep = adaptive && optimize && !no_coefs;
    if (adaptive) {
      for (int p = 0; p < nparts; ++p) {
        const size_t f0 = part_lo[p], nf = part_lo[p + 1] - f0;
        const sjpeg_hip_source ps = part_source(f0);
        uint32_t* const d_hist = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(sc.d_hist) + f0 * kHist);
        engine->replay_first = static_cast<int>(f0);
        int rc = sjpeg_hip_scan_histogram_src(engine, &ps, width, height, yuv_mode, static_cast<int>(nf), d_hist, stream);
        if (rc != 0) return rc;
        // the analysis sums in TWO groups, each with its read-back and event: the host fits the first half of the part
        // while the second is still on its way (what comes back between a part's histogram pass and its statistics
        // launch is on the critical path of the call: the device waits for that launch)
        for (int h = 0; h < sums_groups(nf); ++h) {
          const size_t g0 = f0 + (nf * h) / sums_groups(nf), gn = f0 + (nf * (h + 1)) / sums_groups(nf) - g0;
          uint8_t* const d_grp = static_cast<uint8_t*>(sc.d_sums) + g0 * (kSums + kTot);      // [gn][kSums] then [gn][kTot]
          rc = sjpeg_hip_adapt_sums(d_hist + (g0 - f0) * (kHist / sizeof(uint32_t)), static_cast<int>(gn), reinterpret_cast<const uint8_t(*)[64]>(&quant[0]), min_quant,
                                    reinterpret_cast<int64_t*>(d_grp), reinterpret_cast<int32_t*>(d_grp + gn * kSums), rs);
          if (rc != 0) return rc;
          if (device_decide) {
            // the regression too (adapt_decide_kernel): 128 bytes a frame come back instead of 52 KB, and the host's 7 us a frame
            // between a part's histogram pass and its statistics launch are gone
            uint8_t* const d_q = static_cast<uint8_t*>(sc.d_sums) + n * (kSums + kTot) + g0 * 128;
            if ((rc = adapt_decide(reinterpret_cast<const int64_t*>(d_grp), reinterpret_cast<const int32_t*>(d_grp + gn * kSums), static_cast<int>(gn),
                                   reinterpret_cast<const uint8_t(*)[64]>(&quant[0]), yuv_mode == SJPEG_HIP_YUV400 ? 1 : 2,
                                   qdelta_max_luma, qdelta_max_chroma, d_q, rs))) return rc;
            if (int rcc = read_back(h_sums + g0 * 128, d_q, gn * 128)) return rcc;
          } else {
            if (int rcc = read_back(h_sums + g0 * (kSums + kTot), d_grp, gn * (kSums + kTot))) return rcc;
          }
          HIP_TRY(hipEventRecord(sc.ev[2 * p + h], rs));
        }
      }
    }
    static const bool batch_debug = getenv("SJPEG_HIP_BATCH_DEBUG") != nullptr;      // (measurement aid: host timeline on stderr)
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char* what, int p) {
      static const auto t_epoch = std::chrono::steady_clock::now();
      if (batch_debug) fprintf(stderr, "batch %-18s part %d  %8.1f us  (abs %10.1f)\n", what, p, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(),
                               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_epoch).count());
    };
    mark("hist launched", -1);
    std::vector<sjpeg_hip_huffman_spec> specs(optimize ? n * 4 : 0);
    // (part p's regression runs while the device works on the histogram of part p + 1 / the statistics of part p - 1)
    for (int p = 0; p < nparts; ++p) {
      const size_t f0 = part_lo[p], nf = part_lo[p + 1] - f0;
      if (adaptive) {
        for (int h = 0; h < sums_groups(nf); ++h) {
          const size_t g0 = f0 + (nf * h) / sums_groups(nf), gn = f0 + (nf * (h + 1)) / sums_groups(nf) - g0;
          mark("wait sums", p);
          HIP_TRY(hipEventSynchronize(sc.ev[2 * p + h]));
          mark("sums here", p);
          const uint8_t* const grp = h_sums + g0 * (kSums + kTot);
          if (device_decide) {
            const int ntab = yuv_mode == SJPEG_HIP_YUV400 ? 1 : 2;
            for (size_t f = g0; f < g0 + gn; ++f) {
              memcpy(&quant[f * 128], h_sums + f * 128, static_cast<size_t>(ntab) * 64);
              sjpeg_hip_finalize_quant(reinterpret_cast<uint8_t(*)[64]>(&quant[f * 128]), min_quant, q_bias, &tables[f]);
            }
            continue;
          }
          for (size_t f = g0; f < g0 + gn; ++f) {
            sjpeg_hip_adapt_quant_sums(reinterpret_cast<const int64_t*>(grp + (f - g0) * kSums),
                                       reinterpret_cast<const int32_t*>(grp + gn * kSums + (f - g0) * kTot), yuv_mode,
                                       reinterpret_cast<uint8_t(*)[64]>(&quant[f * 128]), min_quant, q_bias,
                                       qdelta_max_luma, qdelta_max_chroma, &tables[f]);
          }
        }
      }
      mark("adapted", p);
      if (optimize) {
        // the statistics pass leaves its quantized blocks behind (144 B each) and the encode pass
        // replays them: no second colour conversion / DCT / quantization (the reference's stored
        // run/levels, src/enc.cc:121-129,374-386)
        for (size_t f = f0; f < f0 + nf; ++f) tables[f].flags |= SJPEG_HIP_QUANT_KEEP;
        engine->replay_total = nframes; engine->replay_first = static_cast<int>(f0);
        engine->coefs_use = engine->coefs_keep;
        const sjpeg_hip_source ps = part_source(f0);
        uint32_t* const d_freq = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(sc.d_freq) + f0 * kFreq);
        const int rc = sjpeg_hip_scan_symbol_stats_multi(engine, &ps, width, height, yuv_mode, static_cast<int>(nf), &tables[f0], d_freq, stream);
        if (rc != 0) return rc;
        if (int rcc = read_back(h_freq + f0 * kFreq, d_freq, nf * kFreq)) return rcc;
        dbg_mark("batch: counts read back");
        HIP_TRY(hipEventRecord(sc.ev[2 * kMaxParts + p], rs));
        dbg_mark("batch: event recorded");
        mark("stats launched", p);
      }
    }
    // (part p's table builder runs while the device counts the symbols of part p + 1 / codes part p - 1)
    std::vector<uint8_t> headers;
    std::vector<size_t> offs;
    for (int p = 0; p < nparts; ++p) {
      const size_t f0 = part_lo[p], nf = part_lo[p + 1] - f0;
      if (optimize) {
        mark("wait freq", p);
        HIP_TRY(hipEventSynchronize(sc.ev[2 * kMaxParts + p]));
        mark("freq here", p);
        for (size_t f = f0; f < f0 + nf; ++f) {
          tables[f].flags = (tables[f].flags & ~SJPEG_HIP_QUANT_KEEP) | SJPEG_HIP_QUANT_REPLAY;
          sjpeg_hip_optimize_huffman(reinterpret_cast<const uint32_t*>(h_freq + f * kFreq), yuv_mode, &specs[f * 4], &tables[f]);
        }
        engine->replay_total = nframes; engine->replay_first = static_cast<int>(f0);
      }
      mark("tables built", p);
      headers.clear();
      offs.assign(nf + 1, 0);
      uint8_t one[2048];
      for (size_t f = f0; f < f0 + nf; ++f) {
        const size_t hs = sjpeg_hip_make_header_ex(width, height, yuv_mode, reinterpret_cast<const uint8_t(*)[64]>(&quant[f * 128]),
                                                   optimize ? &specs[f * 4] : nullptr, one, sizeof(one));
        if (hs == 0) return fail(SJPEG_HIP_EINVAL, "header generation failed");
        headers.insert(headers.end(), one, one + hs);
        offs[f - f0 + 1] = headers.size();
      }
      const sjpeg_hip_source ps = part_source(f0);
      if (engine->replay_total > 0) engine->replay_first = static_cast<int>(f0);     // (where this part's tables / headers lie)
      const int rc_enc = sjpeg_hip_encode_scan_multi(engine, &ps, width, height, yuv_mode, static_cast<int>(nf), &tables[f0], headers.data(),
                                                     offs.data(), /*append_eoi=*/1, static_cast<uint8_t*>(d_out) + f0 * out_stride,
                                                     out_stride, d_sizes + f0, stream);
      if (rc_enc != 0) return rc_enc;
      mark("encode launched", p);
    }
    return 0;
  } catch (...) {
    return fail(SJPEG_HIP_ENOMEM, "out of host memory");
  }
}

// ---- exchange step of the multi-device batch path: the coded frames of one call, back to back ----
// (BASELINE.json config #4; the reference is single-threaded and has no counterpart.)  One
// launch, no host round trip: every workgroup sums the (16-byte aligned) sizes of the frames in
// front of its own and copies 16 bytes per thread.

__global__ __launch_bounds__(256) void compact_streams_kernel(const uint8_t* out, size_t out_stride, const unsigned long long* sizes,
                                                              int nframes, uint8_t* packed, unsigned long long capacity,
                                                              unsigned long long* offsets) {
  const int f = blockIdx.y, tid = threadIdx.x;
  __shared__ unsigned long long part[4];
  unsigned long long s = 0;
  for (int g = tid; g < f; g += 256) s += (sizes[g] + 15ull) & ~15ull;
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  const unsigned long long off = part[0] + part[1] + part[2] + part[3];
  const unsigned long long n = sizes[f], n16 = (n + 15ull) & ~15ull;
  if (blockIdx.x == 0 && tid == 0) {
    offsets[f] = off;
    // bytes needed, whether they fit or not; bit 63 says they did not (SJPEG_HIP_PACKED_OVERFLOW: the exchange
    // reads it in the rank's row and refuses on every rank)
    if (f == nframes - 1) offsets[nframes] = (off + n16) | (off + n16 > capacity ? (1ull << 63) : 0ull);
  }
  if (off + n16 > capacity) return;                            // the caller sees it in offsets[nframes]
  const uint4* const src = reinterpret_cast<const uint4*>(out + static_cast<size_t>(f) * out_stride);
  uint4* const dst = reinterpret_cast<uint4*>(packed + off);
  const size_t nch = static_cast<size_t>(n16 >> 4);
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + tid; i < nch; i += static_cast<size_t>(gridDim.x) * 256) {
    uint4 v = src[i];
    if (i == nch - 1 && (n & 15ull) != 0ull) {                 // zero the padding behind the last byte
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
      const uint32_t keep = static_cast<uint32_t>(n & 15ull);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t lo = 4u * k;
        w[k] = keep <= lo ? 0u : (keep >= lo + 4u ? w[k] : (w[k] & ((1u << (8u * (keep - lo))) - 1u)));
      }
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    dst[i] = v;
  }
}

int sjpeg_hip_compact_streams(const void* d_out, size_t out_stride, const uint64_t* d_sizes, int nframes,
                              void* d_packed, size_t packed_capacity, uint64_t* d_offsets, void* stream) {
  if (d_out == nullptr || d_sizes == nullptr || d_packed == nullptr || d_offsets == nullptr) {
    return fail(SJPEG_HIP_EINVAL, "sjpeg_hip_compact_streams: NULL argument");
  }
  if (nframes <= 0 || nframes > 65535) return fail(SJPEG_HIP_EINVAL, "sjpeg_hip_compact_streams: nframes must be 1 .. 65535");
  if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0 || (reinterpret_cast<uintptr_t>(d_packed) & 15u) != 0 || (out_stride & 15u) != 0) {
    return fail(SJPEG_HIP_EINVAL, "sjpeg_hip_compact_streams: d_out, d_packed and out_stride must be multiples of 16");
  }
  hipLaunchKernelGGL(compact_streams_kernel, dim3(64, nframes), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint8_t*>(d_out), out_stride, reinterpret_cast<const unsigned long long*>(d_sizes), nframes,
                     static_cast<uint8_t*>(d_packed), static_cast<unsigned long long>(packed_capacity),
                     reinterpret_cast<unsigned long long*>(d_offsets));
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- measurement aid: what a read-only streaming kernel reaches on this device ------------------
// (SURVEY section 8d asks for the achieved read bandwidth beside the 8 TB/s spec figure)

__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* p, size_t n, uint32_t* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) {
    const uint4 v = p[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x9e3779b9u) sink[0] = r;               // keeps the loads alive, practically never taken
}

int sjpeg_hip_debug_stream_read(const void* d_buf, size_t bytes, uint32_t* d_sink, void* stream) {
  if (d_buf == nullptr || d_sink == nullptr || bytes < 16) return SJPEG_HIP_EINVAL;
  hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(d_buf), bytes / 16, d_sink);
  return hipGetLastError() == hipSuccess ? 0 : SJPEG_HIP_ERUNTIME;
}

// ---- measurement aid: cycles a wave64 VALU instruction occupies a SIMD, per issue class -----------
// (the second roofline of K1, bench.py `roofline.valu`; tools/valu_rate.hip has the table of all ops)

#define SJPEG_VALU_RATE_KERNEL(NAME, ASM)                                                              \
__global__ __launch_bounds__(256) void NAME(uint32_t a, uint32_t b, uint32_t* sink) {                  \
  uint32_t r[8];                                                                                       \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + a;      \
  for (int it = 0; it < 1024; ++it) {                                                                  \
    _Pragma("unroll") for (int u = 0; u < 32; ++u) asm volatile(ASM : "+v"(r[u & 7]) : "v"(a), "v"(b)); \
  }                                                                                                    \
  uint32_t x = 0;                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) x ^= r[i];                                             \
  if (x == 0x12345u) sink[0] = 1;                                                                      \
}
SJPEG_VALU_RATE_KERNEL(valu_rate_kernel_slow, "v_perm_b32 %0, %0, %1, %2")
SJPEG_VALU_RATE_KERNEL(valu_rate_kernel_fast, "v_add_u32 %0, %0, %1")

// The shader clock as the shaders see it: a wave spins for `ticks` ticks of the device-wide 100 MHz counter and
// counts the cycles of the shader-clock counter meanwhile.  Enqueued on the stream of the work it is to describe,
// it runs right behind that work, before the clocks have come down (sysfs / rocm-smi show ~100 MHz the moment the
// queue is empty, and lag behind when it is not).
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
  const unsigned long long c1 = __builtin_readcyclecounter();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}

int sjpeg_hip_debug_shader_clock(float* mhz, void* stream) {
  if (mhz == nullptr) return SJPEG_HIP_EINVAL;
  unsigned long long* d = nullptr;
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)));
  const hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, d, 2000ull);      // 20 us
  const hipError_t e0 = hipGetLastError();
  const hipError_t e1 = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || h[1] == 0) return SJPEG_HIP_ERUNTIME;
  *mhz = static_cast<float>(static_cast<double>(h[0]) / static_cast<double>(h[1]) * 100.0);
  return 0;
}

int sjpeg_hip_debug_valu_rate(float cycles[2], void* stream) {
  if (cycles == nullptr) return SJPEG_HIP_EINVAL;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  uint32_t* d_sink = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_sink), 64));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const int wps = 8;                               // workgroups per CU = waves per SIMD (256 threads each)
  const dim3 grid(prop.multiProcessorCount * wps);
  for (int c = 0; c < 2; ++c) {
    for (int rep = 0; rep < 2; ++rep) {            // first launch warms up
      (void)hipEventRecord(e0, st);
      if (c == 0) hipLaunchKernelGGL(valu_rate_kernel_slow, grid, dim3(256), 0, st, 3u, 5u, d_sink);
      else hipLaunchKernelGGL(valu_rate_kernel_fast, grid, dim3(256), 0, st, 3u, 5u, d_sink);
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    cycles[c] = ms * 1e-3f * 2.4e9f / (1024.0f * 32.0f * wps);   // at the nominal 2.4 GHz
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d_sink);
  return hipGetLastError() == hipSuccess ? 0 : SJPEG_HIP_ERUNTIME;
}

}  // extern "C"
