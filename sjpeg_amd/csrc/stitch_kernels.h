// K2..K5: segment offsets, bit-granular placement into one stream per frame, chunk offsets, 0xFF stuffing.
// Part of the single translation unit scan_engine.hip: included there inside its anonymous
// namespace, after <hip/hip_runtime.h> and sjpeg_hip.h; not a stand-alone header.
// ------------------------------------------------------------------------------------
// K2: per frame, exclusive scan of segment bit lengths

struct StitchArgs {
  int nseg, nframes;
  const uint32_t* seg_nbits;
  unsigned long long* seg_off;       // [nframes][nseg+1]
  const uint32_t* seg_words;         // [nframes][nseg][slot_words]: the first slot_words words of every segment ...
  uint32_t slot_words;
  const uint32_t* pool;              // [nframes][pool_words]: ... the rest of it at seg_xbase[frame][seg] (NULL: slots hold everything)
  uint32_t pool_words;
  const uint32_t* seg_xbase;
  const uint32_t* pool_ctr;          // [nframes][2]: [1] != 0: the frame overran its pool (reports size 0)
  uint32_t* ubuf;                    // [nframes][ubuf_words] un-stuffed stream, MSB-first words
  size_t ubuf_words;
  uint32_t* chunk_ff;                // [nframes][max_chunks]
  unsigned long long* chunk_off;     // [nframes][max_chunks]
  uint32_t max_chunks;
  const uint8_t* header;
  uint32_t header_size;
  const uint32_t* hdr_off;           // per-frame headers: frame f owns header[hdr_off[f] .. hdr_off[f+1]) (else NULL: one for all)
  int append_eoi;
  uint8_t* out;
  size_t out_stride;
  unsigned long long* sizes;
  // packed output (sjpeg_hip_encode_scan_packed_src): frame f starts at out + pack_off[f], a multiple of 16, the
  // frames back to back; NULL: at out + f * out_stride.  out_stride stays the bytes a frame may take.
  unsigned long long* pack_off = nullptr;
  const unsigned long long* seg_nbits64;   // band stitch: lengths as uint64 (else NULL)
  unsigned long long* total_bits_out;      // band encode: where the bit count of the band goes (else NULL)
  uint32_t subs;                           // K3: waves per segment (1 unless segments are whole bands, or wide_subs)
  uint32_t wide_subs;                      // K3: an ordinary call whose (few) segments are cut into sub-ranges of 768 words
  int seg_first;                           // restart mode, band of a frame: frame-level index of segment 0 ...
  int rst_tail;                            // ... and whether the band's last interval gets its marker too (it is not the frame's last)
  // [nframes]: K2 takes the frame's "overran its pool" flag here and leaves the pool's counters at zero for the next
  // call; K3 .. K5 read the copy.  (NULL -- bands, which have no K4 --: K3 reads the counter itself.)
  uint32_t* frame_flags;
  // small launches: K5 works the chunk offsets out itself (every workgroup scans the frame's 0xFF counts, at most
  // 2048 of them) and its first workgroup does what else K4 does -- size, header, EOI: no K4 launch
  int fused_k4;
  // ... and K3 the segment offsets (every workgroup scans the frame's at most 2048 segment lengths into LDS; its first
  // workgroup leaves the total and the overrun flag behind for K5, whose first workgroup then zeroes the pool's counters;
  // K1 has cleared the frame's 0xFF counters): no K2 launch either -- K1, K3, K5
  int fused_k2;
};
// (frame_flags is written by K2 -- or by K3's first workgroup when K2 runs inside K3 -- from pool_ctr, and only when
// BOTH are there: a call with flags but no pool has no overrun to report, and nothing wrote the flags -- ADVICE r05)
__device__ __forceinline__ bool frame_overran(const StitchArgs& a, int frame) {
  if (a.pool_ctr == nullptr) return false;
  return a.frame_flags != nullptr ? a.frame_flags[frame] != 0u : a.pool_ctr[2 * frame + 1] != 0u;
}

__global__ __launch_bounds__(kThreads) void scan_seg_offsets(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  const int frame = blockIdx.x;
  const uint32_t* nb = a.seg_nbits + static_cast<size_t>(frame) * a.nseg;
  unsigned long long* off = a.seg_off + static_cast<size_t>(frame) * (a.nseg + 1);
  // (a band is shorter than 2^32 bits: sjpeg_hip_stitch_bands checks its capacity)
  auto len_of = [&](int i) -> uint32_t {
    return a.seg_nbits64 != nullptr ? static_cast<uint32_t>(a.seg_nbits64[static_cast<size_t>(frame) * a.nseg + i]) : nb[i];
  };
  // A thread owns a RUN of consecutive segments (8 at most per round; one round up to 2048 segments, i.e.
  // every frame but the very large ones): it adds them up, the workgroup scans the 256 sums, and the
  // thread writes the offsets of its run -- one scan (two barriers) per 2048 segments instead of one
  // per 256 (an 8K 4:4:4 frame has 6172 segments: 13.8 us with a scan per 256).
  constexpr int kRun = 8;
  unsigned long long running = 0;
  for (int base = 0; base < a.nseg; base += kThreads * kRun) {
    const int i0 = base + static_cast<int>(threadIdx.x) * kRun;
    uint32_t v[kRun];
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      v[j] = (i0 + j < a.nseg) ? len_of(i0 + j) : 0u;
      mine += v[j];
    }
    // (a segment has at most 425 000 bits: the 2048 of a round stay below 2^32; bands may be 2^32 - 1
    // bits each: their sums are scanned as two halves)
    unsigned long long at, round_total;
    if (a.seg_nbits64 == nullptr) {
      uint32_t total;
      const uint32_t ex = wg_exclusive_scan<kThreads>(static_cast<uint32_t>(mine), scratch, &total);
      at = running + ex;
      round_total = total;
    } else {
      uint32_t total_lo, total_hi;
      const uint32_t ex_lo = wg_exclusive_scan<kThreads>(static_cast<uint32_t>(mine & 0xffffffu), scratch, &total_lo);
      const uint32_t ex_hi = wg_exclusive_scan<kThreads>(static_cast<uint32_t>(mine >> 24), scratch, &total_hi);
      at = running + ex_lo + (static_cast<unsigned long long>(ex_hi) << 24);
      round_total = total_lo + (static_cast<unsigned long long>(total_hi) << 24);
    }
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      if (i0 + j < a.nseg) off[i0 + j] = at;
      at += v[j];
    }
    running += round_total;
  }
  if (threadIdx.x == 0) {
    off[a.nseg] = running;
    if (a.total_bits_out != nullptr) a.total_bits_out[frame] = running;
  }
  // K3 accumulates the 0xFF counts of the chunks with atomics: clear the ones this frame uses
  const unsigned long long U = (running + 7) >> 3;
  unsigned long long nch64 = (U + kChunkBytes - 1) / kChunkBytes;
  if (nch64 > a.max_chunks) nch64 = a.max_chunks;             // (such a frame is not stitched: K3, K4)
  const uint32_t nchunks = static_cast<uint32_t>(nch64);
  uint32_t* ff = a.chunk_ff + static_cast<size_t>(frame) * a.max_chunks;
  for (uint32_t i = threadIdx.x; i < nchunks; i += kThreads) ff[i] = 0;
  if (a.frame_flags != nullptr && a.pool_ctr != nullptr && threadIdx.x == 0) {
    a.frame_flags[frame] = a.pool_ctr[2 * frame + 1];
    const_cast<uint32_t*>(a.pool_ctr)[2 * frame] = 0u;
    const_cast<uint32_t*>(a.pool_ctr)[2 * frame + 1] = 0u;
  }
}

// ------------------------------------------------------------------------------------
// K3: place every segment in the continuous bit stream, and count 0xFF bytes per 4 KiB chunk

// 0x80 in every byte of w that equals 0xFF (exact, no carries between bytes)
__device__ __forceinline__ uint32_t ff_bytes(uint32_t w) {
  // low seven bits all set <=> + 1 carries into bit 7 (never out of the byte), and bit 7 itself set
  return ((w & 0x7f7f7f7fu) + 0x01010101u) & w & 0x80808080u;
}
__device__ __forceinline__ uint32_t count_ff(uint32_t w, int nbytes /*valid leading bytes, MSB first*/) {
  uint32_t n = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b < nbytes && ((w >> (24 - 8 * b)) & 0xffu) == 0xffu) ++n;
  }
  return n;
}

// One WAVE per SEGMENT (scatter form): a segment knows where its bits go (seg_off); its
// words are read coalesced, funnel-shifted to the destination alignment, and every word of the
// continuous stream whose FIRST bit lies inside the segment is written.  Only the last of those
// words needs bits of the following segment(s), or the final 1-bit padding
// (src/bit_writer.cc:107-116).  0xFF bytes are counted per 4 KiB chunk of the stream (atomics;
// cleared by K2).
// The kernel is latency-bound by construction (a few KB per workgroup), so the dependent chain
// is cut to ONE round trip: destination word i always needs source words i and i + 1 whatever
// the offset (only the shift depends on it), so the first kSpec batches of source words are
// requested before the offsets have arrived.  Earlier forms (a workgroup per 4 KiB chunk with a
// search; per group of segments) spent 35-50 us in chains of 3-5 dependent loads.
constexpr int kSpec = 12;                                   // speculative batches of 64 words: segments up to 3 KiB
constexpr int kPlaceLanes = 64;                             // one WAVE per segment, four segments per workgroup
constexpr int kWideSpec = 3;                                // wide form: speculative batches of 256 words
struct __attribute__((packed, aligned(4))) Words4 { uint32_t w[4]; };   // 16 bytes at any word address
constexpr int kFusedSegs = 2048;                            // K2 inside K3: frames of up to 2048 segments (FUSED = 1)
constexpr int kFusedSegsBig = 16384;                        // FUSED = 2, ONE large frame (8K 4:4:4 = 6172 segments): sums on demand
template <int FUSED>
__global__ __launch_bounds__(kThreads) void place_segments(const StitchArgs a) {
  __shared__ uint32_t off_lds[FUSED == 1 ? kFusedSegs + 1 : 16];   // (32 bits: a fused frame has less than 2^32 bits of stream)
  __shared__ uint32_t scratch_k2[16];
  const int frame = blockIdx.y;
  const uint32_t* const nb_f = a.seg_nbits + static_cast<size_t>(frame) * a.nseg;
  // FUSED = 2: the first segment this workgroup places; off_lds[0..7] = offsets of segments sc_first .. sc_first + 7,
  // off_lds[8] = the frame's total
  const int sc_first = static_cast<int>((blockIdx.x * (kThreads / 64)) / a.subs);
  if (FUSED == 1) {
    // K2's scan, by every workgroup for itself (eight lengths per thread, one scan; the sums stay below 2^32: such a
    // frame has at most 8 MiB of stream); the loads go out with the speculative ones below
    const uint32_t* const nb = nb_f;
    constexpr int kRun = kFusedSegs / kThreads;
    const int i0 = static_cast<int>(threadIdx.x) * kRun;
    uint32_t v[kRun], mine = 0;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      v[j] = (i0 + j < a.nseg) ? nb[i0 + j] : 0u;
      mine += v[j];
    }
    uint32_t total;
    uint32_t at = wg_exclusive_scan<kThreads>(mine, scratch_k2, &total);
#pragma unroll
    for (int j = 0; j < kRun; ++j) { off_lds[i0 + j] = at; at += v[j]; }
    if (threadIdx.x == 0) {
      off_lds[a.nseg] = total;
      if (blockIdx.x == 0) {
        a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg] = total;      // (K5 reads the total here)
        if (a.frame_flags != nullptr && a.pool_ctr != nullptr) a.frame_flags[frame] = a.pool_ctr[2 * frame + 1];
      }
    }
    __syncthreads();
  }
  if (FUSED == 2) {
    // One large frame: a workgroup needs the offsets of its own four segments, of the two behind them, and the total --
    // not the frame's whole scan in LDS (6172 segments of an 8K 4:4:4 frame: K2 was a launch of ONE workgroup, 10 us
    // between K1 and K3).  Every thread adds up a strided share of the lengths -- those in front of the workgroup's
    // first segment, and all of them --, two workgroup sums, and eight threads finish the window.  Coalesced loads of
    // an array that lies in the L2.
    uint32_t front = 0, all = 0;
    for (int i = threadIdx.x; i < a.nseg; i += kThreads) {
      const uint32_t v = nb_f[i];
      all += v;
      front += i < sc_first ? v : 0u;
    }
    uint32_t front_total, all_total;
    (void)wg_exclusive_scan<kThreads>(front, scratch_k2, &front_total);
    (void)wg_exclusive_scan<kThreads>(all, scratch_k2, &all_total);
    if (threadIdx.x < 8) {
      uint32_t at = front_total;
      for (int j = 0; j < static_cast<int>(threadIdx.x); ++j) at += sc_first + j < a.nseg ? nb_f[sc_first + j] : 0u;
      off_lds[threadIdx.x] = at;
    }
    if (threadIdx.x == 8) {
      off_lds[8] = all_total;
      if (blockIdx.x == 0) {
        a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg] = all_total;  // (K5 reads the total here)
        if (a.frame_flags != nullptr && a.pool_ctr != nullptr) a.frame_flags[frame] = a.pool_ctr[2 * frame + 1];
      }
    }
    __syncthreads();
  }
  // a wave takes words [sub * kSpec * 64, ...) of one segment; normal segments have one wave
  // (subs == 1, the loop below takes the rare longer rest), whole bands are cut into many
  const uint32_t unit = blockIdx.x * (kThreads / kPlaceLanes) + (threadIdx.x >> 6);
  const int sc0 = static_cast<int>(unit / a.subs);
  const uint32_t ibase = (unit % a.subs) * (a.wide_subs ? kWideSpec * 256u : kSpec * kPlaceLanes);
  if (sc0 >= a.nseg) return;
  const unsigned long long* const off_g = a.seg_off + static_cast<size_t>(frame) * (a.nseg + 1);
  auto off = [&](int i) -> unsigned long long {
    if (FUSED == 1) return off_lds[i];
    if (FUSED == 2) {
      if (i >= a.nseg) return off_lds[8];
      if (i - sc_first < 8) return off_lds[i - sc_first];
      unsigned long long at = off_lds[7];          // (a word finished from segments further behind: short ones, rare)
      for (int j = sc_first + 7; j < i; ++j) at += nb_f[j];
      return at;
    }
    return off_g[i];
  };
  const uint32_t* segw = a.seg_words + static_cast<size_t>(frame) * a.nseg * a.slot_words;
  const uint32_t* src = segw + static_cast<size_t>(sc0) * a.slot_words;
  // Two forms of the same loads.  WIDE (every ordinary call): a lane takes 4 consecutive words per batch
  // of 256 -- one 16-byte load plus the word behind them -- so that a segment costs 6 load and 2-3
  // store instructions instead of 24 and 11: with one dword per lane the kernel was bound by the
  // number of memory instructions, not by bytes.  NARROW (bands cut into sub-ranges, slots that are
  // not a multiple of 16 bytes): one word per lane.
  // (wide_subs: an ordinary call with few segments cuts them into sub-ranges of kWideSpec * 256 words too,
  // one wave each, so that a long segment is not one wave's chain of round trips)
  const bool wide = (a.subs == 1u || a.wide_subs) && (a.slot_words & 3u) == 0u && a.slot_words >= 776u;      // uniform
  uint32_t spec[kSpec][2];
  uint4 wq[kWideSpec];
  uint32_t wx[kWideSpec];
  if (wide) {
#pragma unroll
    for (int k = 0; k < kWideSpec; ++k) {                    // words 0 .. 771: inside the slot whatever the length
      // (a last sub-range may reach behind the slot: clamped like the batches of the loop below)
      const uint32_t i = min(ibase + 256u * k + 4u * (threadIdx.x & 63), a.slot_words - 4u);
      wq[k] = *reinterpret_cast<const uint4*>(src + i);
      wx[k] = src[min(i + 4u, a.slot_words - 1u)];
    }
  } else {
#pragma unroll
    for (int k = 0; k < kSpec; ++k) {                        // inside the slot whatever the length
      const uint32_t i = min(ibase + k * kPlaceLanes + (threadIdx.x & 63), a.slot_words - 2u);
      spec[k][0] = src[i];
      spec[k][1] = src[i + 1];
    }
  }
  // what the segment's LAST word needs, requested with everything else: the first word of the segment
  // behind it, and where that one ends (is it long enough to fill the word?)
  const bool has_next = sc0 + 1 < a.nseg;                   // uniform
  const uint32_t next_first = has_next ? src[a.slot_words] : 0u;
  const unsigned long long b0 = off(sc0), b1 = off(sc0 + 1);
  const unsigned long long b2 = off(has_next ? sc0 + 2 : sc0 + 1);
  const unsigned long long T = off(a.nseg);                 // total bits
  const unsigned long long U = (T + 7) >> 3;                // bytes incl. 1-bit padding
  // a frame whose stream is longer than the scratch sized from out_stride cannot fit its output slot
  // either; one that overran its pool has words missing: K4 reports size 0 for both, nothing to place
  // (FUSED: K2's copy of the flag does not exist yet -- this kernel's first workgroup makes it, for K5)
  const bool overran = FUSED ? (a.pool_ctr != nullptr && a.pool_ctr[2 * frame + 1] != 0u) : frame_overran(a, frame);
  if (((U + 3) >> 2) + 1 > a.ubuf_words || overran) return;
  // word i of segment sc: in its slot, or (the rare long segment) in the frame's pool
  const uint32_t* const pool_f = a.pool == nullptr ? nullptr : a.pool + static_cast<size_t>(frame) * a.pool_words;
  const uint32_t* const xbase_f = a.seg_xbase == nullptr ? nullptr : a.seg_xbase + static_cast<size_t>(frame) * a.nseg;
  auto seg_word = [&](int sc, uint32_t i) -> uint32_t {
    if (i < a.slot_words) return segw[static_cast<size_t>(sc) * a.slot_words + i];
    if (xbase_f == nullptr) return 0u;
    const uint32_t xb = xbase_f[sc];
    return xb == 0xffffffffu ? 0u : pool_f[xb + (i - a.slot_words)];
  };
  uint32_t* ub = a.ubuf + static_cast<size_t>(frame) * a.ubuf_words;
  uint32_t* cff = a.chunk_ff + static_cast<size_t>(frame) * a.max_chunks;
  const int lane = threadIdx.x & 63;
  const unsigned long long wbeg = (b0 + 31) >> 5;
  const unsigned long long wend = (sc0 == a.nseg - 1) ? ((U + 3) >> 2) : ((b1 + 31) >> 5);
  const uint32_t nwords = static_cast<uint32_t>(wend - wbeg);
  const uint32_t lead = static_cast<uint32_t>(wbeg * 32 - b0);           // bits of the segment in front of word wbeg (< 32)
  const uint32_t len = static_cast<uint32_t>(b1 - b0);
  uint32_t* dst = ub + wbeg;
  const uint32_t wbase = static_cast<uint32_t>(wbeg);                     // < 2^32 words per frame
  uint32_t ff_acc = 0;                                      // 0xFF bytes seen by this lane in chunk ff_chunk
  uint32_t ff_chunk = 0xffffffffu;                          // wave-uniform
  auto ff_flush = [&]() {
    uint32_t sum = ff_acc;
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if (lane == 0 && sum != 0u) atomicAdd(&cff[ff_chunk], sum);
    ff_acc = 0;
  };
  // destination word i of this segment, bit by bit from wherever its bits are: the word that runs over the
  // end of the segment is finished from the following ones, and from 1-bits behind the frame's last
  auto gather_word = [&](uint32_t i) -> uint32_t {
    uint32_t outw = 0;
    int need = 32, sc = sc0;
    unsigned long long p = (wbeg + i) * 32, c_beg = b0, c_end = b1;
    while (need > 0 && p < T) {
      while (p >= c_end) { ++sc; c_beg = c_end; c_end = off(sc + 1); }
      const unsigned long long avail = c_end - p;
      const int take = avail < static_cast<unsigned long long>(need) ? static_cast<int>(avail) : need;
      const uint32_t rr = static_cast<uint32_t>(p - c_beg);
      const unsigned long long two = (static_cast<unsigned long long>(seg_word(sc, rr >> 5)) << 32) | seg_word(sc, (rr >> 5) + 1);
      const uint32_t bits = static_cast<uint32_t>((two << (rr & 31)) >> (64 - take));
      outw |= bits << (need - take);
      need -= take;
      p += take;
    }
    if (need > 0) outw |= (need == 32) ? 0xffffffffu : ((1u << need) - 1u);   // past the end: 1-bits
    return outw;
  };
  auto one = [&](uint32_t i, uint32_t v0, uint32_t v1) {
    uint32_t ffs = 0;
    if (i < nwords) {
      const uint32_t r = lead + 32u * i;                                  // first source bit of this word
      uint32_t outw;
      if (r + 32u <= len) {
        outw = lead ? __builtin_amdgcn_alignbit(v0, v1, 32u - lead) : v0;   // (v0:v1) >> (32 - lead)
      } else {
        outw = gather_word(i);
      }
      dst[i] = outw;
      const unsigned long long byte0 = (wbeg + i) * 4;
      const int valid = byte0 >= U ? 0 : (U - byte0 >= 4 ? 4 : static_cast<int>(U - byte0));
      ffs = valid == 4 ? static_cast<uint32_t>(__popc(ff_bytes(outw))) : count_ff(outw, valid);
    }
    // the 64 words of a wave sit in one chunk unless they straddle a boundary
    const uint32_t chunk = (wbase + i) >> 10;
    const uint32_t chunk0 = __builtin_amdgcn_readfirstlane(chunk);
    if (chunk0 != ff_chunk) {                                // uniform
      if (ff_chunk != 0xffffffffu) ff_flush();
      ff_chunk = chunk0;
    }
    if (chunk == chunk0) ff_acc += ffs;
    else if (ffs != 0u) atomicAdd(&cff[chunk], ffs);
  };
  // source words this wave reads: up to word `nwords` of its segment.  All inside the slot (every
  // ordinary segment): the speculative loads are the data.  Otherwise every word goes through the
  // slot / pool mapping (wave-uniform branch: the common path is the one without it).
  const bool long_seg = ((len + lead + 31u) >> 5) + 2u > a.slot_words;
  if (!long_seg && wide) {
    // destination words that lie entirely inside the segment: the funnel shift is all there is to them
    const uint32_t n_int = len >= lead + 32u ? (len - lead) >> 5 : 0u;
    // At most ONE word of a segment runs over its end (index n_int): its low bits are the first bits of
    // the segment behind (or the 1-bit padding behind the frame's last segment).  Both are at hand
    // (next_first above) unless the segment behind is shorter than what the word lacks -- tiny
    // segments of tiny pictures, empty bands -- which takes the general path of one().
    const uint32_t edge_bits = len > lead ? (len - lead) & 31u : 0u;      // bits of the edge word that are this segment's
    const bool edge_easy = !has_next || (b2 - b1) >= 32u - edge_bits;
    // bytes of the edge word that belong to the stream: four, unless the frame ends inside it -- which
    // is the LAST segment's edge word, or an earlier segment's when everything behind it is so short
    // that it starts no word of its own (flat pictures with one-bit codes: found by the soak)
    const unsigned long long edge_byte0 = (wbeg + n_int) * 4;
    const uint32_t edge_valid = edge_byte0 >= U ? 0u : (U - edge_byte0 >= 4 ? 4u : static_cast<uint32_t>(U - edge_byte0));
    auto batch = [&](uint32_t i0, uint4 q, uint32_t x) {     // destination words i0 .. i0 + 255, four per lane
      const uint32_t i = i0 + 4u * lane;
      const uint32_t v[5] = {q.x, q.y, q.z, q.w, x};
      const bool full = i0 + 256u <= n_int;                  // (uniform) all of them inside the segment
      Words4 o;
      uint32_t f[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o.w[u] = lead ? __builtin_amdgcn_alignbit(v[u], v[u + 1], 32u - lead) : v[u];
        f[u] = static_cast<uint32_t>(__popc(ff_bytes(o.w[u])));
      }
      if (full) {
        *reinterpret_cast<Words4*>(dst + i) = o;
      } else {
        const bool edge_here = n_int < nwords && n_int >= i0 && n_int < i0 + 256u;      // uniform
        // (general form: every lane works the same word out of the same addresses -- no divergence in
        // front of the wave-wide accounting below -- and the lane that owns it keeps it)
        const uint32_t gathered = (edge_here && !edge_easy) ? gather_word(n_int) : 0u;
        const uint32_t hi_mask = edge_bits ? ~(0xffffffffu >> edge_bits) : 0u;
        const uint32_t tail = (has_next ? next_first : 0xffffffffu) >> edge_bits;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t idx = i + u;
          if (edge_here && idx == n_int) {
            o.w[u] = edge_easy ? ((o.w[u] & hi_mask) | tail) : gathered;
            f[u] = edge_valid == 4u ? static_cast<uint32_t>(__popc(ff_bytes(o.w[u]))) : count_ff(o.w[u], static_cast<int>(edge_valid));
          }
          if (idx < nwords) dst[idx] = o.w[u]; else f[u] = 0;
        }
      }
      const uint32_t ffs = f[0] + f[1] + f[2] + f[3];
      const uint32_t c_lo = (wbase + i) >> 10, c_hi = (wbase + i + 3u) >> 10;
      const uint32_t chunk0 = __builtin_amdgcn_readfirstlane(c_lo);
      if (chunk0 != ff_chunk) {                              // uniform
        if (ff_chunk != 0xffffffffu) ff_flush();
        ff_chunk = chunk0;
      }
      if (c_lo == chunk0 && c_hi == chunk0) {
        ff_acc += ffs;
      } else if (ffs != 0u) {                                // the lane's words straddle a chunk boundary, or lie behind it
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t c = (wbase + i + u) >> 10;
          if (f[u] != 0u) { if (c == chunk0) ff_acc += f[u]; else atomicAdd(&cff[c], f[u]); }
        }
      }
    };
#pragma unroll
    for (int k = 0; k < kWideSpec; ++k) {
      if (ibase + 256u * k < nwords) batch(ibase + 256u * k, wq[k], wx[k]);
    }
    if (a.wide_subs) { if (ff_chunk != 0xffffffffu) ff_flush(); return; }   // (the other sub-ranges have their own waves)
    // longer segments: the words of batch k + 1 are requested before batch k is worked on
    // A lane whose four words start inside the slot must get exactly those (a segment that is not
    // `long_seg` has up to slot_words - 2 of them); only the fifth word of the slot's last lane,
    // needed by destination word slot_words - 1 alone, and the lanes behind the slot, all of
    // whose words are unused, are clamped.
    auto fetch = [&](uint32_t i0, uint4* q, uint32_t* x) {
      const uint32_t i = min(i0 + 4u * lane, a.slot_words - 4u);
      *q = *reinterpret_cast<const uint4*>(src + i);
      *x = src[min(i + 4u, a.slot_words - 1u)];
    };
    uint4 qn = make_uint4(0, 0, 0, 0);
    uint32_t xn = 0;
    if (256u * kWideSpec < nwords) fetch(256u * kWideSpec, &qn, &xn);
    for (uint32_t i0 = 256u * kWideSpec; i0 < nwords; i0 += 256u) {
      const uint4 qc = qn;
      const uint32_t xc = xn;
      if (i0 + 256u < nwords) fetch(i0 + 256u, &qn, &xn);
      batch(i0, qc, xc);
    }
  } else if (!long_seg) {
#pragma unroll
    for (int k = 0; k < kSpec; ++k) {
      if (ibase + static_cast<uint32_t>(k) * kPlaceLanes < nwords) one(ibase + k * kPlaceLanes + lane, spec[k][0], spec[k][1]);
    }
    if (a.subs == 1u) {
      for (uint32_t i0 = kSpec * kPlaceLanes; i0 < nwords; i0 += kPlaceLanes) {
        const uint32_t i = i0 + lane;
        one(i, src[i], src[i + 1]);
      }
    }
  } else {
    // (a long segment of a call with wide sub-ranges: the last of them takes what lies behind the slot)
    const bool takes_rest = a.subs == 1u || (a.wide_subs && (unit % a.subs) == a.subs - 1u);
    const uint32_t iend = takes_rest ? nwords : min(nwords, ibase + kSpec * kPlaceLanes);
    for (uint32_t i0 = ibase; i0 < iend; i0 += kPlaceLanes) {
      const uint32_t i = i0 + lane;
      one(i, seg_word(sc0, i), seg_word(sc0, i + 1));
    }
  }
  if (ff_chunk != 0xffffffffu) ff_flush();
}

__device__ __forceinline__ uint8_t* frame_out(const StitchArgs& a, int frame) {
  return a.pack_off != nullptr ? a.out + a.pack_off[frame] : a.out + static_cast<size_t>(frame) * a.out_stride;
}

// ------------------------------------------------------------------------------------
// K4: per frame, exclusive scan of per-chunk 0xFF counts; final stream size

__global__ __launch_bounds__(kThreads) void scan_chunk_offsets(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  const int frame = blockIdx.x;
  const unsigned long long T = a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg];
  const unsigned long long U = (T + 7) >> 3;
  unsigned long long nch64 = (U + kChunkBytes - 1) / kChunkBytes;
  if (nch64 > a.max_chunks) nch64 = a.max_chunks;             // (a frame longer than the scratch: size 0 below)
  const uint32_t nchunks = static_cast<uint32_t>(nch64);
  const uint32_t* ff = a.chunk_ff + static_cast<size_t>(frame) * a.max_chunks;
  unsigned long long* co = a.chunk_off + static_cast<size_t>(frame) * a.max_chunks;
  // (runs of eight chunks per thread, like K2: one scan per 2048 chunks = 8 MiB of stream)
  constexpr uint32_t kRun = 8;
  unsigned long long running = 0;
  for (uint32_t base = 0; base < nchunks; base += kThreads * kRun) {
    const uint32_t i0 = base + threadIdx.x * kRun;
    uint32_t v[kRun], mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < kRun; ++j) {
      v[j] = (i0 + j < nchunks) ? ff[i0 + j] : 0u;           // (at most 4096 each)
      mine += v[j];
    }
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan<kThreads>(mine, scratch, &total);
    unsigned long long at = running + ex;
#pragma unroll
    for (uint32_t j = 0; j < kRun; ++j) {
      if (i0 + j < nchunks) co[i0 + j] = at;
      at += v[j];
    }
    running += total;
  }
  // a frame that does not fit the caller's slot reports size 0 and is not written
  const unsigned long long body = U + running;
  const uint32_t hoff = a.hdr_off ? a.hdr_off[frame] : 0u;
  const uint32_t hsize = a.hdr_off ? a.hdr_off[frame + 1] - hoff : a.header_size;
  const unsigned long long size = hsize + body + (a.append_eoi ? 2 : 0);
  const bool fits = size <= a.out_stride && ((U + 3) >> 2) + 1 <= a.ubuf_words && !frame_overran(a, frame);
  // (without K2's copy of the flag this is the last reader of the frame's pool counters: it leaves them at zero)
  __syncthreads();
  if (a.frame_flags == nullptr && a.pool_ctr != nullptr && threadIdx.x == 0) {
    const_cast<uint32_t*>(a.pool_ctr)[2 * frame] = 0u;
    const_cast<uint32_t*>(a.pool_ctr)[2 * frame + 1] = 0u;
  }
  if (threadIdx.x == 0) a.sizes[frame] = fits ? size : 0ull;
  // (packed output: where the frame starts is only known once every frame has its size -- pack_frames())
  if (a.pack_off != nullptr) return;
  uint8_t* dst = a.out + static_cast<size_t>(frame) * a.out_stride;
  if (threadIdx.x == 0 && fits && a.append_eoi) {
    dst[hsize + body] = 0xff;
    dst[hsize + body + 1] = 0xd9;
  }
  // header bytes in front of the entropy segment
  if (fits) {
    for (uint32_t i = threadIdx.x; i < hsize; i += kThreads) dst[i] = a.header[hoff + i];
  }
}

// K4b (packed output only): where every frame starts -- exclusive scan of the sizes, each rounded up to 16 --,
// then header, EOI and the zero padding behind it, one workgroup per frame (blockIdx.x - 1; workgroup 0 scans).
// Two launches: pack_frame_offsets<<<1>>> then pack_frame_edges<<<nframes>>>.
__global__ __launch_bounds__(kThreads) void pack_frame_offsets(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  unsigned long long running = 0;
  for (int f0 = 0; f0 < a.nframes; f0 += kThreads) {
    const int f = f0 + static_cast<int>(threadIdx.x);
    const unsigned long long mine = f < a.nframes ? ((a.sizes[f] + 15ull) & ~15ull) : 0ull;
    // (a frame is below 4 GiB: out_stride is checked by the host; the scan runs on 16-byte units in 32 bits)
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan<kThreads>(static_cast<uint32_t>(mine >> 4), scratch, &total);
    if (f < a.nframes) a.pack_off[f] = running + (static_cast<unsigned long long>(ex) << 4);
    running += static_cast<unsigned long long>(total) << 4;
  }
  if (threadIdx.x == 0) a.pack_off[a.nframes] = running;
}

__global__ __launch_bounds__(kThreads) void pack_frame_edges(const StitchArgs a) {
  const int frame = blockIdx.x;
  const unsigned long long size = a.sizes[frame];
  if (size == 0) return;
  const uint32_t hoff = a.hdr_off ? a.hdr_off[frame] : 0u;
  const uint32_t hsize = a.hdr_off ? a.hdr_off[frame + 1] - hoff : a.header_size;
  uint8_t* const dst = a.out + a.pack_off[frame];
  for (uint32_t i = threadIdx.x; i < hsize; i += kThreads) dst[i] = a.header[hoff + i];
  if (threadIdx.x == 0 && a.append_eoi) { dst[size - 2] = 0xff; dst[size - 1] = 0xd9; }
  const uint32_t pad = static_cast<uint32_t>((16u - (size & 15u)) & 15u);
  if (threadIdx.x < pad) dst[size + threadIdx.x] = 0;
}

// ------------------------------------------------------------------------------------
// K5: byte stuffing into the caller's slot

// A thread takes 16 bytes of the chunk; a workgroup scan of the 0xFF counts gives every thread the place
// of its bytes in the stuffed chunk, which is staged in LDS laid out like the destination (LDS byte i
// <-> destination byte dalign + i, dalign 16-byte aligned) and copied out with aligned 16-byte stores.
// Staging is done in ALIGNED DWORDS by the rule of K3: a thread writes every dword whose FIRST byte is
// one of its own, and completes the last of them with the first bytes of the thread behind it -- which
// it has loaded itself (a fifth source word), so no two threads ever exchange anything.  94 % of the
// threads hold no 0xFF byte (1 byte in 256 of an entropy-coded stream): four byte-aligns and two
// ds_write2_b32 instead of the sixteen byte stores of the general path, which the others keep.
constexpr uint32_t kFusedChunks = 2048;                     // K4 inside K5: frames of up to 8 MiB of un-stuffed stream (FUSED = 1)
constexpr uint32_t kFusedChunksBig = 1u << 17;              // FUSED = 2, ONE large frame of up to 512 MiB of stream: sums on demand
template <int FUSED>
__global__ __launch_bounds__(kThreads) void stuff_chunks(const StitchArgs a) {
  constexpr uint32_t kCapChunks = FUSED == 2 ? kFusedChunksBig : kFusedChunks;
  __shared__ uint32_t scratch[16];
  __shared__ __attribute__((aligned(16))) uint8_t stage[2 * kChunkBytes + 64];
  __shared__ uint32_t fused_off[FUSED == 1 ? kFusedChunks : 1];   // FUSED = 1: 0xFF bytes in front of every chunk of the frame
  uint32_t big_off = 0;                                      // FUSED = 2: 0xFF bytes in front of this workgroup's chunk at hand
  const int frame = blockIdx.y;
  const unsigned long long T = a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg];
  const unsigned long long U = (T + 7) >> 3;
  const uint32_t nchunks = static_cast<uint32_t>((U + kChunkBytes - 1) / kChunkBytes);
  const uint32_t* ub = a.ubuf + static_cast<size_t>(frame) * a.ubuf_words;
  const unsigned long long* co = a.chunk_off + static_cast<size_t>(frame) * a.max_chunks;
  const uint32_t hoff = a.hdr_off ? a.hdr_off[frame] : 0u;
  const uint32_t hsize = a.hdr_off ? a.hdr_off[frame + 1] - hoff : a.header_size;
  const uint32_t* const ff_f = a.chunk_ff + static_cast<size_t>(frame) * a.max_chunks;
  if (FUSED) {
    // K4's scan, by every workgroup for itself (the counts of at most 2048 chunks: eight per thread, one scan): no
    // launch between K3 and this kernel.  Workgroup 0 also reports the size and writes header and EOI.
    const uint32_t* ff = ff_f;
    const uint32_t nch = nchunks < a.max_chunks ? nchunks : a.max_chunks;     // (more: the frame does not fit, below)
    uint32_t total;
    if (FUSED == 1) {
      constexpr uint32_t kRun = kFusedChunks / kThreads;
      const uint32_t i0 = threadIdx.x * kRun;
      uint32_t v[kRun], mine = 0;
#pragma unroll
      for (uint32_t j = 0; j < kRun; ++j) {
        v[j] = (i0 + j < nch) ? ff[i0 + j] : 0u;
        mine += v[j];
      }
      uint32_t at = wg_exclusive_scan<kThreads>(mine, scratch, &total);
#pragma unroll
      for (uint32_t j = 0; j < kRun; ++j) { fused_off[i0 + j] = at; at += v[j]; }
    } else {
      // one large frame (an 8K 4:4:4 q90 frame: 6204 chunks): the workgroup needs the count in front of ITS chunk and
      // the total, not the whole scan in LDS -- strided loads of an array in the L2, two workgroup sums; the counts
      // between this chunk and the workgroup's next one are added in the loop below
      uint32_t front = 0, all = 0;
      for (uint32_t i = threadIdx.x; i < nch; i += kThreads) {
        const uint32_t v = ff[i];
        all += v;
        front += i < blockIdx.x ? v : 0u;
      }
      (void)wg_exclusive_scan<kThreads>(front, scratch, &big_off);
      (void)wg_exclusive_scan<kThreads>(all, scratch, &total);
    }
    const unsigned long long body = U + total;
    const unsigned long long size = hsize + body + (a.append_eoi ? 2 : 0);
    const bool fits = nchunks <= kCapChunks && size <= a.out_stride && ((U + 3) >> 2) + 1 <= a.ubuf_words && !frame_overran(a, frame);
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0) a.sizes[frame] = fits ? size : 0ull;
      if (a.fused_k2 && a.pool_ctr != nullptr && threadIdx.x == 0) {    // (K3, their last reader, is through)
        const_cast<uint32_t*>(a.pool_ctr)[2 * frame] = 0u;
        const_cast<uint32_t*>(a.pool_ctr)[2 * frame + 1] = 0u;
      }
      if (fits) {
        uint8_t* const dst = a.out + static_cast<size_t>(frame) * a.out_stride;
        if (threadIdx.x == 0 && a.append_eoi) { dst[hsize + body] = 0xff; dst[hsize + body + 1] = 0xd9; }
        for (uint32_t i = threadIdx.x; i < hsize; i += kThreads) dst[i] = a.header[hoff + i];
      }
    }
    if (!fits) return;                                      // (uniform over the frame's workgroups)
    __syncthreads();                                        // fused_off is complete
  } else {
    if (a.sizes[frame] == 0) return;                        // did not fit (see K4)
  }
  uint8_t* const dst0 = frame_out(a, frame) + hsize;
  // the 16 bytes of this thread in the NEXT chunk of the workgroup (and the word behind them) are
  // requested while the current ones are stuffed
  auto fetch = [&](uint32_t chunk, uint4* q, uint32_t* behind, int* valid, unsigned long long* off) {
    *q = make_uint4(0, 0, 0, 0);
    *behind = 0;
    *valid = 0;
    *off = 0;
    if (chunk >= nchunks) return;
    const unsigned long long w0 = static_cast<unsigned long long>(chunk) * kChunkWords + threadIdx.x * 4;
    const unsigned long long byte0 = w0 * 4;
    *off = FUSED == 1 ? static_cast<unsigned long long>(fused_off[chunk]) : FUSED == 2 ? 0ull : co[chunk];
    if (byte0 < U) {
      *q = *reinterpret_cast<const uint4*>(ub + w0);
      *valid = (U - byte0 >= 16) ? 16 : static_cast<int>(U - byte0);
      if (byte0 + 16 < U) *behind = ub[w0 + 4];              // (inside the stream: the buffer is longer)
    }
  };
  uint4 q_next;
  uint32_t behind_next;
  int valid_next;
  unsigned long long off_next;
  fetch(blockIdx.x, &q_next, &behind_next, &valid_next, &off_next);
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint4 q = q_next;
    const uint32_t behind = behind_next;
    const int valid = valid_next;
    const unsigned long long chunk_off = FUSED == 2 ? static_cast<unsigned long long>(big_off) : off_next;
    fetch(chunk + gridDim.x, &q_next, &behind_next, &valid_next, &off_next);
    if (FUSED == 2 && chunk + gridDim.x < nchunks) {          // (uniform) the counts between this chunk and the next one
      uint32_t mine = 0, step;
      for (uint32_t i = chunk + threadIdx.x; i < chunk + gridDim.x; i += kThreads) mine += ff_f[i];
      (void)wg_exclusive_scan<kThreads>(mine, scratch, &step);
      big_off += step;
    }
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t ffs = 0;
    if (valid == 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) ffs += __popc(ff_bytes(w[j]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) ffs += count_ff(w[j], valid - 4 * j);
    }
    uint32_t total_ff;
    const uint32_t ex = wg_exclusive_scan<kThreads>(ffs, scratch, &total_ff);
    uint8_t* const dchunk = dst0 + static_cast<unsigned long long>(chunk) * kChunkBytes + chunk_off;
    const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dchunk) & 15u);
    const uint32_t o = mis + threadIdx.x * 16 + ex;          // where this thread's bytes start in `stage`
    // the first three bytes the thread behind will produce: its first source bytes, stuffed
    const uint32_t n0 = behind >> 24, n1 = (behind >> 16) & 0xffu, n2 = (behind >> 8) & 0xffu;
    const bool behind_plain = (ff_bytes(behind) & 0x80808000u) == 0u;
    if (valid == 16 && ffs == 0u && behind_plain) {
      // memory-order dwords of the 16 bytes and of the word behind them
      const uint32_t m[5] = {__builtin_bswap32(w[0]), __builtin_bswap32(w[1]), __builtin_bswap32(w[2]),
                             __builtin_bswap32(w[3]), __builtin_bswap32(behind)};
      const uint32_t up = (0u - o) & 3u;                     // bytes in front of the first dword that starts here
      uint32_t d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = __builtin_amdgcn_alignbyte(m[k + 1], m[k], up);
      uint32_t* const sp = reinterpret_cast<uint32_t*>(stage + o + up);
      sp[0] = d[0]; sp[1] = d[1]; sp[2] = d[2]; sp[3] = d[3];
      if (threadIdx.x == 0) {                                // nobody in front: the chunk's first bytes one by one
        for (uint32_t j = 0; j < up; ++j) stage[o + j] = static_cast<uint8_t>(m[0] >> (8 * j));
      }
    } else {
      uint8_t* sp = stage + o;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < valid) {
          const uint8_t b = static_cast<uint8_t>(w[j >> 2] >> (24 - 8 * (j & 3)));
          *sp++ = b;
          if (b == 0xff) *sp++ = 0x00;
        }
      }
      // ... and the head of the thread behind (the dword its first byte shares with this thread's last
      // bytes is this thread's to complete; what lies behind the stream's end is never copied out)
      sp[0] = static_cast<uint8_t>(n0);
      sp[1] = static_cast<uint8_t>(n0 == 0xffu ? 0u : n1);
      sp[2] = static_cast<uint8_t>(n0 == 0xffu ? n1 : (n1 == 0xffu ? 0u : n2));
    }
    __syncthreads();
    const unsigned long long rest = U - static_cast<unsigned long long>(chunk) * kChunkBytes;
    const uint32_t nbytes = static_cast<uint32_t>(rest < kChunkBytes ? rest : kChunkBytes) + total_ff;
    // bytes [mis, mis + nbytes) of `stage` go to dchunk - mis + [mis, ...): whole 16-byte units in the
    // middle, single bytes at the two ragged ends
    uint8_t* const dalign = dchunk - mis;
    const uint32_t lo = mis, hi = mis + nbytes;
    const uint32_t first_full = (lo + 15u) & ~15u, last_full = hi & ~15u;
    if (first_full <= last_full) {
      for (uint32_t i = first_full / 16 + threadIdx.x; i < last_full / 16; i += kThreads) {
        reinterpret_cast<uint4*>(dalign)[i] = reinterpret_cast<const uint4*>(stage)[i];
      }
      if (threadIdx.x < first_full - lo) dalign[lo + threadIdx.x] = stage[lo + threadIdx.x];
      if (threadIdx.x < hi - last_full) dalign[last_full + threadIdx.x] = stage[last_full + threadIdx.x];
    } else {
      if (threadIdx.x < nbytes) dalign[lo + threadIdx.x] = stage[lo + threadIdx.x];
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------
// K6 (restart mode only): the RSTn markers.  K1 left 16 zero bits behind every restart interval but
// the last; they went through the stitch as two 0x00 bytes.  One wave per marker finds where they
// ended up -- their position in the un-stuffed stream plus the 0xFF bytes in front of it (the chunk's
// offset from K4 plus a count over the part of the chunk in front of the marker) -- and writes FF D0+n.

__global__ __launch_bounds__(kThreads) void patch_restart_markers(const StitchArgs a) {
  const int frame = blockIdx.y;
  const int s = static_cast<int>(blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (s >= a.nseg - (a.rst_tail ? 0 : 1) || a.sizes[frame] == 0) return;
  const unsigned long long end_bits = a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + s + 1];
  const unsigned long long p = (end_bits >> 3) - 2;          // byte position of the placeholder, un-stuffed stream
  const unsigned long long chunk = p / kChunkBytes, cstart = chunk * kChunkBytes;
  const uint32_t* ub = a.ubuf + static_cast<size_t>(frame) * a.ubuf_words;
  uint32_t ffs = 0;
  for (unsigned long long b = cstart + 4ull * lane; b < p; b += 256) {
    const unsigned long long left = p - b;
    ffs += count_ff(ub[b >> 2], left >= 4 ? 4 : static_cast<int>(left));
  }
  for (int d = 32; d > 0; d >>= 1) ffs += __shfl_down(ffs, d, 64);
  if (lane == 0) {
    const uint32_t hsize = a.hdr_off ? a.hdr_off[frame + 1] - a.hdr_off[frame] : a.header_size;
    uint8_t* dst = frame_out(a, frame) + hsize + p +
                   a.chunk_off[static_cast<size_t>(frame) * a.max_chunks + chunk] + ffs;
    dst[0] = 0xff;
    dst[1] = static_cast<uint8_t>(0xd0 + ((s + a.seg_first) & 7));
  }
}
