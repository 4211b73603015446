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
  const unsigned long long* seg_nbits64;   // band stitch: lengths as uint64 (else NULL)
  unsigned long long* total_bits_out;      // band encode: where the bit count of the band goes (else NULL)
  uint32_t subs;                           // K3: waves per segment (1 unless segments are whole bands)
  int seg_first;                           // restart mode, band of a frame: frame-level index of segment 0 ...
  int rst_tail;                            // ... and whether the band's last interval gets its marker too (it is not the frame's last)
};

__global__ __launch_bounds__(kThreads) void scan_seg_offsets(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  const int frame = blockIdx.x;
  const uint32_t* nb = a.seg_nbits + static_cast<size_t>(frame) * a.nseg;
  unsigned long long* off = a.seg_off + static_cast<size_t>(frame) * (a.nseg + 1);
  unsigned long long running = 0;
  for (int base = 0; base < a.nseg; base += kThreads) {
    const int i = base + threadIdx.x;
    // (a band is shorter than 2^32 bits: sjpeg_hip_stitch_bands checks its capacity)
    const uint32_t x = i >= a.nseg ? 0u
                     : a.seg_nbits64 != nullptr ? static_cast<uint32_t>(a.seg_nbits64[static_cast<size_t>(frame) * a.nseg + i])
                                                : nb[i];
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan<kThreads>(x, scratch, &total);
    if (i < a.nseg) off[i] = running + ex;
    running += total;
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
}

// ------------------------------------------------------------------------------------
// K3: place every segment in the continuous bit stream, and count 0xFF bytes per 4 KiB chunk

// 0x80 in every byte of w that equals 0xFF (exact, no carries between bytes)
__device__ __forceinline__ uint32_t ff_bytes(uint32_t w) {
  const uint32_t z = ~w;                                     // 0x00 where w has 0xFF
  return ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z | 0x7f7f7f7fu);
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
__global__ __launch_bounds__(kThreads) void place_segments(const StitchArgs a) {
  const int frame = blockIdx.y;
  // a wave takes words [sub * kSpec * 64, ...) of one segment; normal segments have one wave
  // (subs == 1, the loop below takes the rare longer rest), whole bands are cut into many
  const uint32_t unit = blockIdx.x * (kThreads / kPlaceLanes) + (threadIdx.x >> 6);
  const int sc0 = static_cast<int>(unit / a.subs);
  const uint32_t ibase = (unit % a.subs) * (kSpec * kPlaceLanes);
  if (sc0 >= a.nseg) return;
  const unsigned long long* off = a.seg_off + static_cast<size_t>(frame) * (a.nseg + 1);
  const uint32_t* segw = a.seg_words + static_cast<size_t>(frame) * a.nseg * a.slot_words;
  const uint32_t* src = segw + static_cast<size_t>(sc0) * a.slot_words;
  // Two forms of the same loads.  WIDE (every ordinary call): a lane takes 4 consecutive words per batch
  // of 256 -- one 16-byte load plus the word behind them -- so that a segment costs 6 load and 2-3
  // store instructions instead of 24 and 11: with one dword per lane the kernel was bound by the
  // number of memory instructions, not by bytes.  NARROW (bands cut into sub-ranges, slots that are
  // not a multiple of 16 bytes): one word per lane.
  const bool wide = a.subs == 1u && (a.slot_words & 3u) == 0u && a.slot_words >= 776u;      // uniform
  uint32_t spec[kSpec][2];
  uint4 wq[kWideSpec];
  uint32_t wx[kWideSpec];
  if (wide) {
#pragma unroll
    for (int k = 0; k < kWideSpec; ++k) {                    // words 0 .. 771: inside the slot whatever the length
      const uint32_t i = 256u * k + 4u * (threadIdx.x & 63);
      wq[k] = *reinterpret_cast<const uint4*>(src + i);
      wx[k] = src[i + 4];
    }
  } else {
#pragma unroll
    for (int k = 0; k < kSpec; ++k) {                        // inside the slot whatever the length
      const uint32_t i = min(ibase + k * kPlaceLanes + (threadIdx.x & 63), a.slot_words - 2u);
      spec[k][0] = src[i];
      spec[k][1] = src[i + 1];
    }
  }
  const unsigned long long b0 = off[sc0], b1 = off[sc0 + 1];
  const unsigned long long T = off[a.nseg];                 // total bits
  const unsigned long long U = (T + 7) >> 3;                // bytes incl. 1-bit padding
  // a frame whose stream is longer than the scratch sized from out_stride cannot fit its output slot
  // either; one that overran its pool has words missing: K4 reports size 0 for both, nothing to place
  if (((U + 3) >> 2) + 1 > a.ubuf_words || (a.pool_ctr != nullptr && a.pool_ctr[2 * frame + 1] != 0u)) return;
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
  auto one = [&](uint32_t i, uint32_t v0, uint32_t v1) {
    uint32_t ffs = 0;
    if (i < nwords) {
      const uint32_t r = lead + 32u * i;                                  // first source bit of this word
      uint32_t outw;
      if (r + 32u <= len) {
        outw = lead ? __builtin_amdgcn_alignbit(v0, v1, 32u - lead) : v0;   // (v0:v1) >> (32 - lead)
      } else {
        // the word that runs over the end of the segment: finish it from the next ones
        outw = 0;
        int need = 32, sc = sc0;
        unsigned long long p = (wbeg + i) * 32, c_beg = b0, c_end = b1;
        while (need > 0 && p < T) {
          while (p >= c_end) { ++sc; c_beg = c_end; c_end = off[sc + 1]; }
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
    auto batch = [&](uint32_t i0, uint4 q, uint32_t x) {     // destination words i0 .. i0 + 255, four per lane
      const uint32_t i = i0 + 4u * lane;
      const uint32_t v[5] = {q.x, q.y, q.z, q.w, x};
      if (i0 + 256u <= n_int) {                              // (uniform)
        Words4 o;
        uint32_t ffs = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o.w[u] = lead ? __builtin_amdgcn_alignbit(v[u], v[u + 1], 32u - lead) : v[u];
          ffs += static_cast<uint32_t>(__popc(ff_bytes(o.w[u])));
        }
        *reinterpret_cast<Words4*>(dst + i) = o;
        const uint32_t c_lo = (wbase + i) >> 10, c_hi = (wbase + i + 3u) >> 10;
        const uint32_t chunk0 = __builtin_amdgcn_readfirstlane(c_lo);
        if (chunk0 != ff_chunk) {                            // uniform
          if (ff_chunk != 0xffffffffu) ff_flush();
          ff_chunk = chunk0;
        }
        if (c_lo == chunk0 && c_hi == chunk0) {
          ff_acc += ffs;
        } else if (ffs != 0u) {                              // the lane's words straddle a chunk boundary, or lie behind it
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t f = static_cast<uint32_t>(__popc(ff_bytes(o.w[u])));
            const uint32_t c = (wbase + i + u) >> 10;
            if (f != 0u) { if (c == chunk0) ff_acc += f; else atomicAdd(&cff[c], f); }
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u, v[u], v[u + 1]);
      }
    };
#pragma unroll
    for (int k = 0; k < kWideSpec; ++k) {
      if (256u * k < nwords) batch(256u * k, wq[k], wx[k]);
    }
    for (uint32_t i0 = 256u * kWideSpec; i0 < nwords; i0 += 256u) {
      // A lane whose four words start inside the slot must get exactly those (a segment that is not
      // `long_seg` has up to slot_words - 2 of them); only the fifth word of the slot's last lane,
      // needed by destination word slot_words - 1 alone, and the lanes behind the slot, all of
      // whose words are unused, are clamped.
      const uint32_t i = min(i0 + 4u * lane, a.slot_words - 4u);
      batch(i0, *reinterpret_cast<const uint4*>(src + i), src[min(i + 4u, a.slot_words - 1u)]);
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
    const uint32_t iend = a.subs == 1u ? nwords : min(nwords, ibase + kSpec * kPlaceLanes);
    for (uint32_t i0 = ibase; i0 < iend; i0 += kPlaceLanes) {
      const uint32_t i = i0 + lane;
      one(i, seg_word(sc0, i), seg_word(sc0, i + 1));
    }
  }
  if (ff_chunk != 0xffffffffu) ff_flush();
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
  unsigned long long running = 0;
  for (uint32_t base = 0; base < nchunks; base += kThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t x = i < nchunks ? ff[i] : 0u;
    uint32_t total;
    const uint32_t ex = wg_exclusive_scan<kThreads>(x, scratch, &total);
    if (i < nchunks) co[i] = running + ex;
    running += total;
  }
  // a frame that does not fit the caller's slot reports size 0 and is not written
  const unsigned long long body = U + running;
  const uint32_t hoff = a.hdr_off ? a.hdr_off[frame] : 0u;
  const uint32_t hsize = a.hdr_off ? a.hdr_off[frame + 1] - hoff : a.header_size;
  const unsigned long long size = hsize + body + (a.append_eoi ? 2 : 0);
  const bool fits = size <= a.out_stride && ((U + 3) >> 2) + 1 <= a.ubuf_words &&
                    (a.pool_ctr == nullptr || a.pool_ctr[2 * frame + 1] == 0u);
  // last reader of the frame's pool counters: leave them at zero for the next call (scan_engine.hip)
  __syncthreads();
  if (a.pool_ctr != nullptr && threadIdx.x == 0) {
    const_cast<uint32_t*>(a.pool_ctr)[2 * frame] = 0u;
    const_cast<uint32_t*>(a.pool_ctr)[2 * frame + 1] = 0u;
  }
  uint8_t* dst = a.out + static_cast<size_t>(frame) * a.out_stride;
  if (threadIdx.x == 0) {
    if (fits && a.append_eoi) {
      dst[hsize + body] = 0xff;
      dst[hsize + body + 1] = 0xd9;
    }
    a.sizes[frame] = fits ? size : 0ull;
  }
  // header bytes in front of the entropy segment
  if (fits) {
    for (uint32_t i = threadIdx.x; i < hsize; i += kThreads) dst[i] = a.header[hoff + i];
  }
}

// ------------------------------------------------------------------------------------
// K5: byte stuffing into the caller's slot

__global__ __launch_bounds__(kThreads) void stuff_chunks(const StitchArgs a) {
  __shared__ uint32_t scratch[16];
  // stuffed bytes of one chunk (<= 2 * 4 KiB), placed so that LDS words line up with the
  // 4-byte words of the destination: the copy-out is aligned dword stores
  __shared__ __attribute__((aligned(16))) uint8_t stage[2 * kChunkBytes + 16];
  const int frame = blockIdx.y;
  const unsigned long long T = a.seg_off[static_cast<size_t>(frame) * (a.nseg + 1) + a.nseg];
  const unsigned long long U = (T + 7) >> 3;
  const uint32_t nchunks = static_cast<uint32_t>((U + kChunkBytes - 1) / kChunkBytes);
  const uint32_t* ub = a.ubuf + static_cast<size_t>(frame) * a.ubuf_words;
  const unsigned long long* co = a.chunk_off + static_cast<size_t>(frame) * a.max_chunks;
  const uint32_t hsize = a.hdr_off ? a.hdr_off[frame + 1] - a.hdr_off[frame] : a.header_size;
  uint8_t* const dst0 = a.out + static_cast<size_t>(frame) * a.out_stride + hsize;
  if (a.sizes[frame] == 0) return;                          // did not fit (see K4)
  // the 16 bytes of this thread in the NEXT chunk of the workgroup are requested while the current
  // ones are stuffed
  auto fetch = [&](uint32_t chunk, uint4* q, int* valid, unsigned long long* off) {
    *q = make_uint4(0, 0, 0, 0);
    *valid = 0;
    *off = 0;
    if (chunk >= nchunks) return;
    const unsigned long long w0 = static_cast<unsigned long long>(chunk) * kChunkWords + threadIdx.x * 4;
    const unsigned long long byte0 = w0 * 4;
    *off = co[chunk];
    if (byte0 < U) {
      *q = *reinterpret_cast<const uint4*>(ub + w0);
      *valid = (U - byte0 >= 16) ? 16 : static_cast<int>(U - byte0);
    }
  };
  uint4 q_next;
  int valid_next;
  unsigned long long off_next;
  fetch(blockIdx.x, &q_next, &valid_next, &off_next);
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint4 q = q_next;
    const int valid = valid_next;
    const unsigned long long chunk_off = off_next;
    fetch(chunk + gridDim.x, &q_next, &valid_next, &off_next);
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
    const uint32_t mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dchunk) & 3u);
    uint8_t* sp = stage + mis + threadIdx.x * 16 + ex;
    // (4-byte stores at the lanes' odd offsets and a permute-based expansion of the words that
    // hold 0xFF bytes were tried: bit-exact, but the unaligned LDS stores made the kernel 47 %
    // slower, 125 against 85 us; a padded buffer without bank conflicts: 102 us; a pre-cleared
    // buffer with branch-free placement by popcount: 87 us -- neither the conflicts nor the
    // per-byte control flow is what bounds this kernel)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < valid) {
        const uint8_t b = static_cast<uint8_t>(w[j >> 2] >> (24 - 8 * (j & 3)));
        *sp++ = b;
        if (b == 0xff) *sp++ = 0x00;
      }
    }
    __syncthreads();
    const unsigned long long rest = U - static_cast<unsigned long long>(chunk) * kChunkBytes;
    const uint32_t nbytes = static_cast<uint32_t>(rest < kChunkBytes ? rest : kChunkBytes) + total_ff;
    // bytes [mis, mis + nbytes) of `stage` go to dchunk - mis + [mis, ...): whole words in
    // the middle, single bytes at the two ragged ends
    uint8_t* const dalign = dchunk - mis;
    const uint32_t lo = mis, hi = mis + nbytes;
    const uint32_t first_full = (lo + 3u) & ~3u, last_full = hi & ~3u;
    if (first_full <= last_full) {
      for (uint32_t i = first_full / 4 + threadIdx.x; i < last_full / 4; i += kThreads) {
        reinterpret_cast<uint32_t*>(dalign)[i] = reinterpret_cast<const uint32_t*>(stage)[i];
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
    uint8_t* dst = a.out + static_cast<size_t>(frame) * a.out_stride + hsize + p +
                   a.chunk_off[static_cast<size_t>(frame) * a.max_chunks + chunk] + ffs;
    dst[0] = 0xff;
    dst[1] = static_cast<uint8_t>(0xd0 + ((s + a.seg_first) & 7));
  }
}
