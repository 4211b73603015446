// Small kernels behind the statistics passes: sums of the per-workgroup partials, the quantization-error
// total, the bin loops of AnalyseHisto.
// Part of the single translation unit scan_engine.hip: included there inside its anonymous
// namespace, after <hip/hip_runtime.h> and sjpeg_hip.h; not a stand-alone header.
// ------------------------------------------------------------------------------------
// Sums the per-workgroup partial statistics of one frame: out[frame][i] = sum over segments.
// BYTES: partial words hold four 8-bit counters (histogram) -> four u32 outputs per word.
template <bool BYTES>
__global__ __launch_bounds__(kThreads) void reduce_partials(const uint32_t* part, int nseg, int words,
                                                           uint32_t* out) {
  // blockIdx.z = slice of the segments: a thread adds up its slice (independent loads, unrolled)
  // and the slices meet in the output with atomics (cleared by the caller).  One thread walking
  // all ~800 partials of a 4K frame was a chain of loads: 0.15 ms of a 0.23 ms histogram pass.
  const int frame = blockIdx.y;
  const int w = blockIdx.x * kThreads + threadIdx.x;
  if (w >= words) return;
  const int per = (nseg + gridDim.z - 1) / gridDim.z;
  const int s0 = blockIdx.z * per, s1 = min(nseg, s0 + per);
  const uint32_t* src = part + static_cast<size_t>(frame) * nseg * words + w;
  if (BYTES) {
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll 8
    for (int s = s0; s < s1; ++s) {
      const uint32_t v = src[static_cast<size_t>(s) * words];
      c0 += v & 0xffu; c1 += (v >> 8) & 0xffu; c2 += (v >> 16) & 0xffu; c3 += v >> 24;
    }
    uint32_t* dst = out + (static_cast<size_t>(frame) * words + w) * 4;
    if (c0) atomicAdd(&dst[0], c0);
    if (c1) atomicAdd(&dst[1], c1);
    if (c2) atomicAdd(&dst[2], c2);
    if (c3) atomicAdd(&dst[3], c3);
  } else {
    uint32_t sum = 0;
#pragma unroll 8
    for (int s = s0; s < s1; ++s) sum += src[static_cast<size_t>(s) * words];
    if (sum) atomicAdd(&out[static_cast<size_t>(frame) * words + w], sum);
  }
}

__global__ __launch_bounds__(kThreads) void reduce_error(const unsigned long long* part, int nseg,
                                                        unsigned long long* out) {
  __shared__ unsigned long long red[kThreads / 64];
  const int frame = blockIdx.x;
  unsigned long long sum = 0;
  for (int s = threadIdx.x; s < nseg; s += kThreads) sum += part[static_cast<size_t>(frame) * nseg + s];
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) out[frame] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------
// The bin loops of AnalyseHisto (reference src/histogram.cc:150-205) on the device-resident
// histogram: one small workgroup per (frame, table, position), one thread per candidate step.
// Integer sums only (see jpeg_host.cc AdaptSums: the reference's double accumulators hold exactly
// these integers); the regression and the choice of the step stay on the host.
struct AdaptArgs {
  const uint32_t* hist;             // [nframes][2][64][128]
  long long* sums;                  // [nframes][2][64][25][2]: bits, distortion (INT64_MIN = not a candidate)
  int* totlast;                     // [nframes][2][64][2]: population, highest occupied bin + 1
  uint8_t quant[2][64], min_quant[2][64];
};
__global__ __launch_bounds__(64) void adapt_sums_kernel(const AdaptArgs a) {
  // one wave per (frame, table, position): the 128 bins go to LDS once (two per lane; population and
  // highest occupied bin by a wave reduction), then lane d < 25 walks them for its candidate step
  __shared__ uint32_t bins[128];
  const int pos = blockIdx.x, idx = blockIdx.y, frame = blockIdx.z, delta = threadIdx.x;
  const uint32_t* const h = a.hist + ((static_cast<size_t>(frame) * 2 + idx) * 64 + pos) * 128;
  const uint2 mine = reinterpret_cast<const uint2*>(h)[delta];          // bins 2 * lane, 2 * lane + 1
  bins[2 * delta] = mine.x;
  bins[2 * delta + 1] = mine.y;
  int total = static_cast<int>(mine.x + mine.y);
  int last = mine.y ? 2 * delta + 2 : (mine.x ? 2 * delta + 1 : 0);
  for (int d = 32; d > 0; d >>= 1) {
    total += __shfl_xor(total, d, 64);
    last = max(last, __shfl_xor(last, d, 64));
  }
  __syncthreads();
  const size_t cell = (static_cast<size_t>(frame) * 2 + idx) * 64 + pos;
  if (delta == 0) { a.totlast[cell * 2] = total; a.totlast[cell * 2 + 1] = last; }
  if (delta >= 25) return;
  const int dq = static_cast<int>(a.quant[idx][pos]) + (delta - 12);
  long long bsum = 0, dsum = 0;
  if (dq < static_cast<int>(a.min_quant[idx][pos]) || dq > 255) {
    dsum = static_cast<long long>(0x8000000000000000ull);
  } else {
    const uint32_t idq = static_cast<uint32_t>(((1 << 16) + dq - 1) / dq);
    for (int i = 0; i < last; ++i) {
      const uint32_t hi = bins[i];
      const uint32_t v = (static_cast<uint32_t>(i) << 2) + 2;
      const uint32_t qv = (v * idq + (1u << 16 >> 1)) >> 16;
      const uint32_t bits = 32u - __clz(qv);                        // 0 for qv == 0
      const uint32_t d = v - qv * static_cast<uint32_t>(dq);
      bsum += static_cast<int>(hi * bits);
      dsum += static_cast<int>(hi * (d * d));
    }
  }
  a.sums[(cell * 25 + delta) * 2] = bsum;
  a.sums[(cell * 25 + delta) * 2 + 1] = dsum;
}
