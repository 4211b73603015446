// Small kernels behind the statistics passes: sums of the per-workgroup partials, the quantization-error
// total, the bin loops of AnalyseHisto.
// Part of the single translation unit scan_engine.hip: included there inside its anonymous
// namespace, after <hip/hip_runtime.h> and sjpeg_hip.h; not a stand-alone header.
// ------------------------------------------------------------------------------------
// Sums the per-workgroup partial symbol counts of one frame (statistics kinds): out[frame][i] = sum over segments.
__global__ __launch_bounds__(kThreads) void reduce_partials(const uint32_t* part, int nseg, int words, uint32_t* out) {
  // blockIdx.z = slice of the segments: a thread adds up its slice (independent loads, unrolled)
  // and the slices meet in the output with atomics (cleared by the caller).
  const int frame = blockIdx.y;
  const int w = blockIdx.x * kThreads + threadIdx.x;
  if (w >= words) return;
  const int per = (nseg + gridDim.z - 1) / gridDim.z;
  const int s0 = blockIdx.z * per, s1 = min(nseg, s0 + per);
  const uint32_t* src = part + static_cast<size_t>(frame) * nseg * words + w;
  uint32_t sum = 0;
#pragma unroll 8
  for (int s = s0; s < s1; ++s) sum += src[static_cast<size_t>(s) * words];
  if (sum) atomicAdd(&out[static_cast<size_t>(frame) * words + w], sum);
}

// Sums the partials of the histogram kind (scan_segments.h: one per persistent workgroup, `groups` per frame; 16-bit
// counters, [8][256] pieces of 16 bytes -- piece j of thread t holds the 8-bit counters of words 16 t + 2 j and
// 16 t + 2 j + 1 of the kernel's [2][16][128] words as (byte 0 | byte 2 << 16), (byte 1 | byte 3 << 16) twice; word
// (table, q, bin) counts that bin of the positions 2q, 2q + 1, 32 + 2q, 33 + 2q in its four bytes):
// out[frame][table][position][bin].  A workgroup takes 64 threads' worth of one piece, its four waves a quarter of the
// groups each (independent loads, sixteen in flight) and meet in LDS: no atomics, and `out` needs no clearing -- unless
// the launch has slices (blockIdx.z: one big frame with a thousand partials), which meet with atomics in a cleared `out`.
__global__ __launch_bounds__(kThreads) void reduce_partials16(const uint4* part, int groups, uint32_t* out) {
  __shared__ uint32_t red[3][8][64];
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x >> 2, t = (blockIdx.x & 3) * 64 + lane;   // piece, thread of the scan kernel
  const int per_slice = (groups + gridDim.z - 1) / gridDim.z;
  const int s0 = blockIdx.z * per_slice, s1 = min(groups, s0 + per_slice);
  const int per = (s1 - s0 + 3) >> 2;
  const int g0 = s0 + wave * per, g1 = min(s1, g0 + per);
  const uint4* src = part + (static_cast<size_t>(frame) * groups * 8 + j) * 256 + t;
  uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 16
  for (int g = g0; g < g1; ++g) {
    const uint4 v = src[static_cast<size_t>(g) * (8 * 256)];
    c[0] += v.x & 0xffffu; c[1] += v.y & 0xffffu; c[2] += v.x >> 16; c[3] += v.y >> 16;
    c[4] += v.z & 0xffffu; c[5] += v.w & 0xffffu; c[6] += v.z >> 16; c[7] += v.w >> 16;
  }
  if (wave != 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave - 1][i][lane] = c[i];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] += red[0][i][lane] + red[1][i][lane] + red[2][i][lane];
  // words 16 t + 2 j, + 1: table and q from t >> 3, bins 16 (t & 7) + 2 j, + 1
  const int tq = t >> 3, bin = 16 * (t & 7) + 2 * j;
  uint32_t* dst = out + static_cast<size_t>(frame) * (2 * 64 * 128) + ((tq >> 4) * 64 + 2 * (tq & 15)) * 128 + bin;
#pragma unroll
  for (int b = 0; b < 4; ++b) {                      // byte b: position 2q + (b & 1) + 32 (b >> 1)
    uint32_t* const d2 = dst + ((b & 1) + 32 * (b >> 1)) * 128;
    if (gridDim.z == 1) {
      *reinterpret_cast<uint2*>(d2) = make_uint2(c[b], c[4 + b]);
    } else {
      if (c[b]) atomicAdd(&d2[0], c[b]);
      if (c[4 + b]) atomicAdd(&d2[1], c[4 + b]);
    }
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

// ------------------------------------------------------------------------------------
// The float / double half of AnalyseHisto (reference src/histogram.cc:169-312; this library's host form: AdaptDecide,
// jpeg_host.cc) on the device, behind adapt_sums_kernel: the host's regression -- 7 us a frame, read out of 52 KB of
// pinned memory a frame -- sat between a part's histogram pass and its statistics launch, which the device waits for.
// One wave per (table, frame), a lane per coefficient position.  Every accumulated term and the score are the host
// form's expressions in the host form's ORDER (normative for the rounding; no contraction: -ffp-contract=off): a lane fits
// its own position's two clouds over the 25 candidate steps; the two slope sums over the live positions are added up by
// ONE lane in ascending position order, as the host's loop does; then every lane picks its step.  IEEE double add / mul /
// div, int64 -> double and double -> float conversions round to nearest even on this device as on the host.
struct DecideArgs {
  const long long* sums;            // [nframes][2][64][25][2] (adapt_sums_kernel)
  const int* totlast;               // [nframes][2][64][2]
  uint8_t* quant_out;               // [nframes][2][64]: the adapted matrices (tables the launch does not run keep quant_in)
  uint8_t quant_in[2][64];
  int last_step[2];                 // qdelta_max of the table - kQDeltaMin
};
__global__ __launch_bounds__(64) void adapt_decide_kernel(const DecideArgs a) {
  __shared__ double cov_d[64], cov_r[64];
  __shared__ double lam;
  const int idx = blockIdx.x, frame = blockIdx.y, pos = threadIdx.x;
  constexpr float kWeight[25] = {0, 0, 0, 0, 0, 1, 5, 16, 43, 94, 164, 228, 255, 228, 164, 94, 43, 16, 5, 1, 0, 0, 0, 0, 0};
  const size_t cell = (static_cast<size_t>(frame) * 2 + idx) * 64 + pos;
  bool live = !((0x103ull >> pos) & 1ull);                        // DC and its two neighbours are never touched
  if (live && a.totlast[cell * 2] < 0.5 * a.totlast[cell * 2 + 1]) live = false;   // sparse histogram
  float dist[25], rate[25];
  double cd = 0., cr = 0.;
  if (live) {
    double w = 0., x = 0., xx = 0., d = 0., dd = 0., xd = 0., r = 0., xr = 0.;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
      const long long s0 = a.sums[(cell * 25 + k) * 2], s1 = a.sums[(cell * 25 + k) * 2 + 1];
      if (s1 == static_cast<long long>(0x8000000000000000ull)) {   // quantizer out of range
        dist[k] = 3.402823466e+38f;
        rate[k] = 0.f;
        continue;
      }
      const double ra = static_cast<double>(s0), di = static_cast<double>(s1);
      dist[k] = static_cast<float>(di);
      rate[k] = static_cast<float>(ra);
      if (kWeight[k] > 0.f) {
        const double weight = kWeight[k], step = static_cast<double>(k - 12);
        w += weight;
        x += weight * step;
        xx += weight * step * step;
        d += weight * di;
        dd += weight * di * di;
        r += weight * ra;
        xd += weight * di * step;
        xr += weight * ra * step;
      }
    }
    cd = w * xd - x * d;
    cr = w * xr - x * r;
    if (cd * cd < 0.5 * (w * xx - x * x) * (w * dd - d * d)) live = false;     // not (nearly) linear in the step
  }
  // (a position that is not live adds +0.0: x + 0.0 == x, so all 64 are added, in order, without a test -- the reads pipeline)
  cov_d[pos] = live ? cd : 0.;
  cov_r[pos] = live ? cr : 0.;
  __syncthreads();
  if (pos == 0) {
    double sd = 0., sr = 0.;
#pragma unroll
    for (int p = 0; p < 64; ++p) { sd += cov_d[p]; sr += cov_r[p]; }
    double lambda = 128.;
    if (sd > 1000. && sr < -10.) {
      lambda = -sd / sr;
      if (lambda < 1.) lambda = 1.;
    }
    lam = lambda;
  }
  __syncthreads();
  int q = a.quant_in[idx][pos];
  if (live) {
    const double lambda = lam;
    float best = 3.402823466e+38f;
    int step = 0;
    const int last = a.last_step[idx];
#pragma unroll
    for (int k = 0; k < 25; ++k) {
      if (k > last) break;
      if (!(dist[k] < 3.402823466e+38f)) continue;
      const float score = dist[k] + lambda * rate[k];
      if (score < best) { best = score; step = k - 12; }
    }
    q += step;
  }
  a.quant_out[cell] = static_cast<uint8_t>(q);
}
