// Pixel-load pattern microbenchmark for gfx950 (MI355X): what does K1's P1 access pattern -- every lane 24 bytes of
// a picture row as dwordx4 + dwordx2, two rows, consecutive lanes 24 bytes apart -- reach against fully coalesced
// 16-byte loads of the same bytes, as a pure read stream?  (round 6: is the histogram kind bound by the number of
// cache-line requests its loads make?)   hipcc --offload-arch=gfx950 -O3 tools/load_pattern.hip -o gpurun_out/load_pattern
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 3840, H = 2160, kRowBytes = W * 3;
constexpr int kSegMcus = 42, kSegBytesRow = kSegMcus * 48;        // 2016 bytes of a row per segment
// MODE 0: strip pattern (lane = one 8-pixel strip: x4 + x2 per row, 2 rows per row pair, 3 row pairs in flight)
// MODE 1: the same bytes, coalesced: a wave reads 1 KiB pieces of a row with one dwordx4 per lane
// MODE 2: strip pattern with three dwordx2 per row instead of x4 + x2
template <int MODE>
__global__ __launch_bounds__(256) void reader(const uint8_t* px, int nframes, int nseg, uint32_t* sink) {
  const int tid = threadIdx.x;
  uint32_t acc = 0;
  for (int s = blockIdx.x; s < nframes * nseg; s += gridDim.x) {
    const int frame = s / nseg, seg = s - frame * nseg;
    const int mcu0 = seg * 41;
    const int my = mcu0 / 240, mx = mcu0 - my * 240;              // (a segment that wraps reads past the row: same bytes)
    const uint8_t* base = px + (size_t)frame * kRowBytes * H + (size_t)my * 16 * kRowBytes + mx * 48;
    if (MODE == 0 || MODE == 2) {
      const int yp0 = tid / 84, strip = tid - yp0 * 84;
      if (yp0 < 3) {
        uint4 a[3][2]; uint2 b[3][2];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
          const int yp = yp0 + 3 * it;
          if (yp < 8) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const uint8_t* p = base + (size_t)(2 * yp + r) * kRowBytes + strip * 24;
              if (MODE == 0) { a[it][r] = *(const uint4*)p; b[it][r] = *(const uint2*)(p + 16); }
              else { const uint2 u = *(const uint2*)p, v = *(const uint2*)(p + 8); a[it][r] = make_uint4(u.x, u.y, v.x, v.y); b[it][r] = *(const uint2*)(p + 16); }
            }
          }
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
          if (yp0 + 3 * it < 8) {
#pragma unroll
            for (int r = 0; r < 2; ++r) acc += a[it][r].x ^ a[it][r].y ^ a[it][r].z ^ a[it][r].w ^ b[it][r].x ^ b[it][r].y;
          }
        }
      }
    } else {
      // 16 rows x 2016 bytes = 126 pieces of 16 bytes per row: thread t takes pieces t, t + 256, ... of the 2016 pieces
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int piece = tid + 256 * k;
        const int row = piece / 126, col = piece - row * 126;
        v[k] = make_uint4(0, 0, 0, 0);
        if (piece < 2016) v[k] = *(const uint4*)(base + (size_t)row * kRowBytes + col * 16);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, const uint8_t* d, int nframes, uint32_t* sink, int wgs) {
  const int nseg = 791;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(reader<MODE>, dim3(wgs), dim3(256), 0, 0, d, nframes, nseg, sink);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(reader<MODE>, dim3(wgs), dim3(256), 0, 0, d, nframes, nseg, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const double bytes = (double)nframes * nseg * 16 * 2016;
  printf("%-34s wgs %5d  %.3f ms  %.2f TB/s\n", name, wgs, ms, bytes / ms / 1e9);
}

int main() {
  const int nframes = 16;
  uint8_t* d; uint32_t* sink;
  CHECK(hipMalloc(&d, (size_t)nframes * kRowBytes * H + (1 << 20)));
  CHECK(hipMemset(d, 1, (size_t)nframes * kRowBytes * H + (1 << 20)));
  CHECK(hipMalloc(&sink, 64));
  for (int wgs : {512, 768, 1024, 2048, 12656}) {
    run<0>("strips x4 + x2 (K1's P1)", d, nframes, sink, wgs);
    run<2>("strips 3 x dwordx2", d, nframes, sink, wgs);
    run<1>("coalesced dwordx4", d, nframes, sink, wgs);
  }
  return 0;
}
