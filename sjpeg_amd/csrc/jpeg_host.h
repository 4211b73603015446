// jpeg_host.h -- host-side (CPU, tiny) preparation shared by the C++ API and the C-ABI
// helpers: quantizer set-up, canonical Huffman codes, JFIF header bytes.  These stay on the
// host in the reference too (BASELINE.json north_star: src/headers.cc, SjpegEncodeParam
// plumbing) and must reproduce its bytes exactly; float code keeps the reference's types
// and operation order (build with -ffp-contract=off, never -ffast-math).
#ifndef SJPEG_AMD_JPEG_HOST_H_
#define SJPEG_AMD_JPEG_HOST_H_

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "sjpeg_hip.h"

namespace sjpeg_host {

extern const uint8_t kZigzag[64];           // zig-zag position -> natural index
extern const uint8_t kAnnexK1[2][64];       // default luma/chroma matrices, natural order

// A Huffman table in DHT form: BITS (codes per length 1..16) + HUFFVAL (symbols).
typedef sjpeg_hip_huffman_spec HuffSpec;
const HuffSpec& DefaultHuff(int type /*0 DC, 1 AC*/, int comp /*0 luma, 1 chroma*/);

// reference: GetQFactor / SetQuantMatrix, src/quantize.cc:77-96
float QualityToScale(float quality);
void ScaleMatrix(const uint8_t in[64], float scale_percent, uint8_t out[64]);

// reference: SetMinQuantMatrix, src/quantize.cc:98-104
void MinMatrix(const uint8_t in[64], int tolerance, uint8_t out[64]);

// reference: Encoder::FinalizeQuantMatrix, src/quantize.cc:123-148.  Clamps quant[] to
// min_quant[] in place and fills the scan-table columns for table `idx`.
void FinalizeQuantizer(uint8_t quant[64], const uint8_t min_quant[64], int q_bias, int idx,
                       sjpeg_hip_scan_tables* tables);

// reference: BuildHuffmanTable, src/entropy.cc:98-112.  tab[sym] = (code << 16) | len.
int BuildCodes(const HuffSpec& spec, uint32_t* tab);
// installs the codes of four specs (DC/AC x luma/chroma) into the scan tables
void InstallCodes(const HuffSpec* dc[2], const HuffSpec* ac[2], int ntables,
                  sjpeg_hip_scan_tables* tables);

struct FrameLayout {      // reference: Encoder::InitComponents, src/encoders.cc:32-88
  int nb_comps, mcu_blocks, block_w, block_h;
  int sampling[3], quant_idx[3];
};
bool LayoutFor(int yuv_mode, FrameLayout* L);

struct Metadata {
  std::string app_markers, exif, iccp, xmp;
  uint16_t xmp_split = 0;
};

// Appends SOI+APP0, metadata, DQT, SOF0, DHT, SOS in the reference's order
// (src/enc.cc:415-443, src/headers.cc).  Returns false on invalid metadata
// (src/headers.cc:77,95,122-125).
bool AppendHeaders(int W, int H, int yuv_mode, const uint8_t quant[2][64],
                   const HuffSpec* dc[2], const HuffSpec* ac[2], const Metadata* meta,
                   std::vector<uint8_t>* out);

}  // namespace sjpeg_host


// ---- adaptive quantization and optimised Huffman tables (host analysis; the statistics they
// consume are collected by the GPU: sjpeg_hip_scan_histogram / sjpeg_hip_scan_symbol_stats) ----
namespace sjpeg_host {

// reference: Encoder::AnalyseHisto, src/histogram.cc:126-315.  hist[idx][pos][bin] counts the
// coefficients of table idx (0 luma, 1 chroma) at NATURAL position pos with |c| >> 2 == bin
// (bin < 128).  Rewrites quant[idx][pos] (within min_quant and the delta limits).
enum { kAdaptDeltas = 25 };        // candidate steps per position: -12 .. +12 (sjpegi.h:269-273)
void AdaptSums(const uint32_t hist[64][128], const uint8_t quant[64], const uint8_t min_quant[64],
               int64_t sums[64][kAdaptDeltas][2], int32_t totlast[64][2]);
void AdaptDecide(const int64_t sums[2][64][kAdaptDeltas][2], const int32_t totlast[2][64][2], int nb_comps,
                 uint8_t quant[2][64], int qdelta_max_luma, int qdelta_max_chroma);
void AdaptQuantMatrices(const uint32_t hist[2][64][128], int nb_comps, uint8_t quant[2][64],
                        const uint8_t min_quant[2][64], int qdelta_max_luma, int qdelta_max_chroma);

// reference: BuildOptimalTable, src/entropy.cc:254-430.  freq[size] -> DHT spec.
void BuildOptimalSpec(const uint32_t* freq, int size, HuffSpec* out);

}  // namespace sjpeg_host

#endif  // SJPEG_AMD_JPEG_HOST_H_
