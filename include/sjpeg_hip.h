/* sjpeg_hip.h -- C-ABI of the MI355X (gfx950) scan engine.
 *
 * This is the one device boundary of the library: the host encoder (include/sjpeg.h,
 * sjpeg_amd/csrc/host_*.cc) prepares quantizers, Huffman codes and JFIF headers exactly
 * as the reference does, then hands whole frames (or a batch of frames) to
 * sjpeg_hip_encode_scan(), which replaces the reference's per-MCU hot loop
 *
 *     Encoder::SinglePassScan()            /root/reference/src/enc.cc:276-307
 *       -> GetSamples()                    src/encoders.cc:170-182,206-215,239-248
 *       -> fDCT_()                         src/fdct.cc:596-609
 *       -> quantize_block_()               src/quantize.cc:288-320
 *       -> GenerateDCDiffCode()/CodeBlock  src/entropy.cc:133-198
 *       -> BitWriter::PutBits/FlushBits    src/bit_writer.h:172-209, bit_writer.cc:107-116
 *
 * with hand-written HIP kernels.  Plain pointers and sizes only: no C++ types, no torch
 * types.  A binding for any host language (cgo, JNI, ctypes, N-API) binds exactly these
 * symbols; see INTEGRATION.md.
 *
 * All functions return 0 on success or a negative SJPEG_HIP_E* code; the message of the
 * last failure on the calling thread is available from sjpeg_hip_last_error().
 * Nothing here ever falls back to a CPU implementation: without a usable gfx950 device
 * the calls fail.
 */
#ifndef SJPEG_HIP_H_
#define SJPEG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SJPEG_HIP_ABI_VERSION 18

enum {
  SJPEG_HIP_OK = 0,
  SJPEG_HIP_EINVAL = -1,      /* bad argument (null pointer, size, mode, ...)        */
  SJPEG_HIP_ENODEV = -2,      /* no usable HIP device / runtime                      */
  SJPEG_HIP_ENOMEM = -3,      /* device or host allocation failed                    */
  SJPEG_HIP_ERUNTIME = -4,    /* a HIP call or kernel failed                         */
  SJPEG_HIP_ECAPACITY = -5    /* caller's output slots are smaller than required     */
};

/* Values of SjpegYUVMode (include/sjpeg.h) that the scan engine codes directly. */
enum { SJPEG_HIP_YUV420 = 1, SJPEG_HIP_YUV444 = 3, SJPEG_HIP_YUV400 = 4 };

/* Everything the scan needs besides pixels, in the reference's own units.
 * Index 0 = luma tables, 1 = chroma tables (reference quant_idx_, src/encoders.cc:37-39).
 * iquant/bias are the reference Quantizer fields (src/sjpegi.h:228-235) in NATURAL order
 * as produced by Encoder::FinalizeQuantMatrix (src/quantize.cc:123-148); qthresh is not
 * needed: by construction (src/quantize.cc:144-145) a >= qthresh <=> QUANTIZE(a) > 0.
 * dc_codes/ac_codes are the packed (code << 16) | length words of
 * BuildHuffmanTable (src/entropy.cc:98-112). */
typedef struct sjpeg_hip_scan_tables {
  uint16_t iquant[2][64];
  uint16_t bias[2][64];
  uint32_t dc_codes[2][12];
  uint32_t ac_codes[2][256];
  uint8_t quant[2][64];        /* the (final, clamped) quantizer steps; read by the quantization-
                                  error pass (src/quantize.cc:553-565) and the trellis */
  uint8_t trellis_len[2][256]; /* SJPEG_HIP_QUANT_TRELLIS: AC code LENGTHS the rate term is priced
                                  with (what Quantizer::codes_ points at, src/quantize.cc:151,396):
                                  the standard tables in the reference's single-pass flow */
  uint32_t flags;              /* SJPEG_HIP_QUANT_* */
} sjpeg_hip_scan_tables;

/* flags: quantize with the trellis search of the reference's methods 7 / 8
 * (Encoder::TrellisQuantizeBlock, src/quantize.cc:325-457) instead of plain rounding.  Applies to
 * the encode and symbol-statistics passes. */
#define SJPEG_HIP_QUANT_TRELLIS 1u
/* Two-pass flows quantize every block twice (statistics pass, then encode pass with the tables
 * compiled from it).  KEEP: the statistics pass leaves its quantized blocks in the engine (144 B per
 * block); REPLAY: the encode pass entropy-codes those instead of converting, transforming and
 * quantizing the pixels again -- what the reference's stored run/levels do (reuse_run_levels_,
 * src/enc.cc:121-129).  REPLAY needs a KEEP statistics pass of the same geometry right before it on
 * the same engine, and ignores iquant / bias / quant / trellis_len / the pixel source.  Worth it
 * where quantization is expensive: the host API uses it for the trellis methods. */
#define SJPEG_HIP_QUANT_KEEP 2u
#define SJPEG_HIP_QUANT_REPLAY 4u
/* OPTIONAL restart-marker mode (encode calls).  NOT the reference's bytes -- it never writes DRI / RSTn
 * (src/sjpegi.h:68-74) -- but the same picture: same tables, same coefficients, identical pixels
 * after decoding.  Every segment of the engine (sjpeg_hip_restart_interval() MCUs) becomes a restart
 * interval: its bits padded to a byte with 1-bits, RSTn (FF D0+(n & 7)) behind it, DC predictors
 * reset (ITU-T T.81 B.2.4.4 / F.1.2.3).  The header passed to the call must carry the DRI segment:
 * sjpeg_hip_header_add_restart().  What it is for: intervals are independently decodable and byte
 * aligned, so a frame's intervals can be produced on different devices and concatenated by the host
 * with no bit-level stitch. */
#define SJPEG_HIP_RESTART_MARKERS 8u

/* Pixel sources.  Packed colour and gray use plane[0]; planar YUV uses Y, U, V; NV12/NV21 use
 * Y and the interleaved chroma plane in plane[1].  Chroma planes of the 4:2:0 layouts are
 * (width+1)/2 x (height+1)/2 samples.  These replace the reference's input adapters
 * Encoder420/444/400 (src/encoders.cc:157-253, packed RGB/BGRA/RGBA), Encoder400G (:256-276),
 * EncoderNV12 (:281-344), EncoderYUV444 (:384-419) and EncoderYUV420 (:442-490). */
enum {
  SJPEG_HIP_SRC_RGB = 0,       /* 3 bytes/pixel R,G,B     -> any of 420 / 444 / 400 */
  SJPEG_HIP_SRC_BGRA = 1,      /* 4 bytes/pixel B,G,R,A   -> any of 420 / 444 / 400 */
  SJPEG_HIP_SRC_RGBA = 2,      /* 4 bytes/pixel R,G,B,A   -> any of 420 / 444 / 400 */
  SJPEG_HIP_SRC_GRAY = 3,      /* 1 plane                 -> 400 */
  SJPEG_HIP_SRC_YUV444 = 4,    /* 3 full-size planes      -> 444 */
  SJPEG_HIP_SRC_YUV420 = 5,    /* Y + subsampled U, V     -> 420 */
  SJPEG_HIP_SRC_NV12 = 6,      /* Y + interleaved U,V     -> 420 */
  SJPEG_HIP_SRC_NV21 = 7       /* Y + interleaved V,U     -> 420 */
};
typedef struct sjpeg_hip_source {
  int32_t format;              /* SJPEG_HIP_SRC_* */
  int32_t reserved;            /* 0 */
  const void* plane[3];        /* DEVICE pointers */
  int64_t row_stride[3];       /* bytes between rows of each plane; may be negative */
  int64_t frame_stride[3];     /* bytes between consecutive frames of a batch, per plane */
} sjpeg_hip_source;

/* A Huffman table in DHT form: BITS (number of codes of each length 1..16) and HUFFVAL
 * (symbols by increasing code length), reference struct HuffmanTable (src/sjpegi.h:221-225). */
typedef struct sjpeg_hip_huffman_spec {
  uint8_t bits[16];
  uint8_t syms[256];
  int32_t nsyms;
} sjpeg_hip_huffman_spec;

/* Opaque engine: one HIP device, cached device scratch.  Not thread-safe; use one engine
 * per host thread (they may share a device).  Every call is asynchronous on the stream it is
 * given; the scratch is shared by all calls, so a call on another stream than the previous one
 * first waits (on the device) for that one's work -- use one engine per stream to overlap.
 * A stream handed to a call must stay alive until the engine's NEXT call has been issued (that call
 * orders itself behind it with an event); if it was destroyed earlier the engine falls back to a
 * device-wide synchronisation instead of failing.
 * The scratch of an encode is sized from the bytes the caller gives every frame (out_stride), per frame of the
 * batch; a batch that would take more than SJPEG_HIP_SCRATCH_LIMIT_BYTES of it (environment, read when the engine
 * is created; default 16 GiB) is coded as several launches of as many frames as fit, in order, on the same
 * stream -- the caller sees one call. */
typedef struct sjpeg_hip_engine sjpeg_hip_engine;

int sjpeg_hip_abi_version(void);
int sjpeg_hip_device_count(void);                 /* 0 when no device/runtime */
const char* sjpeg_hip_last_error(void);

int sjpeg_hip_engine_create(int device, sjpeg_hip_engine** engine);
void sjpeg_hip_engine_destroy(sjpeg_hip_engine* engine);

/* Worst-case size in bytes of ONE coded frame (header + stuffed entropy data + EOI):
 * the minimum legal out_stride for sjpeg_hip_encode_scan().  Follows the reference's
 * own bound of 2560 bytes per MCU (src/enc.cc:206-209).  0 on invalid arguments. */
size_t sjpeg_hip_frame_bound(int width, int height, int yuv_mode, size_t header_size);

/* Codes `nframes` frames that are RESIDENT IN DEVICE MEMORY, all with the same geometry
 * and tables, each into its own complete JPEG byte stream:
 *
 *   d_out + f*out_stride : [ header bytes | entropy-coded segment | FF D9 ]
 *   d_sizes[f]           : number of bytes written for frame f
 *
 *   d_rgb         packed 8-bit R,G,B; pixel (x,y) of frame f at
 *                 d_rgb + f*frame_stride + y*row_stride + 3*x.  row_stride may be negative
 *                 (bottom-up images, as the reference allows: src/api.cc:35-36).
 *   header        HOST pointer to the bytes that precede the entropy segment (SOI..SOS,
 *                 written by the host exactly like src/headers.cc); may be NULL/0, in which
 *                 case each stream starts directly with entropy data.
 *   append_eoi    non-zero: terminate each stream with FF D9 (src/headers.cc:262-268).
 *   d_out         device buffer, nframes*out_stride bytes.  out_stride >= frame_bound always
 *                 suffices; a smaller slot is legal (the engine's scratch is sized from it, about
 *                 3.5 x out_stride per frame instead of the worst case of the geometry): a frame
 *                 whose stream does not fit its slot reports d_sizes[f] = 0 and is not written,
 *                 the other frames of the batch are unaffected.
 *   d_sizes       device array of nframes uint64.
 *   stream        hipStream_t (as void*) on which everything is enqueued; NULL = default
 *                 stream.  The call is asynchronous: results are valid once the stream
 *                 has drained.
 *
 * The output is bit-identical to what the reference writes between (and including) the
 * same header and EOI for the same pixels and tables. */
int sjpeg_hip_encode_scan(sjpeg_hip_engine* engine,
                          const void* d_rgb, int64_t row_stride, int64_t frame_stride,
                          int width, int height, int yuv_mode, int nframes,
                          const sjpeg_hip_scan_tables* tables,
                          const void* header, size_t header_size, int append_eoi,
                          void* d_out, size_t out_stride, uint64_t* d_sizes,
                          void* stream);

/* Stage taps for tests and profiling (same arguments as above where named alike).
 * d_coeffs receives the quantized coefficients of every block in stream order,
 * 64 int16 per block in zig-zag order, element 0 = quantized DC *value*
 * (blocks per frame = mcu count * {6,3,1}).  Pure function of pixels + tables. */
int sjpeg_hip_scan_coeffs(sjpeg_hip_engine* engine,
                          const void* d_rgb, int64_t row_stride, int64_t frame_stride,
                          int width, int height, int yuv_mode, int nframes,
                          const sjpeg_hip_scan_tables* tables,
                          int16_t* d_coeffs, void* stream);

/* Statistics passes for the reference's methods 1..6 (same pixel arguments as above).
 *
 * sjpeg_hip_scan_histogram: replaces Encoder::CollectHistograms (src/histogram.cc:317-339 with
 * StoreHisto :56-108).  d_hist receives, per frame, uint32 [2][64][128]: for quantizer table
 * t (0 luma, 1 chroma) and NATURAL coefficient position p, the number of blocks whose
 * un-quantized coefficient c has |c| >> 2 == bin (bin < 128).
 *
 * sjpeg_hip_scan_symbol_stats: replaces the statistics half of SinglePassScanOptimized
 * (src/enc.cc:323-372, AddEntropyStats src/entropy.cc:208-227).  Needs tables->iquant/bias.
 * d_freq receives, per frame, uint32 [2][272]: [t][0..255] AC symbol counts (run << 4 | size,
 * 0xF0 = ZRL, 0x00 = EOB), [t][256 + n] DC size-category counts. */
int sjpeg_hip_scan_histogram(sjpeg_hip_engine* engine,
                             const void* d_rgb, int64_t row_stride, int64_t frame_stride,
                             int width, int height, int yuv_mode, int nframes,
                             uint32_t* d_hist, void* stream);
int sjpeg_hip_scan_symbol_stats(sjpeg_hip_engine* engine,
                                const void* d_rgb, int64_t row_stride, int64_t frame_stride,
                                int width, int height, int yuv_mode, int nframes,
                                const sjpeg_hip_scan_tables* tables,
                                uint32_t* d_freq, void* stream);

/* Replaces Encoder::ComputePSNR's inner sum (src/dichotomy.cc:302-323 with QuantizeError,
 * src/quantize.cc:553-565): d_err receives, per frame, the uint64 sum over all blocks and
 * coefficients of ((|c| >> 4) - quant * level)^2 (each block's sum wrapped to 32 bits like the
 * reference).  Needs tables->iquant / bias / quant. */
int sjpeg_hip_scan_quant_error_src(sjpeg_hip_engine* engine, const struct sjpeg_hip_source* src,
                                   int width, int height, int yuv_mode, int nframes,
                                   const sjpeg_hip_scan_tables* tables, uint64_t* d_err,
                                   void* stream);

/* Number of entropy-coded bits (before byte stuffing and padding) of each frame of the most
 * recent sjpeg_hip_encode_scan*() call on this engine, copied to HOST memory `bits[nframes]`.
 * Synchronises the device.  With the coded size this gives what the reference's BitCounter
 * reports (src/bit_writer.h:292-365).  (Not defined behind sjpeg_hip_encode_batch_src, whose
 * jobs may run on the engine's child engines.) */
int sjpeg_hip_engine_entropy_bits(sjpeg_hip_engine* engine, uint64_t* bits, int nframes);

/* ---- SJPEG_YUV_SHARP: the iterative sharp RGB -> YUV 4:2:0 conversion -----------------------------
 * Replaces sjpeg::ApplySharpYUVConversion (src/yuv_convert.cc:674-697; the pre-pass of
 * EncoderSharp420, src/encoders.cc:512-541).  `src` is packed RGB / BGRA / RGBA in device memory;
 * the result is three tightly packed 8-bit planes per frame: Y width x height, U and V
 * ((width+1)/2) x ((height+1)/2), frames y_frame_stride / uv_frame_stride bytes apart -- exactly
 * what a SJPEG_HIP_SRC_YUV420 source of sjpeg_hip_encode_scan_src() then takes.  The row pairs of
 * a sweep are sequential by construction (a row pair reads the row above as this sweep left it); the
 * columns of a row pair, the up to four sweeps of a picture (a pipeline a few row pairs apart) and
 * the pictures of a batch run in parallel.  Any width.  d_workspace: sjpeg_hip_sharp_workspace()
 * bytes of device memory.  (Environment, read once, for A/B runs only: SJPEG_HIP_SHARP_STRIPS=0 takes
 * the kernel with one workgroup per picture and sweep, SJPEG_HIP_SHARP_INPLACE=1 the in-place sweeps;
 * same planes.) */
size_t sjpeg_hip_sharp_workspace(int width, int height, int nframes);
int sjpeg_hip_sharp_yuv(const sjpeg_hip_source* src, int width, int height, int nframes,
                        uint8_t* d_y, uint8_t* d_u, uint8_t* d_v, int64_t y_frame_stride,
                        int64_t uv_frame_stride, void* d_workspace, size_t workspace_size,
                        void* stream);

/* ---- SJPEG_YUV_AUTO / SjpegRiskiness (src/jpeg_tools.cc:170-236) -------------------------------
 * The decision between 4:2:0 / sharp / 4:4:4 / 4:0:0 is made of three sums of a stencil over the
 * picture; the stencil looks pairs of 7x7x7 YUV cells up in a 343 x 343 byte table.  That table
 * is trained data of the reference (src/score_7.cc, `sjpeg::kSharpnessScore`): it ships beside the
 * library as riskiness.bin (the reference's bytes, Apache-2.0: riskiness.NOTICE) and is found there;
 * another copy goes in with sjpeg_hip_set_riskiness_table() (117649 bytes, host memory; copied) or
 * through the environment variable SJPEG_HIP_RISKINESS_TABLE (a file holding those bytes).  A
 * library installed without any: SJPEG_YUV_AUTO, SjpegCompress() and SjpegRiskiness() fail.
 * sjpeg_hip_riskiness_sums: device part; d_sums[nframes][3] = sum of the scores above the noise
 * level, their count, the count of neutral-chroma samples. */
#define SJPEG_HIP_RISKINESS_TABLE_SIZE 117649
int sjpeg_hip_set_riskiness_table(const uint8_t* table, size_t size);
int sjpeg_hip_has_riskiness_table(void);
int sjpeg_hip_riskiness_sums(const sjpeg_hip_source* src, int width, int height, int nframes,
                             const uint8_t* d_table, uint64_t* d_sums, void* stream);

/* ---- one frame over several GPUs ------------------------------------------
 * The reference codes a frame as ONE entropy segment (src/enc.cc:276-307; no restart markers,
 * src/sjpegi.h:68-74), so a frame can only be shared between devices at BIT granularity.  The
 * engine's unit of independent work is the segment (a fixed number of consecutive MCUs in scan
 * order); sjpeg_hip_segment_count() says how many a frame has.  Rank r of P codes the band of
 * segments [r*n/P, (r+1)*n/P): it needs the pixel rows of those MCUs plus the one MCU in front of
 * the band (DC predictors; `src` is addressed as the whole frame, rows outside are not read).
 * It gets the band's un-stuffed bit string (MSB-first 32-bit words) and its length in bits.  The
 * root gathers strings and lengths (RCCL gather / send-recv), and sjpeg_hip_stitch_bands() shifts
 * every band to its bit offset, pads with 1-bits, stuffs 0xFF bytes, adds header and EOI: the
 * bytes of sjpeg_hip_encode_scan_src() on one device, i.e. of the reference. */
int sjpeg_hip_segment_count(int width, int height, int yuv_mode);
/* capacity in 32-bit words a band buffer must have (0 on bad arguments) */
size_t sjpeg_hip_band_bound(int width, int height, int yuv_mode, int seg_begin, int seg_end);
int sjpeg_hip_encode_band_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                              int width, int height, int yuv_mode,
                              const sjpeg_hip_scan_tables* tables, int seg_begin, int seg_end,
                              uint32_t* d_words, size_t cap_words, uint64_t* d_nbits, void* stream);
/* d_words: nbands buffers of band_stride_words words each, in band order (device memory of
 * `engine`'s device); d_nbits[nbands]; the rest as sjpeg_hip_encode_scan() for one frame. */
int sjpeg_hip_stitch_bands(sjpeg_hip_engine* engine, int nbands, const uint32_t* d_words,
                           size_t band_stride_words, const uint64_t* d_nbits,
                           const void* header, size_t header_size, int append_eoi,
                           void* d_out, size_t out_cap, uint64_t* d_size, void* stream);

/* The same four operations for any pixel source (the functions above are these with
 * format = SJPEG_HIP_SRC_RGB).  yuv_mode must match the source where it is implied. */
int sjpeg_hip_encode_scan_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                              int width, int height, int yuv_mode, int nframes,
                              const sjpeg_hip_scan_tables* tables,
                              const void* header, size_t header_size, int append_eoi,
                              void* d_out, size_t out_stride, uint64_t* d_sizes, void* stream);
/* The same call with PACKED output: frame f is written at d_out + d_offsets[f], the frames back to back, every
 * one at a multiple of 16 with zero padding behind it; d_offsets[nframes] = the bytes the batch takes.
 * out_stride (a multiple of 16, as d_out) stays what ONE frame may take -- a frame that needs more reports
 * size 0 and takes no room --, so d_out needs nframes * out_stride bytes at most.  This is what
 * sjpeg_hip_compact_streams() makes of a strided batch, without the extra pass over the bytes: the block
 * d_out[0 .. d_offsets[nframes]) is ready for sjpeg_hip_gather_rows / _bytes (multi-device batch path).
 * Reference: none (src/enc.cc:391-448 codes one picture into one sink); config #4 of BASELINE.json. */
int sjpeg_hip_encode_scan_packed_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                                     int width, int height, int yuv_mode, int nframes,
                                     const sjpeg_hip_scan_tables* tables,
                                     const void* header, size_t header_size, int append_eoi,
                                     void* d_out, size_t out_stride, uint64_t* d_sizes,
                                     uint64_t* d_offsets, void* stream);
int sjpeg_hip_scan_coeffs_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                              int width, int height, int yuv_mode, int nframes,
                              const sjpeg_hip_scan_tables* tables, int16_t* d_coeffs, void* stream);
int sjpeg_hip_scan_histogram_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                                 int width, int height, int yuv_mode, int nframes,
                                 uint32_t* d_hist, void* stream);
int sjpeg_hip_scan_symbol_stats_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                                    int width, int height, int yuv_mode, int nframes,
                                    const sjpeg_hip_scan_tables* tables, uint32_t* d_freq,
                                    void* stream);

/* A whole batch the way the reference codes ONE picture with its default parameters
 * (Encoder::Encode, src/enc.cc:391-448): per-picture adapted quantizer (method >= 3:
 * CollectHistograms + AnalyseHisto) and per-picture optimised Huffman codes (method not 0 / 3:
 * the statistics half of SinglePassScanOptimized), then headers and the scan -- one launch per
 * device pass for all frames (histograms, analysis sums, symbol statistics, encode), the float
 * regression and BuildOptimalTable per frame on the host in between (two synchronisations of
 * `stream`).  quant = the two starting matrices (natural order, e.g. sjpeg_hip_quality_matrices),
 * min_quant NULL = ones, q_bias / qdelta_max_* as EncoderParam (0x78, 12, 1).  method 0..6 as
 * SjpegEncode (trellis methods: host API).  Output as sjpeg_hip_encode_scan_src (complete JPEGs,
 * EOI included).  A batch of 90 Mpixels or more is coded as TWO JOBS -- halves of the batch, each a complete
 * sequence of passes -- side by side: one on `stream`, one on a stream (and a child engine, with its own scratch)
 * the ENGINE owns; the call's host thread drives both, and `stream` ends behind both (the usual asynchronous
 * contract: outputs are complete when `stream` is).  The three passes have different bottlenecks, side by side they
 * fill each other's gaps: 32 4K frames 1.12 -> 1.00 ms.  Make the engine early in the life of the
 * process -- a stream made late shares a hardware queue with an older one, the caller's as a rule, and nothing overlaps. */
int sjpeg_hip_encode_batch_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                               int width, int height, int yuv_mode, int nframes,
                               const uint8_t quant[2][64], const uint8_t* min_quant /*[2][64]*/, int q_bias,
                               int method, int qdelta_max_luma, int qdelta_max_chroma,
                               void* d_out, size_t out_stride, uint64_t* d_sizes, void* stream);

/* Pipelined mode, for back-to-back encode calls on one engine (a service coding batch after
 * batch): K1 of a call runs on the caller's stream, the stitch kernels K2..K5 on a stream of the
 * engine, over two sets of segment buffers, so the stitch of call i (HBM-bound) runs under the
 * K1 of call i + 1 (ALU-bound).  In this mode d_out / d_sizes of an encode call are complete
 * only after sjpeg_hip_engine_wait(engine, stream) -- which makes `stream` wait for everything
 * the engine has in flight -- or a device synchronisation; do not touch them in between.
 * Results are the same bytes.  The other entry points stay ordered on the caller's stream (they
 * wait for the engine's stream first).  Off by default; switching it off drains the engine.  (The
 * engine's stream is made by the first call that switches the mode on: do that early too.) */
int sjpeg_hip_engine_set_pipelined(sjpeg_hip_engine* engine, int on);
int sjpeg_hip_engine_wait(sjpeg_hip_engine* engine, void* stream);

/* A batch whose frames each carry their OWN tables and header -- what a batch of the reference's
 * default encodes is (method 4: per-image adapted quantizer and per-image optimised Huffman
 * codes, src/enc.cc:801-812 + 323-372, one Encoder per image there, one launch here).
 * tables[nframes] (host); headers = the frames' header bytes back to back (host), frame f owns
 * [header_offsets[f], header_offsets[f+1]); flags must agree between the frames.  The rest as
 * sjpeg_hip_encode_scan_src() / sjpeg_hip_scan_symbol_stats_src(). */
int sjpeg_hip_encode_scan_multi(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                                int width, int height, int yuv_mode, int nframes,
                                const sjpeg_hip_scan_tables* tables /*[nframes]*/,
                                const void* headers, const size_t* header_offsets /*[nframes+1]*/,
                                int append_eoi, void* d_out, size_t out_stride, uint64_t* d_sizes,
                                void* stream);
int sjpeg_hip_scan_symbol_stats_multi(sjpeg_hip_engine* engine, const sjpeg_hip_source* src,
                                      int width, int height, int yuv_mode, int nframes,
                                      const sjpeg_hip_scan_tables* tables /*[nframes]*/,
                                      uint32_t* d_freq, void* stream);

/* ---- host-side helpers (tiny CPU work, no device needed) -----------------------------
 * They produce exactly what the reference's host code would hand to its hot loop, so that
 * a non-C++ binding can drive sjpeg_hip_encode_scan() without re-implementing them. */

/* quality -> the two 8-bit matrices (natural order) of Encoder::SetQuality
 * (src/enc.cc:100-104, src/quantize.cc:77-96). */
void sjpeg_hip_quality_matrices(float quality, uint8_t quant[2][64]);

/* Encoder::FinalizeQuantMatrix for both tables (src/quantize.cc:123-148): clamps quant to
 * min_quant (NULL = all ones) IN PLACE and fills tables->iquant / tables->bias.
 * q_bias is the reference's quantization_bias (default 0x78). */
void sjpeg_hip_finalize_quant(uint8_t quant[2][64], const uint8_t* min_quant /*[2][64]*/,
                              int q_bias, sjpeg_hip_scan_tables* tables);

/* Installs the default (JPEG Annex K.3) Huffman codes, as WriteDHT/InitCodes does for
 * method 0 (src/headers.cc:221-222, src/entropy.cc:84-128). */
void sjpeg_hip_default_huffman(sjpeg_hip_scan_tables* tables);

/* Writes SOI+APP0, DQT, SOF0, DHT (default tables), SOS for a frame without metadata into
 * buf (capacity cap), bytes identical to src/headers.cc.  Returns the size, 0 on error. */
size_t sjpeg_hip_make_header(int width, int height, int yuv_mode, const uint8_t quant[2][64],
                             uint8_t* buf, size_t cap);

/* Encoder::AnalyseHisto (src/histogram.cc:126-315) on ONE frame's histogram (host memory,
 * uint32 [2][64][128]): adapts quant IN PLACE, then re-finalises both tables into `tables`
 * exactly like the reference (quantization clamped to min_quant, NULL = all ones).
 * qdelta_max_* are EncoderParam::qdelta_max_luma / _chroma (defaults 12 / 1). */
void sjpeg_hip_adapt_quant(const uint32_t* hist, int yuv_mode, uint8_t quant[2][64],
                           const uint8_t* min_quant /*[2][64]*/, int q_bias,
                           int qdelta_max_luma, int qdelta_max_chroma,
                           sjpeg_hip_scan_tables* tables);

/* The same analysis with its bin loops on the device (src/histogram.cc:150-205): for every table,
 * position and candidate step (-12 .. +12 around quant) the rate and distortion sums over the
 * histogram, d_sums[nframes][2][64][25][2] (int64: bits, distortion; distortion == INT64_MIN marks a
 * step outside [min_quant, 255]), and d_totlast[nframes][2][64][2] (population, highest occupied
 * bin + 1).  Integer sums: exactly what the reference's double accumulators hold.  quant / min_quant
 * are host arrays (min_quant NULL = ones).  sjpeg_hip_adapt_quant_sums() is the float half on the
 * host (regression, lambda, choice of the step) for ONE frame's sums; it updates quant and tables
 * like sjpeg_hip_adapt_quant(). */
int sjpeg_hip_adapt_sums(const uint32_t* d_hist, int nframes, const uint8_t quant[2][64],
                         const uint8_t* min_quant /*[2][64]*/, int64_t* d_sums, int32_t* d_totlast,
                         void* stream);
void sjpeg_hip_adapt_quant_sums(const int64_t* sums, const int32_t* totlast, int yuv_mode,
                                uint8_t quant[2][64], const uint8_t* min_quant /*[2][64]*/, int q_bias,
                                int qdelta_max_luma, int qdelta_max_chroma,
                                sjpeg_hip_scan_tables* tables);
/* ... and the float half on the DEVICE too, behind sjpeg_hip_adapt_sums() on the same stream (src/histogram.cc:169-312:
 * the two-cloud line fit per position over the candidate steps, lambda from the slopes summed over the live positions
 * in ascending order, the choice of a step per position -- every expression and accumulation in the reference's order,
 * IEEE double, no contraction: the result is the host's, bit for bit): d_quant_out[nframes][2][64] = the adapted
 * matrices (4:0:0: table 0 only is written).  quant = the starting matrices the sums were made for (host array);
 * sjpeg_hip_finalize_quant() turns a frame's result into its tables.  What sjpeg_hip_encode_batch_src runs: 128 bytes a
 * frame come back from the device instead of 52 KB, and no regression on the host sits between two device passes. */
int sjpeg_hip_adapt_decide(const int64_t* d_sums, const int32_t* d_totlast, int nframes, const uint8_t quant[2][64],
                           int yuv_mode, int qdelta_max_luma, int qdelta_max_chroma, uint8_t* d_quant_out, void* stream);

/* CompileEntropyStats / BuildOptimalTable (src/entropy.cc:254-444) on ONE frame's symbol
 * statistics (host memory, uint32 [2][272]): fills specs[4] = {DC luma, DC chroma, AC luma,
 * AC chroma} (chroma untouched for 4:0:0) and installs their codes into `tables`. */
void sjpeg_hip_optimize_huffman(const uint32_t* freq, int yuv_mode,
                                sjpeg_hip_huffman_spec specs[4], sjpeg_hip_scan_tables* tables);

/* sjpeg_hip_make_header with explicit Huffman tables (specs as above; NULL = Annex K defaults). */
size_t sjpeg_hip_make_header_ex(int width, int height, int yuv_mode, const uint8_t quant[2][64],
                                const sjpeg_hip_huffman_spec* specs, uint8_t* buf, size_t cap);

/* The same with the metadata segments of the reference's EncoderParam (src/sjpeg.h:258-266) in
 * front of the tables, in its order: raw application markers (verbatim), EXIF (one APP1), ICC
 * profile (numbered APP2 chunks), XMP (one APP1, or main packet + extension chunks tied by the
 * MD5 of the extension when it exceeds 64 KiB): src/headers.cc:63-180.  Returns 0 on invalid
 * metadata (EXIF > 64 KiB, ICC >= 256 chunks, malformed extended XMP) or if cap is too small. */
typedef struct sjpeg_hip_metadata {
  const void* app_markers; size_t app_markers_size;
  const void* exif;        size_t exif_size;
  const void* iccp;        size_t iccp_size;
  const void* xmp;         size_t xmp_size;
  uint16_t xmp_split_point;                     /* 0 = default split of a long XMP packet */
} sjpeg_hip_metadata;
size_t sjpeg_hip_make_header_meta(int width, int height, int yuv_mode, const uint8_t quant[2][64],
                                  const sjpeg_hip_huffman_spec* specs,
                                  const sjpeg_hip_metadata* meta, uint8_t* buf, size_t cap);

/* Restart mode (SJPEG_HIP_RESTART_MARKERS): MCUs per restart interval for a colour mode (41 / 82 / 246),
 * and the DRI segment (FF DD 00 04 Ri) inserted in front of the SOS segment of a header made by any of
 * the builders above (in place; returns the new size, 0 if cap < size + 6 or no SOS is found). */
int sjpeg_hip_restart_interval(int yuv_mode);
size_t sjpeg_hip_header_add_restart(uint8_t* header, size_t size, size_t cap, int yuv_mode);

/* Restart mode, one frame over several devices: codes the restart intervals [seg_begin, seg_end) of the
 * frame (segments as counted by sjpeg_hip_segment_count()) into d_out: stuffed entropy bytes, RSTn
 * between the intervals AND behind the last one unless it is the frame's last; no header, no EOI.
 * Intervals are byte aligned and self-contained, so the file is
 *   header with DRI | bytes of band 0 | bytes of band 1 | ... | FF D9
 * put together by plain concatenation (host or device) in band order -- the "segmented at restart
 * markers, gathered, concatenated" form; bit-identical to the one-device restart-mode stream.
 * tables->flags must carry SJPEG_HIP_RESTART_MARKERS.  A band that does not fit out_cap reports size 0. */
int sjpeg_hip_encode_intervals_src(sjpeg_hip_engine* engine, const sjpeg_hip_source* src, int width, int height,
                                   int yuv_mode, const sjpeg_hip_scan_tables* tables, int seg_begin, int seg_end,
                                   void* d_out, size_t out_cap, uint64_t* d_size, void* stream);

/* Exchange step of the multi-device batch path (BASELINE.json config #4; the reference runs on one
 * thread and has no counterpart): packs the `nframes` coded streams an encode call left at
 * d_out + f*out_stride / d_sizes[f] back to back into d_packed, so that ONE collective (RCCL gather over
 * xGMI) or one copy moves them.  Frame f starts at d_offsets[f], a multiple of 16 (the up to 15 bytes
 * of padding behind a frame are zero); d_offsets[nframes] & ~SJPEG_HIP_PACKED_OVERFLOW is the number of
 * bytes the batch needs, and bit 63 (SJPEG_HIP_PACKED_OVERFLOW) is set when that is more than
 * packed_capacity -- a frame that would end behind the capacity is not copied.  The exchange below reads the
 * flag in the rank's row: every rank then returns SJPEG_HIP_ECAPACITY before anything is sent.  One launch on
 * `stream`, no host synchronisation.  d_out, d_packed and out_stride must be multiples of 16.
 * (sjpeg_hip_encode_scan_packed_src() writes this layout directly and needs no second pass.) */
#define SJPEG_HIP_PACKED_OVERFLOW (1ull << 63)
int sjpeg_hip_compact_streams(const void* d_out, size_t out_stride, const uint64_t* d_sizes, int nframes,
                              void* d_packed, size_t packed_capacity, uint64_t* d_offsets /* nframes + 1 */,
                              void* stream);

/* The exchange itself, for a C / C++ caller with one process per GPU: gathers the packed streams of every
 * rank (what sjpeg_hip_compact_streams left in d_packed / d_offsets) into the root's d_gathered over RCCL
 * (xGMI inside a node).  RCCL is resolved at run time (librccl.so.1: the copy already loaded in the
 * process, e.g. PyTorch's, else the one the loader finds, else /opt/rocm/lib): a process that never
 * gathers never loads it.
 *
 * A communicator is either created here -- rank 0 calls sjpeg_hip_comm_unique_id() and hands the 128 bytes
 * to the other ranks by whatever means the application has (MPI, a socket, torch.distributed's store),
 * then every rank calls sjpeg_hip_comm_create() with its device current -- or adopted from an ncclComm_t
 * the application already owns (sjpeg_hip_comm_adopt; not destroyed with the wrapper).
 *
 * sjpeg_hip_gather_streams(), called by every rank with the same frames_per_rank_max, root and
 * gathered_capacity:
 *   1. one ncclAllGather of a row of frames_per_rank_max + 2 uint64 per rank: the rank's packed bytes
 *      (d_offsets[nframes_local], a multiple of 16), its number of frames and its frame sizes (d_sizes,
 *      zero beyond nframes_local) -> d_rows [world][frames_per_rank_max + 2] on every rank;
 *   2. ONE small device-to-host read per rank and call: the rows (8 x world x (frames_per_rank_max + 2)
 *      bytes) -> h_rows, because RCCL's send / receive counts are host values; no per-frame
 *      synchronisation;
 *   3. grouped ncclSend / ncclRecv of EXACT lengths (no padding to the largest rank; RCCL has no
 *      gatherv): rank r's bytes land at d_gathered + h_rank_offsets[r] on the root, in rank order, the
 *      root's own by a device copy.
 * h_rows (host, [world][frames_per_rank_max + 2]) and h_rank_offsets (host, [world + 1]) are filled on
 * every rank; frame k of rank r is at h_rank_offsets[r] + the sum of the 16-aligned sizes of the rank's
 * earlier frames.  Every rank sees the same rows and therefore takes the same decision: a frame of size
 * 0 among the frames of some rank (it did not fit its slot), a rank whose d_offsets[nframes] carries
 * SJPEG_HIP_PACKED_OVERFLOW (its d_packed was too small: size it for nframes x out_stride) or is not the sum
 * of its frames, or a total above gathered_capacity, makes the call return SJPEG_HIP_ECAPACITY on EVERY rank
 * before anything is sent (never a hang).
 * d_gathered is only used on the root.  Everything is enqueued on `stream`; the host read waits for that
 * stream, the byte transfers do not.  Returns 0 or SJPEG_HIP_E*. */
typedef struct sjpeg_hip_comm sjpeg_hip_comm;
#define SJPEG_HIP_COMM_ID_BYTES 128
int sjpeg_hip_comm_unique_id(uint8_t id[SJPEG_HIP_COMM_ID_BYTES]);
int sjpeg_hip_comm_create(const uint8_t id[SJPEG_HIP_COMM_ID_BYTES], int rank, int world, sjpeg_hip_comm** comm);
int sjpeg_hip_comm_adopt(void* nccl_comm /* ncclComm_t */, sjpeg_hip_comm** comm);
/* A communicator on the LOCAL transport: its ranks are threads of ONE process (a server that drives the
 * GPUs of a node from one thread each, or several engines on one GPU), no RCCL in the process.  `id`: any
 * 128 bytes the application picks, the same on every rank of the group; every rank calls this once, with
 * the device it works on current.  The gather functions below are the same code on either transport; here
 * a transfer is a copy on the receiver's stream (a peer copy between devices), ordered behind the sender's
 * stream by an event and the sender's stream behind the copy by another -- the calls of a matched pair meet
 * on the host, so a send returns once its receive has been enqueued, and a rank that does not show up within
 * 60 s makes the call fail with SJPEG_HIP_ERUNTIME on the ranks that wait for it (never a hang). */
int sjpeg_hip_comm_create_local(const uint8_t id[SJPEG_HIP_COMM_ID_BYTES], int rank, int world,
                                sjpeg_hip_comm** comm);
void sjpeg_hip_comm_destroy(sjpeg_hip_comm* comm);
int sjpeg_hip_comm_rank(const sjpeg_hip_comm* comm);
int sjpeg_hip_comm_world(const sjpeg_hip_comm* comm);
/* The two halves of sjpeg_hip_gather_streams() for a root that sizes its buffer from the ACTUAL total:
 * steps 1-2 (every rank; h_rank_offsets[world] = bytes the root will receive; SJPEG_HIP_ECAPACITY on every
 * rank for a frame of size 0 or an overflowed d_packed), then step 3 (every rank).  Step 3 checks the ROOT's own
 * arguments on the root only (d_gathered NULL, gathered_capacity below the total it was just told, d_packed NULL):
 * the peers have queued their sends by then, so such an error leaves them unmatched and the communicator must
 * be destroyed -- size the buffer from h_rank_offsets[world] first (sjpeg_hip_gather_streams(), which knows the
 * capacity on every rank, refuses before anybody sends).  A root whose d_packed IS d_gathered +
 * h_rank_offsets[root] (it coded straight into place with sjpeg_hip_encode_scan_packed_src) copies nothing. */
int sjpeg_hip_gather_rows(sjpeg_hip_comm* comm, const uint64_t* d_offsets, const uint64_t* d_sizes, int nframes_local,
                          int frames_per_rank_max, uint64_t* d_rows, uint64_t* h_rows, uint64_t* h_rank_offsets,
                          void* stream);
int sjpeg_hip_gather_bytes(sjpeg_hip_comm* comm, int root, const void* d_packed, int frames_per_rank_max,
                           const uint64_t* h_rows, const uint64_t* h_rank_offsets, void* d_gathered,
                           size_t gathered_capacity, void* stream);
int sjpeg_hip_gather_streams(sjpeg_hip_comm* comm, int root, const void* d_packed, const uint64_t* d_offsets,
                             const uint64_t* d_sizes, int nframes_local, int frames_per_rank_max,
                             uint64_t* d_rows /* [world + 1][frames_per_rank_max + 2]: the last row is scratch */,
                             void* d_gathered, size_t gathered_capacity,
                             uint64_t* h_rows, uint64_t* h_rank_offsets /* [world + 1] */, void* stream);

/* Duration in milliseconds of the dominant kernel (the fused colour+fDCT+quant+entropy
 * kernel) in the most recent sjpeg_hip_encode_scan() call on this engine, measured with
 * HIP events on the caller's stream.  Timing is recorded only after
 * sjpeg_hip_engine_set_timing(engine, 1).  Negative if unavailable.  Synchronises. */
int sjpeg_hip_engine_set_timing(sjpeg_hip_engine* engine, int enable);
float sjpeg_hip_engine_last_scan_ms(sjpeg_hip_engine* engine);
float sjpeg_hip_engine_last_total_ms(sjpeg_hip_engine* engine);

/* Device memory the engine currently holds (it grows to what the largest call needed and is released
 * by sjpeg_hip_engine_trim / sjpeg_hip_engine_destroy).  The segment scratch of an encode call is sized from the caller's
 * out_stride: per frame about 3.5 x out_stride (segment slots + pool + the un-stuffed stream), capped at
 * the worst case of the geometry -- not the worst case itself. */
size_t sjpeg_hip_engine_scratch_bytes(sjpeg_hip_engine* engine);

/* Gives the engine's scratch back to the device (waits for the device's work first; tables and header
 * buffer included).  The next call allocates what it needs again -- for a service that has just coded
 * an unusually large frame or batch and does not want to keep its high-water mark. */
int sjpeg_hip_engine_trim(sjpeg_hip_engine* engine);

/* The host API (include/sjpeg.h) keeps one device context per calling thread: pixel, stream and plane
 * buffers plus an engine, grown on demand.  A context whose cached device memory exceeds
 * SJPEG_HIP_HOST_CACHE_BYTES (environment, default 1 GiB) after a call that itself needed less than
 * half of it releases it before returning (a steady stream of frames that need more than the limit
 * keeps its buffers: set the limit to what the service may hold, not below what one frame needs);
 * sjpeg_hip_host_trim() releases the calling thread's cache now and returns the bytes it held.
 * The host API first codes against an output capacity of half a byte per sample (0.75 B per pixel in
 * 4:2:0) and repeats the frame against sjpeg_hip_frame_bound() if that was too small
 * (SJPEG_HIP_HOST_FIRST_CAPACITY=bound: worst case from the first pass). */
size_t sjpeg_hip_host_trim(void);

/* Measurement aids of bench.py (no counterpart in the reference, not part of the encode path).
 * sjpeg_hip_debug_stream_read: a read-only streaming kernel over `bytes` of d_buf -- what the device's
 * HBM delivers to the simplest possible reader, beside the 8 TB/s specification figure.
 * sjpeg_hip_debug_valu_rate: cycles (at the nominal 2.4 GHz) a wave64 VALU instruction occupies a SIMD
 * at 8 waves per SIMD, for the two issue classes found on gfx950: cycles[0] = v_perm_b32 (every VOP3,
 * packed, multiply, dot and permute instruction), cycles[1] = v_add_u32 (simple 32-bit integer and
 * f32 instructions).  Synchronises on `stream`. */
int sjpeg_hip_debug_stream_read(const void* d_buf, size_t bytes, uint32_t* d_sink, void* stream);
int sjpeg_hip_debug_valu_rate(float cycles[2], void* stream);
/* sjpeg_hip_debug_shader_clock: the shader clock in MHz as a wave on `stream` measures it (cycle counter against the
 * device-wide 100 MHz counter over 20 us), i.e. the clock of the work that was just queued there.  Synchronises. */
int sjpeg_hip_debug_shader_clock(float* mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* SJPEG_HIP_H_ */
