"""sjpeg_amd -- Python binding (ctypes) of the MI355X-native sjpeg-compatible JPEG encoder.

The product is the C/C++ library ``sjpeg_amd/csrc/libsjpeg_amd.so`` (public API
``include/sjpeg.h``, device C-ABI ``include/sjpeg_hip.h``).  This module only binds those
symbols -- it contains no encoder logic and no CPU fallback: if the shared library is not
built, importing the binding raises; if no gfx950 device is present, every encode call fails.

torch is used (optionally) for device memory and streams only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SJPEG_AMD_LIB") or os.path.join(CSRC, "libsjpeg_amd.so")   # (override: A/B builds in tools/)

YUV_AUTO, YUV_420, YUV_SHARP, YUV_444, YUV_400 = range(5)

_u8p = C.POINTER(C.c_uint8)


class ScanTables(C.Structure):
    """struct sjpeg_hip_scan_tables (include/sjpeg_hip.h)."""
    _fields_ = [("iquant", (C.c_uint16 * 64) * 2),
                ("bias", (C.c_uint16 * 64) * 2),
                ("dc_codes", (C.c_uint32 * 12) * 2),
                ("ac_codes", (C.c_uint32 * 256) * 2),
                ("quant", (C.c_uint8 * 64) * 2),
                ("trellis_len", (C.c_uint8 * 256) * 2),
                ("flags", C.c_uint32)]


QUANT_TRELLIS = 1
QUANT_KEEP = 2
QUANT_REPLAY = 4
RESTART_MARKERS = 8      # optional restart-marker mode (sjpeg_hip.h): not the reference's bytes, the same pixels


SRC_RGB, SRC_BGRA, SRC_RGBA, SRC_GRAY, SRC_YUV444, SRC_YUV420, SRC_NV12, SRC_NV21 = range(8)
_IMPLIED_MODE = {SRC_GRAY: YUV_400, SRC_YUV444: YUV_444, SRC_YUV420: YUV_420, SRC_NV12: YUV_420,
                 SRC_NV21: YUV_420}


class Source(C.Structure):
    """struct sjpeg_hip_source (include/sjpeg_hip.h)."""
    _fields_ = [("format", C.c_int32), ("reserved", C.c_int32), ("plane", C.c_void_p * 3),
                ("row_stride", C.c_int64 * 3), ("frame_stride", C.c_int64 * 3)]


def sharp_yuv(fmt, frames):
    """SJPEG_YUV_SHARP conversion of device-resident packed frames [F, H, row_bytes] (fmt = SRC_RGB /
    SRC_BGRA / SRC_RGBA) -> (y [F, H, W], u [F, ch, cw], v [F, ch, cw]) uint8 CUDA tensors."""
    import torch
    assert frames.is_cuda and frames.dim() == 3
    f, h, row_bytes = frames.shape
    bpp = 3 if fmt == SRC_RGB else 4
    w = row_bytes // bpp
    cw, ch = (w + 1) // 2, (h + 1) // 2
    src, _ = make_source(fmt, [frames])
    y = torch.empty((f, h, w), dtype=torch.uint8, device=frames.device)
    u = torch.empty((f, ch, cw), dtype=torch.uint8, device=frames.device)
    v = torch.empty((f, ch, cw), dtype=torch.uint8, device=frames.device)
    L = lib()
    L.sjpeg_hip_sharp_workspace.restype = C.c_size_t
    L.sjpeg_hip_sharp_workspace.argtypes = [C.c_int, C.c_int, C.c_int]
    L.sjpeg_hip_sharp_yuv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    wsz = L.sjpeg_hip_sharp_workspace(w, h, f)
    work = torch.empty(max(wsz, 16), dtype=torch.uint8, device=frames.device)
    rc = L.sjpeg_hip_sharp_yuv(C.byref(src), w, h, f, y.data_ptr(), u.data_ptr(), v.data_ptr(), h * w, ch * cw,
                               work.data_ptr(), wsz, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise SjpegError("sjpeg_hip_sharp_yuv failed (%d)" % rc)
    return y, u, v


def segment_count(w, h, yuv_mode):
    n = lib().sjpeg_hip_segment_count(w, h, yuv_mode)
    if n <= 0:
        raise SjpegError("sjpeg_hip_segment_count: " + lib().sjpeg_hip_last_error().decode())
    return n


def band_bound(w, h, yuv_mode, seg_begin, seg_end):
    n = lib().sjpeg_hip_band_bound(w, h, yuv_mode, seg_begin, seg_end)
    if n == 0:
        raise SjpegError("sjpeg_hip_band_bound: bad arguments")
    return n


def make_source(fmt, planes):
    """planes: CUDA uint8 tensors [F, rows, row_bytes] (one per plane of the layout).
    Returns (Source, nframes); the tensors must outlive the calls that use it."""
    s = Source()
    s.format = fmt
    for i, t in enumerate(planes):
        assert t.is_cuda and t.dim() == 3 and t.stride(2) == 1
        s.plane[i] = t.data_ptr()
        s.row_stride[i] = t.stride(1)
        s.frame_stride[i] = t.stride(0)
    return s, planes[0].shape[0]


class HuffmanSpec(C.Structure):
    """struct sjpeg_hip_huffman_spec (include/sjpeg_hip.h)."""
    _fields_ = [("bits", C.c_uint8 * 16), ("syms", C.c_uint8 * 256), ("nsyms", C.c_int32)]


class SjpegError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP/C++ library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", CSRC] + ([] if verbose else ["-s"]))
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    try:
        # When torch is around it must come first: it ships its own HIP runtime, and one
        # process must hold exactly one (device pointers and streams are shared with torch).
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make -C sjpeg_amd/csrc` "
                          "(or __graft_entry__.build()); there is no pure-Python fallback")
    L = C.CDLL(LIB_PATH, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    L.SjpegVersion.restype = C.c_uint32
    L.SjpegHipLastError.restype = C.c_char_p
    L.SjpegEncode.restype = C.c_size_t
    L.SjpegEncode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p), C.c_float,
                              C.c_int, C.c_int]
    L.SjpegCompress.restype = C.c_size_t
    L.SjpegCompress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(_u8p)]
    L.SjpegFreeBuffer.argtypes = [_u8p]
    L.SjpegDimensions.restype = C.c_bool
    L.SjpegDimensions.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 3
    L.SjpegFindQuantizer.restype = C.c_int
    L.SjpegFindQuantizer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.SjpegEstimateQuality.restype = C.c_float
    L.SjpegEstimateQuality.argtypes = [C.c_void_p, C.c_bool]
    L.SjpegQuantMatrix.argtypes = [C.c_float, C.c_bool, C.c_void_p]
    L.sjpeg_hip_abi_version.restype = C.c_int
    L.sjpeg_hip_device_count.restype = C.c_int
    L.sjpeg_hip_last_error.restype = C.c_char_p
    L.sjpeg_hip_engine_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.sjpeg_hip_engine_destroy.argtypes = [C.c_void_p]
    L.sjpeg_hip_frame_bound.restype = C.c_size_t
    L.sjpeg_hip_frame_bound.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t]
    L.sjpeg_hip_encode_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.POINTER(ScanTables),
                                        C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p]
    L.sjpeg_hip_scan_coeffs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.POINTER(ScanTables),
                                        C.c_void_p, C.c_void_p]
    L.sjpeg_hip_quality_matrices.argtypes = [C.c_float, C.c_void_p]
    L.sjpeg_hip_finalize_quant.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(ScanTables)]
    L.sjpeg_hip_default_huffman.argtypes = [C.POINTER(ScanTables)]
    L.sjpeg_hip_make_header.restype = C.c_size_t
    L.sjpeg_hip_make_header.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.sjpeg_hip_scan_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_scan_symbol_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.POINTER(ScanTables),
                                              C.c_void_p, C.c_void_p]
    srcp = C.POINTER(Source)
    L.sjpeg_hip_encode_scan_src.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(ScanTables), C.c_void_p, C.c_size_t, C.c_int,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_scan_coeffs_src.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(ScanTables), C.c_void_p, C.c_void_p]
    L.sjpeg_hip_scan_histogram_src.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]
    L.sjpeg_hip_adapt_sums.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    L.sjpeg_hip_adapt_sums.restype = C.c_int
    L.sjpeg_hip_adapt_quant_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p]
    L.sjpeg_hip_adapt_quant_sums.restype = None
    L.sjpeg_hip_encode_batch_src.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_encode_batch_src.restype = C.c_int
    L.sjpeg_hip_engine_set_pipelined.argtypes = [C.c_void_p, C.c_int]
    L.sjpeg_hip_engine_set_pipelined.restype = C.c_int
    L.sjpeg_hip_engine_wait.argtypes = [C.c_void_p, C.c_void_p]
    L.sjpeg_hip_engine_wait.restype = C.c_int
    L.sjpeg_hip_encode_scan_multi.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.c_int,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_encode_scan_multi.restype = C.c_int
    L.sjpeg_hip_scan_symbol_stats_multi.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_scan_symbol_stats_multi.restype = C.c_int
    L.sjpeg_hip_scan_symbol_stats_src.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(ScanTables), C.c_void_p, C.c_void_p]
    L.sjpeg_hip_scan_quant_error_src.argtypes = [C.c_void_p, srcp, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(ScanTables), C.c_void_p, C.c_void_p]
    L.sjpeg_hip_engine_entropy_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.sjpeg_hip_segment_count.argtypes = [C.c_int, C.c_int, C.c_int]
    L.sjpeg_hip_band_bound.restype = C.c_size_t
    L.sjpeg_hip_band_bound.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.sjpeg_hip_encode_band_src.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_stitch_bands.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_char_p,
                                         C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.sjpeg_hip_adapt_quant.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.POINTER(ScanTables)]
    L.sjpeg_hip_optimize_huffman.argtypes = [C.c_void_p, C.c_int, C.POINTER(HuffmanSpec),
                                             C.POINTER(ScanTables)]
    L.sjpeg_hip_make_header_ex.restype = C.c_size_t
    L.sjpeg_hip_make_header_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(HuffmanSpec), C.c_void_p, C.c_size_t]
    L.sjpeg_hip_engine_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.sjpeg_hip_engine_last_scan_ms.restype = C.c_float
    L.sjpeg_hip_engine_last_scan_ms.argtypes = [C.c_void_p]
    L.sjpeg_hip_engine_last_total_ms.restype = C.c_float
    L.sjpeg_hip_engine_last_total_ms.argtypes = [C.c_void_p]
    _lib = L
    return L


EXPORTED_C_SYMBOLS = [
    # include/sjpeg.h (extern "C" part)
    "SjpegVersion", "SjpegCompress", "SjpegEncode", "SjpegFreeBuffer", "SjpegDimensions",
    "SjpegFindQuantizer", "SjpegEstimateQuality", "SjpegQuantMatrix", "SjpegRiskiness", "SjpegHipLastError",
    # include/sjpeg_hip.h
    "sjpeg_hip_abi_version", "sjpeg_hip_device_count", "sjpeg_hip_last_error",
    "sjpeg_hip_engine_create", "sjpeg_hip_engine_destroy", "sjpeg_hip_frame_bound",
    "sjpeg_hip_encode_scan", "sjpeg_hip_scan_coeffs", "sjpeg_hip_quality_matrices",
    "sjpeg_hip_finalize_quant", "sjpeg_hip_default_huffman", "sjpeg_hip_make_header",
    "sjpeg_hip_scan_histogram", "sjpeg_hip_scan_symbol_stats", "sjpeg_hip_adapt_quant",
    "sjpeg_hip_adapt_sums", "sjpeg_hip_adapt_quant_sums", "sjpeg_hip_adapt_decide",
    "sjpeg_hip_encode_scan_src", "sjpeg_hip_scan_coeffs_src", "sjpeg_hip_scan_histogram_src",
    "sjpeg_hip_scan_symbol_stats_src", "sjpeg_hip_scan_quant_error_src", "sjpeg_hip_engine_entropy_bits",
    "sjpeg_hip_engine_trim", "sjpeg_hip_host_trim",
    "sjpeg_hip_encode_scan_multi", "sjpeg_hip_scan_symbol_stats_multi",
    "sjpeg_hip_engine_set_pipelined", "sjpeg_hip_engine_wait", "sjpeg_hip_encode_batch_src",
    "sjpeg_hip_optimize_huffman", "sjpeg_hip_make_header_ex", "sjpeg_hip_make_header_meta",
    "sjpeg_hip_sharp_workspace", "sjpeg_hip_sharp_yuv",
    "sjpeg_hip_set_riskiness_table", "sjpeg_hip_has_riskiness_table", "sjpeg_hip_riskiness_sums",
    "sjpeg_hip_segment_count", "sjpeg_hip_band_bound", "sjpeg_hip_encode_band_src", "sjpeg_hip_stitch_bands",
    "sjpeg_hip_engine_set_timing", "sjpeg_hip_engine_last_scan_ms",
    "sjpeg_hip_engine_last_total_ms", "sjpeg_hip_engine_scratch_bytes", "sjpeg_hip_compact_streams",
    "sjpeg_hip_debug_stream_read", "sjpeg_hip_debug_valu_rate", "sjpeg_hip_debug_shader_clock",
    "sjpeg_hip_restart_interval", "sjpeg_hip_header_add_restart", "sjpeg_hip_encode_intervals_src",
    "sjpeg_hip_comm_unique_id", "sjpeg_hip_comm_create", "sjpeg_hip_comm_create_local", "sjpeg_hip_comm_adopt", "sjpeg_hip_comm_destroy",
    "sjpeg_hip_comm_rank", "sjpeg_hip_comm_world", "sjpeg_hip_gather_rows", "sjpeg_hip_gather_bytes",
    "sjpeg_hip_gather_streams", "sjpeg_hip_encode_scan_packed_src",
]


def restart_interval(yuv_mode: int) -> int:
    """MCUs per restart interval of the optional restart mode = per K1 segment (41 / 82 / 246)."""
    return int(lib().sjpeg_hip_restart_interval(int(yuv_mode)))


def header_add_restart(header: bytes, yuv_mode: int) -> bytes:
    """The header with the DRI segment of the restart mode in front of SOS."""
    buf = (C.c_uint8 * (len(header) + 6)).from_buffer_copy(header + b"\0" * 6)
    f = lib().sjpeg_hip_header_add_restart
    f.restype = C.c_size_t
    n = f(buf, C.c_size_t(len(header)), C.c_size_t(len(header) + 6), int(yuv_mode))
    if n == 0:
        raise SjpegError("sjpeg_hip_header_add_restart failed")
    return bytes(buf[:n])


def compact_streams(out, sizes, nframes=None, capacity=None, packed=None, offsets=None):
    """sjpeg_hip_compact_streams: the first `nframes` coded frames of `out` ([F, stride] uint8 CUDA,
    stride a multiple of 16) / `sizes` ([F] int64 CUDA) back to back, every frame at a multiple of 16.
    Returns (packed uint8 [capacity], offsets int64 [nframes + 1]); offsets[nframes] = bytes needed.
    One launch on the current stream, no synchronisation."""
    import torch
    n = int(out.shape[0] if nframes is None else nframes)
    stride = int(out.stride(0))
    if capacity is None:
        capacity = n * stride
    if packed is None:
        packed = torch.empty(int(capacity), dtype=torch.uint8, device=out.device)
    if offsets is None:
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=out.device)
    rc = lib().sjpeg_hip_compact_streams(C.c_void_p(out.data_ptr()), C.c_size_t(stride), C.c_void_p(sizes.data_ptr()), n,
                                         C.c_void_p(packed.data_ptr()), C.c_size_t(int(packed.numel())),
                                         C.c_void_p(offsets.data_ptr()),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise SjpegError(f"sjpeg_hip_compact_streams: {lib().sjpeg_hip_last_error().decode()}")
    return packed, offsets


def comm_unique_id() -> bytes:
    """sjpeg_hip_comm_unique_id: the 128 bytes rank 0 hands to the other ranks."""
    buf = (C.c_uint8 * 128)()
    if lib().sjpeg_hip_comm_unique_id(buf) != 0:
        raise SjpegError(f"sjpeg_hip_comm_unique_id: {lib().sjpeg_hip_last_error().decode()}")
    return bytes(buf)


class Comm:
    """The library's RCCL communicator (sjpeg_hip_comm_create): one per process, the device current."""

    def __init__(self, unique_id: bytes, rank: int, world: int, local: bool = False):
        """local: the ranks are threads of THIS process (sjpeg_hip_comm_create_local; any 128 bytes as the id)."""
        self._c = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        make = lib().sjpeg_hip_comm_create_local if local else lib().sjpeg_hip_comm_create
        if make(buf, int(rank), int(world), C.byref(self._c)) != 0:
            raise SjpegError(f"sjpeg_hip_comm_create{'_local' if local else ''}: {lib().sjpeg_hip_last_error().decode()}")
        self.rank, self.world = int(rank), int(world)

    def close(self):
        if self._c:
            lib().sjpeg_hip_comm_destroy(self._c)
            self._c = C.c_void_p()

    def gather_rows(self, offsets, sizes, n_local, per_max, rows_dev):
        """sjpeg_hip_gather_rows on the current stream (waits for it).  Returns (rows [world][per_max + 2]
        numpy uint64, rank_offsets [world + 1] numpy uint64) on every rank."""
        import torch
        rows = np.zeros((self.world, per_max + 2), np.uint64)
        offs = np.zeros(self.world + 1, np.uint64)
        rc = lib().sjpeg_hip_gather_rows(
            self._c, C.c_void_p(offsets.data_ptr()), C.c_void_p(sizes.data_ptr() if n_local > 0 else 0),
            int(n_local), int(per_max), C.c_void_p(rows_dev.data_ptr()), C.c_void_p(rows.ctypes.data),
            C.c_void_p(offs.ctypes.data), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_gather_rows: {lib().sjpeg_hip_last_error().decode()}")
        return rows, offs

    def gather_bytes(self, root, packed, per_max, rows, offs, gathered):
        """sjpeg_hip_gather_bytes on the current stream: enqueues the exact-length transfers."""
        import torch
        rc = lib().sjpeg_hip_gather_bytes(
            self._c, int(root), C.c_void_p(packed.data_ptr()), int(per_max), C.c_void_p(rows.ctypes.data),
            C.c_void_p(offs.ctypes.data), C.c_void_p(gathered.data_ptr() if gathered is not None else 0),
            C.c_size_t(int(gathered.numel()) if gathered is not None else 0),
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_gather_bytes: {lib().sjpeg_hip_last_error().decode()}")

    def gather_streams(self, root, packed, offsets, sizes, n_local, per_max, rows_dev, gathered):
        """sjpeg_hip_gather_streams: rows + the capacity check on EVERY rank + the transfers, one call.
        gathered (the root's buffer; its numel() is the capacity every rank passes) may be a tensor or an int
        capacity on the ranks that are not the root.  Returns (rows, rank_offsets) like gather_rows."""
        import torch
        rows = np.zeros((self.world, per_max + 2), np.uint64)
        offs = np.zeros(self.world + 1, np.uint64)
        is_t = hasattr(gathered, "data_ptr")
        rc = lib().sjpeg_hip_gather_streams(
            self._c, int(root), C.c_void_p(packed.data_ptr() if packed is not None else 0),
            C.c_void_p(offsets.data_ptr()), C.c_void_p(sizes.data_ptr() if n_local > 0 else 0), int(n_local),
            int(per_max), C.c_void_p(rows_dev.data_ptr()), C.c_void_p(gathered.data_ptr() if is_t else 0),
            C.c_size_t(int(gathered.numel()) if is_t else int(gathered)), C.c_void_p(rows.ctypes.data),
            C.c_void_p(offs.ctypes.data), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_gather_streams: {lib().sjpeg_hip_last_error().decode()}")
        return rows, offs


# ---------------------------------------------------------------- host API (sjpeg.h)

def last_error() -> str:
    return (lib().SjpegHipLastError() or b"").decode()


def host_trim() -> int:
    """Releases the device memory the host API caches for the calling thread (sjpeg_hip_host_trim);
    returns the number of bytes given back."""
    f = lib().sjpeg_hip_host_trim
    f.restype = C.c_size_t
    f.argtypes = []
    return int(f())


def set_riskiness_table(table: bytes) -> None:
    """Installs the reference's riskiness score table (117649 bytes) for SJPEG_YUV_AUTO."""
    buf = (C.c_uint8 * len(table)).from_buffer_copy(table)
    if lib().sjpeg_hip_set_riskiness_table(buf, len(table)) != 0:
        raise SjpegError("sjpeg_hip_set_riskiness_table: wrong size")


def SjpegRiskiness(rgb: np.ndarray):
    """(SjpegYUVMode, risk) like the reference's SjpegRiskiness; (YUV_AUTO, -1.0) if the table is missing."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[0], rgb.shape[1]
    risk = C.c_float(0)
    L = lib()
    L.SjpegRiskiness.restype = C.c_int
    L.SjpegRiskiness.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    mode = L.SjpegRiskiness(rgb.ctypes.data, w, h, 3 * w, C.byref(risk))
    return int(mode), float(risk.value)


def SjpegCompress(rgb: np.ndarray, quality: float = 75.0):
    """SjpegCompress() of include/sjpeg.h (method 4, SJPEG_YUV_AUTO).  Returns bytes or None."""
    return SjpegEncode(rgb, quality, 4, YUV_AUTO)


def SjpegEncode(rgb: np.ndarray, quality: float = 75.0, method: int = 0,
                yuv_mode: int = YUV_420, stride=None):
    """SjpegEncode() of include/sjpeg.h on a host image (H, W, 3) uint8.  Returns bytes or None."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[0], rgb.shape[1]
    out = _u8p()
    base = rgb.ctypes.data
    if stride is None:
        stride = rgb.strides[0]
    elif stride < 0:
        base = rgb.ctypes.data + (h - 1) * (-stride)     # first row = last in memory
    n = lib().SjpegEncode(base, w, h, stride, C.byref(out), quality, method, yuv_mode)
    if n == 0:
        return None
    data = bytes((C.c_ubyte * n).from_address(C.addressof(out.contents)))   # (string_at: 2 GiB limit)
    lib().SjpegFreeBuffer(out)
    return data


# ---------------------------------------------------------------- device C-ABI (sjpeg_hip.h)

def device_count() -> int:
    return lib().sjpeg_hip_device_count()


def make_tables(quality=None, quant=None, min_quant=None, q_bias=0x78):
    """(ScanTables, final quant[2][64]) exactly as the reference's host code prepares them."""
    L = lib()
    q = np.zeros((2, 64), np.uint8)
    if quant is None:
        L.sjpeg_hip_quality_matrices(float(quality), q.ctypes.data)
    else:
        q[:] = np.asarray(quant, np.uint8).reshape(2, 64)
    t = ScanTables()
    mq = None
    if min_quant is not None:
        mq = np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
    L.sjpeg_hip_finalize_quant(q.ctypes.data, mq.ctypes.data if mq is not None else None, q_bias,
                               C.byref(t))
    L.sjpeg_hip_default_huffman(C.byref(t))
    return t, q


def make_header(w, h, yuv_mode, quant) -> bytes:
    buf = np.zeros(2048, np.uint8)
    q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
    n = lib().sjpeg_hip_make_header(w, h, yuv_mode, q.ctypes.data, buf.ctypes.data, buf.size)
    if n == 0:
        raise SjpegError("sjpeg_hip_make_header failed")
    return buf[:n].tobytes()


def adapt_quant(hist: np.ndarray, yuv_mode, quant, min_quant=None, q_bias=0x78, dmax_luma=12,
                dmax_chroma=1):
    """Host AnalyseHisto on one frame's histogram [2][64][128]; returns (ScanTables, new quant)."""
    h = np.ascontiguousarray(hist, np.uint32).reshape(2, 64, 128)
    q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64).copy()
    mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
    t = ScanTables()
    lib().sjpeg_hip_finalize_quant(q.ctypes.data, mq.ctypes.data if mq is not None else None, q_bias,
                                   C.byref(t))
    lib().sjpeg_hip_adapt_quant(h.ctypes.data, yuv_mode, q.ctypes.data,
                                mq.ctypes.data if mq is not None else None, q_bias, dmax_luma,
                                dmax_chroma, C.byref(t))
    lib().sjpeg_hip_default_huffman(C.byref(t))
    return t, q


def adapt_quant_device_batch(hists_dev, yuv_mode, quant, min_quant=None, q_bias=0x78, dmax_luma=12,
                             dmax_chroma=1):
    """AnalyseHisto for a batch with the bin loops on the GPU: hists_dev = CUDA int32 tensor
    [F, 2, 64, 128] (Engine.scan_histogram()); ONE launch, the sums come back (52 KB per frame),
    the float half runs on the host per frame.  Returns [(ScanTables, quant[2][64])] * F."""
    import torch
    q0 = np.ascontiguousarray(quant, np.uint8).reshape(2, 64).copy()
    mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
    mqp = mq.ctypes.data if mq is not None else None
    f = hists_dev.shape[0]
    sums = torch.zeros((f, 2, 64, 25, 2), dtype=torch.int64, device=hists_dev.device)
    totlast = torch.zeros((f, 2, 64, 2), dtype=torch.int32, device=hists_dev.device)
    L = lib()
    rc = L.sjpeg_hip_adapt_sums(hists_dev.contiguous().data_ptr(), f, q0.ctypes.data, mqp, sums.data_ptr(),
                                totlast.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise SjpegError("sjpeg_hip_adapt_sums: " + L.sjpeg_hip_last_error().decode())
    s_host = np.ascontiguousarray(sums.cpu().numpy())
    t_host = np.ascontiguousarray(totlast.cpu().numpy())
    res = []
    for k in range(f):
        q = q0.copy()
        t = ScanTables()
        L.sjpeg_hip_finalize_quant(q.ctypes.data, mqp, q_bias, C.byref(t))
        L.sjpeg_hip_adapt_quant_sums(s_host[k].ctypes.data, t_host[k].ctypes.data, yuv_mode, q.ctypes.data, mqp,
                                     q_bias, dmax_luma, dmax_chroma, C.byref(t))
        L.sjpeg_hip_default_huffman(C.byref(t))
        res.append((t, q))
    return res


def adapt_quant_on_device(hists_dev, yuv_mode, quant, min_quant=None, dmax_luma=12, dmax_chroma=1, q_bias=0x78):
    """AnalyseHisto for a batch with BOTH halves on the GPU (sjpeg_hip_adapt_sums + sjpeg_hip_adapt_decide): hists_dev =
    CUDA int32 tensor [F, 2, 64, 128]; returns the adapted matrices, uint8 [F, 2, 64] (4:0:0: table 1 = quant's).  The
    starting matrices are finalized first (raised to min_quant), as adapt_quant() and the batch path do."""
    import torch
    q0 = np.ascontiguousarray(quant, np.uint8).reshape(2, 64).copy()
    mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
    mqp = mq.ctypes.data if mq is not None else None
    lib().sjpeg_hip_finalize_quant(q0.ctypes.data, mqp, q_bias, C.byref(ScanTables()))
    f = hists_dev.shape[0]
    sums = torch.zeros((f, 2, 64, 25, 2), dtype=torch.int64, device=hists_dev.device)
    totlast = torch.zeros((f, 2, 64, 2), dtype=torch.int32, device=hists_dev.device)
    out = torch.from_numpy(np.broadcast_to(q0, (f, 2, 64)).copy()).to(hists_dev.device)
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    L.sjpeg_hip_adapt_decide.restype = C.c_int
    L.sjpeg_hip_adapt_decide.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rc = L.sjpeg_hip_adapt_sums(hists_dev.contiguous().data_ptr(), f, q0.ctypes.data, mqp, sums.data_ptr(), totlast.data_ptr(), st)
    if rc == 0:
        rc = L.sjpeg_hip_adapt_decide(sums.data_ptr(), totlast.data_ptr(), f, q0.ctypes.data, yuv_mode, dmax_luma, dmax_chroma,
                                      out.data_ptr(), st)
    if rc != 0:
        raise SjpegError("sjpeg_hip_adapt_sums / _decide: " + L.sjpeg_hip_last_error().decode())
    return out.cpu().numpy()


def adapt_quant_device(hist_dev, yuv_mode, quant, min_quant=None, q_bias=0x78, dmax_luma=12, dmax_chroma=1):
    """One frame of adapt_quant_device_batch(): hist_dev = CUDA int32 tensor [2, 64, 128]."""
    return adapt_quant_device_batch(hist_dev.unsqueeze(0), yuv_mode, quant, min_quant, q_bias, dmax_luma,
                                    dmax_chroma)[0]


def optimize_huffman(freq: np.ndarray, yuv_mode, tables: ScanTables):
    """Host BuildOptimalTable on one frame's symbol statistics [2][272]; installs the codes into
    `tables` and returns the four HuffmanSpec (DC luma, DC chroma, AC luma, AC chroma)."""
    f = np.ascontiguousarray(freq, np.uint32).reshape(2, 272)
    specs = (HuffmanSpec * 4)()
    lib().sjpeg_hip_optimize_huffman(f.ctypes.data, yuv_mode, specs, C.byref(tables))
    return specs


def make_header_ex(w, h, yuv_mode, quant, specs) -> bytes:
    buf = np.zeros(4096, np.uint8)
    q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
    n = lib().sjpeg_hip_make_header_ex(w, h, yuv_mode, q.ctypes.data, specs, buf.ctypes.data, buf.size)
    if n == 0:
        raise SjpegError("sjpeg_hip_make_header_ex failed")
    return buf[:n].tobytes()


class Metadata(C.Structure):
    """struct sjpeg_hip_metadata (include/sjpeg_hip.h)."""
    _fields_ = [("app_markers", C.c_char_p), ("app_markers_size", C.c_size_t),
                ("exif", C.c_char_p), ("exif_size", C.c_size_t),
                ("iccp", C.c_char_p), ("iccp_size", C.c_size_t),
                ("xmp", C.c_char_p), ("xmp_size", C.c_size_t),
                ("xmp_split_point", C.c_uint16)]


def make_header_meta(w, h, yuv_mode, quant, specs=None, app_markers=b"", exif=b"", iccp=b"", xmp=b"",
                     xmp_split_point=0):
    """Header bytes SOI..SOS with metadata segments; None if the metadata is invalid."""
    m = Metadata(app_markers or None, len(app_markers), exif or None, len(exif), iccp or None, len(iccp),
                 xmp or None, len(xmp), xmp_split_point)
    cap = 4096 + len(app_markers) + len(exif) + len(iccp) + len(xmp) + 64 * (len(iccp) // 65000 + len(xmp) // 65000 + 4)
    buf = np.zeros(cap, np.uint8)
    q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
    L = lib()
    L.sjpeg_hip_make_header_meta.restype = C.c_size_t
    L.sjpeg_hip_make_header_meta.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t]
    n = L.sjpeg_hip_make_header_meta(w, h, yuv_mode, q.ctypes.data, specs, C.byref(m), buf.ctypes.data, cap)
    return buf[:n].tobytes() if n else None


def frame_bound(w, h, yuv_mode, header_size) -> int:
    return lib().sjpeg_hip_frame_bound(w, h, yuv_mode, header_size)


class Engine:
    """sjpeg_hip_engine bound to one device; drives device-resident (torch) frames."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = lib().sjpeg_hip_engine_create(device, C.byref(self._h))
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_engine_create: {lib().sjpeg_hip_last_error().decode()}")
        self.device = device

    def close(self):
        if self._h:
            lib().sjpeg_hip_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timing(self, on: bool):
        lib().sjpeg_hip_engine_set_timing(self._h, int(on))

    def set_pipelined(self, on: bool):
        """Back-to-back encode calls overlap (stitch of call i under K1 of call i + 1); outputs are
        complete after wait() or a device synchronisation (include/sjpeg_hip.h)."""
        self._chk(lib().sjpeg_hip_engine_set_pipelined(self._h, int(on)), "sjpeg_hip_engine_set_pipelined")

    def wait(self):
        """Makes the current torch stream wait for everything the engine has in flight."""
        self._chk(lib().sjpeg_hip_engine_wait(self._h, self._stream()), "sjpeg_hip_engine_wait")

    def last_scan_ms(self) -> float:
        return lib().sjpeg_hip_engine_last_scan_ms(self._h)

    def last_total_ms(self) -> float:
        return lib().sjpeg_hip_engine_last_total_ms(self._h)

    def scratch_bytes(self) -> int:
        """Device memory the engine holds right now (sjpeg_hip_engine_scratch_bytes)."""
        f = lib().sjpeg_hip_engine_scratch_bytes
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p]
        return int(f(self._h))

    def trim(self) -> None:
        """Give the engine's scratch back to the device (sjpeg_hip_engine_trim); the next call
        allocates what it needs again."""
        f = lib().sjpeg_hip_engine_trim
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        self._chk(f(self._h), "sjpeg_hip_engine_trim")

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _check_frames(frames):
        """The C-ABI takes only a row and a frame stride: pixels must be packed RGB, 3 bytes apart."""
        import torch
        if not (frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
                and frames.stride(3) == 1 and frames.stride(2) == 3):
            raise SjpegError("frames must be a CUDA uint8 tensor [F, H, W, 3] of packed RGB pixels "
                             "(stride 1 over the channels, 3 over x); call .contiguous() on a permuted or sliced view")

    def encode_frames(self, frames, tables: ScanTables, header: bytes, yuv_mode: int,
                      out=None, sizes=None, out_stride=None, append_eoi=True):
        """frames: torch.uint8 CUDA tensor [F, H, W, 3] (contiguous rows).  Returns
        (out [F, out_stride] uint8 CUDA tensor, sizes [F] int64 CUDA tensor).  Asynchronous
        on the current torch stream."""
        import torch
        self._check_frames(frames)
        f, h, w, _ = frames.shape
        # header None: the C-ABI's NULL / 0 -- every stream starts directly with its entropy-coded data (the call
        # shape of INTEGRATION.md section B, where the reference has written SOI..SOS itself)
        hdr_len = 0 if header is None else len(header)
        if out_stride is None:
            out_stride = frame_bound(w, h, yuv_mode, hdr_len)
        if out is None:
            out = torch.empty((f, out_stride), dtype=torch.uint8, device=frames.device)
        if sizes is None:
            sizes = torch.zeros(f, dtype=torch.int64, device=frames.device)
        rc = lib().sjpeg_hip_encode_scan(self._h, frames.data_ptr(), frames.stride(1),
                                         frames.stride(0), w, h, yuv_mode, f, C.byref(tables),
                                         header, hdr_len, int(append_eoi), out.data_ptr(),
                                         out_stride, sizes.data_ptr(), self._stream())
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_encode_scan: {lib().sjpeg_hip_last_error().decode()}")
        return out, sizes

    def encode_frames_packed(self, frames, tables: ScanTables, header: bytes, yuv_mode: int,
                             out, sizes, offsets, out_stride, append_eoi=True):
        """sjpeg_hip_encode_scan_packed_src: like encode_frames(), but frame k is written at out + offsets[k], the
        frames back to back at multiples of 16 (what compact_streams() makes of a strided batch, without the extra
        pass).  out: flat uint8 CUDA tensor of at least F * out_stride bytes (16-byte aligned; it may be longer --
        the root of an exchange codes straight into the buffer the other ranks' streams are gathered behind),
        sizes [F] int64, offsets [F + 1] int64, out_stride a multiple of 16 = what one frame may take."""
        self._check_frames(frames)
        f, h, w, _ = frames.shape
        assert out.is_cuda and out.dtype.itemsize == 1 and out.numel() >= f * out_stride and offsets.numel() >= f + 1
        src, _ = make_source(SRC_RGB, [frames.view(f, h, w * 3)])
        L = lib()
        L.sjpeg_hip_encode_scan_packed_src.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_int,
            C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = L.sjpeg_hip_encode_scan_packed_src(self._h, C.byref(src), w, h, yuv_mode, f, C.byref(tables), header,
                                                len(header), int(append_eoi), out.data_ptr(), int(out_stride),
                                                sizes.data_ptr(), offsets.data_ptr(), self._stream())
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_encode_scan_packed_src: {lib().sjpeg_hip_last_error().decode()}")
        return out, sizes, offsets

    # ---- any pixel source (sjpeg_hip_source) -------------------------------------------------
    def _chk(self, rc, what):
        if rc != 0:
            raise SjpegError(f"{what}: {lib().sjpeg_hip_last_error().decode()}")

    def encode_source(self, src: Source, nframes, w, h, tables: ScanTables, header: bytes, yuv_mode,
                      out_stride=None, append_eoi=True, device="cuda"):
        import torch
        if out_stride is None:
            out_stride = frame_bound(w, h, yuv_mode, len(header))
        out = torch.empty((nframes, out_stride), dtype=torch.uint8, device=device)
        sizes = torch.zeros(nframes, dtype=torch.int64, device=device)
        self._chk(lib().sjpeg_hip_encode_scan_src(self._h, C.byref(src), w, h, yuv_mode, nframes,
                                                  C.byref(tables), header, len(header),
                                                  int(append_eoi), out.data_ptr(), out_stride,
                                                  sizes.data_ptr(), self._stream()),
                  "sjpeg_hip_encode_scan_src")
        return out, sizes

    def scan_histogram_source(self, src: Source, nframes, w, h, yuv_mode, device="cuda"):
        import torch
        out = torch.zeros((nframes, 2, 64, 128), dtype=torch.int32, device=device)
        self._chk(lib().sjpeg_hip_scan_histogram_src(self._h, C.byref(src), w, h, yuv_mode, nframes,
                                                     out.data_ptr(), self._stream()),
                  "sjpeg_hip_scan_histogram_src")
        return out

    def scan_symbol_stats_source(self, src: Source, nframes, w, h, tables, yuv_mode, device="cuda"):
        import torch
        out = torch.zeros((nframes, 2, 272), dtype=torch.int32, device=device)
        self._chk(lib().sjpeg_hip_scan_symbol_stats_src(self._h, C.byref(src), w, h, yuv_mode, nframes,
                                                        C.byref(tables), out.data_ptr(), self._stream()),
                  "sjpeg_hip_scan_symbol_stats_src")
        return out

    def encode_source_multi(self, src: Source, nframes, w, h, tables_list, headers, yuv_mode,
                            out_stride=None, append_eoi=True, device="cuda"):
        """Frame f is coded with tables_list[f] and gets headers[f] in front (one launch)."""
        import torch
        assert len(tables_list) == nframes and len(headers) == nframes
        arr = (ScanTables * nframes)(*tables_list)
        offs = (C.c_size_t * (nframes + 1))()
        for i, hd in enumerate(headers):
            offs[i + 1] = offs[i] + len(hd)
        blob = b"".join(headers)
        if out_stride is None:
            out_stride = frame_bound(w, h, yuv_mode, max(len(hd) for hd in headers))
        out = torch.empty((nframes, out_stride), dtype=torch.uint8, device=device)
        sizes = torch.zeros(nframes, dtype=torch.int64, device=device)
        self._chk(lib().sjpeg_hip_encode_scan_multi(self._h, C.byref(src), w, h, yuv_mode, nframes,
                                                    C.cast(arr, C.c_void_p), blob, offs, int(append_eoi),
                                                    out.data_ptr(), out_stride, sizes.data_ptr(),
                                                    self._stream()),
                  "sjpeg_hip_encode_scan_multi")
        return out, sizes

    def scan_symbol_stats_multi(self, src: Source, nframes, w, h, tables_list, yuv_mode, device="cuda"):
        import torch
        assert len(tables_list) == nframes
        arr = (ScanTables * nframes)(*tables_list)
        out = torch.zeros((nframes, 2, 272), dtype=torch.int32, device=device)
        self._chk(lib().sjpeg_hip_scan_symbol_stats_multi(self._h, C.byref(src), w, h, yuv_mode, nframes,
                                                          C.cast(arr, C.c_void_p), out.data_ptr(),
                                                          self._stream()),
                  "sjpeg_hip_scan_symbol_stats_multi")
        return out

    def encode_batch(self, src: Source, nframes, w, h, yuv_mode, quant, method=4, min_quant=None, q_bias=0x78,
                     dmax_luma=12, dmax_chroma=1, out_stride=None, device="cuda", out=None, sizes=None):
        """The reference's per-picture analysis (methods 0..6) for a whole batch in one C call
        (sjpeg_hip_encode_batch_src).  Returns (out [F, stride] uint8, sizes [F] int64)."""
        import torch
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
        if out_stride is None:
            out_stride = frame_bound(w, h, yuv_mode, 2048)
        if out is None:
            out = torch.empty((nframes, out_stride), dtype=torch.uint8, device=device)
        if sizes is None:
            sizes = torch.zeros(nframes, dtype=torch.int64, device=device)
        self._chk(lib().sjpeg_hip_encode_batch_src(self._h, C.byref(src), w, h, yuv_mode, nframes, q.ctypes.data,
                                                   mq.ctypes.data if mq is not None else None, q_bias, int(method),
                                                   dmax_luma, dmax_chroma, out.data_ptr(), out_stride,
                                                   sizes.data_ptr(), self._stream()),
                  "sjpeg_hip_encode_batch_src")
        return out, sizes

    def scan_quant_error_source(self, src: Source, nframes, w, h, tables, yuv_mode, device="cuda"):
        import torch
        out = torch.zeros(nframes, dtype=torch.int64, device=device)
        self._chk(lib().sjpeg_hip_scan_quant_error_src(self._h, C.byref(src), w, h, yuv_mode, nframes,
                                                       C.byref(tables), out.data_ptr(), self._stream()),
                  "sjpeg_hip_scan_quant_error_src")
        return out

    def encode_band(self, src: Source, w, h, tables: ScanTables, yuv_mode, seg_begin, seg_end,
                    cap_words=None, device="cuda"):
        """One band of a frame shared between GPUs (include/sjpeg_hip.h): un-stuffed bit string
        (int32 tensor of MSB-first words) + its length in bits (int64 tensor [1])."""
        import torch
        need = band_bound(w, h, yuv_mode, seg_begin, seg_end)
        cap_words = need if cap_words is None else cap_words
        words = torch.zeros(cap_words, dtype=torch.int32, device=device)
        nbits = torch.zeros(1, dtype=torch.int64, device=device)
        self._chk(lib().sjpeg_hip_encode_band_src(self._h, C.byref(src), w, h, yuv_mode, C.byref(tables),
                                                  seg_begin, seg_end, words.data_ptr(), cap_words,
                                                  nbits.data_ptr(), self._stream()),
                  "sjpeg_hip_encode_band_src")
        return words, nbits

    def encode_intervals(self, src: Source, w, h, tables: ScanTables, yuv_mode, seg_begin, seg_end,
                         out_cap=None, device="cuda"):
        """Restart mode: the restart intervals [seg_begin, seg_end) of one frame as stuffed bytes with
        their RSTn markers (sjpeg_hip_encode_intervals_src).  Returns (out uint8 [out_cap], size int64 [1])."""
        import torch
        if out_cap is None:
            out_cap = 4 * band_bound(w, h, yuv_mode, seg_begin, seg_end) * 2 + 4096
        out = torch.empty(int(out_cap), dtype=torch.uint8, device=device)
        size = torch.zeros(1, dtype=torch.int64, device=device)
        self._chk(lib().sjpeg_hip_encode_intervals_src(self._h, C.byref(src), w, h, yuv_mode, C.byref(tables),
                                                       seg_begin, seg_end, C.c_void_p(out.data_ptr()),
                                                       C.c_size_t(int(out_cap)), C.c_void_p(size.data_ptr()),
                                                       self._stream()),
                  "sjpeg_hip_encode_intervals_src")
        return out, size

    def stitch_bands(self, words, nbits, header: bytes, append_eoi=True, out_cap=None):
        """words [nbands, stride] int32, nbits [nbands] int64 (this engine's device) -> JPEG bytes."""
        import torch
        assert words.dim() == 2 and words.is_contiguous() and nbits.numel() == words.shape[0]
        nb, stride = words.shape
        if out_cap is None:
            out_cap = len(header) + 2 * 4 * nb * stride + 4096
        out = torch.empty(out_cap, dtype=torch.uint8, device=words.device)
        size = torch.zeros(1, dtype=torch.int64, device=words.device)
        self._chk(lib().sjpeg_hip_stitch_bands(self._h, nb, words.data_ptr(), stride, nbits.data_ptr(),
                                               header, len(header), int(append_eoi), out.data_ptr(),
                                               out_cap, size.data_ptr(), self._stream()),
                  "sjpeg_hip_stitch_bands")
        n = int(size.item())
        if n == 0:
            raise SjpegError("sjpeg_hip_stitch_bands: output buffer too small")
        return bytes(out[:n].cpu().numpy())

    def entropy_bits(self, nframes):
        bits = np.zeros(nframes, np.uint64)
        self._chk(lib().sjpeg_hip_engine_entropy_bits(self._h, bits.ctypes.data, nframes),
                  "sjpeg_hip_engine_entropy_bits")
        return bits

    def scan_histogram(self, frames, yuv_mode: int):
        """[F, 2, 64, 128] uint32 (as int32 tensor) coefficient histograms (adaptive quantization)."""
        import torch
        f, h, w, _ = frames.shape
        out = torch.zeros((f, 2, 64, 128), dtype=torch.int32, device=frames.device)
        self._check_frames(frames)
        rc = lib().sjpeg_hip_scan_histogram(self._h, frames.data_ptr(), frames.stride(1),
                                            frames.stride(0), w, h, yuv_mode, f, out.data_ptr(),
                                            self._stream())
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_scan_histogram: {lib().sjpeg_hip_last_error().decode()}")
        return out

    def scan_symbol_stats(self, frames, tables: ScanTables, yuv_mode: int):
        """[F, 2, 272] symbol counts (256 AC + 16 DC per table) for Huffman optimisation."""
        import torch
        f, h, w, _ = frames.shape
        out = torch.zeros((f, 2, 272), dtype=torch.int32, device=frames.device)
        self._check_frames(frames)
        rc = lib().sjpeg_hip_scan_symbol_stats(self._h, frames.data_ptr(), frames.stride(1),
                                               frames.stride(0), w, h, yuv_mode, f, C.byref(tables),
                                               out.data_ptr(), self._stream())
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_scan_symbol_stats: {lib().sjpeg_hip_last_error().decode()}")
        return out

    def scan_coeffs(self, frames, tables: ScanTables, yuv_mode: int):
        import torch
        f, h, w, _ = frames.shape
        px = 16 if yuv_mode == YUV_420 else 8
        per = {YUV_420: 6, YUV_444: 3, YUV_400: 1}[yuv_mode]
        nb = ((w + px - 1) // px) * ((h + px - 1) // px) * per
        coeffs = torch.zeros((f, nb, 64), dtype=torch.int16, device=frames.device)
        self._check_frames(frames)
        rc = lib().sjpeg_hip_scan_coeffs(self._h, frames.data_ptr(), frames.stride(1),
                                         frames.stride(0), w, h, yuv_mode, f, C.byref(tables),
                                         coeffs.data_ptr(), self._stream())
        if rc != 0:
            raise SjpegError(f"sjpeg_hip_scan_coeffs: {lib().sjpeg_hip_last_error().decode()}")
        return coeffs


def encode_device_method(frames, quality=75.0, yuv_mode=YUV_420, method=4, engine=None, quant=None,
                         min_quant=None, q_bias=0x78, dmax_luma=12, dmax_chroma=1):
    """Per-frame adaptive quantization / optimised Huffman tables (reference methods 0..6) for a
    batch of device-resident frames [F, H, W, 3]: one call of sjpeg_hip_encode_batch_src (every
    device pass is one launch over the batch).  Returns a list of JPEG byte strings."""
    eng = engine or Engine(frames.device.index or 0)
    method = max(0, min(int(method), 8))
    if method >= 7:
        raise SjpegError("trellis methods: use the host API (SjpegEncode / sjpeg::Encode)")
    f, h, w, _ = frames.shape
    assert frames.stride(3) == 1 and frames.stride(2) == 3
    rows = frames.as_strided((f, h, w * 3), (frames.stride(0), frames.stride(1), 1))
    src, _ = make_source(SRC_RGB, [rows])
    if quant is None:
        q = np.zeros((2, 64), np.uint8)
        lib().sjpeg_hip_quality_matrices(float(quality), q.ctypes.data)
    else:
        q = np.asarray(quant, np.uint8).reshape(2, 64)
    out, sizes = eng.encode_batch(src, f, w, h, yuv_mode, q, method, min_quant, q_bias, dmax_luma, dmax_chroma,
                                  device=frames.device)
    eng.wait()                                   # (pipelined mode: the output is complete after this)
    return _fetch_frames(out, sizes)


def _fetch_frames(out, sizes):
    """The batch's JPEGs as byte strings: the used part of every slot, one pinned staging buffer,
    one synchronisation."""
    import torch
    sz = sizes.cpu().numpy()
    if (sz <= 0).any():
        raise SjpegError("frame %d did not fit its output slot (the device reported size 0): raise out_stride"
                         % int(np.argmax(sz <= 0)))
    offs = np.concatenate([[0], np.cumsum(sz)]).astype(np.int64)
    stage = torch.empty(int(offs[-1]), dtype=torch.uint8, pin_memory=True)
    for k in range(len(sz)):
        stage[int(offs[k]):int(offs[k + 1])].copy_(out[k, :int(sz[k])], non_blocking=True)
    torch.cuda.synchronize()
    host = stage.numpy()
    return [host[int(offs[k]):int(offs[k + 1])].tobytes() for k in range(len(sz))]


def encode_source_method(fmt, planes, w, h, quality=75.0, yuv_mode=YUV_420, method=0, engine=None,
                         quant=None, min_quant=None, q_bias=0x78, dmax_luma=12, dmax_chroma=1):
    """One frame in any source layout (planes: CUDA uint8 tensors [1, rows, row_bytes]) with the
    reference's method 0..6 semantics, through the C-ABI.  Returns the JPEG bytes."""
    import torch
    eng = engine or Engine(planes[0].device.index or 0)
    yuv_mode = _IMPLIED_MODE.get(fmt, yuv_mode)
    method = max(0, min(int(method), 8))
    adaptive, optimize = method >= 3, method not in (0, 3)
    src, n = make_source(fmt, planes)
    assert n == 1
    tables, q = make_tables(quality=quality, quant=quant, min_quant=min_quant, q_bias=q_bias)
    if adaptive:
        hist = eng.scan_histogram_source(src, 1, w, h, yuv_mode).cpu().numpy().view(np.uint32)[0]
        tables, q = adapt_quant(hist, yuv_mode, q, min_quant, q_bias, dmax_luma, dmax_chroma)
    specs = None
    if optimize:
        freq = eng.scan_symbol_stats_source(src, 1, w, h, tables, yuv_mode).cpu().numpy().view(np.uint32)[0]
        specs = optimize_huffman(freq, yuv_mode, tables)
    header = make_header_ex(w, h, yuv_mode, q, specs)
    out, sizes = eng.encode_source(src, 1, w, h, tables, header, yuv_mode)
    return _fetch_frames(out, sizes)[0]


def encode_device(frames, quality=75.0, yuv_mode=YUV_420, engine=None, quant=None):
    """Convenience: list of JPEG byte strings for device-resident frames [F, H, W, 3]."""
    import torch
    eng = engine or Engine(frames.device.index or 0)
    tables, q = make_tables(quality=quality, quant=quant)
    f, h, w, _ = frames.shape
    header = make_header(w, h, yuv_mode, q)
    out, sizes = eng.encode_frames(frames, tables, header, yuv_mode)
    eng.wait()                                   # (pipelined mode: the output is complete after this)
    return _fetch_frames(out, sizes)
