"""ctypes loader for oracle/liboracle.so (plain-C restatement, oracle/sjpeg_oracle.c).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORC_SO = os.path.join(_HERE, "liboracle.so")
YUV_420, YUV_444, YUV_400 = 1, 3, 4
_u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


SRC_RGB, SRC_BGRA, SRC_RGBA, SRC_GRAY, SRC_YUV444, SRC_YUV420, SRC_NV12, SRC_NV21 = range(8)


class Source(C.Structure):
    """orc_source (oracle/sjpeg_oracle.h)."""
    _fields_ = [("format", C.c_int), ("plane", C.c_void_p * 3), ("stride", C.c_int * 3)]


def make_source(fmt, planes):
    """planes: list of 2-D (or 3-D packed) uint8 arrays with contiguous rows; returns
    (Source, keepalive)."""
    ps = [np.ascontiguousarray(p, np.uint8) for p in planes]
    s = Source()
    s.format = fmt
    for i, p in enumerate(ps):
        s.plane[i] = p.ctypes.data
        s.stride[i] = p.strides[0]
    return s, ps


class Quantizer(C.Structure):
    """orc_quantizer (oracle/sjpeg_oracle.h) == reference struct Quantizer minus codes_."""
    _fields_ = [("quant", C.c_uint8 * 64), ("min_quant", C.c_uint8 * 64),
                ("iquant", C.c_uint16 * 64), ("qthresh", C.c_uint16 * 64),
                ("bias", C.c_uint16 * 64)]


class Oracle:
    def __init__(self):
        if not os.path.exists(ORC_SO):
            build()
        self.lib = lib = C.CDLL(ORC_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        lib.orc_encode_rst.restype = C.c_size_t
        lib.orc_encode_rst.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                       C.POINTER(_u8p)]
        lib.orc_encode.restype = C.c_size_t
        lib.orc_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.POINTER(_u8p)]
        lib.orc_encode_matrices.restype = C.c_size_t
        lib.orc_encode_matrices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_int, C.POINTER(_u8p)]
        lib.orc_scan_bits.restype = C.c_size_t
        lib.orc_scan_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int, C.POINTER(_u8p)]
        lib.orc_scan_coeffs.restype = C.c_size_t
        lib.orc_scan_coeffs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_headers.restype = C.c_size_t
        lib.orc_headers.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.orc_free.argtypes = [C.c_void_p]
        lib.orc_fdct.argtypes = [C.c_void_p, C.c_int]
        lib.orc_get_samples.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]
        lib.orc_quality_matrices.argtypes = [C.c_float, C.c_void_p]
        lib.orc_default_codes.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_finalize_quant.argtypes = [C.POINTER(Quantizer), C.c_int]
        lib.orc_encode_method.restype = C.c_size_t
        lib.orc_encode_method.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                          C.c_int, C.POINTER(_u8p)]
        lib.orc_encode_full.restype = C.c_size_t
        lib.orc_encode_full.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p)]
        lib.orc_histogram.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.orc_encode_src.restype = C.c_size_t
        lib.orc_encode_src.argtypes = [C.POINTER(Source), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p)]
        lib.orc_encode_search.restype = C.c_size_t
        lib.orc_encode_search.argtypes = [C.POINTER(Source), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                          C.POINTER(_u8p)]
        lib.orc_histogram_src.argtypes = [C.POINTER(Source), C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.orc_symbol_stats_src.argtypes = [C.POINTER(Source), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_int, C.c_void_p]
        lib.orc_scan_coeffs_src.restype = C.c_size_t
        lib.orc_scan_coeffs_src.argtypes = [C.POINTER(Source), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p]
        lib.orc_symbol_stats.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_int, C.c_void_p]
        lib.orc_build_optimal.restype = C.c_int
        lib.orc_build_optimal.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.orc_adapt_quant.argtypes = [C.c_void_p, C.c_int, C.POINTER(Quantizer), C.c_int, C.c_int,
                                        C.c_int]

    def finalize_quant(self, quant64, min_quant64=None, q_bias=0x78) -> Quantizer:
        q = Quantizer()
        q.quant[:] = list(np.asarray(quant64, np.uint8).reshape(64))
        mq = np.ones(64, np.uint8) if min_quant64 is None else np.asarray(min_quant64, np.uint8)
        q.min_quant[:] = list(mq.reshape(64))
        self.lib.orc_finalize_quant(C.byref(q), q_bias)
        return q

    def _take(self, n, out):
        if n == 0:
            return None
        data = bytes((C.c_ubyte * n).from_address(C.addressof(out.contents)))   # (string_at: 2 GiB limit)
        self.lib.orc_free(out)
        return data

    @staticmethod
    def _img(rgb, stride):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        return rgb, rgb.shape[1], rgb.shape[0], (stride if stride is not None else rgb.strides[0])

    def quality_matrices(self, quality):
        m = np.zeros((2, 64), np.uint8)
        self.lib.orc_quality_matrices(quality, m.ctypes.data)
        return m

    def default_codes(self):
        dc = np.zeros((2, 12), np.uint32)
        ac = np.zeros((2, 256), np.uint32)
        self.lib.orc_default_codes(dc.ctypes.data, ac.ctypes.data)
        return dc, ac

    def encode(self, rgb, quality=75.0, yuv_mode=YUV_420, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        out = _u8p()
        n = self.lib.orc_encode(rgb.ctypes.data, w, h, stride, quality, yuv_mode, C.byref(out))
        return self._take(n, out)

    def encode_rst(self, rgb, quality=75.0, yuv_mode=YUV_420, ri=41, stride=None):
        """Restart-marker variant (not the reference's bytes; see sjpeg_oracle.c)."""
        rgb, w, h, stride = self._img(rgb, stride)
        out = _u8p()
        n = self.lib.orc_encode_rst(rgb.ctypes.data, w, h, stride, quality, yuv_mode, ri, C.byref(out))
        return self._take(n, out)

    def encode_method(self, rgb, quality=75.0, yuv_mode=YUV_420, method=0, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        out = _u8p()
        n = self.lib.orc_encode_method(rgb.ctypes.data, w, h, stride, quality, yuv_mode, method,
                                       C.byref(out))
        return self._take(n, out)

    def encode_full(self, rgb, quant, min_quant=None, q_bias=0x78, dmax_luma=12, dmax_chroma=1,
                    yuv_mode=YUV_420, method=0, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
        out = _u8p()
        n = self.lib.orc_encode_full(rgb.ctypes.data, w, h, stride, q.ctypes.data,
                                     mq.ctypes.data if mq is not None else None, q_bias, dmax_luma,
                                     dmax_chroma, yuv_mode, method, C.byref(out))
        return self._take(n, out)

    def encode_src(self, fmt, planes, w, h, quant, min_quant=None, q_bias=0x78, dmax_luma=12,
                   dmax_chroma=1, yuv_mode=YUV_420, method=0):
        src, keep = make_source(fmt, planes)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
        out = _u8p()
        n = self.lib.orc_encode_src(C.byref(src), w, h, q.ctypes.data,
                                    mq.ctypes.data if mq is not None else None, q_bias, dmax_luma,
                                    dmax_chroma, yuv_mode, method, C.byref(out))
        return self._take(n, out)

    def encode_search(self, fmt, planes, w, h, quant, yuv_mode=YUV_420, huffman=True, adaptive=True,
                      target_mode=1, target_value=0.0, passes=10, tolerance=1.0, qmin=0.0, qmax=100.0,
                      min_quant=None, q_bias=0x78, dmax_luma=12, dmax_chroma=1, trellis=False):
        src, keep = make_source(fmt, planes)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
        out = _u8p()
        n = self.lib.orc_encode_search(C.byref(src), w, h, q.ctypes.data,
                                       mq.ctypes.data if mq is not None else None, q_bias, dmax_luma,
                                       dmax_chroma, yuv_mode, int(huffman), int(adaptive), target_mode,
                                       C.c_float(target_value), C.c_int(passes), C.c_float(tolerance), C.c_float(qmin), C.c_float(qmax),
                                       C.c_int(int(trellis)), C.byref(out))
        return self._take(n, out)

    def sharp_yuv(self, rgb):
        """(y, u, v) planes of orc_sharp_yuv (SJPEG_YUV_SHARP conversion)."""
        rgb, w, h, stride = self._img(rgb, None)
        cw, ch = (w + 1) // 2, (h + 1) // 2
        y = np.zeros((h, w), np.uint8); u = np.zeros((ch, cw), np.uint8); v = np.zeros((ch, cw), np.uint8)
        self.lib.orc_sharp_yuv.restype = None
        self.lib.orc_sharp_yuv(C.c_void_p(rgb.ctypes.data), C.c_int(w), C.c_int(h), C.c_int(stride),
                               C.c_void_p(y.ctypes.data), C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data))
        return y, u, v

    def riskiness(self, rgb, table: bytes, stride=None):
        """(SjpegYUVMode, risk) of orc_riskiness; table = the reference's 117649-byte score table."""
        rgb, w, h, stride = self._img(rgb, stride)
        assert len(table) == 117649
        tab = np.frombuffer(table, np.uint8)
        risk = C.c_float(0)
        self.lib.orc_riskiness.restype = C.c_int
        mode = self.lib.orc_riskiness(C.c_void_p(rgb.ctypes.data), C.c_int(w), C.c_int(h), C.c_int(stride),
                                      C.c_void_p(tab.ctypes.data), C.byref(risk))
        return int(mode), float(risk.value)

    def quant_error(self, fmt, planes, w, h, quant, yuv_mode=YUV_420, q_bias=0x78):
        src, keep = make_source(fmt, planes)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        self.lib.orc_quant_error_src.restype = C.c_uint64
        return int(self.lib.orc_quant_error_src(C.byref(src), C.c_int(w), C.c_int(h), C.c_int(yuv_mode),
                                                C.c_void_p(q.ctypes.data), C.c_int(q_bias)))

    def counted_bits(self, fmt, planes, w, h, quant, yuv_mode=YUV_420, q_bias=0x78):
        src, keep = make_source(fmt, planes)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        self.lib.orc_counted_bits_src.restype = C.c_uint64
        return int(self.lib.orc_counted_bits_src(C.byref(src), C.c_int(w), C.c_int(h), C.c_int(yuv_mode),
                                                 C.c_void_p(q.ctypes.data), C.c_int(q_bias)))

    def histogram(self, rgb, yuv_mode=YUV_420, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        hist = np.zeros((2, 64, 128), np.uint32)
        self.lib.orc_histogram(rgb.ctypes.data, w, h, stride, yuv_mode, hist.ctypes.data)
        return hist

    def symbol_stats(self, rgb, quant, q_bias=0x78, yuv_mode=YUV_420, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        freq = np.zeros((2, 272), np.uint32)
        self.lib.orc_symbol_stats(rgb.ctypes.data, w, h, stride, yuv_mode, q.ctypes.data, q_bias,
                                  freq.ctypes.data)
        return freq

    def build_optimal(self, freq, size):
        f = np.ascontiguousarray(freq, np.uint32)
        bits = np.zeros(16, np.uint8)
        syms = np.zeros(256, np.uint8)
        n = self.lib.orc_build_optimal(f.ctypes.data, size, bits.ctypes.data, syms.ctypes.data)
        return bits, syms[:n].copy(), n

    def adapt_quant(self, hist, nb_comps, quant, min_quant=None, q_bias=0x78, dmax_luma=12,
                    dmax_chroma=1):
        qs = (Quantizer * 2)()
        for c in range(2):
            qs[c] = self.finalize_quant(np.asarray(quant)[c], None if min_quant is None
                                        else np.asarray(min_quant)[c], q_bias)
        h = np.ascontiguousarray(hist, np.uint32)
        self.lib.orc_adapt_quant(h.ctypes.data, nb_comps, qs, q_bias, dmax_luma, dmax_chroma)
        return np.array([list(qs[0].quant), list(qs[1].quant)], np.uint8), qs

    def encode_matrices(self, rgb, quant, min_quant=None, q_bias=0x78, yuv_mode=YUV_420,
                        stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        mq = None if min_quant is None else np.ascontiguousarray(min_quant, np.uint8).reshape(2, 64)
        out = _u8p()
        n = self.lib.orc_encode_matrices(rgb.ctypes.data, w, h, stride, q.ctypes.data,
                                         mq.ctypes.data if mq is not None else None, q_bias,
                                         yuv_mode, C.byref(out))
        return self._take(n, out)

    def scan_bits(self, rgb, quant, q_bias=0x78, yuv_mode=YUV_420, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        out = _u8p()
        n = self.lib.orc_scan_bits(rgb.ctypes.data, w, h, stride, yuv_mode, q.ctypes.data, q_bias,
                                   C.byref(out))
        return self._take(n, out)

    def scan_coeffs(self, rgb, quant, q_bias=0x78, yuv_mode=YUV_420, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        bw = 16 if yuv_mode == YUV_420 else 8
        per = {YUV_420: 6, YUV_444: 3, YUV_400: 1}[yuv_mode]
        nb = ((w + bw - 1) // bw) * ((h + bw - 1) // bw) * per
        zz = np.zeros((nb, 64), np.int16)
        n = self.lib.orc_scan_coeffs(rgb.ctypes.data, w, h, stride, yuv_mode, q.ctypes.data,
                                     q_bias, zz.ctypes.data)
        assert n == nb
        return zz

    def headers(self, w, h, yuv_mode, quant):
        q = np.ascontiguousarray(quant, np.uint8).reshape(2, 64)
        buf = np.zeros(1024, np.uint8)
        n = self.lib.orc_headers(w, h, yuv_mode, q.ctypes.data, buf.ctypes.data)
        return buf[:n].tobytes()

    def fdct(self, blocks):
        c = np.ascontiguousarray(blocks, np.int16).copy()
        self.lib.orc_fdct(c.ctypes.data, c.size // 64)
        return c

    def get_samples(self, yuv_mode, rgb, mb_x, mb_y, stride=None):
        rgb, w, h, stride = self._img(rgb, stride)
        per = {YUV_420: 6, YUV_444: 3, YUV_400: 1}[yuv_mode]
        out = np.zeros(per * 64, np.int16)
        self.lib.orc_get_samples(yuv_mode, rgb.ctypes.data, w, h, stride, mb_x, mb_y,
                                 out.ctypes.data)
        return out


_o = None


def oracle() -> Oracle:
    global _o
    if _o is None:
        _o = Oracle()
    return _o
