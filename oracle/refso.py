"""ctypes loader for oracle/_ref/libsjpeg_ref.so (the REAL reference, see oracle/Makefile).

TEST INFRASTRUCTURE ONLY: used by tests/, tests/golden/make_golden.py and bench.py's
cpu_baseline leg.  Loaded RTLD_LOCAL so its SjpegEncode & co. never collide with the
product library's same-named exports.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libsjpeg_ref.so")

YUV_AUTO, YUV_420, YUV_SHARP, YUV_444, YUV_400 = range(5)
_u8p = C.POINTER(C.c_uint8)


def available() -> bool:
    return os.path.exists(REF_SO)


class Ref:
    def __init__(self, path: str = REF_SO):
        self.lib = lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        lib.ref_encode.restype = C.c_size_t
        lib.ref_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_int, C.c_int, C.POINTER(_u8p)]
        lib.ref_encode_param.restype = C.c_size_t
        lib.ref_encode_param.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_float, C.c_int, C.c_int, C.POINTER(_u8p)]
        lib.ref_encode_src.restype = C.c_size_t
        lib.ref_encode_src.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                       C.c_int, C.POINTER(_u8p)]
        lib.ref_encode_search.restype = C.c_size_t
        lib.ref_encode_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                          C.c_float, C.c_float, C.c_int, C.POINTER(_u8p)]
        lib.ref_compress.restype = C.c_size_t
        lib.ref_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.POINTER(_u8p)]
        lib.ref_free.argtypes = [_u8p]
        lib.ref_get_block.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.ref_fdct.argtypes = [C.c_void_p, C.c_int]
        lib.ref_force_slow_c.argtypes = [C.c_int]
        lib.ref_quant_matrix.argtypes = [C.c_float, C.c_int, C.c_void_p]
        lib.ref_find_quantizer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.ref_find_quantizer.restype = C.c_int
        lib.ref_dimensions.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 3
        lib.ref_dimensions.restype = C.c_int
        lib.ref_estimate_quality.argtypes = [C.c_void_p, C.c_int]
        lib.ref_estimate_quality.restype = C.c_float
        lib.ref_riskiness.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        lib.ref_riskiness.restype = C.c_int

    def _take(self, n, out):
        if n == 0:
            return None
        data = bytes((C.c_ubyte * n).from_address(C.addressof(out.contents)))   # (string_at: 2 GiB limit)
        self.lib.ref_free(out)
        return data

    @staticmethod
    def _img(rgb, stride):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = rgb.shape[0], rgb.shape[1]
        return rgb, w, h, (stride if stride is not None else rgb.strides[0])

    def encode(self, rgb, quality=75.0, method=0, yuv_mode=YUV_420, stride=None):
        """SjpegEncode() of the reference (src/api.cc:32-49)."""
        rgb, w, h, stride = self._img(rgb, stride)
        out = _u8p()
        n = self.lib.ref_encode(rgb.ctypes.data, w, h, stride, quality, method, yuv_mode,
                                C.byref(out))
        return self._take(n, out)

    def encode_param(self, rgb, quality=75.0, yuv_mode=YUV_420, huffman=False, adaptive=False,
                     trellis=False, quant=None, reduction=100.0, limit_quant=False,
                     quant_bias=-1, stride=None):
        """sjpeg::Encode() with an EncoderParam (src/api.cc:183-191)."""
        rgb, w, h, stride = self._img(rgb, stride)
        q = None
        if quant is not None:
            q = np.ascontiguousarray(quant, dtype=np.uint8).reshape(2, 64)
        out = _u8p()
        n = self.lib.ref_encode_param(rgb.ctypes.data, w, h, stride, quality, yuv_mode,
                                      int(huffman), int(adaptive), int(trellis),
                                      q.ctypes.data if q is not None else None, reduction,
                                      int(limit_quant), quant_bias, C.byref(out))
        return self._take(n, out)

    def encode_src(self, fmt, planes, w, h, quality=75.0, yuv_mode=YUV_420, huffman=False,
                   adaptive=False):
        """EncodeBGRA/RGBA/Gray/YUV444/YUV420/NV12/NV21 of the reference; planes = list of 2-D
        uint8 arrays (rows contiguous), format numbering = oracle ORC_SRC_*."""
        ps = [np.ascontiguousarray(p, np.uint8) for p in planes] + [None, None]
        ptr = [p.ctypes.data if p is not None else None for p in ps[:3]]
        st = [p.strides[0] if p is not None else 0 for p in ps[:3]]
        out = _u8p()
        n = self.lib.ref_encode_src(fmt, ptr[0], ptr[1], ptr[2], st[0], st[1], st[2], w, h, quality,
                                    yuv_mode, int(huffman), int(adaptive), C.byref(out))
        return self._take(n, out)

    def encode_search(self, rgb, quality=75.0, yuv_mode=YUV_420, huffman=True, adaptive=True,
                      target_mode=1, target_value=0.0, passes=10, tolerance=1.0, qmin=0.0, qmax=100.0,
                      stride=None, trellis=False):
        rgb, w, h, stride = self._img(rgb, stride)
        out = _u8p()
        n = self.lib.ref_encode_search(rgb.ctypes.data, w, h, stride, quality, yuv_mode, int(huffman),
                                       int(adaptive), target_mode, target_value, passes, tolerance,
                                       C.c_float(qmin), C.c_float(qmax), C.c_int(int(trellis)), C.byref(out))
        return self._take(n, out)

    def encode_meta(self, rgb, quality=75.0, yuv_mode=YUV_420, app_markers=b"", exif=b"", iccp=b"", xmp=b"",
                    xmp_split_point=0):
        rgb, w, h, stride = self._img(rgb, None)
        out = _u8p()
        self.lib.ref_encode_meta.restype = C.c_size_t
        self.lib.ref_encode_meta.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                             C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p,
                                             C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(_u8p)]
        n = self.lib.ref_encode_meta(rgb.ctypes.data, w, h, stride, quality, yuv_mode,
                                     app_markers or None, len(app_markers), exif or None, len(exif),
                                     iccp or None, len(iccp), xmp or None, len(xmp), xmp_split_point,
                                     C.byref(out))
        return self._take(n, out)

    def sharpness_table(self) -> bytes:
        """The reference's trained riskiness score table (sjpeg::kSharpnessScore, src/score_7.cc), read
        out of the built reference.  Test data only: never written into the repository."""
        arr = (C.c_uint8 * 117649).in_dll(self.lib, "_ZN5sjpeg15kSharpnessScoreE")
        return bytes(arr)

    def riskiness(self, rgb):
        rgb, w, h, stride = self._img(rgb, None)
        risk = C.c_float(0)
        self.lib.ref_riskiness.restype = C.c_int
        mode = self.lib.ref_riskiness(C.c_void_p(rgb.ctypes.data), C.c_int(w), C.c_int(h), C.c_int(stride), C.byref(risk))
        return int(mode), float(risk.value)

    def compress(self, rgb, quality=75.0):
        rgb, w, h, _ = self._img(rgb, None)
        out = _u8p()
        n = self.lib.ref_compress(rgb.ctypes.data, w, h, quality, C.byref(out))
        return self._take(n, out)

    def get_block(self, yuv_mode, rgb_ptr_arr, offset, step, nblocks):
        out = np.zeros(nblocks * 64, np.int16)
        self.lib.ref_get_block(yuv_mode, rgb_ptr_arr.ctypes.data + offset, step, out.ctypes.data)
        return out

    def fdct(self, coeffs):
        c = np.ascontiguousarray(coeffs, dtype=np.int16).copy()
        self.lib.ref_fdct(c.ctypes.data, c.size // 64)
        return c

    def quant_matrix(self, quality, for_chroma):
        m = np.zeros(64, np.uint8)
        self.lib.ref_quant_matrix(quality, int(for_chroma), m.ctypes.data)
        return m

    def find_quantizer(self, jpeg: bytes):
        q = np.zeros((2, 64), np.uint8)
        n = self.lib.ref_find_quantizer(jpeg, len(jpeg), q.ctypes.data)
        return n, q


_ref = None


def ref() -> Ref:
    global _ref
    if _ref is None:
        _ref = Ref()
    return _ref
