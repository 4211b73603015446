"""Canonical synthetic inputs.  TEST/BENCH INFRASTRUCTURE.

Same LCG as the reference's unit tests use for their pictures
(/root/reference/tests/unit_test.cc:72-94): s = 1103515245*s + 12345 (mod 2^32),
R8() = (s >> 16) & 0xff.  Vectorised with numpy through the affine-map closed form
s_n = A_n*s_0 + C_n where A_n = a^n and C_n = c*(1 + a + ... + a^(n-1)), all mod 2^32
(uint32 arithmetic wraps).
"""
import hashlib
import numpy as np

_A = np.uint32(1103515245)
_C = np.uint32(12345)


def lcg_stream(seed: int, n: int) -> np.ndarray:
    """First n states s_1..s_n of the LCG started at s_0 = seed (uint32)."""
    if n == 0:
        return np.zeros(0, np.uint32)
    with np.errstate(over="ignore"):
        a_pow = np.empty(n, np.uint32)          # a^1 .. a^n
        a_pow[:] = _A
        np.multiply.accumulate(a_pow, out=a_pow)
        geo = np.empty(n, np.uint32)            # 1 + a + ... + a^(k-1), k = 1..n
        geo[0] = 1
        if n > 1:
            geo[1:] = a_pow[:-1]
        np.add.accumulate(geo, out=geo)
        return a_pow * np.uint32(seed & 0xFFFFFFFF) + _C * geo


def r8_stream(seed: int, n: int) -> np.ndarray:
    return ((lcg_stream(seed, n) >> np.uint32(16)) & np.uint32(0xFF)).astype(np.uint8)


def g_struct(w: int, h: int, seed: int = 7654321) -> np.ndarray:
    """MakeRGB() of the reference's unit tests: structured, hard-ish to compress."""
    r = r8_stream(seed, 2 * w * h).reshape(h, w, 2)
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    out = np.empty((h, w, 3), np.uint8)
    out[..., 0] = (x * 5 + (r[..., 0] >> 3)) & 0xFF
    out[..., 1] = (y * 3 + (r[..., 1] >> 4)) & 0xFF
    out[..., 2] = (((x // 8) ^ (y // 8)) * 51) & 0xFF
    return out


def g_noise(w: int, h: int, seed: int = 7654321) -> np.ndarray:
    """Every byte = R8() in memory order (entropy-coder stress)."""
    return r8_stream(seed, 3 * w * h).reshape(h, w, 3)


def md5(b) -> str:
    return hashlib.md5(bytes(b) if not isinstance(b, np.ndarray) else b.tobytes()).hexdigest()
