"""Frames up to the reference's own limit of 65535 x 65535 (src/enc.cc:406) through the host API, against
the plain-C oracle (the reference itself addresses MCUs with 32-bit ints and is undefined beyond 2^31
source bytes -- profiles/HISTORY_r01.md).  12.9 GB of pixels for the largest: run on the GPU box only.
Usage: python tools/max_frame_check.py [WxH:mode:q[:method] ...]"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, synth  # noqa: E402


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2**20
    return 0.0


def frame(w, h):
    tile = synth.g_struct(4096, 2048, 4321)
    img = np.empty((h, w, 3), np.uint8)
    for y in range(0, h, 2048):
        rows = min(2048, h - y)
        for x in range(0, w, 4096):
            cols = min(4096, w - x)
            img[y:y + rows, x:x + cols] = tile[:rows, :cols]
        img[y:y + rows:977, ::3] ^= (y // 2048 * 37 + 0x5a) & 255       # no two bands alike
    return img


cases = sys.argv[1:] or ["40000x36000:1:75", "65535x65535:1:75"]
o = orc.oracle()
for c in cases:
    dims, mode, q, method = (c.split(":") + ["0"])[:4]
    w, h = (int(v) for v in dims.split("x"))
    mode, q, method = int(mode), float(q), int(method)
    need = 3 * w * h / 2**30
    avail = mem_available_gb()
    print(f"{c}: {need:.1f} GiB of pixels, {avail:.0f} GiB of host memory available", flush=True)
    if avail < 1.5 * need + 8:
        print("  skipped: not enough host memory", flush=True)
        continue
    img = frame(w, h)
    t0 = time.time()
    got = sj.SjpegEncode(img, q, method, mode)
    t1 = time.time()
    if got is None:
        print("  GPU path failed:", sj.last_error(), flush=True)
        continue
    print(f"  gpu {len(got)} bytes {hashlib.md5(got).hexdigest()[:12]} in {t1 - t0:.1f} s "
          f"(host buffers, copies included); cached {sj.host_trim() / 2**30:.1f} GiB released", flush=True)
    want = o.encode_method(img, q, mode, method) if method else o.encode(img, q, mode)
    t2 = time.time()
    print(f"  oracle {len(want)} bytes {hashlib.md5(want).hexdigest()[:12]} in {t2 - t1:.1f} s | "
          f"equal {got == want}", flush=True)
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        n = min(len(a), len(b))
        first, ndiff = -1, 0
        for at in range(0, n, 1 << 26):
            d = np.nonzero(a[at:at + (1 << 26)] != b[at:at + (1 << 26)])[0]
            if len(d):
                ndiff += len(d)
                if first < 0:
                    first = at + int(d[0])
                last = at + int(d[-1])
        print(f"  first differing byte {first} ({first / n:.4f} of the stream, bit {8 * first / 2**32:.4f} x 2^32), "
              f"last {last}, {ndiff} bytes differ", flush=True)
        print("   gpu   ", a[first - 8:first + 24].tobytes().hex(), flush=True)
        print("   oracle", b[first - 8:first + 24].tobytes().hex(), flush=True)
    del img, got, want
