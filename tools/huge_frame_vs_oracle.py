"""Frames whose source exceeds 2^31 bytes: the reference addresses MCUs with 32-bit ints
(src/encoders.cc:171,207,240: signed overflow, undefined) so it cannot be the checker there; the
plain-C oracle (ptrdiff_t offsets, pinned against the reference below that size) is.
Also the largest frame still below 2^31 bytes against the real reference."""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, refso, synth  # noqa: E402


def frame(w, h):
    tile = synth.g_struct(4000, 3600, 4321)
    return np.ascontiguousarray(np.tile(tile, ((h + 3599) // 3600, (w + 3999) // 4000, 1))[:h, :w])


def md5(b):
    return hashlib.md5(b).hexdigest()[:12]


o, r = orc.oracle(), refso.ref()
# below 2^31 source bytes: all three must agree
for (w, h, mode, q) in ((26000, 27000, 1, 75.0), (26750, 26750, 3, 75.0)):
    img = frame(w, h)
    t0 = time.time(); got = sj.SjpegEncode(img, q, 0, mode); t1 = time.time()
    ref = r.encode(img, q, 0, mode); t2 = time.time()
    orc_out = o.encode(img, q, mode); t3 = time.time()
    print(f"{w}x{h} mode {mode}: source {img.nbytes / 2**31:.3f} x 2^31 B | gpu {len(got)} {md5(got)} | "
          f"reference {len(ref)} {md5(ref)} | oracle {len(orc_out)} {md5(orc_out)} | "
          f"gpu==reference {got == ref} gpu==oracle {got == orc_out} "
          f"(gpu {t1 - t0:.1f}s ref {t2 - t1:.1f}s oracle {t3 - t2:.1f}s)", flush=True)
    del img, got, ref, orc_out
# above 2^31: the oracle is the checker; the reference is shown for the record
for (w, h, mode, q) in ((40000, 35000, 1, 75.0), (36000, 40001, 3, 60.0), (40000, 36000, 4, 75.0)):
    img = frame(w, h)
    got = sj.SjpegEncode(img, q, 0, mode)
    orc_out = o.encode(img, q, mode)
    print(f"{w}x{h} mode {mode}: source {img.nbytes / 2**31:.3f} x 2^31 B | gpu {len(got)} {md5(got)} | "
          f"oracle {len(orc_out)} {md5(orc_out)} | gpu==oracle {got == orc_out}", flush=True)
    del img, got, orc_out
