"""First probe of frames beyond 2^32 bytes of pixels.  KEPT FOR THE RECORD, NOT A VALID CHECK: it
compares with the real reference, which addresses MCUs with 32-bit ints (src/encoders.cc:171,207,
240) and is undefined once the source offset passes 2^31 -- its "MISMATCH" lines are the
reference overflowing, see profiles/HISTORY_r01.md.  Use tools/huge_frame_vs_oracle.py instead.
Usage: python tools/huge_frame_check.py [max]"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import refso, synth  # noqa: E402

r = refso.ref()
cases = [(40000, 36000, 1, 75.0, 0), (36000, 40001, 3, 60.0, 0), (65535, 23000, 1, 75.0, 4)]
if len(sys.argv) > 1 and sys.argv[1] == "max":
    cases.append((65535, 65535, 1, 75.0, 0))
for (w, h, mode, q, m) in cases:
    tile = synth.g_struct(4000, 3600, 4321)
    reps = ((h + 3599) // 3600, (w + 3999) // 4000, 1)
    img = np.ascontiguousarray(np.tile(tile, reps)[:h, :w])
    img[::977, ::3] ^= 0x5a                      # break the exact periodicity a little
    t0 = time.time()
    got = sj.SjpegEncode(img, q, m, mode)
    t1 = time.time()
    want = r.encode(img, q, m, mode)
    t2 = time.time()
    ok = got is not None and len(got) == len(want) and hashlib.md5(got).digest() == hashlib.md5(want).digest()
    print(w, h, mode, q, m, f"{img.nbytes / 2**30:.1f} GiB in", "equal" if ok else "MISMATCH",
          None if got is None else len(got), len(want), f"gpu {t1 - t0:.1f}s ref {t2 - t1:.1f}s",
          "" if got is not None else sj.last_error(), flush=True)
    del img, got, want
