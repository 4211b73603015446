import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sjpeg_amd as sj
from oracle import orc, synth
o = orc.oracle()
for (w, h, mode, q, m) in ((16384, 8192, 1, 75.0, 0), (30001, 3001, 3, 85.0, 4), (65535, 33, 4, 60.0, 1),
                           # round 6: the sharp conversion in 63 and in 171 strips (mode 2), the trellis walk on a large picture
                           (12001, 3001, 2, 75.0, 0), (65535, 70, 2, 80.0, 4), (8192, 4096, 1, 75.0, 7)):
    img = synth.g_struct(w, h, 99)
    t0 = time.time(); got = sj.SjpegEncode(img, q, m, mode); t1 = time.time()
    want = o.encode_method(img, q, mode, m); t2 = time.time()
    print(w, h, mode, q, m, "equal" if got == want else "MISMATCH", len(want), f"gpu {t1-t0:.2f}s cpu-oracle {t2-t1:.2f}s", sj.last_error() if got is None else "")
