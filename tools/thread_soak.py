"""Concurrent use of the drop-in API from many host threads (one engine / mailbox per thread behind
the scenes): every result against the real reference.  Usage: python tools/thread_soak.py [threads] [encodes]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import refso, synth  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
per = int(sys.argv[2]) if len(sys.argv) > 2 else 150
r = refso.ref()
bad = [0] * nthreads
done = [0] * nthreads


def work(t):
    rng = np.random.RandomState(1000 + t)
    for it in range(per):
        w, h = int(rng.randint(1, 700)), int(rng.randint(1, 500))
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8) if rng.rand() < 0.5 else synth.g_struct(w, h, int(rng.randint(1 << 30)))
        q = float(rng.choice([10, 50, 75, 95]))
        method = int(rng.randint(0, 9))
        mode = int(rng.choice([1, 3, 4, 2])) if w * h < 200 * 200 else int(rng.choice([1, 3, 4]))
        got = sj.SjpegEncode(img, q, method, mode)
        want = r.encode(img, q, method, mode)
        done[t] += 1
        if got != want:
            bad[t] += 1
            print("MISMATCH thread", t, w, h, q, method, mode, sj.last_error(), flush=True)


ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
for th in ths:
    th.start()
for th in ths:
    th.join()
print(f"thread soak: {nthreads} threads, {sum(done)} encodes, mismatches: {sum(bad)}")
