"""Latency of the drop-in host API on small pictures (thumbnails), beside the reference on this host.
Usage: python tools/small_image_latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import refso, synth  # noqa: E402

r = refso.ref()
for (w, h) in ((64, 64), (128, 128), (256, 256), (640, 480), (1280, 720), (1920, 1080)):
    img = synth.g_struct(w, h, 11)
    for method in (0, 4):
        for _ in range(5):
            got = sj.SjpegEncode(img, 75.0, method, 1)
        n = 200 if w <= 640 else 50
        t0 = time.perf_counter()
        for _ in range(n):
            got = sj.SjpegEncode(img, 75.0, method, 1)
        t1 = time.perf_counter()
        for _ in range(3):
            want = r.encode(img, 75.0, method, 1)
        t2 = time.perf_counter()
        for _ in range(n):
            want = r.encode(img, 75.0, method, 1)
        t3 = time.perf_counter()
        print(f"{w}x{h} method {method}: gpu host API {1e6 * (t1 - t0) / n:8.1f} us   reference CPU {1e6 * (t3 - t2) / n:8.1f} us   "
              f"{'equal' if got == want else 'MISMATCH'}", flush=True)
