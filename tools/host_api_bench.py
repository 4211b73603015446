"""Throughput of the drop-in host API (SjpegEncode on host buffers, PCIe included), GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sjpeg_amd as sj
from oracle import synth
img = synth.g_struct(3840, 2160, 7654321)
for method, mode, name in ((0, sj.YUV_420, "m0 420"), (4, sj.YUV_420, "m4 420"), (0, sj.YUV_444, "m0 444"), (7, sj.YUV_420, "m7 420 (trellis)"), (0, sj.YUV_SHARP, "m0 sharp")):
    out = sj.SjpegEncode(img, 75.0, method, mode)
    assert out is not None, sj.last_error()
    n, t0 = 0, time.perf_counter()
    while True:
        sj.SjpegEncode(img, 75.0, method, mode)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 2.0 or n >= 200:
            break
    print(f"{name:18s} {dt / n * 1e3:8.2f} ms/frame  {n * 3840 * 2160 / dt / 1e6:9.1f} Mpx/s  ({len(out)} bytes)")
