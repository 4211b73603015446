"""SjpegCompress() -- the reference's one-call entry (AUTO colour mode by riskiness, default parameters) -- host memory
to host memory, for a few picture sizes; the reference's own time beside it where oracle/_ref is built."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sjpeg_amd as sj
from oracle import synth, refso
r = refso.ref() if refso.available() else None
for (w, h) in ((128, 128), (640, 480), (1920, 1080), (3840, 2160)):
    for name, img in (("struct", synth.g_struct(w, h, 7)), ("noise", synth.g_noise(w, h, 7))):
        got = sj.SjpegCompress(img, 75.0)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); sj.SjpegCompress(img, 75.0); ts.append(time.perf_counter() - t0)
        line = "%4dx%-4d %-6s %8.3f ms  %7.1f Mpx/s  %8d bytes" % (w, h, name, np.median(ts) * 1e3, w * h / np.median(ts) / 1e6, len(got))
        if r is not None:
            t0 = time.perf_counter(); want = r.compress(img, 75.0); dt = time.perf_counter() - t0
            line += "  | reference %8.3f ms, equal %s" % (dt * 1e3, want == got)
        print(line)
