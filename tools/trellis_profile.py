"""The trellis path on its own (for rocprofv3): N host-API encodes of one 1080p picture with method 7."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj
from oracle import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
img = synth.g_struct(1920, 1080, 7654321)
sj.SjpegEncode(img, 75.0, 7, sj.YUV_420)
t0 = time.perf_counter()
for _ in range(reps):
    sj.SjpegEncode(img, 75.0, 7, sj.YUV_420)
print("trellis 1080p method 7: %.3f ms per call" % ((time.perf_counter() - t0) / reps * 1e3))
