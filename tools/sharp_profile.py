"""The sharp-YUV path on its own (for rocprofv3): N host-API encodes of one 1080p picture in SJPEG_YUV_SHARP.
Usage: python tools/sharp_profile.py [reps] [method]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sjpeg_amd as sj
from oracle import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
method = int(sys.argv[2]) if len(sys.argv) > 2 else 4
img = synth.g_struct(1920, 1080, 7654321)
sj.SjpegEncode(img, 75.0, method, sj.YUV_SHARP)
t0 = time.perf_counter()
for _ in range(reps):
    sj.SjpegEncode(img, 75.0, method, sj.YUV_SHARP)
dt = (time.perf_counter() - t0) / reps
print("sharp 1080p method %d: %.3f ms per call" % (method, dt * 1e3))
t0 = time.perf_counter()
for _ in range(reps):
    sj.SjpegEncode(img, 75.0, method, sj.YUV_420)
print("4:2:0 1080p method %d: %.3f ms per call" % (method, (time.perf_counter() - t0) / reps * 1e3))
