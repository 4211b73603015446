"""Host-API sharp-YUV encodes in a loop, for rocprofv3 / wall time: python tools/sharp_time.py [w h reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rgb = synth.g_struct(w, h, 7654321)
for mode, name in ((sj.YUV_420, "420"), (sj.YUV_SHARP, "sharp")):
    for _ in range(3):
        out = sj.SjpegEncode(rgb, 75.0, method=0, yuv_mode=mode)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = sj.SjpegEncode(rgb, 75.0, method=0, yuv_mode=mode)
    dt = (time.perf_counter() - t0) / reps
    print(f"{w}x{h} {name}: {dt * 1e3:.3f} ms  {w * h / dt / 1e9:.2f} Gpx/s  {len(out)} bytes")
