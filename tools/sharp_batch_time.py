"""The sharp conversion of a device-resident batch, ms per call: python tools/sharp_batch_time.py  (SJPEG_HIP_SHARP_STRIPS=0: the kernel before)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402
for w, h, n in ((640, 480, 64), (1920, 1080, 16), (1920, 1080, 64), (3840, 2160, 8), (3840, 2160, 32)):
    fr = torch.stack([torch.from_numpy(synth.g_struct(w, h, 100 + k % 4)) for k in range(n)]).cuda().view(n, h, w * 3)
    for _ in range(2):
        y = sj.sharp_yuv(sj.SRC_RGB, fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = sj.sharp_yuv(sj.SRC_RGB, fr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("batch %dx%d x%d: %.3f ms per call, %.2f Gpx/s" % (w, h, n, dt * 1e3, w * h * n / dt / 1e9))
