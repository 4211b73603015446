"""A batch of thousands of small frames in one call (grid.y = frames): every frame against the oracle.
Usage: python tools/many_frames_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc  # noqa: E402

o = orc.oracle()
rng = np.random.RandomState(5)
for (n, w, h, mode, method) in ((5000, 64, 48, 1, 0), (3000, 33, 17, 3, 4), (65535, 8, 8, 4, 0), (2000, 100, 60, 1, 4)):
    imgs = rng.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    imgs[::3] //= 8
    got = sj.encode_device_method(torch.from_numpy(imgs).cuda(), 70.0, mode, method)
    idx = list(range(0, n, max(1, n // 200))) + [n - 1]
    bad = sum(got[k] != o.encode_method(imgs[k], 70.0, mode, method) for k in idx)
    print(f"{n} frames {w}x{h} mode {mode} method {method}: checked {len(idx)}, mismatches {bad}", flush=True)
try:
    sj.encode_device_method(torch.zeros((65536, 8, 8, 3), dtype=torch.uint8, device="cuda"), 70.0, 1, 0)
    print("65536 frames: accepted (unexpected)")
except sj.SjpegError as e:
    print("65536 frames: rejected:", str(e)[:80])
