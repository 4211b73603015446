"""Batched default-parameter (method 4) encode of 4K frames: three launches over the batch +
per-frame host analysis.  Prints wall time per frame and the device share.
Usage: python tools/method4_batch_bench.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
base = [synth.g_struct(3840, 2160, 100 + k) for k in range(4)]
frames = torch.from_numpy(np.stack([base[k % 4] for k in range(n)])).cuda()
eng = sj.Engine(0)
for method in (4, 1, 3, 0):
    sj.encode_device_method(frames[:2], 75.0, 1, method, engine=eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sj.encode_device_method(frames, 75.0, 1, method, engine=eng)
    t1 = time.perf_counter()
    px = n * 3840 * 2160
    print(f"method {method}: {n} frames  {1e3 * (t1 - t0) / n:.3f} ms/frame wall (incl. D2H of the JPEGs) "
          f"{px / (t1 - t0) / 1e9:.1f} Gpx/s  bytes/frame {sum(map(len, out)) // n}")
# the same without fetching the JPEGs to the host: sjpeg_hip_encode_batch_src, output resident in HBM
f, h, w, _ = frames.shape
rows = frames.view(f, h, w * 3)
src, _ = sj.make_source(sj.SRC_RGB, [rows])
q = np.zeros((2, 64), np.uint8)
sj.lib().sjpeg_hip_quality_matrices(75.0, q.ctypes.data)
stride = sj.frame_bound(w, h, 1, 2048)
out_buf = torch.empty((f, stride), dtype=torch.uint8, device="cuda")   # one output buffer for all calls
sizes_buf = torch.zeros(f, dtype=torch.int64, device="cuda")
for method in (4, 1, 3, 0):
    for _ in range(2):                            # scratch allocation happens in the first call
        eng.encode_batch(src, f, w, h, 1, q, method, out_stride=stride, out=out_buf, sizes=sizes_buf)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.encode_batch(src, f, w, h, 1, q, method, out_stride=stride, out=out_buf, sizes=sizes_buf)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"method {method}: encode_batch, output in HBM  {1e3 * (t1 - t0) / 3 / n:.3f} ms/frame  "
          f"{3 * n * 3840 * 2160 / (t1 - t0) / 1e9:.1f} Gpx/s")
# device-only share of method 4: the three launches back to back
t, q = sj.make_tables(quality=75.0)
tabs = [t] * f
hdrs = [sj.make_header(w, h, 1, q)] * f
torch.cuda.synchronize()
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.scan_histogram(frames, 1)
    eng.scan_symbol_stats_multi(src, f, w, h, tabs, 1)
    eng.encode_source_multi(src, f, w, h, tabs, hdrs, 1)
    e1.record()
    torch.cuda.synchronize()
print(f"device passes only: {e0.elapsed_time(e1) / n:.3f} ms/frame")
