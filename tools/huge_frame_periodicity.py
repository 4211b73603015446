"""Finds the first segment of a huge periodic frame whose coded length breaks the picture's
period (no oracle needed): the per-segment bit counts come from sjpeg_hip_encode_band_src over
consecutive segment ranges.  Usage: python tools/huge_frame_periodicity.py W H MODE"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402

w, h, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tile = synth.g_struct(4000, 3600, 4321)
img = np.ascontiguousarray(np.tile(tile, ((h + 3599) // 3600, (w + 3999) // 4000, 1))[:h, :w])
dev = torch.from_numpy(img).cuda()
rows = dev.view(1, h, w * 3)
src, _ = sj.make_source(sj.SRC_RGB, [rows])
eng = sj.Engine(0)
t, q = sj.make_tables(quality=75.0)
nseg = sj.segment_count(w, h, mode)
px = 16 if mode == 1 else 8
mcus_per_row = (w + px - 1) // px
seg_mcus = {1: 41, 3: 85, 4: 256}[mode]
print(f"nseg {nseg}, {mcus_per_row} MCUs per MCU row, {seg_mcus} MCUs per segment", flush=True)
# band bit counts over a coarse grid of segment ranges, then compare ranges one picture-period apart
step = 1000
edges = list(range(0, nseg, step)) + [nseg]
bits = []
for b, e in zip(edges[:-1], edges[1:]):
    words, nb = eng.encode_band(src, w, h, t, mode, b, e)
    bits.append(int(nb.item()))
bits = np.array(bits, np.int64)
print("band bits, first 12 ranges:", bits[:12].tolist(), flush=True)
print("band bits per range: min", int(bits[:-1].min()), "median", int(np.median(bits[:-1])), "max", int(bits[:-1].max()))
bad = np.flatnonzero(bits[:-1] < 0.5 * np.median(bits[:-1]))
print("ranges with less than half the median bits:", bad[:10].tolist(), "... count", len(bad), flush=True)
if len(bad):
    b0 = int(bad[0]) * step
    lo, hi = max(0, b0 - step), b0 + step            # refine: single segments around the first bad range
    per = []
    for s in range(lo, min(hi, nseg)):
        _, nb = eng.encode_band(src, w, h, t, mode, s, s + 1)
        per.append(int(nb.item()))
    per = np.array(per)
    med = np.median(per[:step // 2])
    first = int(np.flatnonzero(per < 0.5 * med)[0]) + lo if (per < 0.5 * med).any() else None
    print("first short segment:", first, "-> MCU", None if first is None else first * seg_mcus,
          "MCU row", None if first is None else first * seg_mcus // mcus_per_row,
          "pixel row", None if first is None else first * seg_mcus // mcus_per_row * px)
    if first is not None:
        y = first * seg_mcus // mcus_per_row * px
        print("byte offset of that pixel row in the source:", y * w * 3, f"= {y * w * 3 / 2**31:.4f} x 2^31")
