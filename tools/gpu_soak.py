"""Soak of the drop-in API against the REAL reference (oracle/_ref) on this host: random sizes up
to 2200 x 2200, every method 0..8, 4:2:0 / 4:4:4 / 4:0:0 / sharp, every other source layout, odd
strides.  Usage: python tools/gpu_soak.py SEED SECONDS"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import refso, synth  # noqa: E402

r = refso.ref()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
eng = sj.Engine(0)
t_end = time.time() + budget
n = bad = 0
px = 0


def picture(w, h):
    k = rng.rand()
    if k < 0.35:
        return rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if k < 0.7:
        return synth.g_struct(w, h, int(rng.randint(1 << 30)))
    if k < 0.85:                                     # saturated blocks: what the sharp conversion is for
        img = np.zeros((h, w, 3), np.uint8)
        img[(np.arange(h)[:, None] // 3 + np.arange(w)[None, :] // 5) % 2 == 0] = (255, 0, 40)
        img[(np.arange(h)[:, None] // 7 + np.arange(w)[None, :] // 2) % 3 == 0] = (0, 250, 255)
        return img
    if k < 0.93:
        return np.full((h, w, 3), int(rng.randint(256)), np.uint8)
    cell = int(rng.choice([1, 8, 16, 64]))           # saturated primaries in cells (pure red / blue: chroma +128)
    idx = rng.randint(0, 2, ((h + cell - 1) // cell, (w + cell - 1) // cell, 3))
    return (np.repeat(np.repeat(idx, cell, 0), cell, 1)[:h, :w] * 255).astype(np.uint8)


while time.time() < t_end:
    big = rng.rand() < 0.25
    w = int(rng.randint(1, 2200 if big else 300))
    h = int(rng.randint(1, 2200 if big else 300))
    q = float(rng.choice([0, 3, 25, 50, 75, 90, 97, 100]))
    if rng.rand() < 0.7:
        img = picture(w, h)
        mode = int(rng.choice([1, 1, 3, 4, 2]))          # 2 = sharp
        method = int(rng.randint(0, 9))
        if mode == 2 and w * h > 400 * 400:
            mode = 1
        pad = int(rng.choice([0, 0, 1, 13]))
        if pad:                                          # odd row stride
            buf = np.zeros((h, 3 * w + pad), np.uint8)
            buf[:, :3 * w] = img.reshape(h, 3 * w)
            view = np.lib.stride_tricks.as_strided(buf, (h, w, 3), (buf.strides[0], 3, 1))
            got = sj.SjpegEncode(view, q, method, mode)
        else:
            got = sj.SjpegEncode(img, q, method, mode)
        want = r.encode(img, q, method, mode)
        what = ("rgb", w, h, q, method, mode, pad)
    else:
        fmt = int(rng.choice([1, 2, 3, 4, 5, 6, 7]))
        cw, ch = (w + 1) // 2, (h + 1) // 2
        shapes = {1: [(h, 4 * w)], 2: [(h, 4 * w)], 3: [(h, w)], 4: [(h, w)] * 3, 5: [(h, w), (ch, cw), (ch, cw)],
                  6: [(h, w), (ch, 2 * cw)], 7: [(h, w), (ch, 2 * cw)]}[fmt]
        planes = [rng.randint(0, 256, s).astype(np.uint8) for s in shapes]
        if rng.rand() < 0.5:
            planes = [(p // 3 + np.arange(p.shape[1])[None, :] // 2).astype(np.uint8) for p in planes]
        mode = int(rng.choice([1, 3, 4])) if fmt in (1, 2) else 1
        huff, adapt = bool(rng.randint(2)), bool(rng.randint(2))
        method = (1 if huff else 0) + (3 if adapt else 0)
        dev = [torch.from_numpy(p).cuda().unsqueeze(0) for p in planes]
        got = sj.encode_source_method(fmt, dev, w, h, q, mode, method, engine=eng)
        want = r.encode_src(fmt, planes, w, h, q, mode, huff, adapt)
        what = ("src", fmt, w, h, q, method, mode)
    n += 1
    px += w * h
    if got != want:
        bad += 1
        print("MISMATCH", what, None if got is None else len(got), len(want), sj.last_error(), flush=True)
print(f"soak: {n} encodes, {px / 1e6:.0f} Mpx, mismatches: {bad}")
