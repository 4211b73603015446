"""Fuzz with EXTREME sample values against the oracle: pictures made of the corners of the RGB cube (and their
neighbours 1 / 254) in cells, stripes and checkerboards, alone or mixed with noise -- chroma +128 (pure red / blue),
luma -128 / +127, the largest DC steps and AC magnitudes the format can carry -- through every colour mode and
method, qualities up to 100 and matrices of ones, RGB / BGRA / RGBA and planar sources with 0 / 255 planes.
(Round 3's row-pass overflow lived here: profiles/HISTORY.md.)  Usage: python tools/extremes_fuzz.py SEED SECONDS"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc  # noqa: E402

o = orc.oracle()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
eng = sj.Engine(0)
LEVELS = np.array([0, 255, 0, 255, 1, 254, 128, 127], np.uint8)


def picture(w, h):
    kind = rng.randint(6)
    cell = int(rng.choice([1, 2, 4, 8, 16, 32]))
    nlev = int(rng.choice([2, 2, 2, 4, 8]))
    gh, gw = (h + cell - 1) // cell, (w + cell - 1) // cell
    if kind == 0:                                       # random corners in cells
        idx = rng.randint(0, nlev, (gh, gw, 3))
    elif kind == 1:                                     # vertical stripes
        idx = np.broadcast_to(rng.randint(0, nlev, (1, gw, 3)), (gh, gw, 3))
    elif kind == 2:                                     # horizontal stripes
        idx = np.broadcast_to(rng.randint(0, nlev, (gh, 1, 3)), (gh, gw, 3))
    elif kind == 3:                                     # checkerboard of two colours
        a, b = rng.randint(0, nlev, 3), rng.randint(0, nlev, 3)
        m = ((np.arange(gh)[:, None] + np.arange(gw)[None, :]) & 1)[:, :, None]
        idx = np.where(m == 0, a, b)
    elif kind == 4:                                     # one solid colour
        idx = np.broadcast_to(rng.randint(0, nlev, (1, 1, 3)), (gh, gw, 3))
    else:                                               # corners with a noisy region
        idx = rng.randint(0, nlev, (gh, gw, 3))
    img = np.repeat(np.repeat(LEVELS[idx], cell, 0), cell, 1)[:h, :w].copy()
    if kind == 5:
        y0, x0 = rng.randint(h), rng.randint(w)
        img[y0:y0 + h // 2 + 1, x0:x0 + w // 2 + 1] = rng.randint(0, 256, img[y0:y0 + h // 2 + 1, x0:x0 + w // 2 + 1].shape)
    return img


t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    w, h = int(rng.randint(1, 400)), int(rng.randint(1, 300))
    if rng.rand() < 0.15:
        w, h = int(rng.choice([8, 16, 17, 64, 257, 640, 1000])), int(rng.choice([1, 2, 8, 9, 16, 64]))
    img = picture(w, h)
    q = float(rng.choice([0, 50, 75, 90, 99, 100]))
    mode = int(rng.choice([1, 1, 3, 3, 4, 2]))
    method = int(rng.randint(0, 9))
    r = rng.rand()
    if r < 0.6:
        got, want = sj.SjpegEncode(img, q, method, mode), o.encode_method(img, q, mode, method)
        what = ("host", w, h, q, mode, method)
    elif r < 0.75 and mode != 2 and method <= 6:            # matrices of ones: the longest codes
        quant = np.ones((2, 64), np.uint8)
        got = sj.encode_device_method(torch.from_numpy(img).cuda().unsqueeze(0), 75.0, mode, method, engine=eng, quant=quant)[0]
        want = o.encode_full(img, quant, yuv_mode=mode, method=method)
        what = ("ones", w, h, mode, method)
    else:                                                    # other source layouts
        fmt = int(rng.choice([1, 2, 3, 4, 5, 6, 7]))
        cw, ch = (w + 1) // 2, (h + 1) // 2
        if fmt in (1, 2):
            x = np.zeros((h, w, 4), np.uint8)
            x[..., :3] = img[..., ::-1] if fmt == 1 else img
            x[..., 3] = rng.randint(0, 256)
            planes, smode = [x.reshape(h, 4 * w)], int(rng.choice([1, 3, 4]))
        else:
            shapes = {3: [(h, w)], 4: [(h, w)] * 3, 5: [(h, w), (ch, cw), (ch, cw)], 6: [(h, w), (ch, 2 * cw)], 7: [(h, w), (ch, 2 * cw)]}[fmt]
            planes = [LEVELS[rng.randint(0, 4, s)] for s in shapes]
            smode = {3: 4, 4: 3, 5: 1, 6: 1, 7: 1}[fmt]
        m2 = int(rng.choice([0, 1, 3, 4]))
        got = sj.encode_source_method(fmt, [torch.from_numpy(p.copy()).cuda().unsqueeze(0) for p in planes], w, h, q, smode, m2, engine=eng)
        want = o.encode_src(fmt, planes, w, h, o.quality_matrices(q), yuv_mode=smode, method=m2)
        what = ("src", fmt, w, h, q, smode, m2)
    n += 1
    if got != want:
        bad += 1
        print("MISMATCH", what, None if got is None else len(got), len(want), flush=True)
print(f"extremes fuzz: {n} encodes, mismatches: {bad}")
