"""Fuzz of the bit-level stitch with LOW-ENTROPY pictures against the oracle: flat pictures, flat with a few
specks, stripes, soft gradients -- streams of very short codes, segments of a few dozen bits, last segments
that start no word of their own, frames that end inside an earlier segment's last word (the case
profiles/HISTORY.md, round 3, describes).  Every size from 1 to 700, every method and colour mode, the host
API and the batch entry (1-3 frames, ordered and pipelined engine).  Usage: python tools/low_entropy_fuzz.py SEED SECONDS"""
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
eng = [sj.Engine(0), sj.Engine(0)]
eng[1].set_pipelined(True)
t_end = time.time() + budget
n = bad = 0


def picture(w, h):
    k = rng.randint(5)
    base = rng.randint(0, 256, 3)
    img = np.empty((h, w, 3), np.uint8)
    img[:] = base
    if k == 1:                                           # a few specks
        for _ in range(int(rng.randint(1, 6))):
            img[rng.randint(h), rng.randint(w)] = rng.randint(0, 256, 3)
    elif k == 2:                                         # stripes of two levels
        p = int(rng.choice([2, 3, 8, 16, 50]))
        img[:, (np.arange(w) // p) % 2 == 1] = rng.randint(0, 256, 3)
    elif k == 3:                                         # soft gradient
        g = (np.arange(w)[None, :] * int(rng.randint(1, 4)) // 8 + np.arange(h)[:, None] // int(rng.randint(4, 40))) & 255
        img[:] = g[:, :, None].astype(np.uint8)
    elif k == 4:                                         # one noisy corner
        ch, cw = max(1, h // int(rng.randint(2, 9))), max(1, w // int(rng.randint(2, 9)))
        img[:ch, :cw] = rng.randint(0, 256, (ch, cw, 3))
    return img


while time.time() < t_end:
    big = rng.rand() < 0.1
    w = int(rng.randint(1, 2600 if big else 700))
    h = int(rng.randint(1, 1200 if big else 700))
    q = float(rng.choice([0, 10, 50, 75, 90, 100]))
    mode = int(rng.choice([1, 1, 3, 4, 2]))
    if mode == 2 and w * h > 300 * 300:
        mode = 1
    method = int(rng.randint(0, 9))
    img = picture(w, h)
    want = o.encode_method(img, q, mode, method)
    got = sj.SjpegEncode(img, q, method, mode)
    n += 1
    if got != want:
        bad += 1
        print("MISMATCH host", w, h, q, mode, method, None if got is None else len(got), len(want), flush=True)
    if method <= 6 and mode != 2:
        f = int(rng.randint(1, 4))
        imgs = [img] + [picture(w, h) for _ in range(f - 1)]
        e = eng[int(rng.randint(2))]
        gots = sj.encode_device_method(torch.from_numpy(np.stack(imgs)).cuda(), q, mode, method, engine=e)
        for k in range(f):
            n += 1
            if gots[k] != (want if k == 0 else o.encode_method(imgs[k], q, mode, method)):
                bad += 1
                print("MISMATCH batch", w, h, q, mode, method, f, k, flush=True)
print(f"low-entropy fuzz: {n} encodes, mismatches: {bad}")
