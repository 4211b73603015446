"""One-off stress of the GPU path against the oracle: extreme qualities, noise, every mode, big and
tiny pictures, custom flat matrices (quant 1 everywhere: the longest codes the format allows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sjpeg_amd as sj
from oracle import orc, synth
o = orc.oracle()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    w, h = int(rng.choice([1, 7, 8, 16, 17, 31, 64, 100, 255, 256, 257, 640, 1000])), int(rng.choice([1, 5, 8, 16, 33, 64, 99, 128, 360]))
    k = rng.rand()
    if k < 0.45:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    elif k < 0.55:
        img = (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
    elif k < 0.6:                                  # saturated primaries in cells (pure red / blue: chroma +128)
        cell = int(rng.choice([1, 4, 8, 16]))
        idx = rng.randint(0, 2, ((h + cell - 1) // cell, (w + cell - 1) // cell, 3))
        img = (np.repeat(np.repeat(idx, cell, 0), cell, 1)[:h, :w] * 255).astype(np.uint8)
    elif k < 0.8:
        img = synth.g_struct(w, h, int(rng.randint(1 << 30)))
    else:
        img = np.full((h, w, 3), int(rng.randint(256)), np.uint8)
    mode = int(rng.choice([1, 3, 4]))
    q = float(rng.choice([0, 1, 50, 95, 99, 100]))
    m = int(rng.choice([0, 0, 1, 3, 4, 7]))
    got = sj.SjpegEncode(img, q, m, mode)
    want = o.encode_method(img, q, mode, m)
    if got != want:
        bad += 1
        print("MISMATCH", w, h, mode, q, m, None if got is None else len(got), len(want), sj.last_error())
# flat matrices of ones: maximal magnitudes and code lengths
eng = sj.Engine(0)
import torch
for it in range(12):
    w, h = int(rng.choice([64, 257, 640])), int(rng.choice([48, 99, 360]))
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    quant = np.ones((2, 64), np.uint8)
    for mode in (1, 3, 4):
        for method in (0, 4):
            got = sj.encode_device_method(torch.from_numpy(img).cuda().unsqueeze(0), 75.0, mode, method, engine=eng, quant=quant)
            got = got[0] if isinstance(got, list) else got
            want = o.encode_full(img, quant, yuv_mode=mode, method=method)
            if got != want:
                bad += 1
                print("MISMATCH ones", w, h, mode, method)
# batches with per-frame tables and headers (sjpeg_hip_*_multi): mixed content in one launch
for it in range(N // 6):
    w, h = int(rng.choice([1, 8, 17, 100, 257, 640])), int(rng.choice([1, 8, 33, 99, 360]))
    f = int(rng.randint(1, 6))
    imgs = []
    for k in range(f):
        u = rng.rand()
        if u < 0.4:
            imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        elif u < 0.7:
            imgs.append(synth.g_struct(w, h, int(rng.randint(1 << 30))))
        else:
            imgs.append(np.full((h, w, 3), int(rng.randint(256)), np.uint8))
    mode = int(rng.choice([1, 3, 4]))
    q = float(rng.choice([1, 30, 75, 95, 100]))
    m = int(rng.choice([0, 1, 3, 4, 6]))
    got = sj.encode_device_method(torch.from_numpy(np.stack(imgs)).cuda(), q, mode, m, engine=eng)
    for k in range(f):
        if got[k] != o.encode_method(imgs[k], q, mode, m):
            bad += 1
            print("MISMATCH batch", w, h, f, k, mode, q, m)
print("fuzz done, mismatches:", bad)
