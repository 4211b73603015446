"""Stress of sjpeg_hip_encode_batch_src against the oracle: batches of 1 .. 40 frames (24 and more are coded in two
parts), mixed content in one batch (noise beside flat and structured pictures: narrow and wide kept blocks, every
statistics kind), every analysis method, every colour mode, qualities 1 .. 100.  Every frame is compared with the
reference's single-picture encode.  Usage: python tools/batch_fuzz.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sjpeg_amd as sj
from oracle import orc, synth
o = orc.oracle()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
eng = sj.Engine(0)
bad = batches = frames_done = 0
t_end = time.time() + seconds
while time.time() < t_end:
    w, h = int(rng.choice([8, 17, 64, 100, 257, 330])), int(rng.choice([8, 33, 48, 99, 200]))
    f = int(rng.choice([1, 2, 5, 16, 23, 24, 25, 31, 40]))
    imgs = []
    for k in range(f):
        u = rng.rand()
        if u < 0.35:
            imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        elif u < 0.7:
            imgs.append(synth.g_struct(w, h, int(rng.randint(1 << 30))))
        elif u < 0.85:
            img = synth.g_struct(w, h, int(rng.randint(1 << 30)))
            img[: h // 2] = rng.randint(0, 256, (h // 2, w, 3))
            imgs.append(img)
        else:
            imgs.append(np.full((h, w, 3), int(rng.randint(256)), np.uint8))
    mode = int(rng.choice([1, 3, 4]))
    q = float(rng.choice([1, 30, 60, 75, 90, 97, 100]))
    m = int(rng.choice([1, 2, 3, 4, 4, 5, 6]))
    got = sj.encode_device_method(torch.from_numpy(np.stack(imgs)).cuda(), q, mode, m, engine=eng)
    for k in range(f):
        if got[k] != o.encode_method(imgs[k], q, mode, m):
            bad += 1
            print("MISMATCH", w, h, f, k, mode, q, m, flush=True)
    batches += 1
    frames_done += f
print(f"batch fuzz: {batches} batches, {frames_done} frames, mismatches: {bad}")
