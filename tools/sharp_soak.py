"""Soak of the pipelined sharp-YUV sweeps (sharp_yuv.hip): batches of different pictures, sizes and batch sizes
alternate on the device for SECONDS seconds, every result against the oracle's planes computed up front --
stale lines of an earlier call's planes, a sweep that runs past its producer, a final plane picked wrong would all
show as a mismatch.   python tools/sharp_soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import orc, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o = orc.oracle()
cases = []
for (w, h, n) in ((640, 360, 9), (200, 120, 24), (1920, 64, 3), (96, 400, 11), (1280, 720, 2)):
    for variant in range(3):
        imgs = []
        for k in range(n):
            kind = (k + variant) % 4
            imgs.append(synth.g_struct(w, h, 10 * variant + k) if kind == 0 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8) if kind == 1
                        else (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8) if kind == 2 else np.full((h, w, 3), (37 * k + variant) & 255, np.uint8))
        dev = torch.from_numpy(np.stack(imgs).reshape(n, h, 3 * w)).cuda()
        cases.append((w, h, n, dev, [o.sharp_yuv(im) for im in imgs]))
t0 = time.time(); calls = bad = 0
while time.time() - t0 < secs:
    w, h, n, dev, want = cases[rng.randint(len(cases))]
    y, u, v = sj.sharp_yuv(sj.SRC_RGB, dev)
    y, u, v = y.cpu().numpy(), u.cpu().numpy(), v.cpu().numpy()
    calls += 1
    for k in range(n):
        if not (np.array_equal(y[k], want[k][0]) and np.array_equal(u[k], want[k][1]) and np.array_equal(v[k], want[k][2])):
            bad += 1
            print("MISMATCH", w, h, n, k)
print("sharp soak: %d calls, %d mismatches" % (calls, bad))
