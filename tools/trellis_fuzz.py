"""Trellis quantization (methods 7, 8) against the oracle on pictures chosen to stress the node walk: dense noise at
qualities 90-100 (up to 126 nodes a block, walks far beyond the nodes held in registers, 32-bit scores that wrap),
sparse pictures at low qualities, flat matrices of ones / of 255, every colour mode.   python tools/trellis_fuzz.py [N] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, synth  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 20260930)
o = orc.oracle()
bad = 0
for it in range(N):
    w, h = int(rng.randint(1, 160)), int(rng.randint(1, 120))
    u = rng.rand()
    if u < 0.45:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)                      # noise: every position has a level
    elif u < 0.6:
        img = (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)                # saturated noise: the largest coefficients
    elif u < 0.85:
        img = synth.g_struct(w, h, int(rng.randint(1 << 30)))
    else:
        img = np.clip(synth.g_struct(w, h, int(rng.randint(1 << 30))).astype(np.int32) + rng.randint(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)
    mode = int(rng.choice([1, 3, 4]))
    q = float(rng.choice([0, 3, 20, 50, 75, 90, 95, 98, 99, 100]))
    m = int(rng.choice([7, 8]))
    got = sj.SjpegEncode(img, q, m, mode)
    want = o.encode_method(img, q, mode, m)
    if got != want:
        bad += 1
        print("MISMATCH", w, h, mode, q, m, None if got is None else len(got), len(want), sj.last_error())
print("trellis fuzz: %d pictures, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
