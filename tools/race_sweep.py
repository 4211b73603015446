"""Race sweep of K1 (needs the stress build: make -C sjpeg_amd/csrc STRESS=1): for every race point
and every wave of the workgroup, that wave is held back ~50 000 cycles at that point, and a small
set of encodes is compared with the oracle.  A wave that may not lag (or whose partners may not
run ahead) without a barrier in between shows up as a mismatch.
Usage: python tools/race_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, synth  # noqa: E402

o = orc.oracle()
rng = np.random.RandomState(1)
cases = [(synth.g_struct(1913, 1071, 3), 1), (rng.randint(0, 256, (360, 640, 3)).astype(np.uint8), 3),
         (synth.g_struct(777, 333, 4), 4), (rng.randint(0, 256, (270, 480, 3)).astype(np.uint8), 1)]
want = {}
for i, (img, mode) in enumerate(cases):
    for method in (0, 4):
        want[(i, method)] = o.encode_method(img, 75.0, mode, method)
dev = [torch.from_numpy(img).cuda().unsqueeze(0) for (img, _) in cases]
bad = runs = 0
for point in range(0, 14):
    for wave in range(8):                          # 0..3: that wave lags; 4..7: that wave runs ahead of the others
        os.environ["SJPEG_HIP_ABLATE"] = str(0x5a000000 | (6 << 16) | (point << 8) | (0x80 if wave >= 4 else 0) | (wave & 3))
        eng = sj.Engine(0)
        for i, (img, mode) in enumerate(cases):
            for method in (0, 4):
                got = sj.encode_device_method(dev[i], 75.0, mode, method, engine=eng)[0]
                runs += 1
                if got != want[(i, method)]:
                    bad += 1
                    print(f"MISMATCH point {point} wave {wave} case {i} method {method}", flush=True)
        eng.close()
os.environ.pop("SJPEG_HIP_ABLATE", None)
print(f"race sweep: {runs} encodes over 14 points x 4 waves, lagging and leading, mismatches: {bad}")
