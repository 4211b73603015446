"""Race sweep (needs the stress build: make -C sjpeg_amd/csrc STRESS=1): for every race point and
every wave of the workgroup, that wave is held back ~50 000 cycles at that point, and a small set
of encodes is compared with the oracle.  A wave that may not lag (or whose partners may not run
ahead) without a barrier in between shows up as a mismatch.  Points 0..25: K1 (encode, histogram,
statistics, replay kinds); points 32..47: the sharp-YUV sweeps (sharp_yuv.hip), driven through the
host API on BASELINE config C1 (SjpegCompress of test128.rgb: AUTO -> sharp, method 4), a small
sharp picture and one wide enough for the general sweep kernel.  The kernels without any
intra-workgroup hand-over (risk_scan, adapt_sums_kernel, reduce_partials: per-thread work + atomics)
have nothing to hold.
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
# (the persistent histogram kind: three workgroups in all, so that every workgroup walks many segments -- the barriers
# at the end of a segment, points 22 / 23, are only met from the second one on)
os.environ["SJPEG_HIP_HISTO_SLOTS"] = "3"
for point in range(0, 26):
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
print(f"race sweep K1: {runs} encodes over 26 points x 4 waves, lagging and leading, mismatches: {bad}")

# ---- the sharp-YUV sweeps, through the host API (the stress code is read per call there)
import hashlib  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
img128 = np.fromfile(os.path.join(ROOT, "tests", "golden", "test128.rgb"), np.uint8).reshape(128, 128, 3)
small = synth.g_struct(801, 203, 9)               # three strips of chroma columns that meet three times (sharp_sweeps_strips)
wide = synth.g_struct(4242, 10, 11)               # chroma rows of 2121 > 2048 columns: sharp_sweeps
want_small = o.encode_method(small, 80.0, 2, 4)
want_wide = o.encode_method(wide, 75.0, 2, 0)
have_table = os.path.exists(os.path.join(sj.CSRC, "riskiness.bin"))
sbad = sruns = 0
for point in list(range(32, 37)) + list(range(40, 48)):
    for wave in range(8):
        os.environ["SJPEG_HIP_ABLATE"] = str(0x5a000000 | (6 << 16) | (point << 8) | (0x80 if wave >= 4 else 0) | (wave & 3))
        checks = [("small", sj.SjpegEncode(small, 80.0, 4, 2), want_small), ("wide", sj.SjpegEncode(wide, 75.0, 0, 2), want_wide)]
        if have_table:
            got = sj.SjpegCompress(img128, 75.0)
            checks.append(("c1", None if got is None else hashlib.md5(got).hexdigest(), "acc8ce8111f5ff4b32b3faa15ad5d994"))
        for name, got, want_ in checks:
            sruns += 1
            if got != want_:
                sbad += 1
                print(f"MISMATCH sharp point {point} wave {wave} case {name}: {sj.last_error()}", flush=True)
os.environ.pop("SJPEG_HIP_ABLATE", None)
print(f"race sweep sharp: {sruns} encodes over 13 points x 4 wave classes, lagging and leading, mismatches: {sbad}")
