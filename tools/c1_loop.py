"""BASELINE config C1 (SjpegCompress of tests/golden/test128.rgb: AUTO -> sharp 4:2:0, method 4) repeated in one
process, every result against the known MD5.  Usage (GPU box, repo root): python tools/c1_loop.py N"""
import hashlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import sjpeg_amd as sj
img = np.fromfile("tests/golden/test128.rgb", np.uint8).reshape(128, 128, 3)
bad = 0
n = int(sys.argv[1])
for i in range(n):
    got = sj.SjpegCompress(img, 75.0)
    if got is None or hashlib.md5(got).hexdigest() != "acc8ce8111f5ff4b32b3faa15ad5d994":
        bad += 1
        print("MISMATCH at", i, None if got is None else (len(got), hashlib.md5(got).hexdigest()), sj.last_error(), flush=True)
        if bad > 5: break
print("c1 loop", n, "bad", bad)
