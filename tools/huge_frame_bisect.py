"""Localises the >4 GiB-frame defect found by tools/huge_frame_check.py: runs each probe in its
own process (a crash in one does not hide the others) and reports the first differing byte.
Usage: python tools/huge_frame_bisect.py            (driver)
       python tools/huge_frame_bisect.py one W H MODE KIND Q   (one probe)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROBES = [  # (label, w, h, yuv_mode, content, quality)
    ("control: both under 2^32", 40000, 35000, 1, "struct", 75.0),
    ("A input bytes > 2^32, few bits", 40000, 36000, 4, "struct", 75.0),
    ("B input < 2^32, stream bits > 2^32", 31000, 31000, 1, "noise", 75.0),
    ("D the 4:4:4 case that dumped core", 36000, 40001, 3, "struct", 60.0),
]


def one(w, h, mode, kind, q):
    import numpy as np
    import sjpeg_amd as sj
    from oracle import refso, synth
    r = refso.ref()
    if kind == "struct":
        tile = synth.g_struct(4000, 3600, 4321)
    else:
        tile = np.random.RandomState(5).randint(0, 256, (3600, 4000, 3)).astype(np.uint8)
    reps = ((h + 3599) // 3600, (w + 3999) // 4000, 1)
    img = np.ascontiguousarray(np.tile(tile, reps)[:h, :w])
    print(f"  input {img.nbytes} B ({img.nbytes / 2**32:.3f} x 2^32)", flush=True)
    t0 = time.time()
    got = sj.SjpegEncode(img, q, 0, mode)
    t1 = time.time()
    if got is None:
        print("  GPU path failed:", sj.last_error(), flush=True)
        return
    want = r.encode(img, q, 0, mode)
    a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
    n = min(len(a), len(b))
    diff = np.flatnonzero(a[:n] != b[:n])
    first = int(diff[0]) if len(diff) else (n if len(a) != len(b) else -1)
    print(f"  got {len(a)} want {len(b)} B; stream bits ~{8 * len(b) / 2**32:.3f} x 2^32; "
          f"{'EQUAL' if first < 0 else f'first difference at byte {first} (= bit {8 * first}, {8 * first / 2**32:.4f} x 2^32)'}"
          f"; gpu {t1 - t0:.1f}s", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "one":
    one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], float(sys.argv[6]))
else:
    for (label, w, h, mode, kind, q) in PROBES:
        print(f"{label}: {w}x{h} mode {mode} {kind}", flush=True)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(w), str(h), str(mode), kind, str(q)],
                           timeout=420)
        if p.returncode != 0:
            print(f"  probe process ended with return code {p.returncode}"
                  f"{' (killed by signal ' + str(-p.returncode) + ')' if p.returncode < 0 else ''}", flush=True)
