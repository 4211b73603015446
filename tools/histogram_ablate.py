"""The histogram kind as the default-parameter batch path runs it (coefficients kept: SJPEG_HIP_FORCE_COEF_KEEP), 16 x 4K
frames, + the sums of its partials on the same stream.  Arguments: SJPEG_HIP_ABLATE values, one engine each; the engines
take turns (ROUNDS rounds of 10 passes), median and minimum per engine."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.environ.get("NOKEEP"):
    os.environ["SJPEG_HIP_FORCE_COEF_KEEP"] = "1"
import sjpeg_amd as sj
from oracle import synth
n = 16
base = [synth.g_struct(3840, 2160, 100 + k) for k in range(4)]
frames = torch.from_numpy(np.stack([base[k % 4] for k in range(n)])).cuda()
codes = sys.argv[1:] or ["0"]
engs = []
for ab in codes:
    os.environ["SJPEG_HIP_ABLATE"] = ab
    engs.append(sj.Engine(0))
for eng in engs:
    for _ in range(3):
        eng.scan_histogram(frames, 1)
torch.cuda.synchronize()
times = [[] for _ in engs]
for rnd in range(int(os.environ.get("ROUNDS", "7"))):
    for k, eng in enumerate(engs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.scan_histogram(frames, 1)
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10)
for ab, t in zip(codes, times):
    print(f"ablate={ab} slots={os.environ.get('SJPEG_HIP_HISTO_SLOTS')}: histogram pass + sums median {np.median(t):.4f} min {min(t):.4f} ms per {n} frames", flush=True)
