import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj
from oracle import synth
n = 16
base = [synth.g_struct(3840, 2160, 100 + k) for k in range(4)]
frames = torch.from_numpy(np.stack([base[k % 4] for k in range(n)])).cuda()
eng = sj.Engine(0)
for _ in range(3):
    eng.scan_histogram(frames, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng.scan_histogram(frames, 1)
e1.record(); torch.cuda.synchronize()
print(f"ablate={os.environ.get('SJPEG_HIP_ABLATE')}: histogram pass {e0.elapsed_time(e1)/10:.3f} ms per {n} frames")
