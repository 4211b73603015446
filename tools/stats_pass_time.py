"""The statistics pass (kKindStats) alone: 16 x 4K frames, standard tables.  SJPEG_HIP_ABLATE=9 replaces its
symbol-count atomics by plain stores (wrong counts): what their serialisation costs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj
from oracle import synth
n = 16
base = [synth.g_struct(3840, 2160, 100 + k) for k in range(4)]
frames = torch.from_numpy(np.stack([base[k % 4] for k in range(n)])).cuda()
f, h, w, _ = frames.shape
src, _ = sj.make_source(sj.SRC_RGB, [frames.view(f, h, w * 3)])
eng = sj.Engine(0)
t, q = sj.make_tables(quality=75.0)
tabs = [t] * f
for _ in range(3):
    eng.scan_symbol_stats_multi(src, f, w, h, tabs, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    eng.scan_symbol_stats_multi(src, f, w, h, tabs, 1)
e1.record(); torch.cuda.synchronize()
print(f"ablate={os.environ.get('SJPEG_HIP_ABLATE')}: statistics pass {e0.elapsed_time(e1)/10:.3f} ms per {n} frames")
