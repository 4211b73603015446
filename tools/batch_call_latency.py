"""One default-parameter batch call at a time, synchronised after each (what bench.py's other_configs report):
median of the calls.  Usage: python tools/batch_call_latency.py [m4|c5m4] [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["m4"])
cfg = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402
eng = sj.Engine(0)
F = 32
f0 = synth.g_struct(3840, 2160, 7654321)
frames = torch.from_numpy(np.stack([f0] * F)).cuda()
rows = frames.view(F, 2160, 3840 * 3)
src, _ = sj.make_source(sj.SRC_RGB, [rows])
qm = np.zeros((2, 64), np.uint8)
quant = None
if cfg == "m4":
    sj.lib().sjpeg_hip_quality_matrices(75.0, qm.ctypes.data)
else:
    import json
    d = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "digests.json")))
    s0 = np.array(d["recompress|r90|m0"]["source_quant"], np.uint8).reshape(2, 64)
    quant = np.clip((s0.astype(np.float64) * 100.0 / 90.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)
    qm[:] = quant
stride = ((3840 * 2160 * 2) // 2 + 4096 + 4095) & ~4095
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
step = lambda: eng.encode_batch(src, F, 3840, 2160, 1, qm, 4, min_quant=quant, out_stride=stride, out=out, sizes=sizes)
for _ in range(4): step()
torch.cuda.synchronize()
t = []
for _ in range(reps):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): step()
torch.cuda.synchronize()
b2b = (time.perf_counter() - t0) / reps * 1e3
print(f"{cfg}: back to back {b2b:.4f} ms  {F * 3840 * 2160 / b2b / 1e6:.1f} Gpx/s")
print(f"{cfg}: median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  {F * 3840 * 2160 / np.median(t) / 1e6:.1f} Gpx/s  bytes/frame {int(sizes[0])}")
