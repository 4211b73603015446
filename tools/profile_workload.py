"""One BASELINE configuration in a loop, for rocprofv3 (tools/profile_configs.sh): every kernel
instantiation gets its own trace, the aggregate per kernel name would mix them otherwise.
Usage: python tools/profile_workload.py <config> [reps]
  c2noise   16 x 4K G_noise q75 4:2:0 method 0        c3x4 / c3x1   8K G_struct q90 4:4:4 (4 frames / 1)
  c4        64 x 1080p G_struct q75 4:2:0             one4k         ONE resident 4K frame per call
  m4        32 x 4K G_struct default parameters (histogram, statistics + keep, replay)
  c5m0 / c5m4   4K recompress matrices (reduction 90), method 0 / default parameters, 16 frames"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402

cfg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = sj.Engine(0)
if os.environ.get("PIPELINED"):          # the engine's pipelined mode (K1 of call n + 1 beside the stitch of call n)
    eng.set_pipelined(True)

def plain(frames_np, q, mode, quant=None):
    F = len(frames_np)
    h, w = frames_np[0].shape[:2]
    frames = torch.stack([torch.from_numpy(f) for f in frames_np]).cuda()
    tables, fq = sj.make_tables(quality=q) if quant is None else sj.make_tables(quant=quant, min_quant=quant)
    header = sj.make_header(w, h, mode, fq)
    bpp = {1: 1.5, 3: 3.0, 4: 1.0}[mode]
    stride = (int(w * h * bpp) // 2 + len(header) + 4095) & ~4095
    out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
    step = lambda: eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
    return step, F * w * h, sizes

def batch(frames_np, mode, method, q=75.0, quant=None):
    F = len(frames_np)
    h, w = frames_np[0].shape[:2]
    frames = torch.stack([torch.from_numpy(f) for f in frames_np]).cuda()
    rows = frames.view(F, h, w * 3)
    src, _ = sj.make_source(sj.SRC_RGB, [rows])
    qm = np.zeros((2, 64), np.uint8)
    if quant is None:
        sj.lib().sjpeg_hip_quality_matrices(float(q), qm.ctypes.data)
    else:
        qm[:] = quant
    stride = (int(w * h * 1.5) // 2 + 4096 + 4095) & ~4095
    out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
    keep = (frames, rows)
    step = lambda: (keep, eng.encode_batch(src, F, w, h, mode, qm, method, min_quant=(quant if quant is not None else None),
                                           out_stride=stride, out=out, sizes=sizes))
    return step, F * w * h, sizes

def c5_quant():
    d = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "digests.json")))
    src = np.array(d["recompress|r90|m0"]["source_quant"], np.uint8).reshape(2, 64)
    return np.clip((src.astype(np.float64) * 100.0 / 90.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)

if cfg == "c2noise":
    step, px, sizes = plain([synth.g_noise(3840, 2160, 7654321 + k) for k in range(4)] * 4, 75.0, 1)
elif cfg == "c3x4":
    step, px, sizes = plain([synth.g_struct(7680, 4320, 7654321)] * 4, 90.0, 3)
elif cfg == "c3x1":
    step, px, sizes = plain([synth.g_struct(7680, 4320, 7654321)], 90.0, 3)
elif cfg == "c4":
    step, px, sizes = plain([synth.g_struct(1920, 1080, 7654321 + k) for k in range(64)], 75.0, 1)
elif cfg == "one4k":
    step, px, sizes = plain([synth.g_struct(3840, 2160, 7654321)], 75.0, 1)
elif cfg == "m4":
    step, px, sizes = batch([synth.g_struct(3840, 2160, 7654321 + k) for k in range(4)] * 8, 1, 4)
elif cfg.startswith("m4n"):                      # m4n<N>: N 4K frames, default parameters
    step, px, sizes = batch([synth.g_struct(3840, 2160, 7654321 + k % 4) for k in range(int(cfg[3:]))], 1, 4)
elif cfg == "c5m0":
    step, px, sizes = plain([synth.g_struct(3840, 2160, 7654321)] * 16, 75.0, 1, quant=c5_quant())
elif cfg == "c5m4":
    step, px, sizes = batch([synth.g_struct(3840, 2160, 7654321)] * 16, 1, 4, quant=c5_quant())
elif cfg == "c5m0b32":                          # the batch entry with method 0 (what bench.py's C5 method-0 line calls)
    step, px, sizes = batch([synth.g_struct(3840, 2160, 7654321)] * 32, 1, 0, quant=c5_quant())
elif cfg == "c5m4x32":
    step, px, sizes = batch([synth.g_struct(3840, 2160, 7654321)] * 32, 1, 4, quant=c5_quant())
else:
    raise SystemExit("unknown config " + cfg)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()

torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
assert int(sizes.min().item()) > 0
print(f"{cfg}: {dt * 1e3:.4f} ms/step  {px / dt / 1e9:.1f} Gpx/s  bytes/frame {int(sizes[0].item())}")
