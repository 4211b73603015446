"""K1 duration of a full encode against a replay encode (quantized blocks kept by a statistics
pass: no colour conversion, DCT or quantization), 64 4K frames.  Usage: python tools/replay_vs_full.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402

n = 64
base = [synth.g_struct(3840, 2160, 7654321 + k) for k in range(8)]
frames = torch.from_numpy(np.stack([base[k % 8] for k in range(n)])).cuda()
f, h, w, _ = frames.shape
src, _ = sj.make_source(sj.SRC_RGB, [frames.view(f, h, w * 3)])
eng = sj.Engine(0)
t, q = sj.make_tables(quality=75.0)
hdr = sj.make_header(w, h, 1, q)
stride = ((w * h * 3) // 2 + len(hdr) + 4095) & ~4095
eng.set_timing(True)
full = []
for _ in range(6):
    out, sizes = eng.encode_source(src, f, w, h, t, hdr, 1, out_stride=stride)
    full.append(eng.last_scan_ms())
ref = bytes(out[0, :int(sizes[0])].cpu().numpy())
t.flags = sj.QUANT_KEEP
eng.scan_symbol_stats_source(src, f, w, h, t, 1)
t.flags = sj.QUANT_REPLAY
rep = []
for _ in range(6):
    out, sizes = eng.encode_source(src, f, w, h, t, hdr, 1, out_stride=stride)
    rep.append(eng.last_scan_ms())
same = bytes(out[0, :int(sizes[0])].cpu().numpy()) == ref
print(f"K1 full {np.mean(full[1:]):.3f} ms   K1 replay {np.mean(rep[1:]):.3f} ms   same bytes: {same}")
