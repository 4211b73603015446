"""One resident 4K frame per call, ordered and in the engine's pipelined mode, back to back (GPU box).
  python tools/one_frame_piped.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = sj.Engine(0)
w, h, mode, q = 3840, 2160, sj.YUV_420, 75.0
frames = torch.from_numpy(synth.g_struct(w, h, 7654321)).cuda().unsqueeze(0)
tables, quant = sj.make_tables(quality=q)
header = sj.make_header(w, h, mode, quant)
stride = ((w * h * 2) // 2 + len(header) + 4095) & ~4095
out = torch.empty((1, stride), dtype=torch.uint8, device="cuda"); sizes = torch.zeros(1, dtype=torch.int64, device="cuda")
step = lambda: eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
res = []
for piped in (False, True, False, True):
    eng.set_pipelined(piped)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    t1 = time.perf_counter()                     # every call has returned: what the HOST needs per call
    torch.cuda.synchronize()
    res.append("%s %.2f us (host %.2f)" % ("pipelined" if piped else "ordered", (time.perf_counter() - t0) / reps * 1e6, (t1 - t0) / reps * 1e6))
eng.set_pipelined(False)
print("  ".join(res))
