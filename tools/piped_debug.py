import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth
eng = sj.Engine(0)
w, h, mode, q = 3840, 2160, sj.YUV_420, 75.0
frames = torch.from_numpy(synth.g_struct(w, h, 7654321)).cuda().unsqueeze(0)
tables, quant = sj.make_tables(quality=q)
header = sj.make_header(w, h, mode, quant)
stride = ((w * h * 2) // 2 + len(header) + 4095) & ~4095
out = torch.empty((1, stride), dtype=torch.uint8, device="cuda"); sizes = torch.zeros(1, dtype=torch.int64, device="cuda")
step = lambda: eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
eng.set_pipelined(True)
for _ in range(30): step()
torch.cuda.synchronize()
sys.stderr.write("==== measured loop\n")
t0 = time.perf_counter()
for _ in range(40): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("pipelined %.2f us (host %.2f)" % ((time.perf_counter() - t0) / 40 * 1e6, (t1 - t0) / 40 * 1e6))
