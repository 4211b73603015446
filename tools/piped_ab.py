"""A/B of the engine's ordered and pipelined modes in one process (alternating, same buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sjpeg_amd as sj
from oracle import synth
W, H, F = 3840, 2160, 64
host = [synth.g_struct(W, H, 7654321 + k) for k in range(4)]
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device="cuda")
for k in range(F):
    frames[k] = torch.from_numpy(host[k % 4]).cuda()
tables, quant = sj.make_tables(quality=75.0)
header = sj.make_header(W, H, sj.YUV_420, quant)
stride = ((W * H * 3) // 2 + len(header) + 4095) & ~4095
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
eng = sj.Engine(0)
step = lambda: eng.encode_frames(frames, tables, header, sj.YUV_420, out=out, sizes=sizes, out_stride=stride)
for _ in range(5):
    step()
torch.cuda.synchronize()
for rep in range(4):
    for mode in (0, 1):
        eng.set_pipelined(bool(mode))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        print("rep %d pipelined %d: %.4f ms/step" % (rep, mode, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    eng.set_pipelined(False)
