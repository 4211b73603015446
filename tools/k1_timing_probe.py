"""K1's duration by the engine's HIP events, 64 resident 4K frames: ordered calls in a fresh engine, the engine's
pipelined mode (the host waits for K1's end only), ordered calls again.  Is the figure the same kernel's in every mode?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth
W, H, F = 3840, 2160, 64
host = [synth.g_struct(W, H, 7654321 + k) for k in range(8)]
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device="cuda")
for k in range(F):
    frames[k] = torch.from_numpy(host[k % 8]).cuda()
tables, quant = sj.make_tables(quality=75.0)
header = sj.make_header(W, H, 1, quant)
stride = (int(W * H * 0.75) + len(header) + 4095) & ~4095
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
eng = sj.Engine(0)
enc = lambda: eng.encode_frames(frames, tables, header, 1, out=out, sizes=sizes, out_stride=stride)

def k1(n=20):
    eng.set_timing(True)
    for _ in range(3): enc()
    v = []
    for _ in range(n):
        enc(); v.append(eng.last_scan_ms())
    eng.set_timing(False)
    torch.cuda.synchronize()
    return np.mean(v), np.min(v)

def step(n=20):
    for _ in range(5): enc()
    eng.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): enc()
    eng.wait(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(20): enc()
torch.cuda.synchronize()
print("ordered, fresh: K1 %.4f (min %.4f) ms, step %.4f" % (*k1(), step()))
eng.set_pipelined(True)
print("pipelined:      K1 %.4f (min %.4f) ms, step %.4f" % (*k1(), step()))
eng.set_pipelined(False)
print("ordered again:  K1 %.4f (min %.4f) ms, step %.4f" % (*k1(), step()))
eng.set_pipelined(True)
print("pipelined:      K1 %.4f (min %.4f) ms, step %.4f" % (*k1(), step()))
