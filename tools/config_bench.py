"""Throughput of the other BASELINE.json configurations (device-resident, one GPU): C3 8K 4:4:4 q90,
C4 64 x 1080p q75 4:2:0, C2 with noise input, C5 recompress matrices (default params = method 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth

eng = sj.Engine(0)

def run(name, frames_np, q, mode, reps=20):
    F = len(frames_np)
    h, w = frames_np[0].shape[:2]
    frames = torch.stack([torch.from_numpy(f) for f in frames_np]).cuda()
    tables, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, mode, quant)
    stride = (sj.frame_bound(w, h, mode, len(header)) // 8 + 4095) & ~4095     # plenty for these pictures
    out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
    for _ in range(3):
        eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert int(sizes.min().item()) > 0
    print(f"{name:34s} {dt * 1e3:8.3f} ms/step  {F * w * h / dt / 1e6:10.1f} Mpx/s  ({int(sizes[0].item())} bytes/frame)")

run("C2 4K G_noise q75 420 x16", [synth.g_noise(3840, 2160, 7654321 + k) for k in range(4)] * 4, 75.0, sj.YUV_420)
run("C3 8K G_struct q90 444 x4", [synth.g_struct(7680, 4320, 7654321)] * 4, 90.0, sj.YUV_444, reps=10)
run("C3 8K G_struct q90 444 x1", [synth.g_struct(7680, 4320, 7654321)], 90.0, sj.YUV_444, reps=20)
run("C4 64 x 1080p G_struct q75 420", [synth.g_struct(1920, 1080, 7654321 + k) for k in range(64)], 75.0, sj.YUV_420)
run("C2 4K G_struct q75 400 x16", [synth.g_struct(3840, 2160, 7654321 + k) for k in range(4)] * 4, 75.0, sj.YUV_400)
