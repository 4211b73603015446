import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth, orc
o = orc.oracle()
eng = sj.Engine(0)
for (w, h, mode, q, reps) in ((3840, 2160, sj.YUV_420, 75.0, 300), (7680, 4320, sj.YUV_444, 90.0, 50), (1920, 1080, sj.YUV_420, 75.0, 300), (640, 480, sj.YUV_420, 75.0, 300)):
    img = synth.g_struct(w, h, 7654321)
    frames = torch.from_numpy(img).cuda().unsqueeze(0)
    tables, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, mode, quant)
    bpp = 3 if mode == sj.YUV_444 else 2
    stride = ((w * h * bpp) // 2 + len(header) + 4095) & ~4095
    out = torch.empty((1, stride), dtype=torch.uint8, device="cuda"); sizes = torch.zeros(1, dtype=torch.int64, device="cuda")
    step = lambda: eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    got = bytes(out[0, :int(sizes[0])].cpu().numpy())
    ok = (got == o.encode(img, q, mode)) if w <= 3840 else None
    print("%dx%d mode %d: %.2f us per call back to back, exact %s" % (w, h, mode, dt * 1e6, ok))
