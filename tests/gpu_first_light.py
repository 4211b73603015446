"""First-light script for the GPU box: parity vs oracle on a handful of images, prints details."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sjpeg_amd as sj
from oracle import synth, orc

o = orc.oracle()
print("devices", sj.device_count(), torch.cuda.get_device_name(0))
eng = sj.Engine(0)
bad = 0
for (w, h) in [(16, 16), (64, 48), (128, 128), (17, 13), (250, 130), (1920, 1080), (3840, 2160)]:
    for mode in (sj.YUV_420, sj.YUV_444, sj.YUV_400):
        for gen in (synth.g_struct, synth.g_noise):
            img = gen(w, h, 99 + w)
            q = 75.0
            want = o.encode(img, q, mode)
            t, qm = sj.make_tables(quality=q)
            frames = torch.from_numpy(img).cuda().unsqueeze(0).contiguous()
            # stage 1: coefficients
            zz = eng.scan_coeffs(frames, t, mode)
            torch.cuda.synchronize()
            zz_ref = o.scan_coeffs(img, qm, 0x78, mode)
            zz_gpu = zz[0].cpu().numpy()
            ncoef_bad = int((zz_gpu != zz_ref).sum())
            got = sj.encode_device(frames, q, mode, engine=eng)[0]
            ok = got == want
            if not ok or ncoef_bad:
                bad += 1
                first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
                print("MISMATCH", w, h, mode, gen.__name__, "coef_bad", ncoef_bad, "len", len(got), len(want), "first diff", first)
                if ncoef_bad:
                    idx = np.argwhere(zz_gpu != zz_ref)[:5]
                    print("   coef diffs at", idx.tolist(), zz_gpu[tuple(idx[0])], zz_ref[tuple(idx[0])])
            else:
                print("ok", w, h, mode, gen.__name__, len(got))
# host API
img = synth.g_struct(640, 480)
a = sj.SjpegEncode(img, 75, 0, sj.YUV_420)
print("host api", a is not None and a == o.encode(img, 75, sj.YUV_420), sj.last_error())
a = sj.SjpegEncode(img[::-1].copy(), 75, 0, sj.YUV_420, stride=-img.strides[0])
print("host api negative stride", a is not None and a == o.encode(img, 75, sj.YUV_420), sj.last_error())
print("BAD", bad)
