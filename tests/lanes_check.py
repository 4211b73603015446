"""Run by tests/test_gpu_parity.py::test_batch_lanes_many_jobs in a process of its own (the knobs are read once):
SJPEG_HIP_BATCH_JOB_MPIX makes sjpeg_hip_encode_batch_src cut SMALL batches into many jobs, so that the lanes of the
batch path -- child engines, four streams, the polling state machine, several jobs per lane -- are exercised with
pictures the oracle codes in milliseconds: every frame has its own content (noise, structure, flat), every method
1..6 and colour mode, back-to-back asynchronous calls, batches with fewer frames than lanes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, synth  # noqa: E402

o = orc.oracle()
eng = sj.Engine(0)
rng = np.random.RandomState(606)
checked = 0
for (w, h, mode, nfr) in ((321, 203, 1, 23), (160, 96, 3, 9), (75, 131, 4, 14), (640, 360, 1, 5), (33, 17, 1, 3)):
    imgs = []
    for k in range(nfr):
        if k % 3 == 0:
            imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        elif k % 3 == 1:
            imgs.append(synth.g_struct(w, h, 1000 + k))
        else:
            imgs.append(np.full((h, w, 3), (37 * k) & 255, np.uint8))
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    src, _ = sj.make_source(sj.SRC_RGB, [frames.view(nfr, h, w * 3)])
    for m, q in ((1, 60.0), (3, 85.0), (4, 75.0), (6, 30.0), (2, 92.0), (5, 50.0)):
        qm = np.zeros((2, 64), np.uint8)
        sj.lib().sjpeg_hip_quality_matrices(float(q), qm.ctypes.data)
        # two calls back to back without a host wait between them, separate outputs
        calls = [eng.encode_batch(src, nfr, w, h, mode, qm, method=m) for _ in range(2)]
        torch.cuda.synchronize()
        want = [o.encode_method(imgs[k], q, mode, m) for k in range(nfr)]
        for out, sizes in calls:
            got = sj._fetch_frames(out, sizes)
            for k in range(nfr):
                assert got[k] == want[k], (w, h, mode, m, k)
                checked += 1
assert eng.scratch_bytes() > 0
eng.trim()
got = sj.encode_device_method(torch.from_numpy(np.stack([synth.g_struct(64, 48, 5)] * 6)).cuda(), 75.0, 1, 4, engine=eng)
assert got[5] == o.encode_method(synth.g_struct(64, 48, 5), 75.0, 1, 4)          # (the lanes come back after a trim)
eng.close()
print("lanes ok:", checked, "frames")
