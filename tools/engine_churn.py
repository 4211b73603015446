"""Create / use / destroy engines in a loop (scratch, streams, events, mailbox): no crash, no growth of
device memory in use.  Usage: python tools/engine_churn.py [iterations]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
o = orc.oracle()
img = synth.g_struct(640, 360, 9)
want = o.encode(img, 75.0, 1)
want4 = o.encode_method(img, 75.0, 1, 4)
frames = torch.from_numpy(img).cuda().unsqueeze(0)
free0 = None
for it in range(n):
    eng = sj.Engine(0)
    if it % 3 == 1:
        eng.set_pipelined(True)
    assert sj.encode_device(frames, 75.0, 1, engine=eng)[0] == want
    assert sj.encode_device_method(frames, 75.0, 1, 4, engine=eng)[0] == want4
    if it % 3 == 1:
        eng.wait()
    eng.close()
    if it % 50 == 10:
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        if free0 is None:
            free0 = free
        print(f"iteration {it}: device memory in use {(total - free) / 2**20:.0f} MiB", flush=True)
torch.cuda.synchronize()
free, total = torch.cuda.mem_get_info()
print(f"done: {n} engines, in use now {(total - free) / 2**20:.0f} MiB, drift {(free0 - free) / 2**20:.1f} MiB")
