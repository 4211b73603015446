"""Which call of a process is the slow one ("a 6-8 ms call some tens of calls in")?  N default-parameter
batch calls, each synchronised and timed; prints every call over 2 x the median with its index, and -- with
SJPEG_HIP_BATCH_DEBUG=1 in the environment -- the library's own host timeline of those calls goes to stderr.
  python tools/slow_call_hunt.py [calls] [frames]          (under rocprofv3 --hip-trace --kernel-trace: the API call that took the time)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = sj.Engine(0)
f0 = synth.g_struct(3840, 2160, 7654321)
frames = torch.from_numpy(np.stack([f0] * F)).cuda()
rows = frames.view(F, 2160, 3840 * 3)
src, _ = sj.make_source(sj.SRC_RGB, [rows])
qm = np.zeros((2, 64), np.uint8)
sj.lib().sjpeg_hip_quality_matrices(75.0, qm.ctypes.data)
stride = ((3840 * 2160 * 2) // 2 + 4096 + 4095) & ~4095
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
step = lambda: eng.encode_batch(src, F, 3840, 2160, 1, qm, 4, out_stride=stride, out=out, sizes=sizes)
torch.cuda.synchronize()
t = []
for i in range(N):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t0))
t = np.array(t) * 1e3
med = np.median(t[:, 1])
print("calls %d x %d frames: median %.3f ms (call returns after %.3f), p99 %.3f, max %.3f" % (N, F, med, np.median(t[:, 0]), np.percentile(t[:, 1], 99), t[:, 1].max()))
slow = [(i, t[i, 0], t[i, 1]) for i in range(N) if t[i, 1] > 2 * med]
print("calls over 2 x median (index, ms until the call returned, ms until the device was done):")
for i, a, b in slow:
    print("  #%d  %.3f  %.3f" % (i, a, b))
