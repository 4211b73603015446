"""Per-phase cycle stamps of the histogram kind (16 x 4K frames, coefficients kept unless NOKEEP=1): where a segment's
time goes inside a persistent workgroup.   python tools/hist_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if not os.environ.get("NOKEEP"):
    os.environ["SJPEG_HIP_FORCE_COEF_KEEP"] = "1"
os.environ["SJPEG_HIP_STAMPS"] = os.environ.get("STAMP_MODE", "1")
import sjpeg_amd as sj
from oracle import synth
F = 16
host = [synth.g_struct(3840, 2160, 100 + k) for k in range(4)]
frames = torch.from_numpy(np.stack([host[k % 4] for k in range(F)])).cuda()
eng = sj.Engine(0)
for _ in range(3):
    eng.scan_histogram(frames, 1)
torch.cuda.synchronize()
L = sj.lib()
L.sjpeg_hip_debug_stamps.restype = C.c_size_t
L.sjpeg_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros(1 << 22, np.uint64)
n = L.sjpeg_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size)
st = buf[:n].reshape(-1, 8).astype(np.int64)
st = st[(st[:, 7] > 0) & (st[:, 0] > 0)]            # (a workgroup's last segment has no stamp 7)
d = np.diff(st, axis=1)
names = ["P1 colour", "dct + stage 0..3", "bin half 0", "dct + stage 4..7", "bin half 1", "fold", "barrier"]
print("segments", len(st), "mean cycles per segment", (st[:, 7] - st[:, 0]).mean())
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p95 {np.percentile(d[:, i], 95):9.0f}")
print("launch span", st[:, 7].max() - st[:, 0].min())
