"""Per-phase cycle stamps of the LAST scan_segments launch of a default-parameter batch call (32 x 4K: the replay
encode of the second part): where a segment's time goes.   python tools/replay_stamps.py [stats]
(`stats`: method 1 -- the call's last launch but one... no: runs scan_symbol_stats_multi from the kept coefficients is
not reachable alone; `stats` times the statistics kind from the pixels instead.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["SJPEG_HIP_STAMPS"] = os.environ.get("STAMP_MODE", "1")
import sjpeg_amd as sj
from oracle import synth
F = 32
host = [synth.g_struct(3840, 2160, 7654321 + k) for k in range(4)]
frames = torch.from_numpy(np.stack([host[k % 4] for k in range(F)])).cuda()
f, h, w, _ = frames.shape
rows = frames.view(f, h, w * 3)
src, _ = sj.make_source(sj.SRC_RGB, [rows])
qm = np.zeros((2, 64), np.uint8)
sj.lib().sjpeg_hip_quality_matrices(75.0, qm.ctypes.data)
stride = (int(w * h * 1.5) // 2 + 4096 + 4095) & ~4095
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
eng = sj.Engine(0)
for _ in range(3):
    eng.encode_batch(src, F, w, h, 1, qm, 4, out_stride=stride, out=out, sizes=sizes)
torch.cuda.synchronize()
L = sj.lib()
L.sjpeg_hip_debug_stamps.restype = C.c_size_t
L.sjpeg_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros(1 << 22, np.uint64)
n = L.sjpeg_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size)
st = buf[:n].reshape(-1, 8).astype(np.int64)
st = st[(st[:, 7] > 0) & (st[:, 0] > 0)]
d = np.diff(st, axis=1)
names = ["load + unpack", "P2 (replay: none)", "dc + sort", "walk (code)", "scan", "stitch", "flush"]
print("segments", len(st), "mean cycles per segment", (st[:, 7] - st[:, 0]).mean())
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p95 {np.percentile(d[:, i], 95):9.0f}")
