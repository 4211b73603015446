"""Per-phase wall-clock (cycle counter) breakdown of one scan_segments launch (GPU box).
SJPEG_HIP_STAMPS=1 python tools/stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth
os.environ["SJPEG_HIP_STAMPS"] = os.environ.get("STAMP_MODE", "1")     # 2: device-wide 100 MHz clock
F = int(os.environ.get("STAMP_FRAMES", "16"))
gen = synth.g_noise if len(sys.argv) > 1 and sys.argv[1] == "noise" else synth.g_struct
host = [gen(3840, 2160, 7654321 + k) for k in range(4)]
frames = torch.empty((F, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
for k in range(F):
    frames[k] = torch.from_numpy(host[k % 4]).cuda()
t, q = sj.make_tables(quality=75)
hdr = sj.make_header(3840, 2160, 1, q)
eng = sj.Engine(0)
out_stride = 3840 * 2160 * 3 // 2
for _ in range(3):
    out, sizes = eng.encode_frames(frames, t, hdr, 1, out_stride=out_stride)
torch.cuda.synchronize()
L = sj.lib()
L.sjpeg_hip_debug_stamps.restype = C.c_size_t
L.sjpeg_hip_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros(1 << 22, np.uint64)
n = L.sjpeg_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size)
st = buf[:n].reshape(-1, 8).astype(np.int64)
d = np.diff(st, axis=1)
names = ["P1 colour", "P2 dct+quant", "dc+sort", "walk (code)", "scan", "stitch", "flush"]
print("workgroups", len(st), "lifetime mean cycles", (st[:, 7] - st[:, 0]).mean())
for i, nm in enumerate(names):
    print(f"  {nm:14s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p95 {np.percentile(d[:, i], 95):9.0f}")
span = st[:, 7].max() - st[:, 0].min()
print("launch span cycles", span)
if os.environ.get("STAMP_MODE") == "3":
    # the last stamp is the segment's bit count: stitch time (cycles) against segment length
    bits = st[:, 7]; stitch = st[:, 6] - st[:, 5]
    order = np.argsort(stitch)
    print("stitch cycles vs segment bits: slowest 12:", [(int(stitch[i]), int(bits[i]), int(i)) for i in order[-12:]])
    print("                               median ones:", [(int(stitch[i]), int(bits[i]), int(i)) for i in order[len(order) // 2 - 3:len(order) // 2 + 3]])
    print("corr(stitch, bits) = %.3f; segments over %d bits (one window): %d of %d" % (
        np.corrcoef(stitch, bits)[0, 1], 1112 * 32, int((bits > 1112 * 32).sum()), len(bits)))
    sys.exit(0)
if os.environ.get("STAMP_MODE") == "2":
    # device-wide clock, 10 ns ticks: when the workgroups of the launch start and end
    t0 = st[:, 0].min()
    starts = np.sort(st[:, 0] - t0) / 100.0; ends = np.sort(st[:, 7] - t0) / 100.0
    pc = lambda v, p: np.percentile(v, p)
    print("starts (us): p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f" % (starts[0], pc(starts, 10), pc(starts, 50), pc(starts, 90), starts[-1]))
    print("ends   (us): p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f" % (ends[0], pc(ends, 10), pc(ends, 50), pc(ends, 90), ends[-1]))
    life = (st[:, 7] - st[:, 0]) / 100.0
    print("lifetime (us): p10 %.2f p50 %.2f p90 %.2f max %.2f" % (pc(life, 10), pc(life, 50), pc(life, 90), life.max()))
