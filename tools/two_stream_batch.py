"""Do the passes of the default-parameter path overlap usefully?  Two host threads, an engine and a stream each, code
16 4K frames per call concurrently; against one thread coding 32 frames per call (two parts).  Gpixels/s of the process."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth

W, H = 3840, 2160
pics = [synth.g_struct(W, H, 7654321 + k) for k in range(4)]
qm = np.zeros((2, 64), np.uint8)
sj.lib().sjpeg_hip_quality_matrices(75.0, qm.ctypes.data)
stride = (W * H * 3 // 4 + 4096 + 4095) & ~4095


def worker(n, reps, barrier, out_t):
    eng = sj.Engine(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        frames = torch.from_numpy(np.stack([pics[k % 4] for k in range(n)])).cuda()
        src, _ = sj.make_source(sj.SRC_RGB, [frames.view(n, H, W * 3)])
        out = torch.empty((n, stride), dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(n, dtype=torch.int64, device="cuda")
        for _ in range(4):
            eng.encode_batch(src, n, W, H, 1, qm, 4, out_stride=stride, out=out, sizes=sizes)
        st.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.encode_batch(src, n, W, H, 1, qm, 4, out_stride=stride, out=out, sizes=sizes)
        st.synchronize()
        out_t.append(time.perf_counter() - t0)
        assert int(sizes.min().item()) > 0


def run(nthreads, n, reps=30):
    barrier = threading.Barrier(nthreads)
    ts = []
    th = [threading.Thread(target=worker, args=(n, reps, barrier, ts)) for _ in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = max(ts)
    print(f"{nthreads} thread(s) x {n} frames per call: {dt / reps * 1e3:.3f} ms per round of {nthreads * n} frames, "
          f"{nthreads * n * W * H * reps / dt / 1e9:.1f} Gpx/s", flush=True)


for _ in range(3):
    for (t, n) in ((1, 32), (2, 16), (3, 11), (4, 8), (8, 4), (1, 16), (2, 8), (4, 4), (1, 64), (4, 16), (8, 8)):
        run(t, n, reps=20)
