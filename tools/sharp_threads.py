"""Sharp conversions of device-resident batches from several host threads at once, a stream each: every launch is sized for
three quarters of what the device holds of the strips kernel, so N threads ask for N times that -- workgroups that wait for
each other (strips of a sweep, sweeps of a picture) must still all get their turn.  Results against the single-thread ones.
  python tools/sharp_threads.py [threads] [rounds]"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shapes = [(3840, 2160, 8), (1920, 1080, 24), (640, 480, 64), (5000, 300, 6), (1280, 720, 40), (3840, 2160, 12)]
batches, want = [], []
for k in range(T):
    w, h, n = shapes[k % len(shapes)]
    fr = torch.stack([torch.from_numpy(synth.g_struct(w, h, 300 + (k * 7 + i) % 5)) for i in range(n)]).cuda().view(n, h, w * 3)
    batches.append(fr)
    want.append([t.clone() for t in sj.sharp_yuv(sj.SRC_RGB, fr)])
torch.cuda.synchronize()
bad = [0] * T
def work(k):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for r in range(R):
            y, u, v = sj.sharp_yuv(sj.SRC_RGB, batches[k])
            st.synchronize()
            if not (torch.equal(y, want[k][0]) and torch.equal(u, want[k][1]) and torch.equal(v, want[k][2])):
                bad[k] += 1
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
for t in th: t.start()
for t in th: t.join()
print("sharp threads: %d threads x %d rounds in %.2f s, mismatches %d" % (T, R, time.perf_counter() - t0, sum(bad)))
sys.exit(1 if sum(bad) else 0)
