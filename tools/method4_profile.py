import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sjpeg_amd as sj
from oracle import synth
img = synth.g_struct(3840, 2160, 7654321)
frames = torch.from_numpy(img).cuda().unsqueeze(0)
eng = sj.Engine(0)
t, q = sj.make_tables(quality=75.0)
def timeit(f, n=50):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("histogram kind  %.3f ms" % timeit(lambda: eng.scan_histogram(frames, 1)))
print("stats kind      %.3f ms" % timeit(lambda: eng.scan_symbol_stats(frames, t, 1)))
hdr = sj.make_header(3840, 2160, 1, q)
print("encode kind     %.3f ms" % timeit(lambda: eng.encode_frames(frames, t, hdr, 1)))
hist = eng.scan_histogram(frames, 1).cpu().numpy().view(np.uint32)[0]
t0 = time.perf_counter()
for _ in range(20): sj.adapt_quant(hist, 1, q, None, 0x78, 12, 1)
print("adapt_quant host %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
freq = eng.scan_symbol_stats(frames, t, 1).cpu().numpy().view(np.uint32)[0]
t0 = time.perf_counter()
for _ in range(20): sj.optimize_huffman(freq, 1, t)
print("optimize_huffman host %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
