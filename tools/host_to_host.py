"""Host buffer in, host buffer out (SjpegEncode of include/sjpeg.h, PCIe both ways): one 4K frame at a time from pageable and
from pinned memory, and a stream of frames from T threads (every thread has its own device context and stream).
  python tools/host_to_host.py [seconds per leg]"""
import ctypes as C, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import synth  # noqa: E402

def measure(budget=1.5, threads=(1, 2, 4, 8), w=3840, h=2160):
    L = sj.lib()
    img = synth.g_struct(w, h, 7654321)
    pinned_t = torch.empty((h, w, 3), dtype=torch.uint8).pin_memory()
    pinned_t.copy_(torch.from_numpy(img))
    srcs = {"pageable": img, "pinned": pinned_t.numpy()}
    def call(arr):
        out = C.POINTER(C.c_ubyte)()
        n = L.SjpegEncode(arr.ctypes.data, w, h, 3 * w, C.byref(out), C.c_float(75.0), 0, sj.YUV_420)
        assert n > 0, sj.last_error()
        L.SjpegFreeBuffer(out)
        return n
    res = {}
    for name, arr in srcs.items():
        for _ in range(3): call(arr)
        ts = []
        t_end = time.perf_counter() + budget
        while time.perf_counter() < t_end:
            t0 = time.perf_counter(); call(arr); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        res["one_frame_%s_ms" % name] = round(float(np.median(ts)), 4)
        res["one_frame_%s_gpx_s" % name] = round(w * h / np.median(ts) / 1e6, 2)
        for T in threads:
            if T == 1: continue
            counts = [0] * T
            stop = [False]
            def worker(k):
                a = arr if name == "pinned" else img.copy()      # (every thread its own pageable picture)
                call(a)
                bar.wait()
                while not stop[0]:
                    call(a); counts[k] += 1
            bar = threading.Barrier(T + 1)
            th = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
            for x in th: x.start()
            bar.wait(); t0 = time.perf_counter(); time.sleep(budget); stop[0] = True
            for x in th: x.join()
            dt = time.perf_counter() - t0
            res["stream_%s_%d_threads_gpx_s" % (name, T)] = round(sum(counts) * w * h / dt / 1e9, 2)
    return res

if __name__ == "__main__":
    r = measure(float(sys.argv[1]) if len(sys.argv) > 1 else 1.5)
    for k, v in r.items(): print("%-40s %s" % (k, v))
