"""bench.py under a stand-in device, for the CPU test of its multi-rank control flow (tests/test_dist_cpu.py,
VERDICT r05 #6): the first 8-GPU run of `bench.py --gpus N` must not be the first time its N > 1 code runs at all.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
           tests/bench_stub.py --gpus 2 --steps 1 --warmup 0 --regions 1 --frames 2 --no-cpu-baseline

Everything bench.py does between "import torch" and the JSON line runs as written -- process group, fences, max over
ranks, the timed regions, the packed-output exchange loop (sjpeg_amd.dist on torch.distributed), the local sinks,
config #4 sharded and gathered, the watchdog, the one line from rank 0 -- with three substitutions made HERE, not in
bench.py: "cuda" tensors are CPU tensors, the process group is gloo, and the engine is a stand-in that writes the
ORACLE's streams (so the parity checks of the line are real checks of the exchange).  The line it prints carries
`"stub": true` and is never a measurement."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Stream:
    cuda_stream = 0

    def wait_event(self, e): pass
    def wait_stream(self, s): pass
    def synchronize(self): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False


class _Event:
    def __init__(self, *a, **k): pass
    def record(self, *a): pass
    def synchronize(self): pass
    def elapsed_time(self, other): return 1.0


def _cpu_device(kw):
    if "device" in kw and kw["device"] is not None and str(kw["device"]).startswith("cuda"):
        kw["device"] = "cpu"
    return kw


def patch_torch():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    type(torch.empty(1)).is_cuda = property(lambda self: False)
    for name in ("empty", "zeros", "tensor", "full", "ones"):
        real = getattr(torch, name)
        setattr(torch, name, (lambda real: lambda *a, **k: real(*a, **_cpu_device(k)))(real))
    real_device = torch.device
    tc = torch.cuda
    tc.synchronize = lambda *a, **k: None
    tc.set_device = lambda *a, **k: None
    tc.device_count = lambda: 8
    tc.current_device = lambda: 0
    tc.empty_cache = lambda: None
    tc.is_available = lambda: True
    tc.Stream = _Stream
    tc.Event = _Event
    tc.current_stream = lambda *a, **k: _Stream()
    tc.stream = lambda s: s
    real_init = dist.init_process_group

    def init(backend=None, **kw):
        kw.pop("device_id", None)
        return real_init("gloo", **kw)
    dist.init_process_group = init
    return real_device


class StubEngine:
    """sjpeg_amd.Engine's calls that bench.py makes, on CPU tensors, coding with the oracle."""
    _cache = {}

    def __init__(self, device=0):
        from oracle import orc
        self.o = orc.oracle()

    def _code(self, frame, header, mode):
        a = frame.numpy()
        key = (a.shape, a[::7, ::5].tobytes(), mode)
        got = StubEngine._cache.get(key)
        if got is None:
            got = StubEngine._cache[key] = self.o.encode(np.ascontiguousarray(a), 75.0, mode)
        return got

    def encode_frames(self, frames, tables, header, yuv_mode, out=None, sizes=None, out_stride=None, append_eoi=True):
        for k in range(frames.shape[0]):
            c = self._code(frames[k], header, yuv_mode)
            assert len(c) <= out_stride
            out[k, :len(c)] = torch.from_numpy(np.frombuffer(c, np.uint8).copy())
            sizes[k] = len(c)
        return out, sizes

    def encode_frames_packed(self, frames, tables, header, yuv_mode, out, sizes, offsets, out_stride, append_eoi=True):
        at = 0
        for k in range(frames.shape[0]):
            c = self._code(frames[k], header, yuv_mode)
            room = (len(c) + 15) & ~15
            out[at:at + room] = 0
            out[at:at + len(c)] = torch.from_numpy(np.frombuffer(c, np.uint8).copy())
            sizes[k], offsets[k] = len(c), at
            at += room
        offsets[frames.shape[0]] = at
        return out, sizes, offsets

    def set_pipelined(self, on): pass
    def set_timing(self, on): pass
    def wait(self): pass
    def last_scan_ms(self): return 1.0
    def last_total_ms(self): return 1.0
    def scratch_bytes(self): return 0
    def close(self): pass


def main():
    patch_torch()
    import sjpeg_amd as sj
    import sjpeg_amd.dist as sd
    import bench
    sj.Engine = StubEngine
    sj.device_count = lambda: 1
    real_steps = sd.overlapped_steps
    sd.overlapped_steps = lambda n, enc, exch, use_streams, keep="all": real_steps(n, enc, exch, False, keep)
    bench.W, bench.H = 64, 48                    # (the headline frames; config #4 keeps its 1080p frames and its MD5)
    bench.device_clocks = lambda *a, **k: None
    real_emit = bench.emit

    def emit(res):
        res["stub"] = True
        res["value"] = 0.0
        real_emit(res)
    bench.emit = emit
    bench.main()


if __name__ == "__main__":
    main()
