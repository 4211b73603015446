#!/usr/bin/env python3
"""bench.py -- headline benchmark (BASELINE.json): Mpixels/s of baseline-JPEG encode,
3840x2160 RGB, q=75, YUV 4:2:0, method 0, bit-exact vs the reference.

A "step" = one pass of the hot path (sjpeg_hip_encode_scan: colour + fDCT + quantize +
Huffman + bit stitching + byte stuffing -> complete JPEG streams) over one batch of
`--frames` DISTINCT synthetic frames that are already resident in HBM.  Every rank codes its
own batch (frames are independent objects: weak scaling, no data-path collective).  The gather
of the finished byte streams to rank 0 (config #4) is run and verified once, outside the timed
region, and reported as `gather_ms`.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (scan_segments) against
the HBM read roofline with ALGORITHMIC bytes = 3 B/pixel; its duration is
measured live with HIP events on the launch stream (engine timing API).  `cpu_baseline` times
the real reference (oracle/_ref, SSE2 path, "reference") or, if that .so cannot load, the
plain-C oracle ("port") on this host's cores on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, QUALITY = 3840, 2160, 75.0
HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)


def cpu_baseline(frames_np, budget_s=12.0):
    """Reference CPU encoder on this host: 1 thread (the reference has no intra-encode
    threading), plus all cores frame-parallel for information."""
    from oracle import orc, refso
    kind, enc = "port", None
    try:
        if refso.available():
            r = refso.ref()
            enc = lambda im: r.encode(im, QUALITY, 0, refso.YUV_420)
            kind = "reference"
    except OSError:
        enc = None
    if enc is None:
        o = orc.oracle()
        enc = lambda im: o.encode(im, QUALITY, orc.YUV_420)
    enc(frames_np[0])                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        enc(frames_np[n % len(frames_np)])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s * 0.5 or n >= 400:
            break
    single = n * W * H / dt / 1e6
    # all cores, one independent frame per thread (ctypes releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    per = max(2, int(budget_s * 0.5 / max(dt / n, 1e-3)))
    per = min(per, 16)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda i: [enc(frames_np[(i + j) % len(frames_np)]) for j in range(per)],
                    range(cores)))
    dt_all = time.perf_counter() - t0
    return {"value": round(single, 1), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": f"{n} encodes of 3840x2160 G_struct q75 420 method 0, 1 thread, {dt:.1f} s",
            "allcores_value": round(cores * per * W * H / dt_all / 1e6, 1), "allcores": cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="device-resident 4K frames per GPU per step")
    ap.add_argument("--input", choices=["struct", "noise"], default="struct")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", type=int, default=0,
                    help="1: engine pipelined mode (stitch of step i under K1 of step i + 1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sjpeg_amd as sj                       # raises if the HIP library is not built
    from oracle import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    if sj.device_count() == 0:
        raise SystemExit("no HIP device: this benchmark has no CPU fallback")
    F = args.frames

    # ---- synthetic batch, resident in HBM before any timing ------------------------------
    gen = synth.g_struct if args.input == "struct" else synth.g_noise
    distinct = min(F, 8)                          # 8 distinct pictures, tiled to F frames
    host = [gen(W, H, 7654321 + rank * distinct + k) for k in range(distinct)]
    frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device="cuda")
    for k in range(F):
        frames[k] = torch.from_numpy(host[k % distinct]).cuda()
    tables, quant = sj.make_tables(quality=QUALITY)
    header = sj.make_header(W, H, sj.YUV_420, quant)
    out_stride = ((W * H * 3) // 2 + len(header) + 4095) & ~4095       # 1.5 B/px slots
    out = torch.empty((F, out_stride), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
    eng = sj.Engine(local)

    def step():
        eng.encode_frames(frames, tables, header, sj.YUV_420, out=out, sizes=sizes,
                          out_stride=out_stride)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.set_pipelined(bool(args.pipelined))
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    piped_scan_ms = None
    if args.pipelined:                            # K1 under the overlap, then back to ordered calls
        eng.set_timing(True)
        acc = []
        for _ in range(4):                        # K1 of the LAST of four back-to-back steps: under the stitch of the third
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            acc.append(eng.last_scan_ms())
        piped_scan_ms = float(np.mean(acc))
        eng.set_timing(False)
        eng.set_pipelined(False)
    # ---- dominant-kernel duration, HIP events on the launch stream -------------------------
    eng.set_timing(True)
    scan_ms, total_ms = [], []
    for _ in range(max(5, min(args.steps, 20))):
        step()
        scan_ms.append(eng.last_scan_ms())
        total_ms.append(eng.last_total_ms())
    eng.set_timing(False)
    scan_avg = float(np.mean(scan_ms)) * 1e-3
    algo_bytes = 3.0 * W * H * F
    achieved = algo_bytes / scan_avg
    # HBM bytes per launch from the committed PMC pass of this same command (profiles/):
    # 2 x FETCH_SIZE KiB (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB.
    traffic = None
    try:
        import re
        txt = open(os.path.join(ROOT, "profiles", "r01", "v9_pmc_summary.txt")).read()
        blk = txt[txt.index("scan_segments<1"):]
        blk = blk[:blk.index("==", 5)] if "==" in blk[5:] else blk
        fetch = float(re.search(r"FETCH_SIZE\s+total=\S+\s+per_dispatch=(\S+)", blk).group(1))
        write = float(re.search(r"WRITE_SIZE\s+total=\S+\s+per_dispatch=(\S+)", blk).group(1))
        if F == 64 and args.input == "struct":
            traffic = int((2.0 * fetch + write) * 1024)
    except Exception:
        traffic = None

    # ---- what a read-only stream kernel reaches on this device (context for `peak`) ----------
    stream_gbps = None
    if rank == 0:
        try:
            import ctypes as C
            L = sj.lib()
            L.sjpeg_hip_debug_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
            probe = frames.reshape(-1)
            sink = torch.zeros(4, dtype=torch.int32, device="cuda")
            nbytes = (probe.numel() // 16) * 16
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                L.sjpeg_hip_debug_stream_read(probe.data_ptr(), nbytes, sink.data_ptr(), st)
            s0.record()
            for _ in range(10):
                L.sjpeg_hip_debug_stream_read(probe.data_ptr(), nbytes, sink.data_ptr(), st)
            s1.record()
            torch.cuda.synchronize()
            stream_gbps = nbytes * 10 / (s0.elapsed_time(s1) * 1e-3) / 1e9
        except Exception:
            stream_gbps = None

    # ---- parity: every coded frame must equal the reference bit for bit -------------------
    torch.cuda.synchronize()
    sz = sizes.cpu().numpy()
    coded = [bytes(out[k, :int(sz[k])].cpu().numpy()) for k in range(F)]
    parity = None
    if rank == 0:
        from oracle import orc
        o = orc.oracle()
        want = [o.encode(host[k], QUALITY, orc.YUV_420) for k in range(distinct)]
        parity = all(coded[k] == want[k % distinct] for k in range(F))
        if args.input == "struct":
            d = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))
            parity = parity and hashlib.md5(coded[0]).hexdigest() == d["struct4k|420|q75|m0"]["md5"]

    # ---- config #4 exchange step: gather the byte streams to rank 0, verified, untimed ----
    gather_ms, gather_error = None, None
    if world > 1:
        from sjpeg_amd.dist import gather_streams
        ids = list(range(rank, F * world, world))
        fence()
        g0 = time.perf_counter()
        try:
            got = gather_streams(out, sizes, ids, F * world, dst=0)
            fence()
            gather_ms = (time.perf_counter() - g0) * 1e3
            if rank == 0:
                parity = parity and all(got[k] == coded[k // world] for k in ids) and \
                    all(g is not None and g[:2] == b"\xff\xd8" for g in got)
        except Exception as exc:                 # the exchange is outside the metric: report, do not lose the line
            gather_error = repr(exc)

    if rank == 0:
        mpix = W * H * F * world * args.steps / dt / 1e6
        res = {
            "metric": "Mpixels/s JPEG encode, 4K RGB q=75 YUV420; bit-exact vs ref",
            "value": round(mpix, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32/int16 (u8 in)", "data": "synthetic",
            "config": {"workload": f"{F} device-resident 3840x2160 G_{args.input} frames per GPU per step, "
                                   "q=75 YUV420 method 0 (standard Huffman), one complete JPEG per frame",
                       "frames_per_gpu": F, "width": W, "height": H, "quality": QUALITY,
                       "yuv_mode": "420", "parallelism": f"frame-sharded x{world}, no data-path collective"},
            "bit_exact": bool(parity),
            "bytes_per_frame": int(sz[0]),
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                         "kernel": "scan_segments<420>", "kernel_ms": round(scan_avg * 1e3, 4),
                         "all_kernels_ms": round(float(np.mean(total_ms)), 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "stream_read_GBps": None if stream_gbps is None else round(stream_gbps, 1),
                         "frac_of_stream_read": None if not stream_gbps else round(achieved / 1e9 / stream_gbps, 4)},
        }
        res["config"]["pipelined"] = bool(args.pipelined)
        if piped_scan_ms is not None:
            res["roofline"]["kernel_ms_pipelined"] = round(piped_scan_ms, 4)
        if gather_ms is not None:
            res["gather_ms"] = round(gather_ms, 2)
        if gather_error is not None:
            res["gather_error"] = gather_error
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(host)
        if not parity:
            res["value"] = 0.0
            res["error"] = "output differs from the reference: throughput not counted"
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
