#!/usr/bin/env python3
"""bench.py -- headline benchmark (BASELINE.json): Mpixels/s of baseline-JPEG encode,
3840x2160 RGB, q=75, YUV 4:2:0, method 0, bit-exact vs the reference.

A "step" = one pass of the hot path (sjpeg_hip_encode_scan: colour + fDCT + quantize +
Huffman + bit stitching + byte stuffing -> complete JPEG streams) over one batch of
`--frames` DISTINCT synthetic frames that are already resident in HBM.  Every rank codes its
own batch (frames are independent objects: weak scaling, no data-path collective in `value`).
The K timed steps are issued back to back in the engine's pipelined mode (the stitch kernels of
step i run on the engine's own stream under the dominant kernel of step i + 1; everything has
finished when the closing fence returns); `ms_per_step_ordered` is the same step with strictly
ordered kernels, and `roofline.kernel_ms` the dominant kernel alone.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N ...          (no WORLD_SIZE in the environment: bench.py starts the N ranks
                                         itself -- it re-executes under torch.distributed.run on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Methodology: after W warm-up steps, `--regions` R (default 11) timed regions of
EXACTLY K steps each, every one bracketed by barrier + device synchronise, the maximum over ranks taken
per region; `ms_per_step` / `value` are the MEDIAN region, `ms_per_step_min` / `_max` the spread, and
`clocks`: the shader clock a probe wave measures WHILE the last region runs (`shader_mhz_under_load`: the chip clocks
down under this kernel's sustained VALU load) and right behind the first and the last region (cycle counter against the
device's 100 MHz counter: sjpeg_hip_debug_shader_clock) and what sysfs reports for sclk / mclk while they run -- on
these boxes the sysfs sclk lags and shows ~100 MHz, the measured figure is the one to read (boxes differ by a few per cent).

Rank 0 prints ONE JSON line.
* `roofline` prices the dominant kernel (scan_segments) against the HBM read roofline with
  ALGORITHMIC bytes = 3 B/pixel; its duration is measured live with HIP
  events on the launch stream (engine timing API) IN THE MODE OF `value` -- pipelined, read once per
  burst of eight back-to-back steps: `frac` / `kernel_ms` are that kernel with the previous step's
  stitch kernels beside it, as in the timed regions (rocprofv3 of the timed regions agrees:
  profiles/r06/timed_region_kernel_stats.csv); `frac_ordered` / `kernel_ms_ordered` are the kernel
  ALONE on an otherwise idle chip (ordered calls, the host waits for every step); `frac_in_region` =
  the same bytes / `ms_per_step` / peak, the whole timed step.
  `engine_scratch_bytes` is what the pipelined mode holds, `engine_scratch_bytes_ordered` what the
  ordered mode (`ms_per_step_ordered`) needs.  `roofline.valu` is the second roofline the
  kernel actually runs into: VALU instructions per wave (from the committed PMC pass named in
  `source`) x the cycles a wave64 instruction occupies a SIMD, measured live by a
  microbenchmark of the two issue classes (tools/valu_rate.hip has the full table).
* `other_configs`: the other BASELINE.json configurations on this device (C2 with noise, C3 8K
  4:4:4 q90, C4 64 x 1080p, one resident 4K frame; 32 4K frames with the reference's default
  parameters = method 4, and C5's recompress matrices with method 0 and with the default
  parameters, through sjpeg_hip_encode_batch_src), each parity-checked against
  tests/golden/digests.json.  Rank 0 at N = 1 only.
* N > 1 (or `--exchange` at N = 1): `value` stays the device-resident figure, gather EXCLUDED.  Beside it
  `with_gather`: the same steps with the exchange step INSIDE the timed region -- the streams of
  every step are packed on the device (sjpeg_hip_compact_streams) and gathered to rank 0 over
  RCCL by the library's own communicator (sjpeg_hip_gather_rows / _bytes, exact lengths) under the next
  step's kernels (sjpeg_amd.dist.exchange_loop) --, and `c4_sharded_gathered`: BASELINE.json config #4 as
  written (64 x 1080p, frame k on rank k % N, gathered to rank 0, concatenation checked against the MD5 of
  SURVEY.md section 8c), with its own gather-excluded and gather-included figures.
* `cpu_baseline` times the real reference (oracle/_ref, SSE2 path, "reference") or, if that .so
  cannot load, the plain-C oracle ("port") on this host's cores on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, QUALITY = 3840, 2160, 75.0
HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9
# the PMC pass of this build the static figures (HBM traffic, VALU instructions per wave) come from
PMC_SUMMARY = os.path.join("profiles", "r06", "final_pmc_summary.txt")


def host_cpus():
    """(model string, one logical CPU per PHYSICAL core this process may run on, logical CPUs it may run on):
    /proc/cpuinfo's (physical id, core id) pairs, restricted to the affinity mask."""
    model, cores, cur = None, {}, {}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur:
                    key = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
                    if int(cur["processor"]) in allowed:
                        cores.setdefault(key, int(cur["processor"]))
                cur = {}
                continue
            k, v = (t.strip() for t in line.split(":", 1))
            cur[k] = v
            if k == "model name" and model is None:
                model = v
        if "processor" in cur and int(cur["processor"]) in allowed:
            cores.setdefault((cur.get("physical id", "0"), cur.get("core id", cur["processor"])), int(cur["processor"]))
    except OSError:
        pass
    one_per_core = sorted(cores.values()) or allowed
    return model, one_per_core, allowed


def cpu_baseline(frames_np, budget_s=12.0):
    """Reference CPU encoder on this host: 1 thread (the reference has no intra-encode threading), plus one
    independent frame stream per PHYSICAL core, every worker pinned to its own core ("all
    physical cores, one independent frame per thread; state nproc and CPU model")."""
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
    model, core_cpus, allowed = host_cpus()
    enc(frames_np[0])                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        enc(frames_np[n % len(frames_np)])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s * 0.5 or n >= 400:
            break
    single = n * W * H / dt / 1e6
    # one independent frame stream per physical core (ctypes releases the GIL), each worker on its own core
    from concurrent.futures import ThreadPoolExecutor
    cores = len(core_cpus)
    per = max(2, int(budget_s * 0.5 / max(dt / n, 1e-3)))
    per = min(per, 16)

    def worker(i):
        try:
            os.sched_setaffinity(0, {core_cpus[i]})     # (pid 0: the calling thread)
        except (AttributeError, OSError):
            pass
        for j in range(per):
            enc(frames_np[(i + j) % len(frames_np)])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(worker, range(cores)))
    dt_all = time.perf_counter() - t0
    return {"value": round(single, 1), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": f"{n} encodes of 3840x2160 G_struct q75 420 method 0, 1 thread, {dt:.1f} s",
            "cpu_model": model, "physical_cores": cores, "logical_cpus": len(allowed),
            "allcores_value": round(cores * per * W * H / dt_all / 1e6, 1), "allcores": cores,
            "allcores_sample": f"{cores} threads (one per physical core, pinned), {per} encodes each, {dt_all:.1f} s"}


def pmc_figures():
    """(HBM bytes per K1 launch, VALU instructions per wave, waves per launch) of the committed PMC pass
    of this build's default bench command, or Nones.  2 x FETCH_SIZE KiB (gfx950 wide-read
    correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB."""
    try:
        txt = open(os.path.join(ROOT, PMC_SUMMARY)).read()
        blk = txt[txt.index("scan_segments<1"):]
        blk = blk[:blk.index("==", 5)] if "==" in blk[5:] else blk
        get = lambda name: float(re.search(name + r"\s+total=\S+\s+per_dispatch=(\S+)", blk).group(1))
        waves = get("SQ_WAVES")
        return int((2.0 * get("FETCH_SIZE") + get("WRITE_SIZE")) * 1024), get("SQ_INSTS_VALU") / waves, waves
    except Exception:
        return None, None, None


def valu_cycles(sj):
    """Cycles a wave64 instruction of the two VALU issue classes occupies a SIMD on this device,
    measured live (8 waves per SIMD, independent chains): [packed / VOP3 / multiply / permute
    class, simple 32-bit integer and f32 class]."""
    import ctypes as C
    import torch
    L = sj.lib()
    out = (C.c_float * 2)()
    rc = L.sjpeg_hip_debug_valu_rate(out, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return (float(out[0]), float(out[1])) if rc == 0 else None


def run_config(sj, torch, eng, frames_np, tile, q, mode, want_md5, reps=10):
    """One of the other BASELINE.json configurations: `tile` copies of the pictures resident in HBM,
    whole path timed over `reps` calls, K1 by HIP events, frame 0 (or the whole batch) against the
    committed digest."""
    h, w = frames_np[0].shape[:2]
    F = len(frames_np) * tile
    frames = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda")
    for k in range(F):
        frames[k] = torch.from_numpy(frames_np[k % len(frames_np)]).cuda()
    tables, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, mode, quant)
    bpp = 3 if mode == sj.YUV_444 else 2
    stride = ((w * h * bpp) // 2 + len(header) + 4095) & ~4095
    out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
    step = lambda: eng.encode_frames(frames, tables, header, mode, out=out, sizes=sizes, out_stride=stride)
    def timed():
        # median of five groups of `reps` calls, one synchronise behind each group: a single stall of the host or the
        # box inside ONE loop of ten 40-microsecond calls once read 0.35 ms for the one-frame configuration (round 6)
        groups = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / reps)
        return float(np.median(groups))
    for _ in range(6):                            # (K1 of a new geometry gets 5 % faster over its first dozen calls)
        step()
    torch.cuda.synchronize()
    dt = timed()
    # the same calls in the engine's pipelined mode (the mode of the headline: K1 of call n + 1 beside the stitch of call n)
    eng.set_pipelined(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    dt_piped = timed()
    eng.set_pipelined(False)
    torch.cuda.synchronize()
    eng.set_timing(True)
    k1 = []
    for _ in range(8):
        step()
        k1.append(eng.last_scan_ms())
    k1 = k1[3:]
    eng.set_timing(False)
    sz = sizes.cpu().numpy()
    if isinstance(want_md5, dict):                       # whole batch, concatenated
        cat = hashlib.md5()
        for k in range(F):
            cat.update(bytes(out[k, :int(sz[k])].cpu().numpy()))
        ok = cat.hexdigest() == want_md5["md5"] and int(sz.sum()) == want_md5["size"]
    else:
        ok = hashlib.md5(bytes(out[0, :int(sz[0])].cpu().numpy())).hexdigest() == want_md5
    k1_s = float(np.mean(k1)) * 1e-3
    return {"frames": F, "width": w, "height": h, "mpix_s": round(F * w * h / dt / 1e6, 1),
            "ms_per_step": round(dt * 1e3, 4), "mpix_s_pipelined": round(F * w * h / dt_piped / 1e6, 1),
            "ms_per_step_pipelined": round(dt_piped * 1e3, 4), "kernel_ms": round(k1_s * 1e3, 4),
            "frac": round(3.0 * w * h * F / k1_s / HBM_PEAK, 4), "bytes_per_frame": int(sz[0]),
            "bit_exact": bool(ok)}


def run_batch_config(sj, torch, eng, frames_np, tile, mode, method, want, quality=75.0, quant=None, reps=25, latency_calls=15):
    """A configuration that goes through the per-picture analysis of the reference (adaptive quantization,
    optimised Huffman tables: sjpeg_hip_encode_batch_src, device passes + host analysis in between), or
    through caller-supplied matrices (C5): `tile` copies resident in HBM, whole call timed, frame 0
    against the committed digest."""
    h, w = frames_np[0].shape[:2]
    F = len(frames_np) * tile
    frames = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda")
    for k in range(F):
        frames[k] = torch.from_numpy(frames_np[k % len(frames_np)]).cuda()
    src, _ = sj.make_source(sj.SRC_RGB, [frames.view(F, h, w * 3)])
    qm = np.zeros((2, 64), np.uint8)
    if quant is None:
        sj.lib().sjpeg_hip_quality_matrices(float(quality), qm.ctypes.data)
    else:
        qm[:] = quant
    stride = ((w * h * 2) // 2 + 4096 + 4095) & ~4095
    out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(F, dtype=torch.int64, device="cuda")
    step = lambda: eng.encode_batch(src, F, w, h, mode, qm, method, min_quant=quant, out_stride=stride, out=out, sizes=sizes)
    first = []
    for _ in range(5):                            # (the first call of a geometry allocates the scratch: ~10 ms)
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        first.append(round((time.perf_counter() - t0) * 1e3, 3))
    # Timed like every other configuration: calls back to back, one synchronise behind a group of them (a call returns
    # when its last pass is launched; the next call's first pass queues behind it).  Beside it the call taken alone
    # -- synchronised after each, nothing of the next one under its tail --, median: the latency of one batch.
    groups = []
    for _ in range(5):                            # (median of five groups)
        t0 = time.perf_counter()
        for _ in range(max(reps // 5, 1)):
            step()
        torch.cuda.synchronize()
        groups.append((time.perf_counter() - t0) / max(reps // 5, 1))
    dt = float(np.median(groups))
    # (`latency_calls` of them: p50 / p99 / max say whether a call of the process stands out -- round 4 had one of
    # 6-8 ms some tens of calls in, the pinned upload blocks growing one by one)
    per_call = []
    for _ in range(latency_calls):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per_call.append(time.perf_counter() - t0)
    dt_alone = float(np.median(per_call))
    sz = sizes.cpu().numpy()
    # every frame (all are copies of one picture; a batch of 24 frames or more is coded in two parts: both are checked)
    ok = len(frames_np) == 1 and all(int(n) == want["size"] for n in sz)
    for k in range(F):
        ok = ok and hashlib.md5(bytes(out[k, :int(sz[k])].cpu().numpy())).hexdigest() == want["md5"]
    return {"frames": F, "width": w, "height": h, "method": method, "mpix_s": round(F * w * h / dt / 1e6, 1),
            "ms_per_step": round(dt * 1e3, 4), "ms_per_call_alone": round(dt_alone * 1e3, 4),
            "call_ms": {"calls": latency_calls, "p50": round(dt_alone * 1e3, 4), "p99": round(float(np.percentile(per_call, 99)) * 1e3, 4),
                        "max": round(float(np.max(per_call)) * 1e3, 4), "first_calls": first},
            "bytes_per_frame": int(sz[0]), "bit_exact": bool(ok)}


def device_clocks(index=0):
    """Current shader / memory clock of device `index` in MHz: the starred line of the amdgpu sysfs
    tables (pp_dpm_sclk / pp_dpm_mclk), else rocm-smi; None where neither can be read."""
    import glob
    out = {}
    cards = getattr(device_clocks, "cards", None)
    if cards is None:                             # (found once: the read itself must stay far below a region's time)
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int(re.search(r"card(\d+)", p).group(1))):
            try:
                if open(os.path.join(d, "vendor")).read().strip() == "0x1002" and os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                    cards.append(d)
            except OSError:
                pass
        device_clocks.cards = cards
    if index < len(cards):
        for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
            try:
                for line in open(os.path.join(cards[index], name)):
                    if "*" in line:
                        out[key] = int(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1))
            except (OSError, AttributeError):
                pass
    if not out:
        try:
            import subprocess
            txt = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--json"], capture_output=True,
                                 text=True, timeout=20).stdout
            card = next(iter(json.loads(txt).values()))
            for key, pat in (("sclk_mhz", "sclk"), ("mclk_mhz", "mclk")):
                for k, v in card.items():
                    if pat in k.lower() and "level" in k.lower():
                        m = re.search(r"(\d+)\s*[Mm][Hh]z", str(v))
                        if m:
                            out[key] = int(m.group(1))
        except Exception:
            pass
    return out or None


def launch_ranks(args):
    """`--gpus N` with N > 1 and no torch.distributed environment: this process becomes the launcher --
    one rank per GPU under torch.distributed.run on 127.0.0.1 with the same arguments; rank 0 of the
    started world prints the line.  Refuses more ranks than devices."""
    import socket
    if not args.launch_check:
        import torch
        have = torch.cuda.device_count()
        if args.gpus > have:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} device(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def launch_check(world, rank):
    """What the CPU-side test of the launcher runs (tests/test_dist_cpu.py): the started world meets on gloo,
    no device is touched; rank 0 prints who came."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank], dtype=torch.int64))
    dist.barrier()
    if rank == 0:
        emit({"launch_check": True, "n_gpus": world, "ranks": [int(t.item()) for t in seen],
              "local_ranks_env": os.environ.get("LOCAL_RANK")})
    dist.destroy_process_group()


def c4_region(sj, torch, eng, rank, world, steps, regions, digests, fence, max_over_ranks):
    """BASELINE.json config #4 as it is written: 64 frames G_struct(1920, 1080, 7654321 + k), q75 4:2:0 method 0,
    frame k coded by rank k % world (64 / world resident frames per rank, one encode call per step), the coded
    streams gathered to rank 0 (device-side packing + the C-ABI exchange, under the next step's kernels).
    Two figures, both medians over `regions` regions of `steps` steps: `encode_only` = device-resident, gather
    EXCLUDED; `with_gather` = the gather inside the timed region.  Rank 0 brings the last step's gathered
    streams to the host, concatenates them in frame order and compares the MD5 with SURVEY.md section 8c."""
    from oracle import synth
    from sjpeg_amd.dist import exchange_loop, shard_frames
    w, h, total = 1920, 1080, 64
    ids = shard_frames(total, rank, world)
    n = len(ids)
    frames = torch.empty((max(n, 1), h, w, 3), dtype=torch.uint8, device="cuda")
    for i, k in enumerate(ids):
        frames[i] = torch.from_numpy(synth.g_struct(w, h, 7654321 + k)).cuda()
    tables, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    stride = ((w * h * 3) // 2 + len(header) + 4095) & ~4095
    stride = (stride + 15) & ~15
    cap = max(n, 1) * stride if rank != 0 else total * stride        # rank 0: its own streams, the others' behind them
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
    sizes = [torch.zeros(max(n, 1), dtype=torch.int64, device="cuda") for _ in range(2)]
    offs = [torch.zeros(max(n, 1) + 1, dtype=torch.int64, device="cuda") for _ in range(2)]

    def encode(b=0):
        if n > 0:
            eng.encode_frames_packed(frames[:n], tables, header, sj.YUV_420, outs[b], sizes[b], offs[b], stride)

    def timed(fn):
        dts = []
        for _ in range(regions):
            fence()
            t0 = time.perf_counter()
            fn()
            fence()
            dts.append(max_over_ranks(time.perf_counter() - t0))
        return float(np.median(dts)), dts

    for _ in range(3):
        encode()
    enc_dt, _ = timed(lambda: [encode(s & 1) for s in range(steps)])
    exchange_loop(2, encode, outs, sizes, ids, total, use_streams=True, keep="last", packed_offsets=offs)
    got = []
    gat_dt, gat_all = timed(lambda: got.append(exchange_loop(steps, encode, outs, sizes, ids, total, use_streams=True,
                                                             keep="last", packed_offsets=offs)))
    ok, nbytes = None, None
    if rank == 0:
        fr = got[-1][-1].frames()
        cat = hashlib.md5()
        for f in fr:
            cat.update(f)
        want = digests["struct1080p_k0..63_concat|420|q75|m0"]
        nbytes = sum(len(f) for f in fr)
        ok = cat.hexdigest() == want["md5"] and nbytes == want["size"]
    px = w * h * total * steps
    return {"frames": total, "frames_per_rank": n, "width": w, "height": h,
            "encode_only": {"mpix_s": round(px / enc_dt / 1e6, 1), "ms_per_step": round(enc_dt / steps * 1e3, 4),
                            "what": "every rank codes its 64 / N resident frames, nothing leaves the device: gather EXCLUDED"},
            "with_gather": {"mpix_s": round(px / gat_dt / 1e6, 1), "ms_per_step": round(gat_dt / steps * 1e3, 4),
                            "ms_per_step_min": round(min(gat_all) / steps * 1e3, 4),
                            "ms_per_step_max": round(max(gat_all) / steps * 1e3, 4),
                            "what": "the same steps with packing + the gather of all 64 streams into rank 0's HBM INSIDE the "
                                    "timed region (the exchange of step s under the kernels of step s + 1)"},
            "gathered_bytes_per_step": nbytes, "bit_exact": ok,
            "check": "rank 0: streams of the last step, host-side concatenation in frame order, MD5 db599667..."}


def emit(res):
    """The ONE JSON line, last on stdout: RCCL prints a version banner through C stdio, which would
    otherwise be flushed behind it at exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="device-resident 4K frames per GPU per step")
    ap.add_argument("--input", choices=["struct", "noise"], default="struct")
    ap.add_argument("--slot-bpp", type=float, default=0.75, help="bytes per pixel of every frame's output slot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--exchange", action="store_true",
                    help="N = 1 only: run the `with_gather` region too (RCCL with a single rank: packing + self-gather)")
    ap.add_argument("--regions", type=int, default=11,
                    help="timed regions of --steps steps each; the line carries the median region and the spread")
    ap.add_argument("--timed-only", action="store_true",
                    help="stop after the timed regions and the parity check: no per-kernel timing calls, no ordered "
                         "steps, no other configurations -- the run tools/profile_gpu.sh traces to compare the "
                         "rocprof mean of the dominant kernel with ms_per_step")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the world exactly as a run would, meet on gloo, print who came, touch no device")
    ap.add_argument("--pipelined", type=int, default=1,
                    help="1 (default): the timed steps run in the engine's pipelined mode (the stitch kernels of "
                         "step i on the engine's own stream, under K1 of step i + 1); 0: ordered calls")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.regions < 1:
        raise SystemExit("bench.py: --gpus, --steps and --regions must be positive")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                        # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: started with WORLD_SIZE={world}, --gpus {args.gpus}: the line reports n_gpus = {world}",
              file=sys.stderr)
    if args.launch_check:
        return launch_check(world, rank)

    import torch
    import torch.distributed as dist
    import sjpeg_amd as sj                       # raises if the HIP library is not built
    from oracle import synth

    if world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} device(s): one rank per GPU")
    if rank != 0:
        # only rank 0's stdout carries the JSON line: whatever the other ranks (or the libraries
        # they load: RCCL prints a banner through C stdio at exit) write goes to stderr
        os.dup2(2, 1)
    exchange = world > 1 or args.exchange
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
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
    # output slots of --slot-bpp bytes per pixel (default 0.75: three times what the structured frames take, a quarter
    # more than the noise frames); the engine's segment scratch follows the caller's slot size
    out_stride = (int(W * H * args.slot_bpp) + len(header) + 4095) & ~4095
    nsets = 2 if exchange else 1                  # the exchange of step s overlaps the encode of step s + 1
    outs = [torch.empty((F, out_stride), dtype=torch.uint8, device="cuda") for _ in range(nsets)]
    sizes_b = [torch.zeros(F, dtype=torch.int64, device="cuda") for _ in range(nsets)]
    out, sizes = outs[0], sizes_b[0]
    eng = sj.Engine(local)

    def encode(b=0):
        eng.encode_frames(frames, tables, header, sj.YUV_420, out=outs[b], sizes=sizes_b[b],
                          out_stride=out_stride)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    eng.set_pipelined(bool(args.pipelined))
    for _ in range(20):                           # settle: buffers of both pipeline sets allocated, clocks up (the
        encode()                                  # frames were generated on the host for seconds before this)
    for _ in range(args.warmup):
        encode()
    fence()
    # (the clocks are read while the steps of a region are queued on the device -- between regions it idles and
    # the shader clock drops to ~100 MHz within microseconds; the read costs the host ~0.1 ms, the device nothing)
    clocks = {} if rank == 0 else None
    probe_stream = torch.cuda.Stream() if rank == 0 else None
    if rank == 0:
        device_clocks(local)                      # (finds the sysfs files: outside the timed regions)
    region_s = []
    for reg in range(args.regions):               # every region: EXACTLY --steps steps between two fences
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            encode()
        if rank == 0 and reg in (0, args.regions - 1):
            clocks["first_region" if reg == 0 else "last_region"] = device_clocks(local)
        if rank == 0 and reg == args.regions - 1:
            # the shader clock UNDER this load: a probe wave on a side stream while the region's steps are queued and
            # running (K1 leaves half of a CU's wave slots free); 20 us of one wave, the host waits for it only
            try:
                import ctypes as C
                mhz = C.c_float(0)
                with torch.cuda.stream(probe_stream):
                    if sj.lib().sjpeg_hip_debug_shader_clock(C.byref(mhz), C.c_void_p(probe_stream.cuda_stream)) == 0:
                        clocks["shader_mhz_under_load"] = round(mhz.value, 1)
            except Exception:
                pass
        fence()
        region_s.append(max_over_ranks(time.perf_counter() - t0))
        if rank == 0 and reg in (0, args.regions - 1):
            # outside the timed region, right behind its last kernel: the shader clock as a wave measures it
            # (the sysfs figure above lags and shows ~100 MHz whenever it samples an empty queue)
            try:
                import ctypes as C
                mhz = C.c_float(0)
                if sj.lib().sjpeg_hip_debug_shader_clock(C.byref(mhz), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0:
                    clocks["shader_mhz_after_first_region" if reg == 0 else "shader_mhz_after_last_region"] = round(mhz.value, 1)
            except Exception:
                pass
    dt = float(np.median(region_s))
    # what the timed steps left behind is what gets checked (fence() waited for the engine's stream too)
    sz = sizes.cpu().numpy()
    coded = [bytes(out[k, :int(sz[k])].cpu().numpy()) for k in range(F)]

    scratch_piped = eng.scratch_bytes()
    algo_bytes = 3.0 * W * H * F
    scan_ms, total_ms, ordered_ms = [], [], None
    traffic = valu_per_wave = waves = None
    stream_gbps, rates = None, None
    scan_ms_piped = []
    if not args.timed_only:
        if args.pipelined:
            # ---- dominant-kernel duration IN THE MODE OF THE HEADLINE: HIP events on the engine's launch
            # stream around K1 while the stitch kernels of the previous step run beside it on the engine's side stream --
            # the host waits for K1's second event only, never for the stitch, so the steps still overlap as they do in
            # the timed regions
            # (read once per BURST of eight back-to-back steps -- the eighth step's K1, in steady state: waiting for every
            # step's events lets the device run dry between steps, and the K1 that follows starts beside the whole stitch
            # of its predecessor instead of its tail -- 0.93 ms by that clock against 0.89 per step)
            eng.set_timing(True)
            for _ in range(max(5, min(args.steps, 20))):
                for _ in range(8):
                    encode()
                scan_ms_piped.append(eng.last_scan_ms())
            eng.set_timing(False)
            fence()
            eng.set_pipelined(False)              # ordered calls for the other per-kernel figures
        # ---- dominant-kernel duration with ordered calls, HIP events on the launch stream -------
        eng.set_timing(True)
        for _ in range(max(5, min(args.steps, 20))):
            encode()
            scan_ms.append(eng.last_scan_ms())
            total_ms.append(eng.last_total_ms())
        eng.set_timing(False)
        fence()
        t1 = time.perf_counter()
        for _ in range(5):
            encode()
        fence()
        ordered_ms = (time.perf_counter() - t1) / 5 * 1e3
        if F == 64 and args.input == "struct":    # the PMC pass is of the default command
            traffic, valu_per_wave, waves = pmc_figures()
    scan_avg_ordered = float(np.mean(scan_ms)) * 1e-3 if scan_ms else None
    # `frac`, `achieved`, `kernel_ms` are of the mode `value` was measured in; the ordered figures stand beside them
    scan_avg = float(np.mean(scan_ms_piped)) * 1e-3 if scan_ms_piped else scan_avg_ordered
    achieved = algo_bytes / scan_avg if scan_avg else None

    # ---- what a read-only stream kernel reaches on this device (context for `peak`) ----------
    if rank == 0 and not args.timed_only:
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
        try:
            rates = valu_cycles(sj)
        except Exception:
            rates = None

    # ---- parity: every coded frame must equal the reference bit for bit -------------------
    torch.cuda.synchronize()
    parity = None
    digests = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))
    if rank == 0:
        from oracle import orc
        o = orc.oracle()
        want = [o.encode(host[k], QUALITY, orc.YUV_420) for k in range(distinct)]
        parity = all(coded[k] == want[k % distinct] for k in range(F))
        if args.input == "struct":
            parity = parity and hashlib.md5(coded[0]).hexdigest() == digests["struct4k|420|q75|m0"]["md5"]

    res = {}
    if rank == 0:
        mpix = W * H * F * world * args.steps / dt / 1e6
        per_step = sorted(r / args.steps * 1e3 for r in region_s)
        roof = {"bound": "hbm", "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                # the algorithmic bytes over the whole timed step (K1 with the stitch kernels of the
                # previous step beside it): what the headline `value` corresponds to
                "frac_in_region": round(algo_bytes / (dt / args.steps) / HBM_PEAK, 4),
                "kernel": "scan_segments<420>", "algorithmic_bytes_per_launch": int(algo_bytes)}
        if scan_avg:
            roof.update({"achieved": round(achieved / 1e9, 1), "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                         "traffic_source": PMC_SUMMARY if traffic is not None else None,
                         # (traffic and valu.instr_per_wave are read from that committed PMC pass of this build's
                         # default command, not collected in this run: counters need rocprofv3 around the process)
                         "traffic_static": traffic is not None,
                         "kernel_ms": round(scan_avg * 1e3, 4),
                         "kernel_ms_min": round(float(np.min(scan_ms_piped if scan_ms_piped else scan_ms)), 4),
                         "kernel_mode": "pipelined (the mode of `value`)" if scan_ms_piped else "ordered calls",
                         "kernel_ms_ordered": round(scan_avg_ordered * 1e3, 4),
                         "frac_ordered": round(algo_bytes / scan_avg_ordered / HBM_PEAK, 4),
                         "all_kernels_ms": round(float(np.mean(total_ms)), 4),
                         "stream_read_GBps": None if stream_gbps is None else round(stream_gbps, 1),
                         "frac_of_stream_read": None if not stream_gbps else round(achieved / 1e9 / stream_gbps, 4)})
        if rates is not None:
            valu = {"cycles_per_instr": {"packed_vop3_mul_perm": round(rates[0], 2), "simple_int_f32": round(rates[1], 2)},
                    "cycles_source": "measured in this run (sjpeg_hip_debug_valu_rate, 8 waves per SIMD, nominal 2.4 GHz)"}
            if valu_per_wave is not None:
                # every instruction priced as the slow class (most of K1 is) / as the fast class
                floor = [valu_per_wave * waves * c / N_SIMD / CLOCK_HZ * 1e3 for c in (rates[1], rates[0])]
                valu.update({"instr_per_wave": round(valu_per_wave, 1), "waves_per_launch": int(waves),
                             "instr_source": PMC_SUMMARY,
                             "floor_ms": [round(floor[0], 4), round(floor[1], 4)],
                             "frac": [round(floor[0] / (scan_avg * 1e3), 4), round(floor[1] / (scan_avg * 1e3), 4)],
                             "note": "floor_ms / frac: all instructions at the fast-class rate, all at the slow-class rate"})
            roof["valu"] = valu
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
            # value / ms_per_step = the median of `regions` timed regions of `steps` steps each (max over ranks per region)
            "regions": args.regions, "ms_per_step_min": round(per_step[0], 4), "ms_per_step_max": round(per_step[-1], 4),
            "ms_per_step_regions": [round(r / args.steps * 1e3, 4) for r in region_s],
            "clocks": clocks,
            "bit_exact": bool(parity),
            "bytes_per_frame": int(sz[0]),
            "engine_scratch_bytes": scratch_piped,
            "roofline": roof,
        }
        res["config"]["pipelined"] = bool(args.pipelined)
        if ordered_ms is not None:
            res["ms_per_step_ordered"] = round(ordered_ms, 4)
        if args.pipelined and not args.timed_only:   # what the strictly ordered mode holds (one set of segment buffers)
            e2 = sj.Engine(local)
            e2.encode_frames(frames, tables, header, sj.YUV_420, out=outs[0], sizes=sizes_b[0], out_stride=out_stride)
            torch.cuda.synchronize()
            res["engine_scratch_bytes_ordered"] = e2.scratch_bytes()
            e2.close()
        if not parity:
            res["value"] = 0.0
            res["error"] = "output differs from the reference: throughput not counted"
    # ---- N > 1: the same steps with the exchange step of config #4 inside a second timed region ----
    # The headline line is complete at this point.  A watchdog prints it anyway if the exchange (RCCL
    # on a node this code has not met before) does not come back: the metric must not be lost with it.
    if exchange and not args.timed_only:
        import threading
        from sjpeg_amd.dist import exchange_loop

        def give_up():
            if rank == 0:
                res.setdefault("with_gather", {"error": "the exchange region did not finish within 240 s"})
                res.setdefault("c4_sharded_gathered", {"error": "the exchange region did not finish within 240 s"})
                emit(res)
            os._exit(0)

        dog = threading.Timer(240.0, give_up)
        dog.daemon = True
        dog.start()
        ids = list(range(rank, F * world, world))
        try:
            # PACKED output (sjpeg_hip_encode_scan_packed_src): no compaction pass; rank 0 codes straight into the
            # buffer the other ranks' streams are gathered behind (F x world frames of the measured size + slack)
            per_frame = (int(sz.max()) * 21 // 20 + 4096 + 15) & ~15      # (the other ranks' pictures differ: 5 % of room)
            cap = F * out_stride if rank != 0 else max(F * out_stride, F * world * per_frame)
            pouts = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
            poffs = [torch.zeros(F + 1, dtype=torch.int64, device="cuda") for _ in range(2)]

            def encode_packed(b=0):
                eng.encode_frames_packed(frames, tables, header, sj.YUV_420, pouts[b], sizes_b[b], poffs[b], out_stride)

            fence()
            exchange_loop(2, encode_packed, pouts, sizes_b, ids, F * world, use_streams=True, keep="last", packed_offsets=poffs)
            fence()
            gts = []
            for _ in range(min(args.regions, 5)):
                fence()
                g0 = time.perf_counter()
                got = exchange_loop(args.steps, encode_packed, pouts, sizes_b, ids, F * world, use_streams=True,
                                    keep="last", packed_offsets=poffs)
                fence()
                gts.append(max_over_ranks(time.perf_counter() - g0))
            gdt = float(np.median(gts))
            ok = True
            if rank == 0:                          # the last step's gathered streams, brought to the host and checked
                fr = got[-1].frames()
                ok = all(fr[k] == coded[k // world] for k in ids) and all(f is not None and f[:2] == b"\xff\xd8" for f in fr)
            with_gather = {"value": round(W * H * F * world * args.steps / gdt / 1e6, 1), "unit": "Mpixels/s",
                           "ms_per_step": round(gdt / args.steps * 1e3, 4), "verified": bool(ok),
                           "ms_per_step_min": round(min(gts) / args.steps * 1e3, 4), "ms_per_step_max": round(max(gts) / args.steps * 1e3, 4),
                           "what": "encode with PACKED output (sjpeg_hip_encode_scan_packed_src: no compaction pass) + the C-ABI "
                                   "exchange (sjpeg_hip_gather_rows: RCCL all-gather of one row of sizes per rank, one small host "
                                   "read; sjpeg_hip_gather_bytes: exact-length ncclSend / ncclRecv into rank 0's HBM, behind rank 0's "
                                   "own streams, which never move), the exchange of step s under the kernels of step s + 1; median of "
                                   "5 regions; host copy / concatenation not included"}
        except Exception as exc:                 # the exchange is outside the headline metric: report, do not lose the line
            with_gather = {"error": repr(exc)}
        if rank == 0:
            res["with_gather"] = with_gather
        # the non-rooted end: every rank brings its own streams to its own (pinned) host memory over its own PCIe
        # link, under the next step's kernels -- no collective, nothing converges on rank 0
        try:
            from sjpeg_amd.dist import overlapped_steps, sink_streams_local
            pins = [torch.empty(F * per_frame, dtype=torch.uint8).pin_memory() for _ in range(2)]
            last = {}

            def sink(b):
                last["n"], last["offs"] = sink_streams_local(pouts[b], poffs[b], F, pins[b])
                last["b"] = b

            overlapped_steps(2, encode_packed, sink, True, keep="last")
            fence()
            sts = []
            for _ in range(min(args.regions, 5)):
                fence()
                g0 = time.perf_counter()
                overlapped_steps(args.steps, encode_packed, sink, True, keep="last")
                fence()
                sts.append(max_over_ranks(time.perf_counter() - g0))
            sdt = float(np.median(sts))
            hb, ho, hs = pins[last["b"]].numpy(), last["offs"].numpy(), sizes_b[last["b"]].cpu().numpy()
            ok = all(hb[int(ho[k]):int(ho[k]) + int(hs[k])].tobytes() == coded[k] for k in range(F))
            local_sink = {"value": round(W * H * F * world * args.steps / sdt / 1e6, 1), "unit": "Mpixels/s",
                          "ms_per_step": round(sdt / args.steps * 1e3, 4), "verified": bool(ok),
                          "host_GBps_per_rank": round(last["n"] * args.steps / sdt / 1e9, 1),
                          "what": "every rank: encode with packed output, then ONE device-to-host copy of its own streams into "
                                  "its own pinned buffer (sjpeg_amd.dist.sink_streams_local) under the next step's kernels; no "
                                  "collective -- what a rank's PCIe link carries bounds it (about 0.25 B per pixel here)"}
        except Exception as exc:
            local_sink = {"error": repr(exc)}
        if rank == 0:
            res["with_local_sink"] = local_sink
            # The model the two figures are to be read against: xGMI is point to
            # point, a peer reaches the root over ONE link, so a ROOTED gather cannot carry more pixels per peer than
            # that link's bytes / the coded bytes per pixel -- whatever the kernels do; the spread sinks are bound by
            # every rank's own PCIe link instead.  At N > 1 `with_local_sink` is the figure to quote beside `value`.
            bpp = float(sz.mean()) / (W * H)
            res["multi_gpu_model"] = {
                "coded_bytes_per_pixel": round(bpp, 4), "xgmi_link_GBps": 153.0, "pcie_host_GBps_per_rank": 54.0,
                "rooted_gather_ceiling_per_peer_mpix_s": round(153.0e9 / bpp / 1e6, 1),
                "rooted_gather_ceiling_mpix_s": round((world - 1) * 153.0e9 / bpp / 1e6 + W * H * F * args.steps / dt / 1e6 / world, 1) if world > 1 else None,
                "local_sink_ceiling_mpix_s": round(world * 54.0e9 / bpp / 1e6, 1),
                "quote_beside_value": "with_local_sink",
                "note": "ceilings, not measurements: (N - 1) peers x one 153 GB/s link each / coded bytes per pixel + the root's own "
                        "share of `value`; N x 54 GB/s of PCIe / coded bytes per pixel for the spread sinks"}
        try:                                      # config #4 as written: 64 x 1080p, 64 / N per rank, gathered to rank 0
            c4 = c4_region(sj, torch, eng, rank, world, args.steps, min(args.regions, 5), digests, fence, max_over_ranks)
        except Exception as exc:
            c4 = {"error": repr(exc)}
        dog.cancel()
        if rank == 0:
            res["c4_sharded_gathered"] = c4
            if isinstance(c4, dict) and c4.get("bit_exact") is False:
                parity = False
    if rank == 0:
        if world == 1 and not args.no_other_configs and not args.timed_only:
            del frames
            torch.cuda.empty_cache()
            oc = {}
            try:
                oc["C2 4K G_noise q75 420 x32"] = run_config(
                    sj, torch, eng, [synth.g_noise(W, H, 7654321 + k) for k in range(2)], 16, 75.0, sj.YUV_420,
                    digests["noise4k|420|q75|m0"]["md5"])
                c3 = [synth.g_struct(7680, 4320, 7654321)]
                oc["C3 8K G_struct q90 444 x1"] = run_config(sj, torch, eng, c3, 1, 90.0, sj.YUV_444,
                                                             digests["struct8k|444|q90|m0"]["md5"], reps=20)
                oc["C3 8K G_struct q90 444 x4"] = run_config(sj, torch, eng, c3, 4, 90.0, sj.YUV_444,
                                                             digests["struct8k|444|q90|m0"]["md5"])
                oc["C4 64 x 1080p G_struct q75 420"] = run_config(
                    sj, torch, eng, [synth.g_struct(1920, 1080, 7654321 + k) for k in range(64)], 1, 75.0,
                    sj.YUV_420, digests["struct1080p_k0..63_concat|420|q75|m0"], reps=20)
                oc["C2 4K G_struct q75 420 x1 (latency)"] = run_config(
                    sj, torch, eng, [host[0]] if args.input == "struct" else [synth.g_struct(W, H, 7654321)], 1, 75.0,
                    sj.YUV_420, digests["struct4k|420|q75|m0"]["md5"], reps=50)
                g4k = [host[0]] if args.input == "struct" else [synth.g_struct(W, H, 7654321)]
                oc["C2 4K G_struct q75 420 default parameters (method 4) x32"] = run_batch_config(
                    sj, torch, eng, g4k, 32, sj.YUV_420, 4, digests["struct4k|420|q75|m4"], latency_calls=500)
                # SURVEY 8c's two other default-parameter known answers: C3 as a batch of four (cd6a30d5...), 4K noise (1eb32ac3...)
                oc["C3 8K G_struct q90 444 default parameters (method 4) x4"] = run_batch_config(
                    sj, torch, eng, c3, 4, sj.YUV_444, 4, digests["struct8k|444|q90|m4"], quality=90.0, reps=10)
                oc["C2 4K G_noise q75 420 default parameters (method 4) x32"] = run_batch_config(
                    sj, torch, eng, [synth.g_noise(W, H, 7654321)], 32, sj.YUV_420, 4, digests["noise4k|420|q75|m4"], reps=10)
                c5q = np.array(digests["recompress|r90|m0"]["source_quant"], np.uint8).reshape(2, 64)
                c5q = np.clip((c5q.astype(np.float64) * 100.0 / 90.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)
                oc["C5 4K recompress r=90 method 0 x32"] = run_batch_config(
                    sj, torch, eng, g4k, 32, sj.YUV_420, 0, digests["recompress|r90|m0"], quant=c5q)
                oc["C5 4K recompress r=90 default parameters x32"] = run_batch_config(
                    sj, torch, eng, g4k, 32, sj.YUV_420, 4, digests["recompress|r90|default"], quant=c5q)
                # the two host-API kinds that are "correct, not fast" by DESIGN.md: trellis quantization (method 7)
                # and the sharp-YUV conversion, ONE 1080p picture each, host memory to host memory (PCIe included)
                from oracle import orc
                o = orc.oracle()
                p1080 = synth.g_struct(1920, 1080, 7654321)
                for name, method, mode, omode in (("trellis (method 7) 1080p, host API", 7, sj.YUV_420, 1),
                                                  ("sharp YUV (method 4) 1080p, host API", 4, sj.YUV_SHARP, 2)):
                    got = sj.SjpegEncode(p1080, 75.0, method, mode)
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        sj.SjpegEncode(p1080, 75.0, method, mode)
                        ts.append(time.perf_counter() - t0)
                    dt = float(np.median(ts))
                    oc[name] = {"frames": 1, "width": 1920, "height": 1080, "method": method,
                                "mpix_s": round(1920 * 1080 / dt / 1e6, 1), "ms_per_call": round(dt * 1e3, 3),
                                "bytes_per_frame": len(got or b""),
                                "bit_exact": bool(got is not None and got == o.encode_method(p1080, 75.0, omode, method))}
            except Exception as exc:
                oc["error"] = repr(exc)
            res["other_configs"] = oc
            # SURVEY section 8d "report separately": host buffer in, host buffer out through the drop-in API (SjpegEncode of
            # include/sjpeg.h, PCIe both ways) -- one 4K frame at a time from pageable and from pinned memory, and a
            # stream of frames from several threads (every thread has its own device context and stream)
            try:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                import host_to_host
                hh = host_to_host.measure(budget=0.6, threads=(1, 4))
                hh["what"] = ("SjpegEncode(rgb, 3840, 2160, ...) q75 4:2:0 method 0, pixels and JPEG in host memory; Gpixels/s; "
                              "PCIe bound of the upload alone at ~54 GB/s: 18 Gpixels/s")
                res["host_to_host"] = hh
            except Exception as exc:
                res["host_to_host"] = {"error": repr(exc)}
            if any(isinstance(v, dict) and v.get("bit_exact") is False for v in oc.values()):
                parity = False
        if not args.no_cpu_baseline and world == 1 and not args.timed_only:
            res["cpu_baseline"] = cpu_baseline(host)
        if not parity:
            res["value"] = 0.0
            res["bit_exact"] = False
            res["error"] = "output differs from the reference: throughput not counted"
    if exchange:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
    if rank == 0:
        emit(res)


if __name__ == "__main__":
    main()
