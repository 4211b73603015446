"""Multi-GPU plumbing.

Batch path (BASELINE.json config #4): frames are independent
objects, so the hot path shards by frame with NO data-path collective; the only exchange is
the gather of the finished per-frame byte streams to rank 0 (RCCL over xGMI on the GPU box,
gloo in CPU tests), followed by a host-side concatenate.

One process per GPU, torch.distributed already initialised by the caller.  torch is plumbing
here (process group + device tensors); the encoder itself is the C library.
"""
from typing import List, Optional, Sequence

import numpy as np  # noqa: F401  (sizes arrive as numpy arrays)
import torch
import torch.distributed as dist


def shard_frames(nframes: int, rank: int, world: int) -> List[int]:
    """Frame k goes to rank k % world: the global indices this rank codes."""
    return list(range(rank, nframes, world))


def _align16(x):
    return (x + 15) & ~15


class GatheredStreams:
    """What `dst` holds after gather_streams(): the per-rank packed buffers (device resident) and
    the sizes of all frames; frames() brings them to the host as byte strings in global order."""

    def __init__(self, recv, sizes_all, nframes, world):
        self.recv, self.sizes_all, self.nframes, self.world = recv, sizes_all, nframes, world

    def frames(self) -> List[bytes]:
        out: List[Optional[bytes]] = [None] * self.nframes
        host = torch.stack(self.recv).cpu().numpy()            # one device -> host copy
        for r in range(self.world):
            ids = shard_frames(self.nframes, r, self.world)
            sz = self.sizes_all[r][:len(ids)]
            offs = [0]
            for n in sz[:-1]:
                offs.append(offs[-1] + _align16(int(n)))
            for k, o, n in zip(ids, offs, sz):
                out[k] = host[r, o:o + int(n)].tobytes()
        return out  # type: ignore[return-value]


def gather_streams(out: torch.Tensor, sizes: torch.Tensor, frame_ids: Sequence[int],
                   nframes: int, dst: int = 0, group=None, compact=None,
                   to_host: bool = True):
    """Gathers variable-length coded frames to `dst`: the exchange step of the batch path.

    out [F_local, stride] uint8 and sizes [F_local] int64 as an encode call left them,
    frame_ids the global index of each local frame.  The frames are packed back to back ON THE
    DEVICE by one kernel (sjpeg_hip_compact_streams: every frame at a multiple of 16, no host
    round trip), then TWO collectives move them: all_gather of the sizes (8 B per frame) and one
    gather of the packed buffers, padded to the largest per-rank total (RCCL has no gatherv; its
    send counts are host values, hence the ONE host read of the gathered sizes).  No per-frame
    host synchronisation anywhere.  `compact(out, sizes, n, capacity)` -> (packed, offsets)
    defaults to the C-ABI kernel (CUDA tensors); the CPU/gloo tests pass a torch restatement.
    Returns on `dst` the nframes byte strings in global order (to_host=True) or a
    GatheredStreams holding the device-resident buffers (to_host=False); None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = out.device
    if compact is None:
        import sjpeg_amd as sj
        compact = sj.compact_streams
    n_local = len(frame_ids)
    max_local = (nframes + world - 1) // world
    local_sizes = torch.zeros(max_local, dtype=torch.int64, device=dev)
    local_sizes[:n_local] = sizes[:n_local]
    all_sizes = [torch.zeros_like(local_sizes) for _ in range(world)]
    dist.all_gather(all_sizes, local_sizes, group=group)
    sizes_all = torch.stack(all_sizes).cpu().numpy()           # the one host read (world x frames int64)
    pad = max(int(_align16(sizes_all).sum(axis=1).max()), 16)
    if n_local > 0:
        packed, _ = compact(out, sizes, n_local, pad)
    else:
        packed = torch.zeros(pad, dtype=torch.uint8, device=dev)
    recv = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(packed[:pad], recv, dst=dst, group=group)
    if rank != dst:
        return None
    got = GatheredStreams(recv, sizes_all, nframes, world)
    return got.frames() if to_host else got


def overlapped_steps(nsteps: int, encode, exchange, use_streams: bool, keep: str = "all"):
    """The multi-rank step loop of bench.py: step s is coded into buffer set s & 1 while the streams
    of step s - 1 are exchanged.  encode(buf) enqueues one encode call into set `buf`;
    exchange(buf) runs the exchange of that set (it may block the host: the next encode is
    already queued).  With use_streams the exchange runs on a side CUDA stream ordered behind the
    encode by an event, so that it overlaps the next step's kernels; on CPU (gloo tests) the
    same order of calls runs inline.  Returns the results of the exchanges, in step order
    (keep="last": only the last one -- a result holds the gathered buffers of a whole step)."""
    class _Results(list):
        def append(self, x):
            if keep == "last":
                self.clear()
            super().append(x)
    results = _Results()
    if use_streams:
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        done = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]
        for s in range(nsteps):
            b = s & 1
            if s >= 2:
                main.wait_event(freed[b])          # set b is being read by the exchange of step s - 2
            encode(b)
            done[b].record(main)
            if s > 0:
                pb = (s - 1) & 1
                side.wait_event(done[pb])
                with torch.cuda.stream(side):
                    results.append(exchange(pb))
                    freed[pb].record(side)
        if nsteps > 0:
            pb = (nsteps - 1) & 1
            side.wait_event(done[pb])
            with torch.cuda.stream(side):
                results.append(exchange(pb))
            main.wait_stream(side)
    else:
        for s in range(nsteps):
            encode(s & 1)
            if s > 0:
                results.append(exchange((s - 1) & 1))
        if nsteps > 0:
            results.append(exchange((nsteps - 1) & 1))
    return results


def exchange_loop(nsteps: int, encode, outs, sizes, frame_ids: Sequence[int], nframes: int,
                  use_streams: bool, dst: int = 0, group=None, compact=None, keep: str = "all"):
    """bench.py's timed multi-rank region, as a function so that the CPU/gloo test runs exactly
    this code: `nsteps` encode calls, double buffered (outs[b], sizes[b], b = 0 / 1), the streams
    of every step gathered to `dst` (device resident there) under the next step's kernels.
    Returns the GatheredStreams of every step on `dst`, a list of None elsewhere."""
    return overlapped_steps(
        nsteps, encode,
        lambda b: gather_streams(outs[b], sizes[b], frame_ids, nframes, dst=dst, group=group,
                                 compact=compact, to_host=False),
        use_streams, keep)


# ---- one frame over several GPUs: bands of consecutive segments -------------

def band_ranges(nseg: int, world: int) -> List[tuple]:
    """Segments [r*nseg/world, (r+1)*nseg/world) for every rank r (empty if nseg < world)."""
    return [(r * nseg // world, (r + 1) * nseg // world) for r in range(world)]


def gather_bands(words: torch.Tensor, nbits: torch.Tensor, stride: int, dst: int = 0, group=None):
    """The one exchange step of the banded path: every rank contributes its band's bit string
    (int32 words, at most `stride`) and bit count; `dst` gets ([world, stride] int32, [world] int64),
    the others (None, None).  Two collectives: all_gather of the lengths (8 B each), gather of the
    strings padded to `stride` (RCCL has no gatherv)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = words.device
    mine = torch.zeros(stride, dtype=torch.int32, device=dev)
    n = min(int(words.numel()), stride)
    mine[:n] = words.reshape(-1)[:n]
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, nbits.reshape(1).to(torch.int64), group=group)
    recv = [torch.zeros(stride, dtype=torch.int32, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(mine, recv, dst=dst, group=group)
    if rank != dst:
        return None, None
    return torch.stack(recv).contiguous(), torch.cat(lens).contiguous()


def encode_frame_banded(engine, src, w: int, h: int, tables, header: bytes, yuv_mode: int,
                        dst: int = 0, group=None) -> Optional[bytes]:
    """One frame coded by all ranks of the group, bit-identical to the single-device encode.
    `src` is this rank's sjpeg_amd.Source addressed as the whole frame (only the rows of the rank's
    band and of the MCU in front of it are read).  Returns the JPEG on `dst`, None elsewhere."""
    import sjpeg_amd as sj
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nseg = sj.segment_count(w, h, yuv_mode)
    ranges = band_ranges(nseg, world)
    stride = max(sj.band_bound(w, h, yuv_mode, b, e) for (b, e) in ranges if e > b)
    b, e = ranges[rank]
    dev = torch.device("cuda", torch.cuda.current_device())
    if e > b:
        words, nbits = engine.encode_band(src, w, h, tables, yuv_mode, b, e)
    else:                                     # fewer segments than ranks: this rank has nothing
        words = torch.zeros(4, dtype=torch.int32, device=dev)
        nbits = torch.zeros(1, dtype=torch.int64, device=dev)
    allw, alln = gather_bands(words, nbits, stride, dst=dst, group=group)
    if rank != dst:
        return None
    return engine.stitch_bands(allw, alln, header)


def encode_frame_restart_banded(engine, src, w: int, h: int, tables, header: bytes, yuv_mode: int,
                                dst: int = 0, group=None) -> Optional[bytes]:
    """One frame coded by all ranks of the group in the optional RESTART mode (sjpeg_hip.h): rank r codes
    the restart intervals of its band into stuffed bytes with their RSTn markers
    (sjpeg_hip_encode_intervals_src); the bands are gathered (gather_streams: one packed buffer per
    rank) and the file is header-with-DRI + bands in order + EOI -- plain concatenation on the host,
    no bit-level stitch.  Byte-identical to the one-device restart-mode stream; NOT the reference's
    bytes (it writes no restart markers), the same pixels.  `tables.flags` must carry
    RESTART_MARKERS, `header` the DRI segment (sjpeg_amd.header_add_restart).  Returns the JPEG on
    `dst`, None elsewhere."""
    import sjpeg_amd as sj
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nseg = sj.segment_count(w, h, yuv_mode)
    b, e = band_ranges(nseg, world)[rank]
    dev = torch.device("cuda", torch.cuda.current_device())
    if e > b:
        out, size = engine.encode_intervals(src, w, h, tables, yuv_mode, b, e)
        out = out[:(out.numel() // 16) * 16].reshape(1, -1)
    else:                                     # fewer intervals than ranks: nothing from this rank
        out = torch.zeros((1, 16), dtype=torch.uint8, device=dev)
        size = torch.zeros(1, dtype=torch.int64, device=dev)
    bands = gather_streams(out, size, [rank], world, dst=dst, group=group)
    if rank != dst:
        return None
    return header + b"".join(bands) + b"\xff\xd9"
