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
    """What `dst` holds after gather_streams(): ONE device buffer with every rank's packed block at
    rank_offsets[r] (exact lengths, rank order) and the rows {bytes, frames, sizes...} of all ranks;
    frames() brings the streams to the host as byte strings in global order."""

    def __init__(self, gathered, rows, rank_offsets, nframes, world):
        self.gathered, self.rows, self.rank_offsets = gathered, rows, rank_offsets
        self.nframes, self.world = nframes, world

    def frames(self) -> List[bytes]:
        out: List[Optional[bytes]] = [None] * self.nframes
        total = int(self.rank_offsets[self.world])
        host = self.gathered[:total].cpu().numpy()              # one device -> host copy
        for r in range(self.world):
            ids = shard_frames(self.nframes, r, self.world)
            o = int(self.rank_offsets[r])
            for j, k in enumerate(ids):
                n = int(self.rows[r][2 + j])
                out[k] = host[o:o + n].tobytes()
                o += _align16(n)
        return out  # type: ignore[return-value]


_COMMS = {}


def rccl_comm(group=None):
    """The library's own RCCL communicator for `group` (sjpeg_hip_comm_create): rank 0 makes the unique
    id, torch.distributed carries its 128 bytes to the others -- the only thing torch does for the
    exchange on the GPU path.  Cached per group."""
    import sjpeg_amd as sj
    key = id(group) if group is not None else 0
    if key not in _COMMS:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [sj.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        _COMMS[key] = sj.Comm(box[0], rank, world)
    return _COMMS[key]


def _check_rows(rows, per_max):
    """The decision every rank takes on the same rows (the C function's rule, sjpeg_hip.h)."""
    total = 0
    offs = []
    for rk in rows:
        offs.append(total)
        total += int(rk[0])
        nf = int(rk[1])
        sizes = [int(x) for x in rk[2:2 + nf]]
        if nf > per_max or any(n == 0 for n in sizes) or sum(_align16(n) for n in sizes) != int(rk[0]):
            raise _capacity_error("a frame of size 0 (it did not fit its output slot or the packed buffer)")
    offs.append(total)
    return offs


def _capacity_error(msg):
    import sjpeg_amd as sj
    return sj.SjpegError("gather_streams: " + msg + " -- nothing was sent")


def gather_streams(out: torch.Tensor, sizes: torch.Tensor, frame_ids: Sequence[int],
                   nframes: int, dst: int = 0, group=None, compact=None,
                   to_host: bool = True, buffers: Optional[dict] = None, reuse_gathered: bool = True,
                   packed_offsets: Optional[torch.Tensor] = None):
    """Gathers variable-length coded frames to `dst`: the exchange step of the batch path.

    out [F_local, stride] uint8 and sizes [F_local] int64 as an encode call left them, frame_ids the
    global index of each local frame.  The frames are packed back to back ON THE DEVICE by one kernel
    (sjpeg_hip_compact_streams: every frame at a multiple of 16, no host round trip); then the protocol
    of sjpeg_hip_gather_streams (include/sjpeg_hip.h): all-gather of one row {bytes, frames, sizes...}
    per rank, ONE small host read of the rows (RCCL's counts are host values), and transfers of EXACT
    lengths into one buffer on `dst` -- no padding to the largest rank.  CUDA tensors go through the
    C-ABI (the library's own RCCL communicator, rccl_comm()); CPU tensors (the gloo tests) through the
    same steps written with torch.distributed, `compact(out, sizes, n, capacity)` -> (packed, offsets)
    standing in for the kernel.  A frame of size 0 raises on every rank before anything is sent; the root
    sizes its buffer from the actual total.  `buffers`: a dict the call keeps its device buffers in (reuse across
    steps); with to_host=False the RESULT lives in the "gathered" buffer, so a caller that keeps the results of
    several calls passes reuse_gathered=False (a fresh buffer per call; only the scratch is reused).
    packed_offsets ([n_local + 1] int64): `out` is ALREADY packed -- a flat buffer written by
    Engine.encode_frames_packed (sjpeg_hip_encode_scan_packed_src) -- and no compaction pass runs; on rank 0 as
    `dst` the other ranks' streams are then received BEHIND its own in that same buffer when it is long enough
    (and reuse_gathered): the root moves none of its own bytes.
    Returns on `dst` the nframes byte strings in global order (to_host=True) or a GatheredStreams holding
    the device-resident buffer (to_host=False); None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = out.device
    n_local = len(frame_ids)
    per_max = (nframes + world - 1) // world
    stride = int(out.stride(0)) if out.dim() == 2 else int(out.numel())
    buffers = buffers if buffers is not None else {}

    def buf(name, n, dtype):
        t = buffers.get(name)
        if t is None or t.numel() < n or t.dtype != dtype or t.device != dev:
            t = torch.empty(int(n), dtype=dtype, device=dev)
            buffers[name] = t
        return t

    if dev.type == "cuda":
        import sjpeg_amd as sj
        if packed_offsets is not None:
            packed, offsets = out.reshape(-1), packed_offsets
        else:
            packed = buf("packed", max(n_local, 1) * _align16(stride), torch.uint8)
            offsets = buf("offsets", per_max + 1, torch.int64)
            if n_local > 0:
                sj.compact_streams(out, sizes, n_local, packed=packed, offsets=offsets[:n_local + 1])
            else:
                offsets[:1].zero_()
        rows_dev = buf("rows", (world + 1) * (per_max + 2), torch.int64)
        comm = rccl_comm(group)
        rows, offs = comm.gather_rows(offsets, sizes, n_local, per_max, rows_dev)
        # the root sizes its buffer from the actual total (kept between steps, grown by halves)
        total = int(offs[world])
        gathered = None
        if rank == dst and packed_offsets is not None and reuse_gathered and int(offs[dst]) == 0 and packed.numel() >= total:
            gathered = packed                       # the root's own streams are where they belong already
        elif rank == dst and not reuse_gathered:
            gathered = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
        elif rank == dst:
            have = buffers.get("gathered")
            gathered = buf("gathered", max(total, 16) if have is not None and have.numel() >= total else total + total // 2 + 16,
                           torch.uint8)
        comm.gather_bytes(dst, packed, per_max, rows, offs, gathered)
        if rank != dst:
            return None
        got = GatheredStreams(gathered, rows.astype(np.int64), offs.astype(np.int64), nframes, world)
        return got.frames() if to_host else got

    # ---- the same protocol on torch.distributed (gloo, CPU tests)
    if packed_offsets is not None:
        # `out` is packed already (the stand-in of encode_frames_packed): no compaction pass here either
        packed, offsets = out.reshape(-1), packed_offsets
        my_bytes = int(offsets[n_local])
    elif n_local > 0:
        packed, offsets = compact(out, sizes, n_local, n_local * _align16(stride))
        my_bytes = int(offsets[n_local])
    else:
        packed, my_bytes = torch.zeros(16, dtype=torch.uint8, device=dev), 0
    row = torch.zeros(per_max + 2, dtype=torch.int64, device=dev)
    row[0], row[1] = my_bytes, n_local
    row[2:2 + n_local] = sizes[:n_local]
    all_rows = [torch.zeros_like(row) for _ in range(world)]
    dist.all_gather(all_rows, row, group=group)
    rows = torch.stack(all_rows).cpu().numpy()                   # the one host read
    offs = _check_rows(rows, per_max)
    if rank != dst:
        if my_bytes > 0:
            dist.send(packed[:my_bytes].contiguous(), dst=dst, group=group)
        return None
    in_place = (packed_offsets is not None and reuse_gathered and offs[dst] == 0 and packed.numel() >= offs[world])
    if in_place:
        gathered = packed                           # the root's own streams are where they belong already
    else:
        gathered = (buf("gathered", max(offs[world], 16), torch.uint8) if reuse_gathered
                    else torch.empty(max(offs[world], 16), dtype=torch.uint8, device=dev))
    reqs = []
    for r in range(world):
        n = int(rows[r][0])
        if n == 0:
            continue
        if r == dst:
            if not in_place:
                gathered[offs[r]:offs[r] + n] = packed[:n]
        else:
            reqs.append(dist.irecv(gathered[offs[r]:offs[r] + n], src=r, group=group))
    for q in reqs:
        q.wait()
    got = GatheredStreams(gathered, rows, offs, nframes, world)
    return got.frames() if to_host else got


def sink_streams_local(packed: torch.Tensor, offsets: torch.Tensor, n_local: int, host: torch.Tensor):
    """The NON-ROOTED end of the batch path ("spread sinks"): every rank brings the streams IT
    coded to ITS OWN host buffer -- one device-to-host copy of the packed block over the rank's own PCIe link --
    instead of funnelling them through rank 0's xGMI links and rank 0's one PCIe link.  No collective at all: a
    consumer that needs the whole batch in one place reads the ranks' host buffers (one node: shared memory).
    `packed` / `offsets` as Engine.encode_frames_packed (or compact_streams) left them, `host` a pinned uint8
    tensor.  One small read of the offsets (the byte count is a host value for the copy), then the copy,
    asynchronous on the current stream.  Returns (bytes, offsets on the host); frame k of this rank is
    host[offsets[k] : offsets[k] + sizes[k]] once the stream has been waited for."""
    offs = offsets[:n_local + 1].cpu()
    total = int(offs[n_local])
    if total < 0 or total > host.numel():
        raise _capacity_error(f"sink_streams_local: {total & ((1 << 63) - 1)} bytes, host buffer {host.numel()}")
    if total > 0:
        host[:total].copy_(packed.reshape(-1)[:total], non_blocking=True)
    return total, offs


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
                  use_streams: bool, dst: int = 0, group=None, compact=None, keep: str = "all",
                  packed_offsets=None):
    """bench.py's timed multi-rank region, as a function so that the CPU/gloo test runs exactly
    this code: `nsteps` encode calls, double buffered (outs[b], sizes[b], b = 0 / 1), the streams
    of every step gathered to `dst` (device resident there) under the next step's kernels.
    Returns the GatheredStreams of every step on `dst` (keep="last": of the last step only), a list of None
    elsewhere.  The exchange's scratch (packed block, offsets, rows) is held per output set and reused; the
    buffer a result lives in is reused only with keep="last" -- with keep="all" every step gets its own, or the
    streams of step s would be overwritten by step s + 2.  packed_offsets (two tensors, one per output set):
    encode(b) wrote PACKED output into outs[b] (Engine.encode_frames_packed): no compaction pass, and a root that
    is rank 0 receives the others behind its own streams in outs[b] itself."""
    held = [{}, {}]                                 # device buffers of the exchange, one set per output set
    return overlapped_steps(
        nsteps, encode,
        lambda b: gather_streams(outs[b], sizes[b], frame_ids, nframes, dst=dst, group=group,
                                 compact=compact, to_host=False, buffers=held[b], reuse_gathered=(keep == "last"),
                                 packed_offsets=None if packed_offsets is None else packed_offsets[b]),
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
