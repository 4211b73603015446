"""Multi-GPU plumbing.

Batch path (BASELINE.json config #4): frames are independent
objects, so the hot path shards by frame with NO data-path collective; the only exchange is
the gather of the finished per-frame byte streams to rank 0 (RCCL over xGMI on the GPU box,
gloo in CPU tests), followed by a host-side concatenate.

One process per GPU, torch.distributed already initialised by the caller.  torch is plumbing
here (process group + device tensors); the encoder itself is the C library.
"""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_frames(nframes: int, rank: int, world: int) -> List[int]:
    """Frame k goes to rank k % world: the global indices this rank codes."""
    return list(range(rank, nframes, world))


def gather_streams(out: torch.Tensor, sizes: torch.Tensor, frame_ids: Sequence[int],
                   nframes: int, dst: int = 0, group=None) -> Optional[List[bytes]]:
    """Gathers variable-length coded frames to `dst`.

    out   [F_local, stride] uint8, sizes [F_local] int64 (same device as the process group
    backend expects), frame_ids the global index of each local frame.  Two collectives:
    all_gather of the sizes (8 B per frame), then one padded gather of the compacted byte
    streams (RCCL has no gatherv; padding is to the largest per-rank total).  Returns the
    nframes byte strings in global frame order on `dst`, None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = out.device
    max_local = (nframes + world - 1) // world
    local_sizes = torch.zeros(max_local, dtype=torch.int64, device=dev)
    local_sizes[:len(frame_ids)] = sizes[:len(frame_ids)]
    all_sizes = [torch.zeros_like(local_sizes) for _ in range(world)]
    dist.all_gather(all_sizes, local_sizes, group=group)
    totals = [int(s.sum().item()) for s in all_sizes]
    pad = max(max(totals), 1)
    # compact this rank's frames back to back
    packed = torch.zeros(pad, dtype=torch.uint8, device=dev)
    pos = 0
    for i in range(len(frame_ids)):
        n = int(sizes[i].item())
        packed[pos:pos + n] = out[i, :n]
        pos += n
    recv = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(world)] \
        if rank == dst else None
    dist.gather(packed, recv, dst=dst, group=group)
    if rank != dst:
        return None
    frames: List[Optional[bytes]] = [None] * nframes
    for r in range(world):
        buf = recv[r].cpu().numpy()
        sz = all_sizes[r].cpu().numpy()
        pos = 0
        for j, k in enumerate(shard_frames(nframes, r, world)):
            n = int(sz[j])
            frames[k] = buf[pos:pos + n].tobytes()
            pos += n
    return frames  # type: ignore[return-value]


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
