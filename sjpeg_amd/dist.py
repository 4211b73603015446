"""Multi-GPU plumbing for the batch path (BASELINE.json config #4): frames are independent
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
