"""world_size-2 gloo test of the multi-GPU plumbing (sjpeg_amd/dist.py): frame sharding and the
gather of variable-length coded frames to rank 0.  Payloads are oracle-coded frames so the
assembled batch is checked against the real expected bytes.  CPU only."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nframes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc, synth
    from sjpeg_amd.dist import gather_streams, shard_frames
    o = orc.oracle()
    ids = shard_frames(nframes, rank, world)
    coded = [o.encode(synth.g_struct(48, 40, 100 + k), 75.0, 1) for k in ids]
    stride = max([len(c) for c in coded] + [1]) + 13
    out = torch.zeros((max(len(ids), 1), stride), dtype=torch.uint8)
    sizes = torch.zeros(max(len(ids), 1), dtype=torch.int64)
    for i, c in enumerate(coded):
        out[i, :len(c)] = torch.from_numpy(np.frombuffer(c, np.uint8).copy())
        sizes[i] = len(c)
    frames = gather_streams(out, sizes, ids, nframes, dst=0)
    if rank == 0:
        want = [o.encode(synth.g_struct(48, 40, 100 + k), 75.0, 1) for k in range(nframes)]
        q.put(frames == want)
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()


def _run(nframes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nframes, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_shard_assignment():
    from sjpeg_amd.dist import shard_frames
    assert shard_frames(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((shard_frames(7, r, 2) for r in range(2)), [])) == list(range(7))
    assert shard_frames(1, 1, 2) == []


def test_gather_even():
    _run(6)


def test_gather_ragged_and_empty_rank():
    _run(5)
    _run(1)
