"""The C-ABI exchange (sjpeg_hip_comm_create + sjpeg_hip_gather_rows / _bytes) with TWO ranks -- on one GPU
if that is all there is (RCCL may refuse two ranks on one device: reported, not a failure of this library),
on two if the node has them.  Rank r codes frames k = r (mod 2) of a small batch; rank 0 checks every
gathered stream against the oracle.  torch.distributed (gloo) only carries the 128-byte unique id.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/exchange_two_ranks.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjpeg_amd as sj  # noqa: E402
from oracle import orc, synth  # noqa: E402
from sjpeg_amd.dist import gather_streams, shard_frames  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
torch.cuda.set_device(rank % ndev)
dist.init_process_group("gloo", rank=rank, world_size=world)
nframes, w, h = 7, 320, 200
ids = shard_frames(nframes, rank, world)
imgs = [synth.g_struct(w, h, 40 + k) for k in ids]
eng = sj.Engine(rank % ndev)
tables, quant = sj.make_tables(quality=75.0)
header = sj.make_header(w, h, sj.YUV_420, quant)
frames = torch.from_numpy(np.stack(imgs)).cuda()
out, sizes = eng.encode_frames(frames, tables, header, sj.YUV_420)
try:
    got = gather_streams(out, sizes, ids, nframes, dst=0)
except sj.SjpegError as e:
    print(f"rank {rank}: {e}", flush=True)
    got = "error"
if rank == 0:
    if got == "error":
        print("two-rank exchange: RCCL did not come up with %d ranks on %d device(s)" % (world, ndev))
    else:
        o = orc.oracle()
        ok = all(got[k] == o.encode(synth.g_struct(w, h, 40 + k), 75.0, 1) for k in range(nframes))
        print("two-rank exchange on %d device(s): %d frames gathered, all equal to the oracle: %s" % (ndev, nframes, ok))
dist.barrier()
dist.destroy_process_group()
