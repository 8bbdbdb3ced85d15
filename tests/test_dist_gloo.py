"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): the batch is split with shard_range, every rank works
on its own slice with no data-path collective, the only exchanges are the ones bench.py / the harness make --
a MAX all-reduce of the step time and a gather of per-rank error counters summed in lane order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nbldpc_amd.shard import shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(B, rank, world)
    # stand-in for the decode of this rank's slice: per-frame error counts that depend only on the global frame index
    frames = np.arange(lo, hi)
    err_sym = torch.tensor((frames * 7 + 3) % 5, dtype=torch.float64)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(t.item() - 0.1 * world) < 1e-12
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo]))
    assert sum(int(s) for s in sizes) == B
    parts = [torch.zeros(int(s), dtype=torch.float64) for s in sizes]
    dist.all_gather(parts, err_sym) if len(set(int(s) for s in sizes)) == 1 else None
    if len(set(int(s) for s in sizes)) == 1:
        total = torch.cat(parts)  # lane order, like the serial Err loop (main.cpp:48-51)
        assert torch.equal(total, torch.tensor((np.arange(B) * 7 + 3) % 5, dtype=torch.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 64, 16), nprocs=2, join=True)
