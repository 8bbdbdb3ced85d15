"""One process per GPU: the few exchanges the sharded decode path makes (main.cpp:46-51).

Frames are independent, so ranks never exchange data on the decode path.  What remains is
  * start-up: rank / world size from the launcher's environment, the process group (backend "nccl" = RCCL on ROCm; "gloo" for
    CPU rehearsals and for two ranks sharing one GPU),
  * timing: barrier + device sync on both sides of the timed steps and the MAX of the elapsed time over ranks,
  * results: per-lane error counters gathered in lane order -- the order in which the reference's serial `Err()` loop
    (main.cpp:48-51, Comm.cpp:446-503) adds them up -- so the sums are the ones a single process would get.
bench.py, the tests (tests/test_dist_gloo.py on gloo, tests/test_gpu_ranks.py on the GPU box) all go through these functions.
"""
import os
import time

import numpy as np

from .shard import shard_range, shard_sizes


class Ranks:
    def __init__(self, rank=0, world=1, local_rank=0, dist=None, backend=None):
        self.rank, self.world, self.local_rank, self.dist, self.backend = rank, world, local_rank, dist, backend

    @property
    def tensor_device(self):
        """where the small control tensors of the collectives live"""
        return f"cuda:{self.local_rank}" if self.backend == "nccl" else "cpu"


def init(backend="nccl", same_device=False, init_group=True):
    """Read RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run sets them) and join the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 or not init_group:
        return Ranks(rank, world, local_rank, None, None)
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return Ranks(rank, world, local_rank, dist, backend)


def finish(rk):
    if rk.dist is not None:
        rk.dist.destroy_process_group()


def timed(rk, step, steps, sync=lambda: None):
    """Run `step` exactly `steps` times between two (barrier + device sync) fences; return the MAX over ranks of the elapsed seconds."""
    sync()
    if rk.dist is not None:
        rk.dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if rk.dist is not None:
        rk.dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if rk.dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=rk.tensor_device)
        rk.dist.all_reduce(t, op=rk.dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def gather_lane_counters(rk, local, B):
    """`local` [hi-lo][k]: per-lane counters of this rank's slice shard_range(B, rank, world).  Returns [B][k] in lane order on
    every rank (all_gather of slices padded to the largest shard; no reduction on the wire, the caller sums in lane order)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if local.ndim == 1:
        local = local[:, None]
    lo, hi = shard_range(B, rk.rank, rk.world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rk.rank} holds {local.shape[0]} lanes, its shard has {hi - lo}")
    if rk.dist is None:
        return local
    import torch
    sizes = shard_sizes(B, rk.world)
    pad = max(sizes)
    buf = torch.zeros((pad, local.shape[1]), dtype=torch.float64, device=rk.tensor_device)
    buf[: hi - lo] = torch.from_numpy(local).to(buf.device)
    parts = [torch.zeros_like(buf) for _ in range(rk.world)]
    rk.dist.all_gather(parts, buf)
    return np.concatenate([p[:n].cpu().numpy() for p, n in zip(parts, sizes)], axis=0)


def sum_in_lane_order(counters):
    """Column sums taken lane after lane (the counters are integer-valued doubles; the order is the reference's)."""
    tot = np.zeros(counters.shape[1], dtype=np.float64)
    for row in counters:
        tot += row
    return tot
