"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo), through the SAME functions bench.py and the GPU two-rank test use
(nbldpc_amd/ranks.py, nbldpc_amd/shard.py): every rank takes its contiguous slice of the lanes of one harness batch, decodes it
with no data-path collective, and the only exchanges are the barrier-fenced MAX of the step time and the gather of per-lane error
counters, summed in lane order like the reference's serial Err() loop (main.cpp:48-51).  Without a GPU the slice is decoded by the
oracle (test infrastructure) -- what is under test is the sharding / gathering protocol, which is identical on the GPU."""
import os
import socket
import sys
import tempfile

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "divsalar.UNBLDPC.128.64.GF.16"
B, EBN0, ITERS = 37, 2.0, 20   # an odd lane count: the two shards differ in size (18 / 19)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _lane_counters(L, tx, lo, hi):
    """[hi-lo][3]: frame error, symbol errors, iterations of lanes lo..hi-1 (BP on the GF(16) code, decoded by the oracle)"""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pyoracle as po
    import nbldpc_amd.datafiles as df
    po.build()
    N, M, q, ev, ec, eh = df.code_edges(CODE)
    dec = po.Decoder(po.Code(edges=(N, M, q, ev, ec, eh)), po.GF(q), po.BP, ITERS, po.CANONICAL)
    rows = []
    for b in range(lo, hi):
        _, out, it = dec.decode(L[b])
        nerr = int((out != tx[b]).sum())
        rows.append([float(nerr > 0), float(nerr), float(it)])
    return np.array(rows, dtype=np.float64).reshape(hi - lo, 3)


def _frames(tmp):
    from nbldpc_amd import hostlib
    import nbldpc_amd.datafiles as df
    c = df.codes()[CODE]
    hostlib.prepare_workdir(tmp, dict(gfq=c["q"], code=CODE, method=1, max_iter=ITERS, parallel=B, constellation="BPSK", random_msg=1,
                                      seed=173), CODE, "BPSK")
    L, tx, _, _ = hostlib.frontend(tmp, EBN0, 1, c["N"], c["N"] - c["M"], c["q"], B)
    return L, tx


def _worker(rank, world, port, tmp, expect):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      NBL_HOST_THREADS="1")
    from nbldpc_amd import ranks
    from nbldpc_amd.shard import shard_range
    rk = ranks.init("gloo")
    assert (rk.rank, rk.world) == (rank, world)
    L, tx = _frames(os.path.join(tmp, f"r{rank}"))   # every rank runs the (deterministic) link chain and keeps its own lanes
    lo, hi = shard_range(B, rk.rank, rk.world)
    local = {}

    def step():
        local["c"] = _lane_counters(L, tx, lo, hi)

    dt = ranks.timed(rk, step, 1)
    assert dt > 0
    allc = ranks.gather_lane_counters(rk, local["c"], B)
    assert allc.shape == (B, 3)
    tot = ranks.sum_in_lane_order(allc)
    assert np.array_equal(allc, expect), "gathered per-lane counters differ from the single-process run"
    assert np.array_equal(tot, ranks.sum_in_lane_order(expect))
    # the elapsed time every rank reports is the same number (MAX over ranks)
    import torch
    t = torch.tensor([dt], dtype=torch.float64)
    parts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    rk.dist.all_gather(parts, t)
    assert all(float(p) == dt for p in parts)
    ranks.finish(rk)


def test_two_rank_sharding_gloo():
    tmp = tempfile.mkdtemp(prefix="nbl_gloo_")
    os.environ["NBL_HOST_THREADS"] = "1"
    L, tx = _frames(os.path.join(tmp, "single"))
    expect = _lane_counters(L, tx, 0, B)
    assert 0 < expect[:, 0].sum() < B, "the batch should hold both failing and converging frames"
    mp.spawn(_worker, args=(2, _free_port(), tmp, expect), nprocs=2, join=True)


def test_shard_ranges_cover_the_batch():
    from nbldpc_amd.shard import shard_range, shard_sizes
    for Bx in (1, 2, 7, 64, 16384):
        for W in (1, 2, 3, 8):
            cuts = [shard_range(Bx, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == Bx
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
            assert sum(shard_sizes(Bx, W)) == Bx and max(shard_sizes(Bx, W)) - min(shard_sizes(Bx, W)) <= 1
