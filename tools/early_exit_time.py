#!/usr/bin/env python3
"""Early-exit decode of one batch (the mode the harness runs in): wall time against the number of (codeword, iteration) pairs that
really ran, for several polling intervals.

usage: python tools/early_exit_time.py [cfg3|cfg2|cfg4] [batch] [EbN0] -- one JSON line per poll_every."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import nbldpc_amd as nb  # noqa: E402
from bench_config import CFG, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    c = CFG[name]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else c["batch"]
    ebn0 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
    dev = torch.device("cuda", 0)
    code = nb.Code(c["code"])
    L = synth(code, B, ebn0, c["mod"], dev).contiguous()
    out = torch.zeros((B, code.N), dtype=torch.int32, device=dev)
    conv = torch.zeros(B, dtype=torch.uint8, device=dev)
    its = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for poll in (1, 2, 5, 10, 0):
        dec = nb.Decoder(code, c["method"], c["iters"], fixed_iters=(1 if poll == 0 else 0), poll_every=poll, max_batch=B, device=0, **c["kw"])
        dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), its.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), its.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        pairs = int(its.sum().item())
        print(json.dumps({"config": name, "batch": B, "ebn0": ebn0, "poll_every": poll if poll else "fixed iterations", "ms_per_batch": dt * 1e3,
                          "codeword_iterations": pairs, "M_pairs_per_s": pairs / dt / 1e6, "converged_frac": float(conv.float().mean().item()),
                          "iterations_per_frame": pairs / B}))
        dec.close()


if __name__ == "__main__":
    main()
