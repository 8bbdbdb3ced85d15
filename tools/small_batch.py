#!/usr/bin/env python3
"""Latency of small batches (the reference's habitual `parallel` of a few lanes): decode time per call vs batch size.
usage: python tools/small_batch.py   (config 3 code, EMS nm=32 nc=3, 50 iterations, early exit as in the reference, host buffers)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import nbldpc_amd as nb  # noqa: E402
from bench import synth_llr, CODE  # noqa: E402

code = nb.Code(CODE)
L = synth_llr(torch, code.q, code.N, 4096, 1.5, 173, torch.device("cuda", 0)).cpu().numpy()
for fixed in (0, 1):
    for B in (1, 8, 64, 512, 4096):
        dec = nb.Decoder(code, nb.METHOD_EMS, 50, ems_nm=32, ems_nc=3, fixed_iters=fixed, poll_every=4, max_batch=B)
        dec.decode(L[:B])
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            out, conv, it = dec.decode(L[:B])
        dt = (time.perf_counter() - t0) / reps
        print(f"fixed={fixed} B={B:5d}  {dt * 1e3:8.3f} ms per call  {B / dt:10.0f} codewords/s  mean iterations {it.mean():.1f}", flush=True)
        dec.close()
