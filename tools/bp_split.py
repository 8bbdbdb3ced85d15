#!/usr/bin/env python3
"""Diagnostic: which share of the log-QSPA convolutions of config 5 takes the plain-double ("narrow") path and which the
mantissa/exponent ("wide") path, iteration by iteration, and the decode rate with early exit.  Needs the stamps build:
  make -C nbldpc_amd/csrc stamps && NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/libnbldpc_hip_stamps.so python tools/bp_split.py [B] [ebn0]"""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import nbldpc_amd as nb  # noqa: E402
from nbldpc_amd import hostlib  # noqa: E402
import nbldpc_amd.datafiles as df  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ebn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
name, cons = "divsalar.CNBLDPC.512.256.GF.256", "GRAY_256QAM"
c = df.codes()[name]
tmp = tempfile.mkdtemp(prefix="bp_split_")
hostlib.prepare_workdir(tmp, dict(gfq=256, code=name, method=1, max_iter=100, parallel=B, nqam=256, constellation=cons, random_msg=0, seed=173), name, cons)
L, tx, _, _ = hostlib.frontend(tmp, ebn0, 1, c["N"], c["N"] - c["M"], 256, B)
code = nb.Code(name)
prev = (0, 0)
print(f"config 5 at Eb/N0 = {ebn0} dB, {B} frames, FIXED iterations (converged frames keep iterating, as in the throughput runs)")
for it in (1, 2, 3, 4, 6, 8, 12, 20, 40, 100):
    dec = nb.Decoder(code, nb.METHOD_BP, it, fixed_iters=1, max_batch=B)
    lib = dec.lib
    lib.nbl_debug_stamps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    out = (C.c_ulonglong * 16)()
    lib.nbl_debug_stamps(dec.h, 1, None)
    _, conv, _ = dec.decode(L)
    lib.nbl_debug_stamps(dec.h, 0, out)
    dec.close()
    n, w = out[0] - prev[0], out[1] - prev[1]
    print(f"  iterations {it:3d}: cumulative narrow {out[0]:10d} wide {out[1]:10d}; since the previous row {100.0 * w / max(n + w, 1):5.1f} % wide; converged so far {conv.mean():.3f}")
    prev = (out[0], out[1])
for fixed in (1, 0):
    dec = nb.Decoder(code, nb.METHOD_BP, 100, fixed_iters=fixed, poll_every=0 if fixed else 2, max_batch=B)
    dec.decode(L)
    t0 = time.perf_counter()
    _, conv, its = dec.decode(L)
    dt = time.perf_counter() - t0
    dec.close()
    print(f"  {'fixed 100 iterations' if fixed else 'early exit (poll every 2)'}: {B / dt:9.0f} codewords/s (host buffers), mean iterations {its.mean():.1f}, converged {conv.mean():.3f}")
