#!/usr/bin/env python3
"""Where is the waterfall?  Convergence fraction and iteration-count quantiles of a configuration over a list of Eb/N0 points
(link-chain frames from nbldpc_amd/host, decoded on the GPU).  usage: python tools/probe_iters.py cfg4|cfg5|cfg3 B ebn0 [ebn0 ...]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import nbldpc_amd as nb  # noqa: E402
from nbldpc_amd import hostlib  # noqa: E402
import nbldpc_amd.datafiles as df  # noqa: E402

CFG = {
    "cfg3": ("divsalar.UNBLDPC.512.256.GF.256", "BPSK", 2, 50, dict(ems_nm=32, ems_nc=3), 1),
    "cfg4": ("BDS.576.288.GF.64", "GRAY_64QAM", 4, 50, dict(tems_nr=2, tems_nc=3), 0),
    "cfg5": ("divsalar.CNBLDPC.512.256.GF.256", "GRAY_256QAM", 1, 100, dict(), 0),
}
name, B = sys.argv[1], int(sys.argv[2])
code_name, cons, method, iters, kw, rm = CFG[name]
c = df.codes()[code_name]
q = c["q"]
code = nb.Code(code_name)
dec = nb.Decoder(code, method, iters, poll_every=5, **kw)
for e in map(float, sys.argv[3:]):
    tmp = tempfile.mkdtemp(prefix="probe_")
    hostlib.prepare_workdir(tmp, dict(gfq=q, code=code_name, method=method, max_iter=iters, parallel=B, nqam=(2 if cons == "BPSK" else q),
                                      constellation=cons, random_msg=rm, seed=173, **kw), code_name, cons)
    L, tx, _, _ = hostlib.frontend(tmp, e, 1, c["N"], c["N"] - c["M"], q, B)
    out, conv, its = dec.decode(L)
    ok = its[conv == 1]
    print(f"{name} Eb/N0 {e}: converged {conv.mean():.3f}, frame errors {(out != tx).any(axis=1).mean():.3f}, iterations of converged frames "
          f"q10/50/90/max = {np.percentile(ok, [10, 50, 90, 100]).tolist() if ok.size else None}", flush=True)
dec.close()
