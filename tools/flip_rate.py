#!/usr/bin/env python3
"""How often does the residue the kernels do not reproduce (the reference's running add-then-subtract sum
in ConstructConf, NBLDPC.cpp:1767/1773) change a hard decision?  GPU (canonical value) against the oracle's LITERAL restatement
(bit-identical to the compiled reference) on never-converging EMS frames of the north-star configuration -- the only frames on
which a difference has ever been seen -- until at least `want` of them have been decoded.

  python tools/flip_rate.py [want=2000] [ebn0=0.6] [chunk=384] [ems|tems64|tems256]   (GPU box: ~20 frames/s of oracle on 16 threads)
`tems64`: the same for T-EMS nr=2 nc=3 on the BDS GF(64) code over 64-QAM (BASELINE config 4), where the kernels' dynamic programme
could in principle differ from the reference in the PATH it keeps among equal-cost ones.
Writes gpurun_out/r03_flip_rate[_tems64].json; the committed copies are under profiles/."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import nbldpc_amd as nb
import nbldpc_amd.datafiles as df
from nbldpc_amd import hostlib
import pyoracle as po


def main():
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    ebn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 384
    which = sys.argv[4] if len(sys.argv) > 4 else "ems"
    tems = which in ("tems64", "tems256")
    t256 = which == "tems256"   # T-EMS on the north-star code: the literal enumeration costs ~3 core-seconds per iteration and frame -> 12 iterations
    n_it = 12 if t256 else 50
    name = "BDS.576.288.GF.64" if which == "tems64" else "divsalar.UNBLDPC.512.256.GF.256"
    cons = "GRAY_64QAM" if which == "tems64" else "BPSK"
    po.build()
    c = df.codes()[name]
    N, M, q, ev, ec, eh = df.code_edges(name)
    ocode, ogf = po.Code(edges=(N, M, q, ev, ec, eh)), po.GF(q)
    code = nb.Code(name)
    if tems:
        mk = lambda: po.Decoder(ocode, ogf, po.TEMS, n_it, po.LITERAL, tems_nr=2, tems_nc=3)  # noqa: E731
        dec = nb.Decoder(code, nb.METHOD_TEMS, n_it, tems_nr=2, tems_nc=3, poll_every=4)
        prof = dict(gfq=q, code=name, method=4, max_iter=n_it, tems_nr=2, tems_nc=3, nqam=(2 if t256 else 64), constellation=cons, random_msg=(1 if t256 else 0))
    else:
        mk = lambda: po.Decoder(ocode, ogf, po.EMS, 50, po.LITERAL, ems_nm=32, ems_nc=3)  # noqa: E731
        dec = nb.Decoder(code, nb.METHOD_EMS, 50, ems_nm=32, ems_nc=3, poll_every=5)
        prof = dict(gfq=256, code=name, method=2, max_iter=50, ems_nm=32, ems_nc=3, constellation=cons, random_msg=1)
    tot = dict(frames=0, never_converged=0, flag_mismatches=0, iter_mismatches=0, symbol_diffs=0, frames_with_diffs=0,
               converged_frames_with_diffs=0, symbols_compared_on_never_converging_frames=0)
    chunks = []
    seed = 9000
    t0 = time.time()
    while tot["never_converged"] < want:
        tmp = tempfile.mkdtemp(prefix="flip_")
        hostlib.prepare_workdir(tmp, dict(prof, parallel=chunk, seed=seed), name, cons)
        L, tx, _, _ = hostlib.frontend(tmp, ebn0, 1, c["N"], c["N"] - c["M"], c["q"], chunk)
        out, conv, iters = dec.decode(L)
        l_out, l_conv, l_it = po.decode_batch(mk, L, nthreads=16)
        nc = conv == 0
        d = out != l_out
        ch = dict(seed=seed, frames=chunk, never_converged=int(nc.sum()), flag_mismatches=int((conv != l_conv).sum()),
                  iter_mismatches=int((iters != l_it).sum()), symbol_diffs=int(d.sum()), frames_with_diffs=int(d.any(axis=1).sum()),
                  converged_frames_with_diffs=int(d[~nc].any(axis=1).sum()))
        chunks.append(ch)
        for k in ("frames", "never_converged", "flag_mismatches", "iter_mismatches", "symbol_diffs", "frames_with_diffs", "converged_frames_with_diffs"):
            tot[k] += ch[k]
        tot["symbols_compared_on_never_converging_frames"] += int(nc.sum()) * N
        seed += 1
        print(f"[{time.time() - t0:6.0f}s] {ch}", flush=True)
    dec.close()
    res = dict(config=f"{name}, {'T-EMS nr=2 nc=3' if tems else 'EMS nm=32 nc=3'}, {n_it} iterations, {cons}, Eb/N0 {ebn0} dB; GPU (canonical) vs oracle LITERAL (= compiled reference)",
               total=tot, worst_chunk_symbol_diffs=max(ch["symbol_diffs"] for ch in chunks), chunk_size=chunk, chunks=chunks,
               symbol_flip_rate_on_never_converging_frames=tot["symbol_diffs"] / max(1, tot["symbols_compared_on_never_converging_frames"]),
               frame_rate_on_never_converging_frames=tot["frames_with_diffs"] / max(1, tot["never_converged"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"r03_flip_rate_{which}.json" if tems else "r03_flip_rate.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "chunks"}, indent=1))


if __name__ == "__main__":
    main()
