#!/usr/bin/env python3
"""Diagnostic for the one FER anchor that differs (bp_gf16_u256_punct3, 3 dB): which frame, and where do the GPU's log-QSPA and
the oracle's LITERAL restatement of the reference (long double, bit-exact to the compiled reference) part ways?
  CPU part (container or box):  python tools/diag_bp_punct.py oracle   -> gpurun_out/diag_bp_punct_ref.npz
  GPU part:                      python tools/diag_bp_punct.py gpu"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import nbldpc_amd as nb
from nbldpc_amd import hostlib

A = json.load(open(os.path.join(ROOT, "tests", "golden", "fer_anchors.json")))["bp_gf16_u256_punct3"]
P, FRAMES, EBN0 = 8, 63, 3.0
OUT = os.path.join(ROOT, "tools", "diag_bp_punct_ref.npz")  # (not gpurun_out/: that directory does not travel to the GPU box)


def inputs():
    tmp = tempfile.mkdtemp(prefix="diag_")
    hostlib.prepare_workdir(tmp, A["profile"], A["code"], A["constellation"])
    c = nb.datafiles.codes()[A["code"]]
    L, tx, _, _ = hostlib.frontend(tmp, EBN0, FRAMES, c["N"], c["N"] - c["M"], c["q"], P)
    return L, tx


def main():
    mode = sys.argv[1]
    L, tx = inputs()
    B = L.shape[0]
    if mode == "oracle":
        import pyoracle as po
        po.build()
        N, M, q, ev, ec, eh = nb.datafiles.code_edges(A["code"])
        code = po.Code(edges=(N, M, q, ev, ec, eh))
        gf = po.GF(q)
        mk = lambda: po.Decoder(code, gf, po.BP, 20, po.LITERAL)  # noqa: E731
        out, ret, it = po.decode_batch(mk, L, nthreads=8)
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        np.savez(OUT, ret=ret, out=out, it=it)
        err = (out != tx).any(axis=1)
        print("oracle LITERAL: frames", B, "error frames", int(err.sum()), "not converged", int((ret == 0).sum()))
        return
    ref = np.load(OUT)
    dec = nb.Decoder(nb.Code(A["code"]), nb.METHOD_BP, 20)
    for variant in (0, 1):
        dec.lib.nbl_debug_force_generic(dec.h, variant)
        out, conv, it = dec.decode(L)
        err = (out != tx).any(axis=1)
        rerr = (ref["out"] != tx).any(axis=1)
        diff = np.nonzero((out != ref["out"]).any(axis=1) | (conv != ref["ret"]) | (it != ref["it"]))[0]
        print(f"variant {variant}: GPU error frames {int(err.sum())}, oracle {int(rerr.sum())}; frames that differ: {diff.tolist()}")
        for b in diff[:6]:
            print(f"   frame {b}: GPU conv {conv[b]} it {it[b]} symbols wrong {int((out[b] != tx[b]).sum())} | oracle conv {ref['ret'][b]} it {ref['it'][b]} "
                  f"symbols wrong {int((ref['out'][b] != tx[b]).sum())}")
    dec.close()


if __name__ == "__main__":
    main()
