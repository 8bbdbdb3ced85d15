#!/usr/bin/env python3
"""End-to-end rate of the harness (nbldpc_sim): link-chain front-end on the host + batched decode on the GPU + error count.

usage: python tools/sim_throughput.py [cfg3|cfg2|cfg4|cfg1|ems16|tems16] [parallel] [cycles] [ebn0]
Runs the driver in a scratch directory for `cycles` simulation cycles (stop rule on the frame count only), prints its phase summary.
NBL_DEVICE_DEMOD=0 builds the symbol LLRs on the host instead of shipping received samples; NBL_HOST_THREADS sets the front-end threads.
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nbldpc_amd import hostlib  # noqa: E402

CFG = {
    "cfg2": ("divsalar.UNBLDPC.128.64.GF.256", "BPSK", dict(gfq=256, method=2, max_iter=50, nqam=2, ems_nm=16, ems_nc=3)),
    "cfg3": ("divsalar.UNBLDPC.512.256.GF.256", "BPSK", dict(gfq=256, method=2, max_iter=50, nqam=2, ems_nm=32, ems_nc=3)),
    "cfg1": ("divsalar.UNBLDPC.128.64.GF.16", "BPSK", dict(gfq=16, method=1, max_iter=20, nqam=2)),
    "ems16": ("divsalar.UNBLDPC.512.256.GF.16", "BPSK", dict(gfq=16, method=2, max_iter=50, nqam=2, ems_nm=8, ems_nc=3)),
    "tems16": ("divsalar.UNBLDPC.512.256.GF.16", "BPSK", dict(gfq=16, method=4, max_iter=50, nqam=2, tems_nr=2, tems_nc=3)),
    "cfg4": ("BDS.576.288.GF.64", "GRAY_64QAM", dict(gfq=64, method=4, max_iter=50, nqam=64, tems_nr=2, tems_nc=3, random_msg=0)),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    code, cons, kw = CFG[name]
    ebn0 = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
    with tempfile.TemporaryDirectory() as td:
        hostlib.prepare_workdir(td, dict(code=code, parallel=P, snr_begin=ebn0, snr_step=1.0, snr_stop=ebn0, constellation=cons,
                                         min_err_frame=-1, min_uerr_frame=-1, min_sim_cycle=(cycles - 1) * P, seed=173, **kw), code, cons)
        exe = os.path.join(ROOT, "nbldpc_amd", "host", "nbldpc_sim")
        r = subprocess.run([exe], cwd=td, capture_output=True, text=True)
        print(r.stdout[-600:])
        print(r.stderr[-600:])


if __name__ == "__main__":
    main()
