#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the headline bench line).

usage: python tools/bench_config.py cfg1|cfg2|cfg3|cfg3nc2|cfg3nm24|cfg3nm48|cfg3nm64|cfg4|cfg5|tems256|ems64|bp64|ems16|tems16|bp16 [batch] [steps] [ebn0]
Prints one JSON line: codewords/s at fixed iterations with HBM-resident inputs, plus the algorithmic-bytes roofline fraction
(8(q-1)[N + I(N + 4E + D E)] + 4N + 4 bytes per codeword, D = 1 for BP / T-EMS).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nbldpc_amd as nb  # noqa: E402

CFG = {
    "cfg1": dict(code="divsalar.UNBLDPC.128.64.GF.16", method=nb.METHOD_BP, iters=20, batch=65536, kw=dict(), D=1, ebn0=2.5, mod="bpsk"),
    "cfg2": dict(code="divsalar.UNBLDPC.128.64.GF.256", method=nb.METHOD_EMS, iters=50, batch=4096, kw=dict(ems_nm=16, ems_nc=3), D=0, ebn0=2.0, mod="bpsk"),
    "cfg3": dict(code="divsalar.UNBLDPC.512.256.GF.256", method=nb.METHOD_EMS, iters=50, batch=16384, kw=dict(ems_nm=32, ems_nc=3), D=0, ebn0=1.0, mod="bpsk"),
    "cfg3nc2": dict(code="divsalar.UNBLDPC.512.256.GF.256", method=nb.METHOD_EMS, iters=50, batch=4096, kw=dict(ems_nm=32, ems_nc=2), D=0, ebn0=1.0, mod="bpsk"),
    "cfg3nm24": dict(code="divsalar.UNBLDPC.512.256.GF.256", method=nb.METHOD_EMS, iters=50, batch=4096, kw=dict(ems_nm=24, ems_nc=3), D=0, ebn0=1.0, mod="bpsk"),
    "cfg3nm48": dict(code="divsalar.UNBLDPC.512.256.GF.256", method=nb.METHOD_EMS, iters=50, batch=4096, kw=dict(ems_nm=48, ems_nc=3), D=0, ebn0=1.0, mod="bpsk"),
    "cfg3nm64": dict(code="divsalar.UNBLDPC.512.256.GF.256", method=nb.METHOD_EMS, iters=50, batch=4096, kw=dict(ems_nm=64, ems_nc=3), D=0, ebn0=1.0, mod="bpsk"),
    "cfg4": dict(code="BDS.576.288.GF.64", method=nb.METHOD_TEMS, iters=50, batch=8192, kw=dict(tems_nr=2, tems_nc=3), D=1, ebn0=3.0, mod="qam"),
    # shapes without a specialised kernel (general kernels)
    "tems256": dict(code="divsalar.UNBLDPC.512.256.GF.256", method=nb.METHOD_TEMS, iters=50, batch=2048, kw=dict(tems_nr=2, tems_nc=3), D=1, ebn0=1.5, mod="bpsk"),
    "ems64": dict(code="BDS.576.288.GF.64", method=nb.METHOD_EMS, iters=50, batch=4096, kw=dict(ems_nm=16, ems_nc=3), D=0, ebn0=3.0, mod="qam"),
    "bp64": dict(code="BDS.576.288.GF.64", method=nb.METHOD_BP, iters=50, batch=2048, kw=dict(), D=1, ebn0=3.0, mod="qam"),
    "ems16": dict(code="divsalar.UNBLDPC.512.256.GF.16", method=nb.METHOD_EMS, iters=50, batch=8192, kw=dict(ems_nm=8, ems_nc=3), D=0, ebn0=2.0, mod="bpsk"),
    "tems16": dict(code="divsalar.UNBLDPC.512.256.GF.16", method=nb.METHOD_TEMS, iters=50, batch=8192, kw=dict(tems_nr=2, tems_nc=3), D=1, ebn0=2.0, mod="bpsk"),
    "bp16": dict(code="divsalar.UNBLDPC.512.256.GF.16", method=nb.METHOD_BP, iters=50, batch=8192, kw=dict(), D=1, ebn0=2.0, mod="bpsk"),
    "cfg5": dict(code="divsalar.CNBLDPC.512.256.GF.256", method=nb.METHOD_BP, iters=100, batch=1024, kw=dict(), D=1, ebn0=10.0, mod="qam"),
}


def synth(code, B, ebn0, mod, dev):
    q, N = code.q, code.N
    p = q.bit_length() - 1
    gen = torch.Generator(device=dev)
    gen.manual_seed(173)
    if mod == "bpsk":
        sigma = 1.0 / np.sqrt(2 * 1 * 0.5 * 10 ** (ebn0 / 10.0))
        rx = 1.0 + sigma * torch.randn((B, N, p), dtype=torch.float64, device=dev, generator=gen)
        bit = -2.0 * rx / sigma ** 2
        a = torch.arange(1, q, device=dev)
        mask = ((a[:, None] >> torch.arange(p, device=dev)[None, :]) & 1).to(torch.float64)
        return torch.matmul(bit, mask.t()).contiguous()
    # q-ary QAM, all-zero codeword: LLR(a) = ((2r - c0 - ca).(ca - c0)) / (2 sigma^2)   (Comm.cpp:394-395)
    name = {64: "GRAY_64QAM", 256: "GRAY_256QAM"}[q]
    pts = sorted(nb.datafiles.constellations()[name])
    c = torch.tensor([[x[1], x[2]] for x in pts], dtype=torch.float64, device=dev)
    sigma = 1.0 / np.sqrt(2 * p * 0.5 * 10 ** (ebn0 / 10.0))
    r = c[0][None, None, :] + sigma * torch.randn((B, N, 2), dtype=torch.float64, device=dev, generator=gen)
    d = c[1:] - c[0]
    return ((2 * r[:, :, None, :] - c[0] - c[1:]) * d).sum(-1) / (2 * sigma ** 2)


KERNEL = {"cfg1": "cn_bp_small_kernel<16, fused>", "cfg2": "cn_ems_q256_dc4_kernel<16, fused>", "cfg3": "cn_ems_q256_dc4_kernel<32, fused>",
          "cfg3nc2": "cn_ems_q256_dc4_kernel<32, fused, nc 2>", "cfg4": "cn_tems_q64_dc4_kernel<fused, 3>", "cfg5": "cn_bp_q256_dc4_kernel<fused>",
          "tems256": "cn_tems_q256_dc4_kernel<fused, 3>", "ems64": "cn_ems_q64_kernel<fused>", "bp64": "cn_bp_q64_dc4_kernel<fused>",
          "ems16": "cn_ems_small_kernel<16, fused>", "tems16": "cn_tems_small_kernel<16, fused>", "bp16": "cn_bp_small_kernel<16, fused>"}


def run_config(name, B=None, steps=2, ebn0=None, device=0):
    """One configuration at fixed iterations with HBM-resident inputs: codewords/s over `steps` timed decodes (wall clock around
    synchronised launches), the check-node kernel's mean launch time (HIP events on the launch stream) and its algorithmic-bytes
    roofline (8(q-1)(N + 4E + D E) bytes per codeword and iteration, D = 1 for the damped methods)."""
    c = CFG[name]
    B = B or c["batch"]
    if ebn0 is not None:
        c = dict(c, ebn0=float(ebn0))
    dev = torch.device("cuda", device)
    code = nb.Code(c["code"])
    L = synth(code, B, c["ebn0"], c["mod"], dev).contiguous()
    dec = nb.Decoder(code, c["method"], c["iters"], fixed_iters=1, max_batch=B, device=device, **c["kw"])
    out = torch.zeros((B, code.N), dtype=torch.int32, device=dev)
    conv = torch.zeros(B, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), None, st)
    torch.cuda.synchronize()
    dec.profiling(True)
    dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), None, st)
    torch.cuda.synchronize()
    phase_ms, launches = dec.last_timing()
    dec.profiling(False)
    t0 = time.perf_counter()
    for _ in range(steps):
        dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), None, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    q, N, E, I = code.q, code.N, code.E, c["iters"]
    bytes_iter = 8 * (q - 1) * (N + 4 * E + c["D"] * E)
    bytes_cw = 8 * (q - 1) * N + I * bytes_iter + 4 * N + 4
    cws = B * steps / dt
    cn_ms = phase_ms[2] / max(launches[2], 1)
    fused = phase_ms[0] == 0.0
    res = {"config": name, "code": c["code"], "method": c["method"], "iters": I, "batch": B, "codewords_per_s": cws,
           "ms_per_batch": dt / steps * 1e3, "algorithmic_GBps": cws * bytes_cw / 1e9, "hbm_frac": cws * bytes_cw / 8e12,
           "converged_frac": float(conv.float().mean().item()), "ebn0": c["ebn0"],
           "phase_ms": {"vn": phase_ms[0], "syndrome": phase_ms[1], "cn": phase_ms[2]}, "launches": launches,
           "cn_ms_per_launch": cn_ms, "fused": fused, "kernel": KERNEL.get(name, "?") if fused else "(unfused)",
           "cn_algorithmic_bytes_per_launch": B * bytes_iter if fused else None,
           "cn_achieved_GBps": (B * bytes_iter / (cn_ms * 1e-3) / 1e9) if (fused and cn_ms > 0) else None}
    dec.close()
    del L, out, conv
    torch.cuda.empty_cache()
    return res


def main():
    name = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else None
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ebn0 = float(sys.argv[4]) if len(sys.argv) > 4 else None
    print(json.dumps(run_config(name, B, steps, ebn0)))


if __name__ == "__main__":
    main()
