#!/usr/bin/env python3
"""Vector-instruction budget of the GF(256) log-QSPA check node (cn_bp_q256_dc4_kernel, narrow path) per check-wave.

The kernel inlines everything, and its control flow (narrow / wide convolutions, fallbacks) makes a section-mark budget like
tools/isa_budget.py's hard to read; its work, however, is a fixed number of calls of a few building blocks.  This tool compiles
those blocks as separate noinline device functions (tools/bp256_budget.hip), counts the vector instructions of each from the
ISA, reads the convolution loops' bodies from the kernel's own ISA, multiplies by the call counts of one check (dc = 4) and
prints the sum beside the PMC counter of the same kernel.
  python tools/bp256_budget.py [profiles/r03_cfg5_4dB_summary.json]"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing", "-mllvm", "-enable-pre=false", "--offload-arch=gfx950", "-x", "hip",
         "--cuda-device-only", "-S"]


def asm(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc"] + FLAGS + ["-o", out, src], stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def func(lines, prefix):
    a = [i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l][0]
    b = [i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    return lines[a + 1:b]


def valu(body):
    c = Counter()
    for x in body:
        x = x.strip()
        if not x or x.startswith((";", ".")) or x.endswith(":"):
            continue
        c[x.split()[0]] += 1
    return sum(v for k, v in c.items() if k.startswith("v_")), c


def main():
    pmc_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_cfg5_4dB_summary.json")
    P = asm(os.path.join(ROOT, "tools", "bp256_budget.hip"))
    pieces = {n: valu(func(P, n + ":"))[0] for n in ("piece_log", "piece_exp2", "piece_decide", "piece_decide_fp64", "piece_xvec_rest")}
    K = func(asm(os.path.join(ROOT, "nbldpc_amd", "csrc", "nbl_cn_bp256.hip")), "_ZN12_GLOBAL__N_121cn_bp_q256_dc4_kernelILb1EEEv")
    labels = {m.group(1): i for i, l in enumerate(K) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    single = pair = None
    for i, l in enumerate(K):
        m = re.match(r"\s+(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and labels[m.group(2)] < i:
            n, c = valu(K[labels[m.group(2)]:i])
            fma = c["v_fma_f64"] + c["v_fmac_f64_e32"]
            if "v_ldexp_f64" in c or not fma:
                continue
            if fma == 64:
                single = (n, fma)
            elif fma == 128:
                pair = (n, fma)
    rows = [
        ("narrow convolution, one pair of operands (p0 [+] p1, p3 [+] p2): 16 trips of four chunks", 2 * 16 * single[0], 2 * 16 * single[1]),
        ("narrow convolutions that share their first operand (F2 with p2, p3; R1 with p1, p0): 16 trips", 2 * 16 * pair[0], 2 * 16 * pair[1]),
        ("short logarithm of every output symbol and of the two normalisations (18 calls)", 18 * pieces["piece_log"], 0),
        ("exp2 of the four incoming vectors' fractions (16 calls)", 16 * pieces["piece_exp2"], 0),
        ("reference / range / floor of the four incoming vectors (4 calls)", 4 * pieces["piece_xvec_rest"], 0),
        ("hard decisions of the fused variable-node stage (keyed; ~6 per check: one per edge + one per first edge; static size, a vector "
         "without a positive entry leaves after 5)", 6 * pieces["piece_decide"], 0),
    ]
    tot = sum(r[1] for r in rows)
    pmc = None
    if os.path.exists(pmc_path):
        for k, v in json.load(open(pmc_path)).get("sq_pmc", {}).items():
            if "cn_bp_q256" in k and "per_wave" in v:
                pmc = v["per_wave"]["valu_insts"]
    out = [f"vector instructions of cn_bp_q256_dc4_kernel<fused> per check-wave, narrow path (tools/bp256_budget.py)", ""]
    out.append(f"pieces (instructions per call): log {pieces['piece_log']}, exp2 {pieces['piece_exp2']}, keyed decision {pieces['piece_decide']} "
               f"(FP64-reduction decision it replaced: {pieces['piece_decide_fp64']} straight-line), vector preparation {pieces['piece_xvec_rest']}; "
               f"convolution trip of four chunks: {single[0]} ({single[1]} FMAs) / shared-operand pair: {pair[0]} ({pair[1]} FMAs)")
    out.append("")
    for name, n, f in rows:
        out.append(f"{n:7d}  ({f:5d} FMAs)  {name}")
    out.append(f"{tot:7d}  sum of the counted pieces")
    if pmc:
        out.append(f"{pmc:7.0f}  SQ_INSTS_VALU per wave ({os.path.relpath(pmc_path, ROOT)}); the difference is the rest of the fused variable-node stage "
                   f"(sums, damping test, stores), operand staging (ldexp), emit and address arithmetic")
    text = "\n".join(out)
    print(text)
    open(os.path.join(ROOT, "profiles", "r03_bp256_budget.txt"), "w").write(text + "\n")


if __name__ == "__main__":
    main()
