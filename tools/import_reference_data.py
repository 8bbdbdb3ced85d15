#!/usr/bin/env python3
"""Import the reference's DATA files (parity-check matrices, constellations) into our own JSON containers.

Runs only in the build container (reads /root/reference).  Output:
  nbldpc_amd/data/codes.json           name -> {N, M, q, maxdv, maxdc, var_rows, chk_rows}
  nbldpc_amd/data/constellations.json  name -> [[index, real, imag], ...] in file order

These are input data (code definitions and modulation points), not source code: the reference's file
formats are re-emitted on demand by nbldpc_amd.datafiles (write_code_file / write_constellation_file) so the
drop-in harness can read them exactly like the reference does (NBLDPC.cpp:147-205, Comm.cpp:113-126).
GF arithmetic tables are NOT imported: they are generated from the primitive polynomial and only compared
against the reference's Arith.Table files in tests/test_host_data.py.
"""
import glob
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nbldpc_amd", "data")


def read_code(path):
    tok = open(path).read().split()
    it = iter(int(t) for t in tok)
    N, M, q = next(it), next(it), next(it)
    maxdv, maxdc = next(it), next(it)
    dv = [next(it) for _ in range(N)]
    dc = [next(it) for _ in range(M)]
    var_rows = [[[next(it), next(it)] for _ in range(dv[n])] for n in range(N)]
    chk_rows = [[[next(it), next(it)] for _ in range(dc[m])] for m in range(M)]
    rest = list(it)
    assert not rest, f"{path}: {len(rest)} trailing tokens"
    # both directions must describe the same edges
    a = sorted((n + 1, c, h) for n, row in enumerate(var_rows) for c, h in row)
    b = sorted((v, m + 1, h) for m, row in enumerate(chk_rows) for v, h in row)
    assert a == b, path
    return dict(N=N, M=M, q=q, maxdv=maxdv, maxdc=maxdc, var_rows=var_rows, chk_rows=chk_rows)


def read_constellation(path):
    pts = []
    for line in open(path).read().replace("\r", "").split("\n"):
        m = re.match(r"\s*\S+\s+(\d+)\s+\S+\s+(\S+)\s+\S+\s+(\S+)\s*$", line)
        if m:
            pts.append([int(m.group(1)), float(m.group(2)), float(m.group(3))])
    return pts


def main():
    os.makedirs(OUT, exist_ok=True)
    codes = {}
    for p in sorted(glob.glob(os.path.join(REF, "divsalar.*.txt")) + glob.glob(os.path.join(REF, "BDS.*.txt"))):
        codes[os.path.basename(p)[:-4]] = read_code(p)
    with open(os.path.join(OUT, "codes.json"), "w") as f:
        json.dump(codes, f, separators=(",", ":"))
    cons = {}
    for name in ("BPSK", "GRAY_64QAM", "GRAY_256QAM"):
        cons[name] = read_constellation(os.path.join(REF, name + ".txt"))
        assert len(cons[name]) in (2, 64, 256)
    with open(os.path.join(OUT, "constellations.json"), "w") as f:
        json.dump(cons, f, separators=(",", ":"))
    print("codes:", ", ".join(f"{k} (N={v['N']},M={v['M']},q={v['q']})" for k, v in codes.items()))
    print("constellations:", {k: len(v) for k, v in cons.items()})


if __name__ == "__main__":
    main()
