"""Materialise the text files the drop-in harness reads, in the reference's own formats.

  write_code_file          parity-check file        (parsed by NBLDPC.cpp:147-205 in the reference)
  write_constellation_file "Point: i Real: x Imag: y" (Comm.cpp:113-126)
  write_gf_tables          ./SRC/Arith.Table.GF.<q>.txt and ./SRC/Mat.Repr.GF.<q>.txt (GF.cpp:81-152)

The code and constellation definitions come from nbldpc_amd/data/*.json (imported once from the reference's
data files by tools/import_reference_data.py); the GF tables are GENERATED here from the primitive polynomial.
"""
import json
import os

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# primitive polynomials quoted in the first line of the reference's Arith.Table.GF.<q>.txt
PRIMITIVE_POLY = {4: 7, 8: 11, 16: 19, 32: 37, 64: 67, 128: 137, 256: 285, 512: 529}

_codes = None
_cons = None


def codes():
    global _codes
    if _codes is None:
        with open(os.path.join(_DATA, "codes.json")) as f:
            _codes = json.load(f)
    return _codes


def constellations():
    global _cons
    if _cons is None:
        with open(os.path.join(_DATA, "constellations.json")) as f:
            _cons = json.load(f)
    return _cons


def code_edges(name):
    """(N, M, q, edge_var, edge_chk, edge_h) with edges in var-major order, 0-based."""
    c = codes()[name]
    ev, ec, eh = [], [], []
    for n, row in enumerate(c["var_rows"]):
        for chk, h in row:
            ev.append(n)
            ec.append(chk - 1)
            eh.append(h)
    return c["N"], c["M"], c["q"], ev, ec, eh


def write_code_file(name, path):
    c = codes()[name]
    lines = [f"{c['N']} {c['M']} {c['q']}", f"{c['maxdv']} {c['maxdc']}",
             " ".join(str(len(r)) for r in c["var_rows"]) + " ",
             " ".join(str(len(r)) for r in c["chk_rows"]) + " "]
    for row in c["var_rows"] + c["chk_rows"]:
        lines.append(" ".join(f"{a} {h}" for a, h in row) + " ")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def write_constellation_file(name, path):
    pts = constellations()[name]
    with open(path, "w") as f:
        f.write("\n".join(f"Point:\t{i}\tReal:\t{re!r}\tImag:\t{im!r}" for i, re, im in pts))
    return path


def gf_tables(q):
    """(mul[q][q], inv[q]) of GF(q) in the polynomial basis of PRIMITIVE_POLY[q]; add is XOR."""
    poly = PRIMITIVE_POLY[q]
    p = q.bit_length() - 1
    mul = [[0] * q for _ in range(q)]
    inv = [0] * q
    for a in range(q):
        for b in range(a, q):
            acc, x = 0, a
            for i in range(p):
                if (b >> i) & 1:
                    acc ^= x
                x <<= 1
                if x & q:
                    x ^= poly
            mul[a][b] = mul[b][a] = acc
            if acc == 1:
                inv[a], inv[b] = b, a
    return mul, inv


def write_gf_tables(q, src_dir):
    """Write Arith.Table.GF.<q>.txt and Mat.Repr.GF.<q>.txt under src_dir (the reference expects ./SRC/)."""
    os.makedirs(src_dir, exist_ok=True)
    mul, inv = gf_tables(q)
    poly = PRIMITIVE_POLY[q]
    p = q.bit_length() - 1
    with open(os.path.join(src_dir, f"Arith.Table.GF.{q}.txt"), "w") as f:
        f.write(f"GF({q}) with Primitive Polynomial: {poly}. \nMultiply Table:\n")
        for a in range(q):
            f.write(" ".join(map(str, mul[a])) + " \n")
        f.write("Add Table:\n")
        for a in range(q):
            f.write(" ".join(str(a ^ b) for b in range(q)) + " \n")
        f.write("Inverse Table:\n" + " ".join(map(str, inv)) + " \n")
    # companion-matrix powers: row i of A^k holds the coordinates of alpha^(k+i) (alpha = 2)
    with open(os.path.join(src_dir, f"Mat.Repr.GF.{q}.txt"), "w") as f:
        f.write(f"GF({q}) with Primitive Polynomial: {poly} \n")
        x = 1
        for k in range(q - 1):
            f.write(f"A^{k} --> order: {k}\tpoly: {x}\n")
            y = x
            for _ in range(p):
                f.write(" ".join(str((y >> j) & 1) for j in range(p)) + " \n")
                y = mul[y][2]
            x = mul[x][2]
    return src_dir


def materialise(dirpath, q, code_name, constellation_name):
    """Everything one profile needs, laid out like the reference's working directory."""
    os.makedirs(dirpath, exist_ok=True)
    write_gf_tables(q, os.path.join(dirpath, "SRC"))
    write_code_file(code_name, os.path.join(dirpath, code_name + ".txt"))
    write_constellation_file(constellation_name, os.path.join(dirpath, constellation_name + ".txt"))
    return dirpath
