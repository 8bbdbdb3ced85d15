"""Host-side data: GF tables generated from the primitive polynomial, code / constellation file writers."""
import os

import numpy as np
import pytest

import nbldpc_amd.datafiles as df
from nbldpc_amd.shard import shard_range, shard_sizes

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


@pytest.mark.parametrize("q", [4, 8, 16, 32, 64, 128, 256])
def test_gf_field_axioms(q):
    mul, inv = df.gf_tables(q)
    mul = np.array(mul)
    assert np.array_equal(mul, mul.T) and np.all(mul[0] == 0) and np.array_equal(mul[1], np.arange(q))
    for a in range(1, q):
        assert mul[a, inv[a]] == 1
        assert sorted(mul[a, 1:]) == list(range(1, q))  # multiplication by a is a permutation
    a, b, c = np.meshgrid(np.arange(q), np.arange(min(q, 16)), np.arange(min(q, 16)), indexing="ij")
    assert np.array_equal(mul[a, b ^ c], mul[a, b] ^ mul[a, c])  # distributive over XOR
    # alpha = 2 is primitive
    x, seen = 1, set()
    for _ in range(q - 1):
        seen.add(x)
        x = mul[x, 2]
    assert len(seen) == q - 1 and x == 1


def test_oracle_gf_equals_python(oracle):
    for q in (16, 64, 256):
        mul, inv = df.gf_tables(q)
        g = oracle.GF(q)
        assert np.array_equal(g.mul, np.array(mul)) and np.array_equal(g.inv[1:], np.array(inv)[1:])


@needs_ref
@pytest.mark.parametrize("q", [4, 8, 16, 32, 64, 128, 256, 512])
def test_generated_tables_equal_reference_files(tmp_path, q):
    df.write_gf_tables(q, str(tmp_path))
    for stem in ("Arith.Table.GF", "Mat.Repr.GF"):
        ours = open(tmp_path / f"{stem}.{q}.txt").read().split()
        theirs = open(f"{REF}/{stem}.{q}.txt").read().split()
        assert ours == theirs


@needs_ref
def test_code_and_constellation_files_equal_reference(tmp_path):
    for name in df.codes():
        p = df.write_code_file(name, str(tmp_path / (name + ".txt")))
        assert open(p).read().split() == open(f"{REF}/{name}.txt").read().split()
    for name in df.constellations():
        p = df.write_constellation_file(name, str(tmp_path / (name + ".txt")))
        a = [x for x in open(p).read().split() if not x.endswith(":")]
        b = [x for x in open(f"{REF}/{name}.txt").read().split() if not x.endswith(":")]
        assert [float(x) for x in a] == [float(x) for x in b]


def test_code_file_roundtrip_through_oracle_loader(tmp_path, oracle):
    for name, c in df.codes().items():
        p = df.write_code_file(name, str(tmp_path / "c.txt"))
        code = oracle.Code(path=p)
        N, M, q, ev, ec, eh = df.code_edges(name)
        assert (code.N, code.M, code.q, code.E) == (N, M, q, len(ev))
        v, ch, h = code.edge_list()
        assert list(v) == ev and list(ch) == ec and list(h) == eh
        # every check row lists its variables in increasing order in the shipped files (relied on by from_edges)
        for row in c["chk_rows"]:
            vs = [x[0] for x in row]
            assert vs == sorted(vs)


def test_shard_ranges_cover_batch():
    for B in (0, 1, 7, 8, 16384, 16385):
        for W in (1, 2, 3, 8):
            rs = [shard_range(B, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            assert max(shard_sizes(B, W)) - min(shard_sizes(B, W)) <= 1
