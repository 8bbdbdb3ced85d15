"""The C++ host layer (nbldpc_amd/host: CSimulation / CComm / CNBLDPC::Encode / CRand) against what the COMPILED
REFERENCE's own link chain produced (tests/golden): channel LLRs bit for bit, transmitted codewords, messages."""
import numpy as np
import pytest

from conftest import load_golden
import nbldpc_amd.datafiles as df
from nbldpc_amd import hostlib

SETS = ["cfg1_bp_gf16", "cfg2_ems_u128", "cfg3_ems_u512", "ems_nc2_shaped", "cfg4_tems_bds", "cfg5_bp_c512"]


@pytest.mark.parametrize("name", SETS)
def test_frontend_bit_exact(tmp_path, name):
    g, meta = load_golden(name)
    p = meta["profile"]
    hostlib.prepare_workdir(str(tmp_path), p, meta["code"], meta["constellation"])
    c = df.codes()[meta["code"]]
    L, tx, msg, sigma = hostlib.frontend(str(tmp_path), meta["ebn0"], meta["frames"], c["N"], c["N"] - c["M"], c["q"], p["parallel"])
    assert sigma == g["sigma"][0]
    assert np.array_equal(tx, g["tx_code"])
    assert np.array_equal(msg, g["tx_msg"])
    assert np.array_equal(L, g["L_ch"])  # bit-identical doubles


def test_encoder_produces_codewords(tmp_path):
    name = "divsalar.UNBLDPC.512.256.GF.256"
    hostlib.prepare_workdir(str(tmp_path), dict(gfq=256, code=name, method=2, constellation="BPSK", random_msg=1), name, "BPSK")
    c = df.codes()[name]
    N, M, q = c["N"], c["M"], c["q"]
    rng = np.random.default_rng(3)
    msgs = rng.integers(0, q, (32, N - M))
    cw = hostlib.encode(str(tmp_path), msgs, N)
    mul = np.array(df.gf_tables(q)[0])
    for m, row in enumerate(c["chk_rows"]):
        s = np.zeros(32, dtype=np.int64)
        for v, h in row:
            s ^= mul[h, cw[:, v - 1]]
        assert not s.any(), m
    # linear: the all-zero message gives the all-zero word
    assert not hostlib.encode(str(tmp_path), np.zeros((1, N - M), dtype=np.int32), N).any()


def test_profile_grammar_is_positional(tmp_path):
    """Label words are free text; only their count matters (Simulation.cpp:58-107)."""
    from nbldpc_amd.profiles import profile_text
    name = "divsalar.UNBLDPC.128.64.GF.16"
    hostlib.prepare_workdir(str(tmp_path), dict(gfq=16, code=name, method=2, ems_nm=8, constellation="BPSK", parallel=2, random_msg=1), name, "BPSK")
    a = hostlib.frontend(str(tmp_path), 2.0, 1, 32, 16, 16, 2)
    text = open(tmp_path / "NBLDPC.Profile.txt").read()
    garbled = "\n".join(" ".join(("x" + w if i < len(line.split()) - 1 else w) for i, w in enumerate(line.split())) for line in text.splitlines())
    open(tmp_path / "NBLDPC.Profile.txt", "w").write(garbled + "\n")
    b = hostlib.frontend(str(tmp_path), 2.0, 1, 32, 16, 16, 2)
    assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
