"""The C-ABI shared library: loads, exports every symbol include/nbldpc.h declares, and rejects bad arguments
with the reference's error semantics -- all without touching a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nbldpc_amd as nb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbldpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbl_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = nb.load_library()
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nbldpc.h but not exported"
    assert sorted(nb.EXPORTS) == syms
    assert lib.nbl_abi_version() == 1


def test_no_oracle_in_product_library():
    """The product must not link or embed the CPU checker."""
    blob = open(nb.LIB_PATH, "rb").read()
    assert b"nblo_" not in blob and b"liboracle" not in blob
    for root, _, files in os.walk(os.path.join(ROOT, "nbldpc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "pyoracle" not in src and "nbl_oracle" not in src, f


def _create(code, **kw):
    return nb.Decoder(code, **kw)


def test_create_rejects_bad_arguments_before_touching_the_device():
    code = nb.Code("divsalar.UNBLDPC.128.64.GF.16")
    with pytest.raises(nb.NblError) as e:
        _create(code, method=nb.METHOD_EMS, max_iter=5, ems_nm=17)  # NBLDPC.cpp:282-286
    assert e.value.status == -1 and "EMS_Nm is too large" in str(e.value)
    for method in (3, 5, 6, 7, 0):  # Min-Max, T-Min-Max: "has not been developed"; OSD / BS-TEMS out of scope
        with pytest.raises(nb.NblError) as e:
            _create(code, method=method, max_iter=5)
        assert e.value.status == -2
    # inconsistent graph: variable side and check side disagree
    bad = nb.Code("divsalar.UNBLDPC.128.64.GF.16")
    bad.var_h = bad.var_h.copy()
    bad.var_h[0] ^= 1
    with pytest.raises(nb.NblError) as e:
        _create(bad, method=nb.METHOD_EMS, max_iter=5, ems_nm=8)
    assert e.value.status == -1
    # GF(512): the reference ships arithmetic tables for it but no code; a valid request this library cannot serve (-2), not a
    # malformed one (-1)
    big = _ring_code(512, 8, 4)
    with pytest.raises(nb.NblError) as e:
        _create(big, method=nb.METHOD_EMS, max_iter=5, ems_nm=8, gf=(np.zeros((512, 512), np.uint16), np.zeros(512, np.uint16)))
    assert e.value.status == -2 and "GF(256)" in str(e.value)
    # GF table that is not a field table
    mul, inv = nb.datafiles.gf_tables(16)
    mul = [row[:] for row in mul]
    mul[3][2] ^= 1
    with pytest.raises(nb.NblError):
        _create(code, method=nb.METHOD_EMS, max_iter=5, ems_nm=8, gf=(mul, inv))


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    code = nb.Code("divsalar.UNBLDPC.128.64.GF.16")
    with pytest.raises(nb.NblError) as e:
        _create(code, method=nb.METHOD_EMS, max_iter=5, ems_nm=8)
    assert e.value.status == -3 and "no CPU decode path" in str(e.value)


def _ring_code(q, M, dc):
    """A synthetic (2, dc)-regular graph (dc even): M checks, N = M dc / 2 variables; variable n joins checks n % M and
    (n % M + 1 + n // M) % M -- two different checks, every check gets dc / 2 variables from each rule."""
    assert dc % 2 == 0 and M > dc // 2
    N = M * dc // 2
    chk_rows = [[] for _ in range(M)]
    var_rows = [[] for _ in range(N)]
    for n in range(N):
        for m in (n % M, (n % M + 1 + n // M) % M):
            h = 1 + (7 * n + 3 * m) % (q - 1)
            var_rows[n].append((m + 1, h))
            chk_rows[m].append((n + 1, h))
    return nb.Code(spec=dict(N=N, M=M, q=q, var_rows=var_rows, chk_rows=chk_rows))


def test_create_refuses_shapes_the_kernels_cannot_run():
    """Limits of the kernels are refused by nbl_create with NBL_ERR_UNSUPPORTED and a message, not at the first decode."""
    code = _ring_code(256, 8, 6)                       # GF(256), check degree 6: 8 * 6 > 32
    assert code.chk_deg.max() == 6 and code.var_deg.max() == 2
    with pytest.raises(nb.NblError) as e:
        _create(code, method=nb.METHOD_TEMS, max_iter=5, tems_nr=2, tems_nc=2)
    assert e.value.status == -2 and "must not exceed 32" in str(e.value)
    # (EMS at the largest supported shape -- GF(256), check degree 8, nm = q, nc = 6: 71 KB of LDS -- is accepted and run by
    #  tests/test_gpu_parity.py::test_generic_ems_beyond_64k_lds)
