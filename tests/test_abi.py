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
