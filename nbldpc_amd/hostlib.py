"""ctypes access to nbldpc_amd/host/libnbldpc_host.so (the reference-compatible C++ host layer) for tests."""
import contextlib
import ctypes as C
import os

import numpy as np

from . import datafiles

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "host", "libnbldpc_host.so")
SIM_BIN = os.path.join(_HERE, "host", "nbldpc_sim")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            raise FileNotFoundError(f"{HOST_LIB} is missing: make -C nbldpc_amd/host")
        try:
            import torch  # noqa: F401  (same reason as in binding.load_library: one HIP runtime per process)
        except Exception:
            pass
        L = C.CDLL(HOST_LIB)
        L.nblh_frontend.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.nblh_channel.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.nblh_simulate.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
        L.nblh_encode.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


@contextlib.contextmanager
def workdir(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def prepare_workdir(path, profile_kwargs, code_name, constellation_name):
    """Lay out SRC/, the code file, the constellation file and NBLDPC.Profile.txt like the reference's working directory."""
    from .profiles import profile_text
    kw = dict(profile_kwargs)
    datafiles.materialise(path, kw["gfq"], code_name, constellation_name)
    kw["code"] = code_name + ".txt"
    kw["constellation"] = constellation_name + ".txt"
    prof = os.path.join(path, "NBLDPC.Profile.txt")
    with open(prof, "w") as f:
        f.write(profile_text(**kw))
    return prof


def frontend(workdir_path, ebn0, frames, N, K, q, P):
    B = frames * P
    L = np.zeros((B, N, q - 1))
    tx = np.zeros((B, N), dtype=np.int32)
    msg = np.zeros((B, K), dtype=np.int32)
    sig = C.c_double(0)
    with workdir(workdir_path):
        rc = load().nblh_frontend(b"NBLDPC.Profile.txt", ebn0, frames, L.ctypes.data, tx.ctypes.data, msg.ctypes.data, C.byref(sig))
    if rc != 0:
        raise RuntimeError(f"nblh_frontend rc={rc}")
    return L, tx, msg, sig.value


def channel(workdir_path, ebn0, frames, L, P):
    """Host link chain up to the AWGN channel: (rx [B][L][2], tx_index [B][L] uint8, state [B][3] uint32, sigma), B = frames * P."""
    B = frames * P
    rx = np.zeros((B, L, 2))
    txi = np.zeros((B, L), dtype=np.uint8)
    state = np.zeros((B, 3), dtype=np.uint32)
    sig = C.c_double(0)
    with workdir(workdir_path):
        rc = load().nblh_channel(b"NBLDPC.Profile.txt", ebn0, frames, rx.ctypes.data, txi.ctypes.data, state.ctypes.data, C.byref(sig))
    if rc != L:
        raise RuntimeError(f"nblh_channel rc={rc} (expected MOD_SYM_LEN {L})")
    return rx, txi, state, sig.value


def simulate(workdir_path, device=0, max_rows=32):
    rows = np.zeros((max_rows, 9))
    with workdir(workdir_path):
        n = load().nblh_simulate(b"NBLDPC.Profile.txt", device, rows.ctypes.data, max_rows)
    if n < 0:
        raise RuntimeError(f"nblh_simulate rc={n}")
    keys = ("EbN0", "errFrame", "errSym", "errBit", "U_errFrame", "frames", "BER", "SER", "FER")
    return [dict(zip(keys, rows[i])) for i in range(n)]


def encode(workdir_path, msgs, N):
    msgs = np.ascontiguousarray(msgs, dtype=np.int32)
    out = np.zeros((msgs.shape[0], N), dtype=np.int32)
    with workdir(workdir_path):
        rc = load().nblh_encode(b"NBLDPC.Profile.txt", msgs.ctypes.data, msgs.shape[0], out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"nblh_encode rc={rc}")
    return out
