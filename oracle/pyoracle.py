"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by nbldpc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BP, EMS, TEMS = 1, 2, 4
LITERAL, CANONICAL, CANONICAL_DFS = 0, 1, 2


class _GF(C.Structure):
    _fields_ = [("q", C.c_int), ("p", C.c_int), ("poly", C.c_int),
                ("mul", C.POINTER(C.c_int)), ("inv", C.POINTER(C.c_int))]


class _Code(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "M", "q", "E", "maxdv", "maxdc")] + \
               [(n, C.POINTER(C.c_int)) for n in ("dv", "dc", "voff", "coff", "v_chk", "v_h", "v_k",
                                                  "c_var", "c_h", "c_d", "c2e")]


class _Params(C.Structure):
    _fields_ = [("method", C.c_int), ("max_iter", C.c_int), ("mode", C.c_int),
                ("ems_nm", C.c_int), ("ems_nc", C.c_int), ("ems_factor", C.c_double), ("ems_offset", C.c_double),
                ("tems_nr", C.c_int), ("tems_nc", C.c_int), ("tems_factor", C.c_double), ("tems_offset", C.c_double),
                ("fixed_iters", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.nblo_gf_build.argtypes = [C.POINTER(_GF), C.c_int]
        L.nblo_gf_load.argtypes = [C.POINTER(_GF), C.c_int, C.c_char_p]
        L.nblo_code_load.restype = C.POINTER(_Code)
        L.nblo_code_load.argtypes = [C.c_char_p]
        L.nblo_code_from_edges.restype = C.POINTER(_Code)
        L.nblo_code_from_edges.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3
        L.nblo_decoder_create.restype = C.c_void_p
        L.nblo_decoder_create.argtypes = [C.POINTER(_Code), C.POINTER(_GF), C.POINTER(_Params)]
        L.nblo_decoder_free.argtypes = [C.c_void_p]
        L.nblo_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.nblo_decode_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        for n in ("nblo_state_post", "nblo_state_v2c", "nblo_state_c2v"):
            getattr(L, n).restype = C.POINTER(C.c_double)
            getattr(L, n).argtypes = [C.c_void_p]
        for n in ("nblo_check_ems", "nblo_check_tems", "nblo_check_bp"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class GF:
    def __init__(self, q, arith_path=None):
        self.s = _GF()
        rc = lib().nblo_gf_load(C.byref(self.s), q, arith_path.encode()) if arith_path else lib().nblo_gf_build(C.byref(self.s), q)
        if rc:
            raise RuntimeError(f"GF({q}) init failed rc={rc}")
        self.q = q

    @property
    def mul(self):
        return np.ctypeslib.as_array(self.s.mul, shape=(self.q, self.q)).copy()

    @property
    def inv(self):
        return np.ctypeslib.as_array(self.s.inv, shape=(self.q,)).copy()


class Code:
    def __init__(self, path=None, edges=None):
        """edges = (N, M, q, edge_var, edge_chk, edge_h) in var-major order."""
        if path is not None:
            self.p = lib().nblo_code_load(path.encode())
        else:
            N, M, q, ev, ec, eh = edges
            ev, ec, eh = (np.ascontiguousarray(x, dtype=np.int32) for x in (ev, ec, eh))
            self.p = lib().nblo_code_from_edges(N, M, q, len(ev), ev.ctypes.data, ec.ctypes.data, eh.ctypes.data)
        if not self.p:
            raise RuntimeError("code load failed")
        s = self.p.contents
        self.N, self.M, self.q, self.E, self.maxdv, self.maxdc = s.N, s.M, s.q, s.E, s.maxdv, s.maxdc

    def arr(self, name, n):
        return np.ctypeslib.as_array(getattr(self.p.contents, name), shape=(n,)).copy()

    def edge_list(self):
        """var-major (edge_var, edge_chk, edge_h)"""
        dv = self.arr("dv", self.N)
        ev = np.repeat(np.arange(self.N, dtype=np.int32), dv)
        return ev, self.arr("v_chk", self.E), self.arr("v_h", self.E)


class Decoder:
    def __init__(self, code, gf, method, max_iter, mode=LITERAL, ems_nm=32, ems_nc=3, ems_factor=1.0, ems_offset=0.0,
                 tems_nr=2, tems_nc=3, tems_factor=1.0, tems_offset=0.0, fixed_iters=0):
        self.code, self.gf = code, gf
        self.prm = _Params(method, max_iter, mode, ems_nm, ems_nc, ems_factor, ems_offset,
                           tems_nr, tems_nc, tems_factor, tems_offset, fixed_iters)
        self.h = lib().nblo_decoder_create(code.p, C.byref(gf.s), C.byref(self.prm))
        if not self.h:
            raise RuntimeError("decoder create failed")
        self.w = code.q - 1

    def __del__(self):
        if getattr(self, "h", None):
            lib().nblo_decoder_free(self.h)
            self.h = None

    def decode(self, L_ch):
        L_ch = np.ascontiguousarray(L_ch, dtype=np.float64)
        assert L_ch.shape == (self.code.N, self.w)
        out = np.zeros(self.code.N, dtype=np.int32)
        it = C.c_int(0)
        r = lib().nblo_decode(self.h, L_ch.ctypes.data, out.ctypes.data, C.byref(it))
        return r, out, it.value

    def state(self):
        n = self.code.E * self.w
        post = np.ctypeslib.as_array(lib().nblo_state_post(self.h), shape=(self.code.N, self.w)).copy()
        v2c = np.ctypeslib.as_array(lib().nblo_state_v2c(self.h), shape=(self.code.E, self.w)).copy()
        c2v = np.ctypeslib.as_array(lib().nblo_state_c2v(self.h), shape=(self.code.E, self.w)).copy()
        del n
        return post, v2c, c2v

    def check(self, m, v2c_in):
        v2c_in = np.ascontiguousarray(v2c_in, dtype=np.float64)
        out = np.zeros_like(v2c_in)
        fn = {BP: lib().nblo_check_bp, EMS: lib().nblo_check_ems, TEMS: lib().nblo_check_tems}[self.prm.method]
        fn(self.h, m, v2c_in.ctypes.data, out.ctypes.data)
        return out


def decode_batch(make_decoder, L_ch, nthreads=1):
    """L_ch [B][N][q-1]; make_decoder() -> Decoder (one per thread). Returns out[B][N], converged[B], iters[B]."""
    decs = [make_decoder() for _ in range(nthreads)]
    L_ch = np.ascontiguousarray(L_ch, dtype=np.float64)
    B, N = L_ch.shape[0], L_ch.shape[1]
    out = np.zeros((B, N), dtype=np.int32)
    conv = np.zeros(B, dtype=np.uint8)
    iters = np.zeros(B, dtype=np.int32)
    arr = (C.c_void_p * nthreads)(*[d.h for d in decs])
    lib().nblo_decode_batch(arr, nthreads, L_ch.ctypes.data, B, out.ctypes.data, conv.ctypes.data, iters.ctypes.data)
    return out, conv, iters
