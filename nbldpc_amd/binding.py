"""ctypes binding of the product library nbldpc_amd/csrc/libnbldpc_hip.so (C ABI: include/nbldpc.h).

This is plumbing for tests/ and bench.py.  There is no Python or CPU fallback: if the HIP library is
missing, or no GPU is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import datafiles

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBL_HIP_LIB") or os.path.join(_HERE, "csrc", "libnbldpc_hip.so")  # NBL_HIP_LIB: A/B builds

METHOD_BP, METHOD_EMS, METHOD_TEMS = 1, 2, 4

# every symbol include/nbldpc.h declares
EXPORTS = ("nbl_abi_version", "nbl_create", "nbl_destroy", "nbl_decode_batch", "nbl_decode_batch_device",
           "nbl_set_demodulator", "nbl_decode_batch_samples", "nbl_decode_batch_noise", "nbl_rand_advance", "nbl_channel_batch", "nbl_decode_batch_resident",
           "nbl_read_state", "nbl_set_record_state", "nbl_set_profiling", "nbl_last_timing", "nbl_last_error",
           "nbl_workspace_bytes")


class NblError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"nbldpc status {status}: {msg}")
        self.status = status


class CodeDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("q", C.c_int32),
                ("var_deg", C.c_void_p), ("chk_deg", C.c_void_p), ("var_chk", C.c_void_p), ("var_h", C.c_void_p),
                ("chk_var", C.c_void_p), ("chk_h", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("method", C.c_int32), ("max_iter", C.c_int32), ("ems_nm", C.c_int32), ("ems_nc", C.c_int32),
                ("ems_factor", C.c_double), ("ems_offset", C.c_double), ("tems_nr", C.c_int32), ("tems_nc", C.c_int32),
                ("tems_factor", C.c_double), ("tems_offset", C.c_double), ("fixed_iters", C.c_int32),
                ("poll_every", C.c_int32), ("max_batch", C.c_int32)]


_lib = None


def load_library():
    """Load libnbldpc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C nbldpc_amd/csrc` "
                                    "(or python -c 'import __graft_entry__ as g; g.build()')")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If torch is going to be used in this process
        # (bench.py, tests) it must be loaded BEFORE our library so that both resolve to the same runtime instance; loading
        # ours first makes a later torch.cuda initialisation fail with "No HIP GPUs are available".
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.nbl_abi_version.restype = C.c_int32
        L.nbl_create.restype = C.c_int
        L.nbl_create.argtypes = [C.POINTER(CodeDesc), C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
        L.nbl_destroy.argtypes = [C.c_void_p]
        L.nbl_destroy.restype = None
        L.nbl_decode_batch.restype = C.c_int
        L.nbl_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nbl_decode_batch_device.restype = C.c_int
        L.nbl_decode_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nbl_read_state.restype = C.c_int
        L.nbl_read_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nbl_set_record_state.argtypes = [C.c_void_p, C.c_int32]
        L.nbl_set_profiling.argtypes = [C.c_void_p, C.c_int32]
        L.nbl_last_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nbl_last_error.restype = C.c_char_p
        L.nbl_last_error.argtypes = [C.c_void_p]
        L.nbl_workspace_bytes.restype = C.c_size_t
        L.nbl_workspace_bytes.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class Code:
    """Tanner graph in the reference's file order (both directions), 0-based."""

    def __init__(self, name=None, spec=None):
        c = spec if spec is not None else datafiles.codes()[name]
        self.name = name
        self.N, self.M, self.q = c["N"], c["M"], c["q"]
        self.var_deg = np.array([len(r) for r in c["var_rows"]], dtype=np.int32)
        self.chk_deg = np.array([len(r) for r in c["chk_rows"]], dtype=np.int32)
        self.var_chk = np.array([x[0] - 1 for r in c["var_rows"] for x in r], dtype=np.int32)
        self.var_h = np.array([x[1] for r in c["var_rows"] for x in r], dtype=np.int32)
        self.chk_var = np.array([x[0] - 1 for r in c["chk_rows"] for x in r], dtype=np.int32)
        self.chk_h = np.array([x[1] for r in c["chk_rows"] for x in r], dtype=np.int32)
        self.E = int(self.var_deg.sum())

    def desc(self):
        return CodeDesc(self.N, self.M, self.q, self.var_deg.ctypes.data, self.chk_deg.ctypes.data,
                        self.var_chk.ctypes.data, self.var_h.ctypes.data, self.chk_var.ctypes.data, self.chk_h.ctypes.data)


class Decoder:
    """Batched decoder handle (nbl_create .. nbl_destroy)."""

    def __init__(self, code, method, max_iter, ems_nm=32, ems_nc=3, ems_factor=1.0, ems_offset=0.0, tems_nr=2, tems_nc=3,
                 tems_factor=1.0, tems_offset=0.0, fixed_iters=0, poll_every=0, max_batch=0, device=0, gf=None):
        self.lib = load_library()
        self.code = code
        mul, inv = gf if gf is not None else datafiles.gf_tables(code.q)
        self._mul = np.ascontiguousarray(np.array(mul, dtype=np.uint16))
        self._inv = np.ascontiguousarray(np.array(inv, dtype=np.uint16))
        self.params = Params(method, max_iter, ems_nm, ems_nc, ems_factor, ems_offset, tems_nr, tems_nc, tems_factor,
                             tems_offset, fixed_iters, poll_every, max_batch)
        desc = code.desc()
        h = C.c_void_p()
        rc = self.lib.nbl_create(C.byref(desc), self._mul.ctypes.data, self._inv.ctypes.data, C.byref(self.params), device, C.byref(h))
        if rc != 0:
            raise NblError(rc, self.lib.nbl_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbl_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise NblError(rc, self.lib.nbl_last_error(self.h).decode())

    def decode(self, L_ch):
        """L_ch: host array [B][N][q-1] float64 -> (out[B][N] int32, converged[B] uint8, iters[B] int32)"""
        L_ch = np.ascontiguousarray(L_ch, dtype=np.float64)
        B = L_ch.shape[0]
        assert L_ch.shape == (B, self.code.N, self.code.q - 1), L_ch.shape
        out = np.zeros((B, self.code.N), dtype=np.int32)
        conv = np.zeros(B, dtype=np.uint8)
        iters = np.zeros(B, dtype=np.int32)
        self._chk(self.lib.nbl_decode_batch(self.h, L_ch.ctypes.data, B, out.ctypes.data, conv.ctypes.data, iters.ctypes.data))
        return out, conv, iters

    def decode_device(self, d_L_ch, B, d_out, d_conv=None, d_iters=None, stream=None):
        """Raw device pointers (ints); asynchronous on `stream` (a hipStream_t as int, None = decoder stream)."""
        self._chk(self.lib.nbl_decode_batch_device(self.h, d_L_ch, B, d_out, d_conv, d_iters, stream))

    def set_demodulator(self, mod_order, n_mod_sym, src, constellation=None):
        """src: int32 sample index per code bit (BPSK) / per code symbol (q-ary), -1 = punctured."""
        class Demod(C.Structure):
            _fields_ = [("mod_order", C.c_int32), ("n_mod_sym", C.c_int32), ("constellation", C.c_void_p), ("src", C.c_void_p)]
        self._dm_src = np.ascontiguousarray(src, dtype=np.int32)
        self._dm_cons = None if constellation is None else np.ascontiguousarray(constellation, dtype=np.float64)
        d = Demod(mod_order, n_mod_sym, None if self._dm_cons is None else self._dm_cons.ctypes.data, self._dm_src.ctypes.data)
        self.lib.nbl_set_demodulator.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.lib.nbl_set_demodulator(self.h, C.byref(d)))

    def decode_samples(self, rx, sigma):
        """rx: [B][L][2] received samples -> (out, converged, iters); L_ch is built on the device."""
        rx = np.ascontiguousarray(rx, dtype=np.float64)
        B = rx.shape[0]
        out = np.zeros((B, self.code.N), dtype=np.int32)
        conv = np.zeros(B, dtype=np.uint8)
        iters = np.zeros(B, dtype=np.int32)
        self.lib.nbl_decode_batch_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.nbl_decode_batch_samples(self.h, rx.ctypes.data, sigma, B, out.ctypes.data, conv.ctypes.data, iters.ctypes.data))
        return out, conv, iters

    def decode_noise(self, tx_index, lane_state, sigma):
        """tx_index [B][L] uint8, lane_state [B][3] uint32 (CRand state before the frame): channel + demodulator + decode on the device"""
        tx_index = np.ascontiguousarray(tx_index, dtype=np.uint8)
        lane_state = np.ascontiguousarray(lane_state, dtype=np.uint32)
        B = tx_index.shape[0]
        out = np.zeros((B, self.code.N), dtype=np.int32)
        conv = np.zeros(B, dtype=np.uint8)
        iters = np.zeros(B, dtype=np.int32)
        self.lib.nbl_decode_batch_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.nbl_decode_batch_noise(self.h, tx_index.ctypes.data, lane_state.ctypes.data, sigma, B, out.ctypes.data, conv.ctypes.data, iters.ctypes.data))
        return out, conv, iters

    def channel_batch(self, slot, tx_index, lane_state, sigma):
        tx_index = np.ascontiguousarray(tx_index, dtype=np.uint8)
        lane_state = np.ascontiguousarray(lane_state, dtype=np.uint32)
        self.lib.nbl_channel_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int32]
        self._chk(self.lib.nbl_channel_batch(self.h, slot, tx_index.ctypes.data, lane_state.ctypes.data, sigma, tx_index.shape[0]))

    def decode_resident(self, slot, sigma, B):
        out = np.zeros((B, self.code.N), dtype=np.int32)
        conv = np.zeros(B, dtype=np.uint8)
        iters = np.zeros(B, dtype=np.int32)
        self.lib.nbl_decode_batch_resident.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.nbl_decode_batch_resident(self.h, slot, sigma, B, out.ctypes.data, conv.ctypes.data, iters.ctypes.data))
        return out, conv, iters

    def channel(self, tx_index, lane_state, sigma):
        """diagnostic: the received samples [B][L][2] the device-side channel forms, and the fraction of log / cos values the host's libm settled"""
        tx_index = np.ascontiguousarray(tx_index, dtype=np.uint8)
        lane_state = np.ascontiguousarray(lane_state, dtype=np.uint32)
        B, L = tx_index.shape
        rx = np.zeros((B, L, 2))
        frac = C.c_double(0)
        self.lib.nbl_debug_channel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
        self._chk(self.lib.nbl_debug_channel(self.h, tx_index.ctypes.data, lane_state.ctypes.data, sigma, B, rx.ctypes.data, C.byref(frac)))
        return rx, frac.value

    def read_lch(self, b):
        L = np.zeros((self.code.N, self.code.q - 1))
        self.lib.nbl_debug_read_lch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        self._chk(self.lib.nbl_debug_read_lch(self.h, b, L.ctypes.data))
        return L

    def record_state(self, on=True):
        self._chk(self.lib.nbl_set_record_state(self.h, int(on)))

    def profiling(self, on=True):
        self._chk(self.lib.nbl_set_profiling(self.h, int(on)))

    def last_timing(self):
        ms = (C.c_double * 4)()
        ln = (C.c_int64 * 3)()
        self._chk(self.lib.nbl_last_timing(self.h, ms, ln))
        return list(ms), list(ln)

    def read_state(self, b, post=True):
        w, N, E = self.code.q - 1, self.code.N, self.code.E
        P = np.zeros((N, w)) if post else None
        V = np.zeros((E, w))
        Cc = np.zeros((E, w))
        self._chk(self.lib.nbl_read_state(self.h, b, P.ctypes.data if post else None, V.ctypes.data, Cc.ctypes.data))
        return P, V, Cc

    def workspace_bytes(self):
        return self.lib.nbl_workspace_bytes(self.h)
