"""SURVEY 8f row 2 on the GPU: the device-side AWGN channel (CRand + Channel_AWGN, Rand.cpp:17-37, Comm.cpp:328-337) must form
the received samples of the host chain BIT FOR BIT, on more than 10^7 samples, and the decode behind it must give the host
path's results."""
import numpy as np
import pytest

import nbldpc_amd as nb
import nbldpc_amd.datafiles as df
from nbldpc_amd import hostlib

pytestmark = pytest.mark.gpu


def _setup(tmp_path, code_name, cons, method, P, seed, **kw):
    c = df.codes()[code_name]
    q = c["q"]
    hostlib.prepare_workdir(str(tmp_path), dict(gfq=q, code=code_name, method=method, max_iter=10, parallel=P, nqam=(2 if cons == "BPSK" else q),
                                                constellation=cons, random_msg=(1 if cons == "BPSK" else 0), seed=seed, **kw), code_name, cons)
    code = nb.Code(code_name)
    pts = sorted(df.constellations()[cons])
    points = np.array([[x[1], x[2]] for x in pts])
    p = q.bit_length() - 1
    L = c["N"] * p if cons == "BPSK" else c["N"]
    return code, points, L, q


@pytest.mark.parametrize("code_name,cons,method,P,frames,ebn0,kw", [
    ("divsalar.UNBLDPC.512.256.GF.256", "BPSK", 2, 5120, 2, 1.0, dict(ems_nm=32, ems_nc=3)),       # 2 x 5120 x 512 x 2 = 1.05e7 samples
    ("BDS.576.288.GF.64", "GRAY_64QAM", 4, 1024, 3, 3.0, dict(tems_nr=2, tems_nc=3)),
    ("divsalar.CNBLDPC.512.256.GF.256", "GRAY_256QAM", 1, 512, 2, 4.0, dict()),
])
def test_device_channel_bit_identical_to_host_chain(tmp_path, code_name, cons, method, P, frames, ebn0, kw):
    code, points, L, q = _setup(tmp_path, code_name, cons, method, P, 173, **kw)
    rx, txi, state, sigma = hostlib.channel(str(tmp_path), ebn0, frames, L, P)
    dec = nb.Decoder(code, method, 10, **kw)
    if cons == "BPSK":
        dec.set_demodulator(2, L, np.arange(L), points)
    else:
        dec.set_demodulator(q, L, np.arange(L), points)
    got, frac = dec.channel(txi, state, sigma)
    assert got.shape == rx.shape
    assert np.array_equal(got.view(np.uint64), rx.view(np.uint64)), f"{int((got.view(np.uint64) != rx.view(np.uint64)).sum())} of {rx.size} samples differ"
    assert 0.06 < frac < 0.12, frac     # share of the log / cos values settled by the host's libm (~5 % of the logs, ~12 % of the cosines)
    # and the whole call: channel + demodulator + decode on the device == host channel, device demodulator + decode
    o1, c1, i1 = dec.decode_noise(txi[:256], state[:256], sigma)
    o2, c2, i2 = dec.decode_samples(rx[:256], sigma)
    assert np.array_equal(o1, o2) and np.array_equal(c1, c2) and np.array_equal(i1, i2)
    dec.close()


def test_rand_advance_matches_the_lanes(tmp_path):
    """nbl_rand_advance(state, 4 L) is the state the host chain reaches after a frame"""
    import ctypes as C
    code, points, L, q = _setup(tmp_path, "divsalar.UNBLDPC.128.64.GF.256", "BPSK", 2, 64, 4242, ems_nm=16, ems_nc=3)
    rx, txi, state, sigma = hostlib.channel(str(tmp_path), 2.0, 3, L, 64)
    lib = nb.load_library()
    lib.nbl_rand_advance.argtypes = [C.c_void_p, C.c_uint64]
    lib.nbl_rand_advance.restype = None
    for lane in (0, 1, 63):
        st = state[lane].copy()
        for f in (1, 2):
            lib.nbl_rand_advance(st.ctypes.data, 4 * L)
            assert np.array_equal(st, state[f * 64 + lane])


def test_two_phase_channel_overlaps_the_decode(tmp_path):
    """nbl_channel_batch (slot s) + nbl_decode_batch_resident (slot s) = nbl_decode_batch_noise; and the channel of the next batch
    may run on another host thread while the current one is decoded -- what the pipelined harness does."""
    import threading
    P, frames = 1024, 4
    code, points, L, q = _setup(tmp_path, "divsalar.UNBLDPC.128.64.GF.256", "BPSK", 2, P, 99, ems_nm=16, ems_nc=3)
    rx, txi, state, sigma = hostlib.channel(str(tmp_path), 2.0, frames, L, P)
    dec = nb.Decoder(code, nb.METHOD_EMS, 10, ems_nm=16, ems_nc=3, poll_every=2)
    dec.set_demodulator(2, L, np.arange(L), points)
    ref = [dec.decode_noise(txi[f * P:(f + 1) * P], state[f * P:(f + 1) * P], sigma) for f in range(frames)]
    # pipelined: channel of frame f+1 under the decode of frame f
    dec.channel_batch(0, txi[:P], state[:P], sigma)
    for f in range(frames):
        slot = f & 1
        th = None
        if f + 1 < frames:
            th = threading.Thread(target=dec.channel_batch, args=(slot ^ 1, txi[(f + 1) * P:(f + 2) * P], state[(f + 1) * P:(f + 2) * P], sigma))
            th.start()
        got = dec.decode_resident(slot, sigma, P)
        if th:
            th.join()
        for a, b in zip(got, ref[f]):
            assert np.array_equal(a, b), f
    with pytest.raises(nb.NblError):
        dec.decode_resident(0, sigma, P // 2)   # the slot holds a batch of another size
    dec.close()


def test_channel_entry_points_edge_cases(tmp_path):
    """Ragged / tiny batches, an empty batch, and the error paths of the device-side channel."""
    import ctypes as C
    P = 7
    code, points, L, q = _setup(tmp_path, "divsalar.UNBLDPC.128.64.GF.256", "BPSK", 2, P, 5, ems_nm=16, ems_nc=3)
    rx, txi, state, sigma = hostlib.channel(str(tmp_path), 2.5, 1, L, P)
    dec = nb.Decoder(code, nb.METHOD_EMS, 10, ems_nm=16, ems_nc=3)
    # the channel needs the constellation points, also for BPSK
    dec.set_demodulator(2, L, np.arange(L))
    with pytest.raises(nb.NblError) as e:
        dec.decode_noise(txi, state, sigma)
    assert e.value.status == -1 and "constellation" in str(e.value)
    dec.set_demodulator(2, L, np.arange(L), points)
    ref = dec.decode_samples(rx, sigma)
    for B in (1, 3, P):                               # batches far below one wave's worth of work, odd sizes
        got = dec.decode_noise(txi[:B], state[:B], sigma)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[:B]), B
        r2, _ = dec.channel(txi[:B], state[:B], sigma)
        assert np.array_equal(r2.view(np.uint64), rx[:B].view(np.uint64))
    # a constellation index beyond the modulation order is refused (it would index past the points on the device)
    bad = txi.copy()
    bad[P - 1, L - 1] = 2
    with pytest.raises(nb.NblError) as e:
        dec.channel(bad, state, sigma)
    assert e.value.status == -1 and "tx_index" in str(e.value)
    # re-configured demodulator with MORE symbols per lane (256-ary, L = N  ->  BPSK, L = 8 N) at the same batch size: the
    # channel's buffers follow the new L (their capacity was keyed on the batch size alone)
    pts256 = np.array([[x[1], x[2]] for x in sorted(df.constellations()["GRAY_256QAM"])])
    dec2 = nb.Decoder(code, nb.METHOD_EMS, 10, ems_nm=16, ems_nc=3)
    dec2.set_demodulator(q, code.N, np.arange(code.N), pts256)
    dec2.channel(np.zeros((P, code.N), np.uint8), state, sigma)
    dec2.set_demodulator(2, L, np.arange(L), points)
    r3, _ = dec2.channel(txi, state, sigma)
    assert np.array_equal(r3.view(np.uint64), rx.view(np.uint64))
    dec2.close()
    out, conv, it = dec.decode_noise(txi[:0], state[:0], sigma)   # empty batch: nothing to do, no error
    assert out.shape == (0, code.N)
    with pytest.raises(nb.NblError):
        dec.decode_noise(txi, state, -1.0)            # sigma must be positive
    lib = nb.load_library()
    lib.nbl_rand_advance.argtypes = [C.c_void_p, C.c_uint64]
    lib.nbl_rand_advance.restype = None
    st = state[0].copy()
    lib.nbl_rand_advance(st.ctypes.data, 0)
    assert np.array_equal(st, state[0])               # zero draws: unchanged
    dec.close()


def test_short_log_and_exp2_on_the_device(tmp_path):
    """The DEVICE build of nbl_fastmath.h -- its short logarithm takes the hardware reciprocal estimate (v_rcp_f64) where the host
    build divides -- against this machine's libm on 4 M arguments: at most one ulp, like the host build (tests/test_ddmath.py)."""
    import json
    import os
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fastmath_device_check")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I", os.path.join(root, "nbldpc_amd", "csrc"),
                           os.path.join(root, "tests", "fastmath_device_check.hip"), "-o", exe])
    r = json.loads(subprocess.check_output([exe], text=True))
    assert r["exp2_worst_ulp"] <= 1.0 and r["log_worst_ulp"] <= 1.0, r


def test_integer_wave_maximum_on_the_device(tmp_path):
    """wave_fmax_nonneg (nbl_device.h: the exact maximum of max(x, 0) over a wave as two unsigned 32-bit DPP reductions, used by
    every hard decision and every most-reliable-symbol search) equals a host evaluation and the FP64 reduction bit for bit on
    200 000 rows: mixed signs and magnitudes, equal high words, all-negative rows, signed zeros, denormals, infinity, integers."""
    import json
    import os
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wave_reduce_device_check")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I", os.path.join(root, "nbldpc_amd", "csrc"),
                           os.path.join(root, "tests", "wave_reduce_device_check.hip"), "-o", exe])
    r = json.loads(subprocess.check_output([exe], text=True))
    assert r["rows"] == 200000 and r["differ_from_host"] == 0 and r["differ_from_fp64_reduction"] == 0 and r["min_differs_from_host"] == 0, r

