#!/usr/bin/env python3
"""Generate the tests/golden npz fixtures from the COMPILED REFERENCE (oracle/_ref, see oracle/Makefile `make ref`).

Build-container only: needs /root/reference.  Each set is one profile (nbldpc_amd/profiles.py grammar) run through
oracle/ref_driver.cpp, which calls the unmodified CSimulation / CComm / CNBLDPC objects and dumps
  L_ch      [B][N][q-1]  channel LLRs produced by the reference's own link chain (frame-major, b = f*P + lane)
  tx_code   [B][N]       transmitted codeword symbols
  out/ret/syn_ok [K][B]  CNBLDPC::Decoding output, return value, zero-syndrome check for maxIter = iters[k]
  st_post/st_v2c/st_c2v  message state after state_iters[k] iterations (frame 0; var-major edge order)
plus FER anchor lines (fer mode).  Large state arrays are subsampled (`state_lanes`) to keep fixtures small.

usage: python tests/golden/make_golden.py [set ...]
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from nbldpc_amd.profiles import profile_text  # noqa: E402

REF = "/root/reference/"
RUN = os.path.join(ROOT, "oracle", "_ref", "run")
GOLD = os.path.join(ROOT, "tests", "golden")

U128_16 = "divsalar.UNBLDPC.128.64.GF.16"
U128_256 = "divsalar.UNBLDPC.128.64.GF.256"
U512_256 = "divsalar.UNBLDPC.512.256.GF.256"
U256_16 = "divsalar.UNBLDPC.256.128.GF.16"
U512_16 = "divsalar.UNBLDPC.512.256.GF.16"
U256_256 = "divsalar.UNBLDPC.256.128.GF.256"
C256_256 = "divsalar.CNBLDPC.256.128.GF.256"
C128_256 = "divsalar.CNBLDPC.128.64.GF.256"
C512_256 = "divsalar.CNBLDPC.512.256.GF.256"
BDS = "BDS.576.288.GF.64"

# name -> (driver build, profile kwargs, EbN0, frames, iters, state_iters, state_lanes)
SETS = {
    # cfg 1: BP, GF(16), BPSK, 20 iterations
    "cfg1_bp_gf16": ("O0", dict(gfq=16, code=U128_16, method=1, max_iter=20, parallel=4, constellation="BPSK"),
                     2.0, 4, [1, 2, 5, 20], [1, 2, 3], [0, 1, 2, 3]),
    # cfg 2: EMS nm=16 nc=3, GF(256) 128.64
    "cfg2_ems_u128": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=50, parallel=4, ems_nm=16, ems_nc=3,
                                 constellation="BPSK"), 2.0, 4, [1, 2, 5, 10, 50], [1, 2], [0, 1]),
    # cfg 3 (north star): EMS nm=32 nc=3, GF(256) 512.256
    "cfg3_ems_u512": ("O2", dict(gfq=256, code=U512_256, method=2, max_iter=50, parallel=2, ems_nm=32, ems_nc=3,
                                 constellation="BPSK"), 1.0, 3, [1, 2, 5, 10, 50], [1, 2], [0]),
    # EMS variants: nc < dc-1, factor/offset dead-zone, irregular GF(16) code with dc = 5
    "ems_nc2_shaped": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=30, parallel=2, ems_nm=16, ems_nc=2,
                                  ems_factor=1.25, ems_offset=0.3, constellation="BPSK"), 2.5, 4, [1, 2, 5, 30], [1, 2], [0]),
    "ems_nc1": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=30, parallel=2, ems_nm=16, ems_nc=1,
                           constellation="BPSK"), 3.0, 4, [1, 2, 5, 30], [1], [0]),
    "ems_gf16_dc5": ("O2", dict(gfq=16, code=U128_16, method=2, max_iter=20, parallel=4, ems_nm=8, ems_nc=2,
                                ems_offset=0.1, constellation="BPSK"), 2.0, 4, [1, 2, 5, 20], [1, 2, 3], [0, 1, 2, 3]),
    "ems_gf16_nc4": ("O2", dict(gfq=16, code=U128_16, method=2, max_iter=20, parallel=4, ems_nm=6, ems_nc=4,
                                constellation="BPSK"), 2.0, 4, [1, 2, 5, 20], [1, 2], [0, 1]),
    # cfg 4: T-EMS, BDS GF(64), 64-QAM, all-zero codeword (QAM bit-order hazard)
    "cfg4_tems_bds": ("O2", dict(gfq=64, code=BDS, method=4, max_iter=50, parallel=2, tems_nr=2, tems_nc=3,
                                 nqam=64, constellation="GRAY_64QAM", random_msg=0), 3.0, 3, [1, 2, 5, 50], [1, 2], [0]),
    "tems_gf16_dc5": ("O2", dict(gfq=16, code=U128_16, method=4, max_iter=20, parallel=4, tems_nr=2, tems_nc=2,
                                 tems_factor=1.1, tems_offset=0.05, constellation="BPSK"), 2.0, 4, [1, 2, 5, 20],
                      [1, 2, 3], [0, 1, 2, 3]),
    # the other shipped codes: mixed check degrees 4 / 5 (GF(16) 256.128 and 512.256), T-EMS over GF(256), the CNBLDPC family
    "ems_gf16_u256_mixed": ("O2", dict(gfq=16, code=U256_16, method=2, max_iter=20, parallel=4, ems_nm=8, ems_nc=3,
                                       constellation="BPSK"), 2.0, 4, [1, 2, 5, 20], [1, 2], [0, 1]),
    "tems_gf16_u512_mixed": ("O2", dict(gfq=16, code=U512_16, method=4, max_iter=20, parallel=4, tems_nr=2, tems_nc=3,
                                        constellation="BPSK"), 2.0, 4, [1, 2, 5, 20], [1, 2], [0, 1]),
    "bp_gf16_u256_mixed": ("O0", dict(gfq=16, code=U256_16, method=1, max_iter=20, parallel=4, constellation="BPSK"),
                           2.0, 4, [1, 2, 5, 20], [1, 2], [0, 1]),
    "tems_gf256_u256": ("O2", dict(gfq=256, code=U256_256, method=4, max_iter=20, parallel=3, tems_nr=2, tems_nc=3,
                                   constellation="BPSK"), 1.5, 3, [1, 2, 5, 20], [1, 2], [0]),
    "ems_c256_qam": ("O2", dict(gfq=256, code=C256_256, method=2, max_iter=30, parallel=3, ems_nm=16, ems_nc=2, nqam=256,
                                constellation="GRAY_256QAM", random_msg=0), 3.5, 3, [1, 2, 5, 30], [1, 2], [0]),
    "tems_c128_nr3": ("O2", dict(gfq=256, code=C128_256, method=4, max_iter=20, parallel=3, tems_nr=3, tems_nc=2, tems_factor=1.05,
                                 tems_offset=0.02, constellation="BPSK"), 2.0, 3, [1, 2, 5, 20], [1, 2], [0]),
    # cfg 5: BP / log-QSPA, GF(256), 256-QAM, all-zero codeword; 2.2 s per iteration per codeword on the CPU
    "cfg5_bp_c512": ("O0", dict(gfq=256, code=C512_256, method=1, max_iter=100, parallel=1, nqam=256,
                                constellation="GRAY_256QAM", random_msg=0), 10.0, 2, [1, 2, 3], [1, 2], [0]),
}

# Deep sets: several reference processes (one lane each, seeds seed0 + k) whose frames are concatenated -- the reference's BP costs
# ~2.5 s per iteration per codeword at -O0, so the frames are spread over the cores.  name -> (build, profile, EbN0, processes,
# frames per process, iters, state_iters of process 0 / frame 0)
DEEP_SETS = {
    # cfg 5 through a waterfall trajectory: 16 frames at 2.8 dB (most need 10-40 iterations, some never converge), outputs after
    # 10 and 30 iterations, full message state after 10 iterations for one frame
    "cfg5_bp_c512_deep": ("O0", dict(gfq=256, code=C512_256, method=1, max_iter=100, parallel=1, nqam=256,
                                     constellation="GRAY_256QAM", random_msg=0), 2.8, 8, 2, [10, 30], [10]),
}

FER_SETS = {
    # cfg 5: 32 frames at 3 dB (waterfall) and at 4 dB (most frames stop within 10 iterations); -O0 build (BP failure path is UB at -O1+)
    "cfg5_bp_c512_3dB": ("O0", dict(gfq=256, code=C512_256, method=1, max_iter=100, parallel=4, nqam=256, constellation="GRAY_256QAM",
                                    random_msg=0, snr_begin=3.0, snr_stop=3.0, min_sim_cycle=28)),
    "cfg5_bp_c512_4dB": ("O0", dict(gfq=256, code=C512_256, method=1, max_iter=100, parallel=4, nqam=256, constellation="GRAY_256QAM",
                                    random_msg=0, snr_begin=4.0, snr_stop=4.0, min_sim_cycle=28)),
    # BASELINE.md anchors, reproduced here from the compiled reference
    "cfg1_bp_gf16": ("O0", dict(gfq=16, code=U128_16, method=1, max_iter=20, parallel=1, ems_nm=16, ems_nc=2, tems_nc=2,
                                snr_begin=2.0, snr_stop=3.0, constellation="BPSK", min_err_frame=20, min_sim_cycle=200)),
    "cfg2_ems_u128": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=50, parallel=1, ems_nm=16, ems_nc=3,
                                 snr_begin=2.0, snr_stop=2.0, constellation="BPSK", min_sim_cycle=200)),
    "cfg2_ems_u128_p8": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=50, parallel=8, ems_nm=16, ems_nc=3,
                                    snr_begin=1.5, snr_step=0.5, snr_stop=2.5, constellation="BPSK", min_sim_cycle=250)),
    # north-star code: 64 frames per point through the waterfall (1.0 / 1.5 / 2.0 dB), 16 lanes
    "cfg3_ems_u512_p16": ("O2", dict(gfq=256, code=U512_256, method=2, max_iter=50, parallel=16, ems_nm=32, ems_nc=3,
                                     snr_begin=1.0, snr_step=0.5, snr_stop=2.0, constellation="BPSK", min_sim_cycle=48)),
    "cfg4_tems_bds": ("O2", dict(gfq=64, code=BDS, method=4, max_iter=50, parallel=4, tems_nr=2, tems_nc=3, nqam=64,
                                 constellation="GRAY_64QAM", random_msg=0, snr_begin=3.0, snr_stop=4.0, min_sim_cycle=28)),
    # GF(16), check degrees 4 / 5 mixed: 2000-4000 frames per point through the waterfall (the small-field kernels' statistics)
    "ems_gf16_u512_p8": ("O2", dict(gfq=16, code=U512_16, method=2, max_iter=20, parallel=8, ems_nm=8, ems_nc=3,
                                    snr_begin=1.5, snr_step=0.5, snr_stop=2.5, constellation="BPSK", min_sim_cycle=2000)),
    "tems_gf16_u256_p8": ("O2", dict(gfq=16, code=U256_16, method=4, max_iter=20, parallel=8, tems_nr=2, tems_nc=3,
                                     snr_begin=1.5, snr_step=0.5, snr_stop=2.5, constellation="BPSK", min_sim_cycle=4000)),
    # the BDS code with the two methods BASELINE config 4 does not use
    "bp_bds_p4": ("O0", dict(gfq=64, code=BDS, method=1, max_iter=50, parallel=4, nqam=64, constellation="GRAY_64QAM", random_msg=0,
                             snr_begin=1.5, snr_step=0.5, snr_stop=2.0, min_sim_cycle=120)),
    "ems_bds_p4": ("O2", dict(gfq=64, code=BDS, method=2, max_iter=50, parallel=4, ems_nm=16, ems_nc=3, nqam=64, constellation="GRAY_64QAM",
                              random_msg=0, snr_begin=1.5, snr_step=0.5, snr_stop=2.5, min_sim_cycle=200)),
    # the CNBLDPC family over 256-QAM (all-zero codeword), EMS with the reference's sample-profile nc = 2
    "ems_c256_qam_p8": ("O2", dict(gfq=256, code=C256_256, method=2, max_iter=50, parallel=8, ems_nm=16, ems_nc=2, nqam=256,
                                   constellation="GRAY_256QAM", random_msg=0, snr_begin=4.0, snr_step=1.0, snr_stop=6.0, min_sim_cycle=200)),
    "ems_c128_qam_p8": ("O2", dict(gfq=256, code=C128_256, method=2, max_iter=50, parallel=8, ems_nm=16, ems_nc=3, nqam=256,
                                   constellation="GRAY_256QAM", random_msg=0, snr_begin=5.0, snr_step=1.0, snr_stop=7.0, min_sim_cycle=200)),
    "ems_u256_p8": ("O2", dict(gfq=256, code=U256_256, method=2, max_iter=50, parallel=8, ems_nm=32, ems_nc=3,
                               snr_begin=1.5, snr_step=0.5, snr_stop=2.0, constellation="BPSK", min_sim_cycle=120)),
    "bp_gf16_u256_p8": ("O0", dict(gfq=16, code=U256_16, method=1, max_iter=20, parallel=8,
                                   snr_begin=2.0, snr_step=0.5, snr_stop=3.0, constellation="BPSK", min_sim_cycle=2000)),
    # the two caller-path features no other anchor touches.
    # Puncturing (NBLDPC.cpp:163-176, Comm.cpp:290-308, :348-357): `Puncture Degree: 3` removes every degree-3 variable of the
    # GF(16) 256.128 code from the channel (its bits are not sent; the demodulator gives them LLR 0).
    "ems_gf16_u256_punct3": ("O2", dict(gfq=16, code=U256_16, puncture_degree=3, method=2, max_iter=20, parallel=8, ems_nm=8, ems_nc=3,
                                        snr_begin=3.0, snr_step=1.0, snr_stop=4.0, constellation="BPSK", min_sim_cycle=500)),
    "tems_gf16_u256_punct3": ("O2", dict(gfq=16, code=U256_16, puncture_degree=3, method=4, max_iter=20, parallel=8, tems_nr=2, tems_nc=3,
                                         snr_begin=3.0, snr_step=1.0, snr_stop=4.0, constellation="BPSK", min_sim_cycle=500)),
    "bp_gf16_u256_punct3": ("O0", dict(gfq=16, code=U256_16, puncture_degree=3, method=1, max_iter=20, parallel=8,
                                       snr_begin=3.0, snr_step=1.0, snr_stop=4.0, constellation="BPSK", min_sim_cycle=500)),
    # CRC-16 and CRC-24 (Comm.cpp:506-636); the reference generates with one CRC-24 polynomial and checks with another (SURVEY
    # hazard 1): the undetected-error column of the crcLen 24 profile holds what that yields -- reproduced, not repaired.
    "ems_u128_crc16": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=50, parallel=8, ems_nm=16, ems_nc=3, crc_len=16,
                                  snr_begin=1.5, snr_step=0.5, snr_stop=2.0, constellation="BPSK", min_sim_cycle=120)),
    "ems_u128_crc24": ("O2", dict(gfq=256, code=U128_256, method=2, max_iter=50, parallel=8, ems_nm=16, ems_nc=3, crc_len=24,
                                  snr_begin=1.5, snr_step=0.5, snr_stop=2.0, constellation="BPSK", min_sim_cycle=120)),
}


def resolve(kw):
    kw = dict(kw)
    code, cons = kw["code"], kw["constellation"]
    kw["code"] = REF + code + ".txt"
    kw["constellation"] = REF + cons + ".txt"
    return kw, code, cons


def run_set(name):
    build, kw, ebn0, frames, iters, st_iters, st_lanes = SETS[name]
    pk, code, cons = resolve(kw)
    tmp = tempfile.mkdtemp(prefix="golden_")
    prof = os.path.join(tmp, "profile.txt")
    open(prof, "w").write(profile_text(**pk))
    t0 = time.time()
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", f"ref_driver_{build}"), "dump", prof, tmp, repr(ebn0),
                           str(frames), ",".join(map(str, iters)), ",".join(map(str, st_iters))], cwd=RUN,
                          stderr=subprocess.DEVNULL)
    arrs = {k[:-4]: np.load(os.path.join(tmp, k)) for k in os.listdir(tmp) if k.endswith(".npy")}
    for k in ("st_post", "st_v2c", "st_c2v"):
        if k in arrs:
            arrs[k] = arrs[k][:, st_lanes]
    arrs["state_lanes"] = np.array(st_lanes, dtype=np.int32)
    meta = dict(profile=dict(kw), code=code, constellation=cons, ebn0=ebn0, frames=frames, build=build,
                reference_flags="-std=c++14 -O2 (NBLDPC.cpp at -%s) -ffp-contract=off, g++ 11.4, x86-64" % build)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), meta=json.dumps(meta), **arrs)
    shutil.rmtree(tmp)
    print(f"{name}: B={arrs['L_ch'].shape[0]} ret={arrs['ret'].tolist()} syn_ok={arrs['syn_ok'].tolist()} "
          f"({time.time() - t0:.1f}s, {os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1e3:.0f} kB)")


def run_deep(name):
    build, kw, ebn0, procs, frames, iters, st_iters = DEEP_SETS[name]
    tmps, ps = [], []
    t0 = time.time()
    for k in range(procs):
        pk, code, cons = resolve(dict(kw, seed=173 + 7 * k))
        tmp = tempfile.mkdtemp(prefix="golden_")
        prof = os.path.join(tmp, "profile.txt")
        open(prof, "w").write(profile_text(**pk))
        ps.append(subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", f"ref_driver_{build}"), "dump", prof, tmp, repr(ebn0),
                                    str(frames), ",".join(map(str, iters)), ",".join(map(str, st_iters)) if k == 0 else ""], cwd=RUN,
                                   stderr=subprocess.DEVNULL))
        tmps.append(tmp)
    for p_ in ps:
        assert p_.wait() == 0
    parts = [{k[:-4]: np.load(os.path.join(t, k)) for k in os.listdir(t) if k.endswith(".npy")} for t in tmps]
    arrs = {}
    for k in ("L_ch", "tx_code", "tx_msg"):
        arrs[k] = np.concatenate([p_[k] for p_ in parts], axis=0)
    for k in ("out", "ret", "syn_ok"):
        arrs[k] = np.concatenate([p_[k] for p_ in parts], axis=1)
    arrs["sigma"] = parts[0]["sigma"]
    arrs["iters"] = parts[0]["iters"]
    arrs["state_iters"] = parts[0]["state_iters"]
    for k in ("st_post", "st_v2c", "st_c2v"):
        arrs[k] = parts[0][k]
    arrs["state_lanes"] = np.array([0], dtype=np.int32)
    meta = dict(profile=dict(kw), code=code, constellation=cons, ebn0=ebn0, frames=frames * procs, build=build,
                seeds=[173 + 7 * k for k in range(procs)],
                reference_flags="-std=c++14 -O2 (NBLDPC.cpp at -%s) -ffp-contract=off, g++ 11.4, x86-64" % build)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), meta=json.dumps(meta), **arrs)
    for t in tmps:
        shutil.rmtree(t)
    print(f"{name}: B={arrs['L_ch'].shape[0]} ret={arrs['ret'].tolist()} syn_ok={arrs['syn_ok'].tolist()} "
          f"({time.time() - t0:.1f}s, {os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1e3:.0f} kB)")


def run_fer(names):
    path = os.path.join(GOLD, "fer_anchors.json")
    anchors = json.load(open(path)) if os.path.exists(path) else {}
    for name in names:
        build, kw = FER_SETS[name]
        pk, code, cons = resolve(kw)
        tmp = tempfile.mkdtemp(prefix="golden_")
        prof = os.path.join(tmp, "profile.txt")
        open(prof, "w").write(profile_text(**pk))
        t0 = time.time()
        out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", f"ref_driver_{build}"), "fer", prof], cwd=RUN,
                             capture_output=True, text=True, check=True).stdout
        pts = [json.loads(line) for line in out.splitlines() if line.startswith("{")]
        for p in pts:
            p.pop("cpu_s", None)
        anchors[name] = dict(profile=dict(kw), code=code, constellation=cons, points=pts)
        shutil.rmtree(tmp)
        print(f"fer {name}: {pts} ({time.time() - t0:.1f}s)")
    json.dump(anchors, open(path, "w"), indent=1)


def main():
    os.makedirs(GOLD, exist_ok=True)
    want = sys.argv[1:] or (list(SETS) + list(DEEP_SETS) + ["fer:" + k for k in FER_SETS])
    for w in want:
        if w.startswith("fer:"):
            run_fer([w[4:]])
        elif w in DEEP_SETS:
            run_deep(w)
        else:
            run_set(w)


if __name__ == "__main__":
    main()
