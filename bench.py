#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NB-LDPC decode hot path on MI355X.

Metric (BASELINE.json): decoded codewords/sec @ 50 iterations, GF(256) 512.256 rate-1/2 code, EMS nm=32, plus the
achieved fraction of the HBM roofline.  A "step" is one pass of the decoder (50 fixed iterations) over one batch
of codewords whose channel LLRs already sit in HBM.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: every rank
decodes its own batch (independent frames, no data-path collective: SURVEY 8e), scaling is weak.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel (fused EMS iteration): algorithmic bytes per launch / mean launch time (HIP events on
                the launch stream inside the timed region) against the 8 TB/s HBM peak; `traffic` / `hbm_actual_GBps` = what
                the kernel really moves (committed PMC passes), `limiter` = what it waits for
  other_configs (N = 1 only) BASELINE configs 2, 4 and 5 at fixed iterations, one short pass each AFTER the headline's timed region:
                codewords/s, ms per launch and the fused kernel's algorithmic-bytes roofline (tools/bench_config.py::run_config)
  cpu_baseline  the compiled reference itself (oracle/_ref, kind "reference") on this host's cores, decode loop only, on a
                bounded sample of the same workload (rank 0, N=1 only); cpu_baseline_port = the oracle's literal restatement
                on codewords of the same batch (kind "port"; also the fallback when oracle/_ref is absent)

`headline_by_convergence` (rank 0, N = 1): the headline configuration once at 0 dB (no codeword converges: every check on the full
path) and once at 2 dB (all converge: exact short lists) -- the kernel's rate depends on how many of the batch's
codewords are past convergence, and the bench's 1 dB sits between the two.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

CODE = "divsalar.UNBLDPC.512.256.GF.256"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_llr(torch, code_q, N, B, ebn0_db, seed, device):
    """Channel LLRs of the all-zero codeword over BPSK/AWGN in the reference's symbol-LLR convention
    (Comm.cpp:157-178 sigma, :328-337 noise, :340-380 bit -> symbol LLR): L[a-1] = sum of bit LLRs over set bits of a."""
    p = code_q.bit_length() - 1
    rate = 0.5
    sigma = 1.0 / np.sqrt(2 * 1 * rate * 10 ** (ebn0_db / 10.0))
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    rx = 1.0 + sigma * torch.randn((B, N, p), dtype=torch.float64, device=device, generator=gen)  # BPSK point 0 -> +1
    bit_llr = -2.0 * rx / (sigma * sigma)
    a = torch.arange(1, code_q, device=device)
    mask = ((a[:, None] >> torch.arange(p, device=device)[None, :]) & 1).to(torch.float64)  # [q-1][p]
    return torch.matmul(bit_llr, mask.t()).contiguous()  # [B][N][q-1]


def harness_llr(nb, B, ebn0, seed, max_iter, nm, nc):
    """Synthetic inputs produced by the harness itself, not random tensors: B lanes of the reference-compatible link
    chain in nbldpc_amd/host (11-stage LFSR message -> CRC-8 -> systematic encode -> BPSK -> AWGN from the 3-LCG generator seeded
    seed+lane -> bit LLRs -> symbol LLRs), bit-identical to what the reference's CComm produces (tests/test_host_frontend.py)."""
    import tempfile
    from nbldpc_amd import hostlib
    tmp = tempfile.mkdtemp(prefix="nbl_bench_")
    hostlib.prepare_workdir(tmp, dict(gfq=256, code=CODE, method=2, max_iter=max_iter, parallel=B, ems_nm=nm, ems_nc=nc,
                                      constellation="BPSK", random_msg=1, seed=seed), CODE, "BPSK")
    c = nb.datafiles.codes()[CODE]
    L, tx, _, _ = hostlib.frontend(tmp, ebn0, 1, c["N"], c["N"] - c["M"], c["q"], B)
    return L, tx


def cpu_baseline_port(nb, L_host, nm, nc, max_iter, threads):
    """The oracle's LITERAL restatement of the reference EMS (same operation order incl. the DFS residue), one decoder per thread."""
    import pyoracle as po
    po.build()
    N, M, q, ev, ec, eh = nb.datafiles.code_edges(CODE)
    code = po.Code(edges=(N, M, q, ev, ec, eh))
    gf = po.GF(q)
    mk = lambda: po.Decoder(code, gf, po.EMS, max_iter, po.LITERAL, ems_nm=nm, ems_nc=nc, fixed_iters=1)  # noqa: E731
    t0 = time.time()
    po.decode_batch(mk, L_host, nthreads=threads)
    dt = time.time() - t0
    return {"value": L_host.shape[0] / dt, "unit": "codewords/s", "cores": threads, "kind": "port",
            "sample": f"{L_host.shape[0]} codewords of the same batch, {max_iter} fixed iterations, oracle literal restatement of the "
                      f"reference EMS (gcc -O2), {dt:.1f} s wall"}


def cpu_baseline_reference(nb, nm, nc, max_iter, threads, frames=4):
    """The COMPILED REFERENCE itself (oracle/_ref/ref_driver_O2, built in the build container from the unmodified sources; the
    binary travels, the sources do not), one process per core like the reference's one CNBLDPC per lane.  The reference has no
    fixed-iteration mode, so it is run at Eb/N0 = -3 dB where every frame fails and all `max_iter` iterations execute.
    Only the simulation of the Eb/N0 point is timed -- the CPU seconds the reference itself reports per point
    (clock() from CSimulation::ClearSimuCount, Simulation.cpp:222-225 / :357) -- not process start, table loads or
    CNBLDPC::Initial; the rate is the sum over the processes of frames / CPU seconds (one process per core)."""
    import subprocess
    import tempfile
    from nbldpc_amd.profiles import profile_text
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver_O2")
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="nbl_cpu_")
    nb.datafiles.materialise(tmp, 256, CODE, "BPSK")
    procs = []
    t0 = time.time()
    for k in range(threads):
        prof = os.path.join(tmp, f"p{k}.txt")
        with open(prof, "w") as f:
            f.write(profile_text(gfq=256, code=CODE + ".txt", method=2, max_iter=max_iter, parallel=1, ems_nm=nm, ems_nc=nc,
                                 snr_begin=-3.0, snr_stop=-3.0, constellation="BPSK.txt", random_msg=1, min_err_frame=-1,
                                 min_sim_cycle=frames - 1, seed=173 + k))
        procs.append(subprocess.Popen([exe, "fer", prof], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    outs = [p.communicate()[0] for p in procs]
    wall = time.time() - t0
    done, rate, cpu_max = 0, 0.0, 0.0
    for o in outs:
        for line in o.splitlines():
            if line.startswith("{"):
                r = json.loads(line)
                if r["errFrame"] != r["frames"]:
                    return None  # a frame converged early: not a fixed-iteration measurement
                done += int(r["frames"])
                rate += r["frames"] / r["cpu_s"]
                cpu_max = max(cpu_max, r["cpu_s"])
    if done == 0:
        return None
    return {"value": rate, "unit": "codewords/s", "cores": threads, "kind": "reference",
            "sample": f"{done} codewords ({threads} processes x {frames}), compiled reference (g++ -O2) EMS nm={nm} nc={nc}, Eb/N0 -3 dB so "
                      f"all {max_iter} iterations run; decode loop only ({cpu_max:.1f} CPU s per process as the reference reports them; "
                      f"{wall:.1f} s wall incl. set-up)"}


PMC_SUMMARY = os.path.join("profiles", "r03_summary.json")
# the other BASELINE.json configurations that fit one GPU: (name in tools/bench_config.py, batch, timed
# decodes, Eb/N0) -- one short fixed-iteration pass each AFTER the headline's timed region, reported under "other_configs"
OTHER_CONFIGS = (("cfg2", 4096, 4, None), ("cfg4", 8192, 2, 3.0), ("cfg5", 2048, 1, 4.0))


def other_configs(device):
    import bench_config
    labels = {"cfg2": "configs[1]: divsalar.UNBLDPC.128.64.GF.256 over BPSK, EMS nm=16 nc=3, 50 fixed iterations, batch 4096",
              "cfg4": "configs[3]: BDS.576.288.GF.64 over GRAY_64QAM, T-EMS nr=2 nc=3, 50 fixed iterations, Eb/N0 3 dB, batch 8192",
              "cfg5": "configs[4]: divsalar.CNBLDPC.512.256.GF.256 over GRAY_256QAM, full log-QSPA, 100 fixed iterations, Eb/N0 4 dB, "
                      "batch 2048 on this GPU (BASELINE: 8192 across 8)"}
    res = []
    for name, B, steps, ebn0 in OTHER_CONFIGS:
        r = bench_config.run_config(name, B, steps, ebn0, device)
        res.append({"workload": labels[name], "value": r["codewords_per_s"], "unit": "codewords/s", "ms_per_step": r["ms_per_batch"],
                    "ms_per_launch": r["cn_ms_per_launch"], "converged_frac": r["converged_frac"], "data": "synthetic (all-zero codeword + randn noise)",
                    "roofline": {"kernel": r["kernel"], "bound": "hbm", "achieved": r["cn_achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (r["cn_achieved_GBps"] or 0.0) / HBM_PEAK_GBS, "bytes_per_launch": r["cn_algorithmic_bytes_per_launch"]}})
    return res


def limiter_text():
    """What the dominant kernel waits for, from the committed counter passes (not measured in this run): vector-ALU busy share and
    LDS-port busy share of the kernel at its four waves per SIMD (16 per CU, one LDS port per CU)."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    for name, v in json.load(open(path)).get("sq_pmc", {}).items():
        if "cn_ems_q256_dc4_kernel<32, true" in name and "per_wave" in v:
            pw = v["per_wave"]
            valu = pw["valu_busy_of_wave_lifetime_x4_waves"]
            lds = 16.0 * pw["lds_active_cycles"] / pw["wave_cycles"]  # 16 waves of a CU share its LDS port
            return (f"fp64 valu issue ({100 * valu:.0f} % busy) + lds port ({100 * lds:.0f} %), not hbm ({PMC_SUMMARY}: "
                    f"{pw['valu_insts']:.0f} VALU / {pw['salu_insts']:.0f} SALU / {pw['lds_insts']:.0f} LDS instructions per check-wave); "
                    "the socket sits at its power limit under this kernel (1.36 kW, shader clock 2.08 of 2.4 GHz: profiles/r03_clocks.txt)")
    return None


def pmc_traffic(fused, B):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside the run, so this is NOT measured live:
    it is taken from the committed rocprofv3 passes of this same command (profiles/r02_summary.json: FETCH_SIZE and WRITE_SIZE in
    separate --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md), scaled from the profiled batch of 16384."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not fused or not os.path.exists(path):
        return None
    for name, v in json.load(open(path)).get("hbm_pmc", {}).items():
        if "cn_ems_q256_dc4_kernel<32, true" in name:
            return v["hbm_bytes_per_launch_corrected"] * B / 16384.0
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16384, help="codewords per GPU per step")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--nm", type=int, default=32)
    ap.add_argument("--nc", type=int, default=3)
    ap.add_argument("--ebn0", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="codewords for the CPU baseline (0 = skip)")
    ap.add_argument("--data", default="harness", choices=["harness", "randn"],
                    help="harness: the reference-compatible link chain generates the codewords and channel LLRs (default); "
                         "randn: all-zero codeword + torch.randn noise")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--other-configs", type=int, default=-1, help="1 / 0: also run BASELINE configs 2, 4, 5 after the timed region (default: at N = 1)")
    args = ap.parse_args()

    import torch
    import nbldpc_amd as nb

    from nbldpc_amd import ranks
    rk = ranks.init(args.backend, args.same_device)
    world, rank, local_rank = rk.world, rk.rank, rk.local_rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    code = nb.Code(CODE)
    B = args.batch
    dec = nb.Decoder(code, nb.METHOD_EMS, args.iters, ems_nm=args.nm, ems_nc=args.nc, fixed_iters=1, max_batch=B, device=local_rank)
    dec.profiling(True)
    if args.data == "harness":
        L_host, tx_code = harness_llr(nb, B, args.ebn0, 173 + 1000003 * rank, args.iters, args.nm, args.nc)
        L = torch.from_numpy(L_host).to(dev)
        tx_dev = torch.from_numpy(tx_code).to(dev)
        del L_host
    else:
        L = synth_llr(torch, code.q, code.N, B, args.ebn0, 173 + rank, dev)
        tx_dev = torch.zeros((B, code.N), dtype=torch.int32, device=dev)
    out = torch.zeros((B, code.N), dtype=torch.int32, device=dev)
    conv = torch.zeros(B, dtype=torch.uint8, device=dev)
    its = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), its.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    timing = {"vn": 0.0, "syn": 0.0, "cn": 0.0, "n_cn": 0}

    def timed_step():
        step()
        ms, launches = dec.last_timing()  # HIP events recorded on the launch stream around every kernel
        timing["vn"] += ms[0]; timing["syn"] += ms[1]; timing["cn"] += ms[2]; timing["n_cn"] += launches[2]

    # barrier + torch.cuda.synchronize() on both sides, MAX over ranks (nbldpc_amd/ranks.py)
    dt = ranks.timed(rk, timed_step, args.steps, torch.cuda.synchronize)
    ms_vn, ms_syn, ms_cn, n_cn = timing["vn"], timing["syn"], timing["cn"], timing["n_cn"]
    # whole-job sanity counters: every rank's per-lane (converged, decoded word == transmitted word) flags, gathered in lane order
    lane_flags = torch.stack([conv.to(torch.float64), (out == tx_dev).all(dim=1).to(torch.float64)], dim=1).cpu().numpy()
    # (ranks decode equal batches: global lane index = rank * B + lane)
    all_flags = ranks.gather_lane_counters(rk, lane_flags, B * world)
    n_conv, n_correct = ranks.sum_in_lane_order(all_flags)

    total_cw = B * args.steps * world
    value = total_cw / dt
    q, N, E = code.q, code.N, code.E
    # bytes per codeword per iteration, w = 8 (FP64): L_ch read (N) + c2v read, v2c write, v2c read, c2v write (4E)
    bytes_iter = 8 * (q - 1) * (N + 4 * E)
    bytes_cw = 8 * (q - 1) * N + args.iters * bytes_iter + 4 * N + 4
    # dominant kernel.  Fused iteration (one launch = variable-node + EMS check-node pass, the default for this code): its
    # algorithmic bytes are the whole iteration's, 8(q-1)(N+4E) per codeword.  Unfused: the check-node kernel alone reads E v2c
    # vectors and writes E c2v vectors, 8(q-1)2E per codeword.
    fused = ms_vn == 0.0
    cn_bytes_launch = B * (bytes_iter if fused else 8 * (q - 1) * 2 * E)
    cn_ms = ms_cn / max(n_cn, 1)
    cn_gbs = cn_bytes_launch / (cn_ms * 1e-3) / 1e9 if cn_ms > 0 else 0.0
    traffic = pmc_traffic(fused, B)
    res = {
        "metric": "decoded codewords/sec @ 50 iters, GF(256) N=512 rate-1/2",
        "value": value, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if args.data == "randn" else "synthetic (harness link chain: LFSR message, CRC-8, encoder, BPSK, AWGN seed 173+lane)",
        "config": {"workload": f"{CODE} over BPSK/AWGN Eb/N0={args.ebn0} dB, EMS nm={args.nm} nc={args.nc}, "
                               f"{args.iters} fixed iterations, batch {B} codewords per GPU",
                   "batch_per_gpu": B, "iters": args.iters, "parallelism": f"frames sharded over {world} GPU(s), no collective"},
        "algorithmic_GBps_whole_job": total_cw * bytes_cw / dt / 1e9,
        "hbm_roofline_frac_whole_job": total_cw * bytes_cw / dt / 1e9 / (HBM_PEAK_GBS * world),
        "converged_frac": n_conv / (B * world),
        # sanity on the decoded words themselves: frames whose output equals the transmitted codeword (all ranks)
        "frames_correct_frac": n_correct / (B * world),
        "phase_ms_per_step": {"vn": ms_vn / args.steps, "syndrome": ms_syn / args.steps, "cn": ms_cn / args.steps},
        # `achieved` is the contract's figure: ALGORITHMIC bytes of one launch (8(q-1)(N+4E) per codeword and iteration,
        # v2c round trip included) over the mean launch time.  The fused kernel never moves the v2c bytes: what it really pulls from
        # HBM is `traffic` (N + 2E vectors in, E out, second reads served by L2), i.e. `hbm_actual_GBps` -- the kernel is limited by
        # FP64 VALU issue and LDS bandwidth, HBM is the roofline it is priced against, not what it waits for.
        "roofline": {"kernel": "cn_ems_q256_dc4_kernel<32, fused>" if fused else "cn_ems_q256_dc4_kernel<32>", "bound": "hbm", "achieved": cn_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": cn_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": PMC_SUMMARY + " (rocprofv3 --pmc passes of this command; not measured in this run)" if traffic else None,
                     "hbm_actual_GBps": (traffic / (cn_ms * 1e-3) / 1e9) if (traffic and cn_ms > 0) else None,
                     "limiter": limiter_text(),
                     "bytes_per_launch": cn_bytes_launch, "ms_per_launch": cn_ms},
    }
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        threads = os.cpu_count() or 1
        try:
            threads = len(os.sched_getaffinity(0))
        except Exception:
            pass
        threads = min(threads, 16)  # the GPU box gives a 1-GPU job a 16-CPU share
        n = args.cpu_sample if args.cpu_sample > 0 else 4 * threads
        ref = cpu_baseline_reference(nb, args.nm, args.nc, args.iters, threads)
        port = cpu_baseline_port(nb, L[:n].cpu().numpy(), args.nm, args.nc, args.iters, threads)
        res["cpu_baseline"] = ref if ref is not None else port
        res["cpu_baseline_port"] = port
    dec.close()
    if rank == 0 and world == 1 and args.other_configs != 0:
        del L, out, conv, its, tx_dev
        torch.cuda.empty_cache()
        res["other_configs"] = other_configs(local_rank)
        # The headline kernel does less on codewords whose syndrome is already zero (exact short lists): the same
        # configuration where nothing converges (every check on the full path) and where everything does, one short pass each
        import bench_config
        res["headline_by_convergence"] = []
        for e in (0.0, 2.0):
            rr = bench_config.run_config("cfg3", B, 1, e, local_rank)
            res["headline_by_convergence"].append({"ebn0_dB": e, "value": rr["codewords_per_s"], "unit": "codewords/s", "converged_frac": rr["converged_frac"],
                                                   "ms_per_launch": rr["cn_ms_per_launch"], "data": "synthetic (all-zero codeword + randn noise)"})
    if rank == 0:
        print(json.dumps(res))
    ranks.finish(rk)


if __name__ == "__main__":
    main()
