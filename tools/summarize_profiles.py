#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh (gpurun_out/prof_<tag>_{stats,sq1,sq2,fetch,write}, rocpd sqlite
databases) into the small, committed summaries under profiles/: <tag>_summary.json and <tag>_kernel_stats.csv.

usage: python tools/summarize_profiles.py <tag> ["description of the profiled command"]
"""
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
KEEP = ("cn_", "vn_kernel", "syn_kernel", "init_kernel", "demod_kernel", "noise_")


def db(tag, part):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{part}", "*", "*.db"))
    return sqlite3.connect(f[0]) if f else None


def counters(con):
    res = {}
    if con is None:
        return res
    for k, c, v, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if any(x in k for x in KEEP):
            res.setdefault(k, {})[c] = v
    return res


def main():
    tag = sys.argv[1]
    what = sys.argv[2] if len(sys.argv) > 2 else ""
    os.makedirs(OUT, exist_ok=True)
    summary = {"source": "rocprofv3 --kernel-trace --stats and --pmc passes (tools/profile.sh, one MI355X): " + what, "kernels": {}}
    st = db(tag, "stats")
    rows = [r for r in st.execute("select name, total_calls, total_duration, average, percentage from top_kernels") if any(x in r[0] for x in KEEP)]
    with open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2] * 1000), r[3] * 1000, r[4]])
            summary["kernels"][r[0]] = {"calls": r[1], "avg_us": r[3], "pct": r[4]}
    fetch, write = counters(db(tag, "fetch")), counters(db(tag, "write"))
    hbm = {}
    for k in set(fetch) | set(write):
        f_kb = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
        w_kb = write.get(k, {}).get("WRITE_SIZE", 0.0)
        # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
        # (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
        hbm[k] = {"FETCH_SIZE_KiB_per_launch": f_kb, "WRITE_SIZE_KiB_per_launch": w_kb, "hbm_bytes_per_launch_corrected": (2 * f_kb + w_kb) * 1024}
    summary["hbm_pmc"] = hbm
    sq = {}
    for part in ("sq1", "sq2"):
        for k, v in counters(db(tag, part)).items():
            sq.setdefault(k, {}).update(v)
    # per-wave figures the design discussion quotes (SQ_* cycle counters are in quad-cycles)
    for k, v in sq.items():
        wv = v.get("SQ_WAVES")
        if wv:
            v["per_wave"] = {"valu_insts": v.get("SQ_INSTS_VALU", 0) / wv, "salu_insts": v.get("SQ_INSTS_SALU", 0) / wv, "lds_insts": v.get("SQ_INSTS_LDS", 0) / wv,
                             "vmem_insts": v.get("SQ_INSTS_VMEM", 0) / wv, "wave_cycles": 4 * v.get("SQ_WAVE_CYCLES", 0) / wv,
                             "valu_busy_of_wave_lifetime_x4_waves": (4 * v.get("SQ_ACTIVE_INST_VALU", 0)) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
                             "lds_active_cycles": v.get("SQ_LDS_IDX_ACTIVE", 0) / wv, "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0) / wv}
    summary["sq_pmc"] = sq
    json.dump(summary, open(os.path.join(OUT, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1)[:6000])


if __name__ == "__main__":
    main()
