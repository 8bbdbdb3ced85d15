#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small, committed summaries under profiles/.

usage: python tools/summarize_profiles.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [<pmc_sq_dir> ...]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")


def counters(d):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    os.makedirs(OUT, exist_ok=True)
    st = glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv"))[0]
    rows = [r for r in csv.DictReader(open(st))]
    keep = [r for r in rows if any(k in r["Name"] for k in ("cn_", "vn_kernel", "syn_kernel", "init_kernel"))]
    with open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    summary = {"source": "rocprofv3 --kernel-trace --stats / --pmc on `python3 bench.py` (1 MI355X, batch 16384, 50 iterations)",
               "kernels": {}}
    for r in keep:
        summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
    fetch, write = counters(fetch_dir), counters(write_dir)
    hbm = {}
    for k in set(fetch) | set(write):
        if not any(x in k for x in ("cn_", "vn_kernel", "syn_kernel", "init_kernel")):
            continue
        f_kb = sum(fetch[k]["FETCH_SIZE"]) / max(len(fetch[k]["FETCH_SIZE"]), 1) if k in fetch else 0.0
        w_kb = sum(write[k]["WRITE_SIZE"]) / max(len(write[k]["WRITE_SIZE"]), 1) if k in write else 0.0
        # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
        # (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
        hbm[k] = {"FETCH_SIZE_KiB_per_launch": f_kb, "WRITE_SIZE_KiB_per_launch": w_kb,
                  "hbm_bytes_per_launch_corrected": (2 * f_kb + w_kb) * 1024}
    summary["hbm_pmc"] = hbm
    for extra in sys.argv[5:]:
        c = counters(extra)
        for k, v in c.items():
            if "cn_" in k or "vn_kernel" in k:
                summary.setdefault("sq_pmc", {}).setdefault(k, {}).update({a: sum(b) / len(b) for a, b in v.items()})
    json.dump(summary, open(os.path.join(OUT, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
