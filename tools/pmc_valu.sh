#!/bin/bash
# vector / scalar instructions per wave of the dominant check-node kernel for library builds: tools/pmc_valu.sh "A B" cfg5 2048 1 4.0
vs=$1; shift
export TMPDIR=/tmp
for v in $vs; do
  NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/lib$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM -d gpurun_out/pmc_$v -- python3 tools/bench_config.py "$@" > /dev/null 2>&1
  python3 - "$v" <<'PY'
import sqlite3, glob, sys
v = sys.argv[1]
con = sqlite3.connect(glob.glob(f"gpurun_out/pmc_{v}/*/*.db")[0])
r = {}
for k, c, val in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
    if "cn_" in k:
        r.setdefault(k, {})[c] = val
for k, d in r.items():
    print(v, k[:70], {c: round(x / d["SQ_WAVES"], 1) for c, x in d.items() if c != "SQ_WAVES"})
PY
  rm -rf gpurun_out/pmc_$v
done
