#!/bin/bash
# A/B of two builds of the HIP library on the same box: tools/ab.sh [bench.py args]; libs in nbldpc_amd/csrc/ab/lib{A,B}.so
for rep in 1 2 3; do
  for v in A B; do
    NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/lib$v.so python bench.py --cpu-sample 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['roofline']['ms_per_launch'],4))"
  done
done
