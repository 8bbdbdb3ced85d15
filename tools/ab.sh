#!/bin/bash
# A/B of library builds on the same box: tools/ab.sh "A B C" [bench.py args]; libs in nbldpc_amd/csrc/ab/lib{A,B,...}.so
vs=$1; shift
for rep in 1 2 3; do
  for v in $vs; do
    NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/lib$v.so python bench.py --cpu-sample 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['roofline']['ms_per_launch'],4), d['frames_correct_frac'])"
  done
done
