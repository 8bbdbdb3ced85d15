#!/bin/bash
# A/B of library builds on another configuration: tools/ab_cfg.sh "A B" cfg4 8192
vs=$1; shift
for rep in 1 2 3; do
  for v in $vs; do
    NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/lib$v.so python tools/bench_config.py "$@" 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['codewords_per_s']), d['phase_ms'], d['converged_frac'])"
  done
done
