#!/bin/bash
# small-field kernels (64 / q checks per wave, fused iteration) against the one-check-per-wave kernels: tools/ab_small.sh [batch] [configs]
B=${1:-8192}
CFGS=${2:-"ems16 tems16 bp16"}
for cfg in $CFGS; do
  for v in small general; do
    if [ $v = general ]; then export NBL_NO_SMALL=1; else unset NBL_NO_SMALL; fi
    python tools/bench_config.py $cfg $B 2 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $v', round(d['codewords_per_s']), d['phase_ms'], d['converged_frac'])"
  done
done
