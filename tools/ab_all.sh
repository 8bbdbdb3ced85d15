#!/bin/bash
# A/B of library builds over every measured configuration: tools/ab_all.sh "A B"   (libs nbldpc_amd/csrc/ab/lib{A,B}.so)
vs=$1
echo "== cfg3 (bench.py)"; tools/ab.sh "$vs" --steps 5 --warmup 1 --other-configs 0 2>&1 | awk '{print $1, $2, $3}'
for cfg in "cfg2 4096 4" "cfg3nc2 4096 2" "cfg4 8192 2 3.0" "cfg5 2048 1 4.0" "tems256 2048 2" "ems64 4096 2" "bp64 4096 2" "ems16 8192 2" "tems16 8192 2" "bp16 8192 2"; do
  echo "== $cfg"; tools/ab_cfg.sh "$vs" $cfg 2>&1 | awk '{print $1, $2}'
done
