#!/bin/bash
# every measured configuration once (tools/bench_config.py), one JSON line each: tools/all_configs.sh > profiles/rNN_configs.jsonl
for cfg in "cfg1 65536 2" "cfg2 4096 4" "cfg3 16384 2" "cfg3nc2 4096 2" "cfg4 8192 2 3.0" "cfg5 2048 1 4.0" "cfg5 2048 1 2.8" "cfg5 1024 1 10.0" "tems256 2048 2" "ems64 4096 2" "bp64 4096 2" "ems16 8192 2" "tems16 8192 2" "bp16 8192 2"; do
  python tools/bench_config.py $cfg 2>/dev/null | grep "^{"
done
