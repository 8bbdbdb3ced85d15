#!/bin/bash
out=gpurun_out/r03_call2.txt
: > $out
echo "== reproducer" >> $out
timeout -k 10 120 tools/repro/tems256_dp_soa >> $out 2>&1; echo "rc=$?" >> $out
echo "== new tests" >> $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "degree_one or small_field or field_sizes or tems_gf64_every or regression" > gpurun_out/r03_t1.log 2>&1; echo "parity subset rc=$? $(tail -1 gpurun_out/r03_t1.log)" >> $out
timeout -k 10 600 python -m pytest tests/test_gpu_noise.py tests/test_abi.py -q -x > gpurun_out/r03_t2.log 2>&1; echo "noise+abi rc=$? $(tail -1 gpurun_out/r03_t2.log)" >> $out
for cfg in "cfg4 8192 2 3.0" "cfg5 2048 1 4.0" "tems256 2048 2" "ems16 8192 2" "tems16 8192 2" "bp16 8192 2" "ems64 4096 2" "bp64 4096 2" "cfg2 4096 2"; do
  echo "== $cfg" >> $out
  tools/ab_cfg.sh "nostrict xcd" $cfg >> $out 2>&1
done
cat $out
