#!/bin/bash
# Round 3, VERDICT item 1: does the restored six-parallel-array DP layout of the GF(256) T-EMS kernel still fail, and does
# -fno-strict-aliasing change that?  Then the cost of -fno-strict-aliasing on every configuration.
out=gpurun_out/r03_alias.txt
: > $out
T="tests/test_gpu_parity.py::test_tems_gf256_nr3_nc2_integer_llr_regression"
for v in tems256_soa_strict tems256_soa_nostrict cur_soa_strict cur_soa_nostrict strict nostrict; do
  NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/lib$v.so timeout -k 10 300 python -m pytest $T -x -q > gpurun_out/r03_alias_$v.log 2>&1
  rc=$?
  echo "$v regression-test rc=$rc : $(tail -1 gpurun_out/r03_alias_$v.log)" >> $out
  if [ $rc -ge 124 ]; then echo "timeout: stop" >> $out; exit 1; fi
done
cat $out
for cfg in "cfg4 8192 2 3.0" "cfg5 2048 1 4.0" "tems256 2048 2" "ems16 8192 2" "tems16 8192 2" "bp16 8192 2" "ems64 4096 2" "bp64 4096 2" "cfg2 4096 2"; do
  echo "== $cfg" >> $out
  tools/ab_cfg.sh "strict nostrict" $cfg >> $out 2>&1
done
echo "== cfg3 (bench.py)" >> $out
tools/ab.sh "strict nostrict" --steps 5 --warmup 1 >> $out 2>&1
cat $out
