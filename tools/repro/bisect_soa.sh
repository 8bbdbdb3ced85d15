#!/bin/bash
# Which optimisation pass turns the six-parallel-array layout of the GF(256) T-EMS kernel's DP state into wrong path codes?
# Runs ON the GPU box: binary search over hipcc's -opt-bisect-limit for tools/repro/tems256_soa.hip, every candidate linked with
# the product's other objects (nbldpc_amd/csrc/build) and judged by the named regression test.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
FLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-result -x hip"
OBJS=$(ls nbldpc_amd/csrc/build/*.o | grep -v nbl_cn_tems256)
T="tests/test_gpu_parity.py::test_tems_gf256_nr3_nc2_integer_llr_regression"
out=gpurun_out/r03_bisect.txt
: > $out
try() { # $1 = limit (-1 = none); returns 0 if the test passes
  hipcc $FLAGS -mllvm -opt-bisect-limit=$1 -c -o /tmp/soa_$1.o tools/repro/tems256_soa.hip 2> /tmp/soa_$1.err || { echo "compile failed at $1" >> $out; return 2; }
  hipcc -fPIC --offload-arch=gfx950 -shared -o /tmp/libsoa.so $OBJS /tmp/soa_$1.o || return 2
  NBL_HIP_LIB=/tmp/libsoa.so timeout -k 10 120 python -m pytest $T -x -q > /tmp/soa_test.log 2>&1
}
try -1; echo "no limit: rc=$?" >> $out
total=$(grep -c "BISECT: running pass" /tmp/soa_-1.err)
echo "passes reported without limit: $total" >> $out
try 0; echo "limit 0: rc=$?" >> $out
lo=0; hi=$total   # invariant: lo passes, hi fails
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(((lo + hi) / 2))
  if try $mid; then lo=$mid; else hi=$mid; fi
  echo "limit $mid -> $( [ $lo -eq $mid ] && echo pass || echo FAIL )   [$lo, $hi]" >> $out
done
echo "first failing limit: $hi" >> $out
try $hi
grep -n "BISECT: running pass ($hi)" /tmp/soa_$hi.err >> $out
grep "BISECT: running pass ($((hi - 1)))\|BISECT: running pass ($((hi + 1)))\|BISECT: NOT running pass ($((hi + 1)))" /tmp/soa_$hi.err >> $out
cat $out
