#!/bin/bash
# Which part of GVN?  The SoA variant of the GF(256) T-EMS kernel (tools/repro/tems256_soa.hip) built with one GVN feature off at a
# time, judged by the named regression test.  Runs on the GPU box (see bisect_soa.sh).
set -u
cd "$(dirname "$0")/../.."
FLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-result -x hip"
OBJS=$(ls nbldpc_amd/csrc/build/*.o | grep -v nbl_cn_tems256)
T="tests/test_gpu_parity.py::test_tems_gf256_nr3_nc2_integer_llr_regression"
out=gpurun_out/r03_gvn_options.txt
: > $out
for opt in "" "-mllvm -enable-pre=false" "-mllvm -enable-load-pre=false" "-mllvm -enable-gvn-memdep=false" "-mllvm -enable-pre=false -mllvm -enable-load-pre=false"; do
  hipcc $FLAGS $opt -c -o /tmp/soa_opt.o tools/repro/tems256_soa.hip 2> /tmp/soa_opt.err || { echo "[$opt] compile failed: $(tail -1 /tmp/soa_opt.err)" >> $out; continue; }
  hipcc -fPIC --offload-arch=gfx950 -shared -o /tmp/libsoa.so $OBJS /tmp/soa_opt.o
  NBL_HIP_LIB=/tmp/libsoa.so timeout -k 10 120 python -m pytest $T -x -q > /tmp/soa_test.log 2>&1
  echo "[${opt:-default}] regression test rc=$? ($(tail -1 /tmp/soa_test.log))" >> $out
done
cat $out
