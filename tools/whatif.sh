#!/bin/bash
# Sensitivity of the headline kernel: diagnostic builds that leave ONE kind of work out (results are wrong on purpose):
#   make -C nbldpc_amd/csrc OBJDIR=build_w$w XFLAGS=-DNBL_WHATIF=$w OUT=ab/libw$w.so   for w = 1 (no gather LDS reads),
#   2 (no gather additions), 3 (no pair-convolution atomics), 4 (no HBM reads), 5 / 6 (three / two waves per SIMD),
#   7 (plain stores instead of the atomics), 10 (every gather read from one address: no LDS bandwidth),
#   11 (channel vectors of codewords 0..7 for every wave: L2 hits instead of HBM first touches);
# then tools/whatif.sh on the GPU box.  ONE iteration per decode (--iters 1): every variant sees the same inputs -- with more
# iterations the wrong results of a variant change the data of the later iterations, and the kernel's time depends on the data.
for rep in 1 2; do
  for v in ${VARIANTS:-abd w1 w2 w3 w4 w5 w6}; do
    NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/lib$v.so python bench.py --cpu-sample 0 --other-configs 0 --iters ${ITERS:-1} --steps 20 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['ms_per_launch'],4))"
  done
done
