#!/bin/bash
# tools/ubench/fp64_power.sh -- builds tools/ubench/fp64_power.hip and runs both variants, sampling rocm-smi once per second
hipcc -O3 --offload-arch=gfx950 tools/ubench/fp64_power.hip -o /tmp/fp64_power || exit 1
for v in valu mfma; do
  /tmp/fp64_power $v 7 > /tmp/fp64_power_$v.txt &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr -s ' ' | sed 's/GPU\[0\]\t\t: //' | tr '\n' '|'; echo
    sleep 1
  done
  wait $pid
  cat /tmp/fp64_power_$v.txt
done
