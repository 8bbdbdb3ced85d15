#!/bin/bash
# Samples the GPU's shader clock, power and temperature (rocm-smi, once per second) while a configuration of tools/bench_config.py runs:
#   tools/clock_sample.sh cfg5 2048 3 4.0        -> one line per second on stdout, the bench's JSON line last
# The child is a plain background job of this shell (no exec after the GPU is initialised).
python tools/bench_config.py "$@" > /tmp/clock_sample_bench.json 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' '|'
  echo
  sleep 1
done
wait $pid
tail -1 /tmp/clock_sample_bench.json | cut -c1-200
