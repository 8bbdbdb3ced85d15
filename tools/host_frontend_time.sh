#!/bin/bash
# builds tools/host_frontend_time.cpp and runs it for config 3's code in a scratch work directory: tools/host_frontend_time.sh [P] [T] [cycles]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
python3 - "$W" "${1:-16384}" <<PY
import sys
sys.path.insert(0, "$ROOT")
from nbldpc_amd import hostlib
hostlib.prepare_workdir(sys.argv[1], dict(gfq=256, code="divsalar.UNBLDPC.512.256.GF.256", method=2, max_iter=50, parallel=int(sys.argv[2]), ems_nm=32, ems_nc=3,
                                          nqam=2, snr_begin=1.5, snr_step=1.0, snr_stop=1.5, constellation="BPSK", seed=173),
                        "divsalar.UNBLDPC.512.256.GF.256", "BPSK")
PY
H=$ROOT/nbldpc_amd/host
g++ -O2 -std=c++17 -ffp-contract=off -I$H -I$ROOT/include $ROOT/tools/host_frontend_time.cpp $H/gf.cpp $H/simulation.cpp $H/nbldpc_host.cpp $H/comm.cpp $H/link.cpp \
    -o $W/hft -pthread -L$ROOT/nbldpc_amd/csrc -lnbldpc_hip -Wl,-rpath,$ROOT/nbldpc_amd/csrc
cd $W && ./hft "${1:-16384}" "${2:-16}" "${3:-4}" "${4:-0}"
rm -rf $W
