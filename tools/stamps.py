"""Diagnostic: in-kernel cycle breakdown of the EMS check-node kernel (not a benchmark).

Needs the stamps build of the library:  make -C nbldpc_amd/csrc stamps  &&
NBL_HIP_LIB=$PWD/nbldpc_amd/csrc/ab/libnbldpc_hip_stamps.so python tools/stamps.py
(the default build leaves the stamps out: their accumulators cost 26 SGPRs and push the kernel into SGPR spills)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nbldpc_amd as nb
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
from bench import synth_llr, CODE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
code = nb.Code(CODE)
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dec = nb.Decoder(code, nb.METHOD_EMS, ITERS, ems_nm=32, ems_nc=3, fixed_iters=1, max_batch=B)
EBN0 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
L = synth_llr(torch, 256, 64, B, EBN0, 173, torch.device("cuda", 0)).cpu().numpy()
dec.decode(L)
lib = dec.lib
lib.nbl_debug_stamps.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
out = (C.c_ulonglong * 16)()
lib.nbl_debug_stamps(dec.h, 1, None)
dec.decode(L)
lib.nbl_debug_stamps(dec.h, 0, out)
n = out[15]
names = ["load + VN pass", "rank 0", "histogram", None, "cut + lists", "staging + conf(q,1)", "pair scatters", "gather convs", "emit"]
tot = sum(out[i] for i in range(9) if names[i])
print("quickselect loop iterations per check:", out[9] / n)
for i, nme in enumerate(names):
    if nme is None:
        continue
    print(f"{nme:14s} {out[i]/n:10.0f} cycles/check  {100*out[i]/tot:5.1f}%")
print("total", tot / n, "cycles per check-wave (s_memtime ticks), samples", n)
# run statistics for tools/isa_budget.py (gather_conv runs twice per check: the trips are per check, both calls together)
import json
counts = dict(qs_trips_per_check=out[9] / n, inexact_edges_per_check=out[10] / n, un4_trips_per_check=out[11] / n, rem_trips_per_check=out[12] / n,
              short_list_checks_frac=out[13] / n, list_entries_per_check=out[14] / n,
              batch=B, samples=int(n), workload=f"bench.py inputs (config 3, {EBN0} dB), iteration 1..{ITERS} of a fixed-iteration decode")
print(json.dumps(counts))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(counts, open("gpurun_out/r03_stamps_counts.json", "w"), indent=1)
