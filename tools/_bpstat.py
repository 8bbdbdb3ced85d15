import sys, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, nbldpc_amd as nb
import bench_config as bc
from nbldpc_amd import binding
lib = binding.load_library()
for name, eb, iters in (("cfg5", 10.0, 100), ("cfg5", 7.0, 100), ("cfg5", 6.0, 30), ("cfg5", 5.0, 30)):
    c = bc.CFG[name]; dev = torch.device("cuda", 0)
    code = nb.Code(c["code"]); B = 256
    L = bc.synth(code, B, eb, c["mod"], dev).contiguous()
    dec = nb.Decoder(code, c["method"], iters, fixed_iters=1, max_batch=B)
    out = torch.zeros((B, code.N), dtype=torch.int32, device=dev); conv = torch.zeros(B, dtype=torch.uint8, device=dev)
    z = (C.c_ulonglong * 8)()
    lib.nbl_debug_bp_stats(z, 1)
    dec.decode_device(L.data_ptr(), B, out.data_ptr(), conv.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.nbl_debug_bp_stats(z, 0)
    print(name, eb, iters, "conv", float(conv.float().mean()), "wide convs", z[0], "fallbacks", z[1], flush=True)
