import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import torch, nbldpc_amd as nb
from bench import synth_llr, CODE
B=16384
code=nb.Code(CODE)
L=synth_llr(torch,256,64,B,1.0,173,torch.device('cuda',0)).cpu().numpy()
dec=nb.Decoder(code, nb.METHOD_EMS, 50, ems_nm=32, ems_nc=3, fixed_iters=1, max_batch=B)
dec.decode(L[:256])
t=time.perf_counter(); dec.decode(L); dt=time.perf_counter()-t
print('host-buffer decode B=%d: %.1f ms -> %.0f codewords/s (pageable host memory, H2D of %.2f GB + D2H included)'%(B,dt*1e3,B/dt,L.nbytes/1e9))
rx=np.random.default_rng(0).normal(size=(B,512,2))
dec.set_demodulator(2,512,np.arange(512))
t=time.perf_counter(); dec.decode_samples(rx,0.9); dt=time.perf_counter()-t
print('sample-input decode (device demodulator) B=%d: %.1f ms -> %.0f codewords/s (H2D of %.3f GB)'%(B,dt*1e3,B/dt,rx.nbytes/1e9))
