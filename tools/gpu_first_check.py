import sys, json, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import nbldpc_amd as nb, pyoracle as po
from conftest import load_golden, decoder_kwargs
def run(name):
    g, meta = load_golden(name); p = meta['profile']; kw = decoder_kwargs(p)
    code = nb.Code(meta['code'])
    L = g['L_ch']; B = L.shape[0]
    oc = po.Code(edges=nb.datafiles.code_edges(meta['code'])[:3] + tuple(nb.datafiles.code_edges(meta['code'])[3:])); ogf = po.GF(code.q)
    for k, it in enumerate(g['iters']):
        dec = nb.Decoder(code, p['method'], int(it), **kw)
        out, conv, iters = dec.decode(L)
        okout = np.array_equal(out, g['out'][k]); okflag = np.array_equal(conv, g['syn_ok'][k])
        print(name, 'maxIter', it, 'out==ref', okout, 'flag==ref', okflag, 'iters', iters.tolist())
        dec.close()
    for k, it in enumerate(g['state_iters']):
        dec = nb.Decoder(code, p['method'], int(it), **kw); dec.record_state(True)
        dec.decode(L)
        od = po.Decoder(oc, ogf, p['method'], int(it), po.CANONICAL, **kw)
        for li, lane in enumerate(g['state_lanes']):
            P,V,Cc = dec.read_state(int(lane))
            od.decode(L[lane]); oP,oV,oC = od.state()
            print('  state it',it,'lane',lane,'GPU==oracle-canonical: post',np.array_equal(P,oP),'v2c',np.array_equal(V,oV),'c2v',np.array_equal(Cc,oC),
                  '| vs ref maxabs c2v %.2e'%np.max(np.abs(Cc-g['st_c2v'][k,li])))
        dec.close()
for n in sys.argv[1:]: run(n)
