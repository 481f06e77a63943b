import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch, time
from oracle import mlp_oracle as mo
from nautilus_amd import device, _lib
d = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(d)
A = rng.normal(size=(d,d)); cov = A@A.T/d + np.eye(d); B = np.linalg.cholesky(cov*0.02)
nets=[mo.glorot_init(d, e)[:2] for e in range(4)]
nbd = device.DeviceBound(d, [], None, False, [dict(ellipsoid=device.member(0.5*np.ones(d), B), score_predict_min=0.0, mlp=dict(mean=np.zeros(d), scale=np.ones(d), nets=nets))])
n = 1 << 20
x = torch.rand((n,d), dtype=torch.float64, device='cuda')
for _ in range(3): nbd.neural_score(x)
torch.cuda.synchronize()
buf = torch.zeros(32, dtype=torch.int64, device='cuda')
lib = _lib.load()
lib.nb_set_eval_counters(buf.data_ptr())
t=time.perf_counter(); nbd.neural_score(x); torch.cuda.synchronize(); dt=time.perf_counter()-t
lib.nb_set_eval_counters(None)
c = buf.cpu().numpy()[8:]
names=['prologue(load+cube)','ell stage+eval','census+tin+first dma','st0 compute','st0 wait+barrier','st1 compute','st1 wait+barrier','epilogue','store','ell: dma+points wait','ell: barrier']
tot=c.sum()
print('launch %.3f ms, clock total %d ticks -> %.1f MHz' % (dt*1e3, tot, tot/dt/1e6))
for nm,v in zip(names,c): print('%-24s %10d  %5.1f%%' % (nm, v, 100*v/tot))
