import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')  # run from the repo root
import numpy as np, torch
from oracle import mlp_oracle as mo
from nautilus_amd import device
for d in [int(v) for v in sys.argv[1:]]:
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d,d)); cov = A@A.T/d + np.eye(d); B = np.linalg.cholesky(cov*0.02)
    nets=[mo.glorot_init(d, e)[:2] for e in range(4)]
    nbd = device.DeviceBound(d, [], None, False, [dict(ellipsoid=device.member(0.5*np.ones(d), B), score_predict_min=0.0, mlp=dict(mean=np.zeros(d), scale=np.ones(d), nets=nets))])
    n = 1 << 20
    x = torch.rand((n,d), dtype=torch.float64, device='cuda')
    for _ in range(3): nbd.neural_score(x)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): nbd.neural_score(x)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print('D=%d %.3f ms %.2f TF' % (d, dt*1e3, 2*(100*d+6020)*4*n/dt/1e12))
