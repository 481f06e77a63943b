import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from helpers import upload
from oracle import bounds_oracle as bo
for d in (8, 20, 50, 64, 100):
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d,d)); cov = A@A.T/d + np.eye(d); B = np.linalg.cholesky(cov*0.02)
    b = upload(bo.OEllipsoid.from_params(0.5*np.ones(d), B))
    n = 1 << 21
    for _ in range(3): b.propose(1, 0, n)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): x = b.propose(1, 0, n)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print('D=%d propose %.3f ms %.2f Gprop/s  checksum %.12f' % (d, dt*1e3, n/dt/1e9, float(x.sum())))
