import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')  # run from the repo root
import numpy as np, torch
from oracle import mlp_oracle as mo
from nautilus_amd import device
d = 50
rng = np.random.default_rng(d)
A = rng.normal(size=(d,d)); cov = A@A.T/d + np.eye(d); B = np.linalg.cholesky(cov*0.02)
nets=[mo.glorot_init(d, e)[:2] for e in range(4)]
nbd = device.DeviceBound(d, [], None, False, [dict(ellipsoid=device.member(0.5*np.ones(d), B), score_predict_min=0.0, mlp=dict(mean=np.zeros(d), scale=np.ones(d), nets=nets))])
n = 1 << 20
x = torch.rand((n,d), dtype=torch.float64, device='cuda')
for _ in range(6): nbd.neural_score(x)
torch.cuda.synchronize()
