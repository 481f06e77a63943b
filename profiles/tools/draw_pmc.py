"""Stand-alone proposal draw for the PMC passes of draw_pmc.sh: one
single-member ellipsoid at n_dim 50, 2^22 proposals per launch."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from helpers import upload
from oracle import bounds_oracle as bo
d = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(d)
A = rng.normal(size=(d, d)); cov = A @ A.T / d + np.eye(d)
B = np.linalg.cholesky(cov * 0.02)
b = upload(bo.OEllipsoid.from_params(0.5 * np.ones(d), B))
n = 1 << 22
for _ in range(6):
    x = b.propose(1, 0, n)
torch.cuda.synchronize()
print('checksum %.12f' % float(x.sum()))
