"""Quick GPU check + timing of the batched MVEE (dev tool)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from nautilus_amd import device, geometry
from helpers import khachiyan_weights_numpy

rng = np.random.default_rng(0)
for n, d in [(100, 3), (2000, 50), (500, 20), (3000, 100), (10000, 100), (400, 128)]:
    pts = rng.normal(size=(n, d)) * rng.uniform(0.5, 2.0, size=d) + 0.3
    x = torch.from_numpy(pts).cuda()
    u = device.mvee_weights(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        u = device.mvee_weights(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    t0 = time.time()
    res = geometry.mvee_batch([pts])
    dt2 = time.time() - t0
    if n * d <= 300000:
        uh = khachiyan_weights_numpy(pts)
        err = np.abs(u.cpu().numpy() - uh).max()
    else:
        err = float('nan')
    c, a, a_inv = res[0]
    r2 = np.einsum('ij,jk,ik->i', pts - c, a, pts - c)
    print('n=%d d=%d  weights %.2f ms  full fit %.2f ms  |du|=%.2e  sum=%.15f  r2max-1=%.1e'
          % (n, d, dt * 1e3, dt2 * 1e3, err, float(u.sum()), r2.max() - 1), flush=True)

# ill-conditioned sets: strong correlation, outlier clusters (reference tests/test_bounds.py trim test)
from oracle import bounds_oracle as bo
sph = rng.normal(size=(200, 3)); sph /= np.linalg.norm(sph, axis=1)[:, None]
for name, pts in [('outliers 1e7', np.vstack([sph, sph + 10, sph[:30] + 1e7])),
                  ('corr 1-1e-10', np.hstack([sph[:, :1], sph[:, :1] + 1e-5 * sph[:, 1:2], sph[:, 2:]])),
                  ('scale 1e-8', sph * np.array([1.0, 1e-8, 1e4]) + 3.0)]:
    c, a, a_inv = geometry.mvee_batch([pts])[0]
    co, ao, aio = bo.mvee(pts)
    r2 = np.einsum('ij,jk,ik->i', pts - c, a, pts - c)
    lv = 0.5 * np.linalg.slogdet(a_inv)[1]; lvo = 0.5 * np.linalg.slogdet(aio)[1]
    print('%-14s r2max-1=%.1e  log-volume %.6f (oracle %.6f)  |dc|=%.2e' % (name, r2.max() - 1, lv, lvo, np.abs(c - co).max()))
