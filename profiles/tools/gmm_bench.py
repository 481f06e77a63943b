"""Time of one ``device.gmm_fit`` (ten restarts, Union.split's mixture fit,
bounds/union.py:185-187) per shape; with the timing build
(NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so) restart 0 also
prints its cycle counts per phase.

    python profiles/tools/gmm_bench.py [d n]...
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from nautilus_amd import device  # noqa: E402

shapes = [(50, 2000), (50, 10000), (20, 2000), (100, 10000), (10, 10000)]
if len(sys.argv) > 2:
    a = [int(v) for v in sys.argv[1:]]
    shapes = list(zip(a[::2], a[1::2]))
for d, n in shapes:
    rng = np.random.default_rng(d + n)
    # an elongated cloud with a weak second mode: what Union.split sees
    x = rng.normal(size=(n, d)) * 0.02 + 0.5
    x[: n // 3] += 0.05 * rng.normal(size=d)
    xt = torch.from_numpy(x).cuda()
    fits = device.gmm_fit(xt, seed=1)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        fits = device.gmm_fit(xt, seed=1 + rep)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print('d=%d n=%d: %.2f ms per fit (min of 5; all: %s)  iterations of the '
          'restarts: %s' % (d, n, 1e3 * min(ts),
                            ' '.join('%.1f' % (1e3 * t) for t in ts),
                            [f['n_iter'] for f in fits]), flush=True)
