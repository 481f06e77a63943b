"""Several ensembles training concurrently (the neural bounds of a multi-modal
NautilusBound): wall time and Adam steps per second."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nautilus_amd import emulator

for m, d, nrow, e in [(1, 50, 5000, 4), (2, 50, 5000, 4), (4, 50, 5000, 4), (6, 50, 3000, 4)]:
    jobs = []
    for j in range(m):
        X = torch.randn((nrow + 100 * j, d), dtype=torch.float64, device='cuda')
        y = torch.rand(nrow + 100 * j, dtype=torch.float64, device='cuda')
        jobs.append(dict(xs=X, y=y, seeds=list(range(e)), max_epochs=32,
                         hparams=dict(n_iter_no_change=100000)))
    torch.cuda.synchronize(); t = time.perf_counter()
    out = emulator.train_ensembles(jobs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    steps = 32 * ((nrow + 199) // 200)
    print('M=%d ensembles x E=%d, D=%d, n=%d: %.3f s, %.1f us per step of the '
          'slowest ensemble' % (m, e, d, nrow, dt, dt / steps * 1e6), flush=True)
