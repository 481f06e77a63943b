"""Several ensembles training in one fleet (the neural bounds of a multi-modal
NautilusBound): wall time per Adam step of the fleet, host start-up removed by
differencing a long and a short run."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nautilus_amd import emulator


def run(m, d, nrow, e, ne):
    jobs = []
    for j in range(m):
        X = torch.randn((nrow, d), dtype=torch.float64, device='cuda')
        y = torch.rand(nrow, dtype=torch.float64, device='cuda')
        jobs.append(dict(xs=X, y=y, seeds=list(range(e)), max_epochs=ne,
                         hparams=dict(n_iter_no_change=100000)))
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_ensembles(jobs)
    torch.cuda.synchronize()
    return time.perf_counter() - t


for m, d, nrow, e in [(1, 50, 20000, 4), (2, 50, 20000, 4), (4, 50, 20000, 4),
                      (1, 100, 20000, 8), (2, 100, 20000, 8)]:
    run(m, d, nrow, e, 2)
    t0, t1 = run(m, d, nrow, e, 16), run(m, d, nrow, e, 64)
    steps = 48 * ((nrow + 199) // 200)
    print('M=%d ensembles x E=%d networks, D=%d, n=%d: %.1f us per step of '
          'the fleet = %.1f us per step and ensemble' % (
              m, e, d, nrow, (t1 - t0) / steps * 1e6,
              (t1 - t0) / steps * 1e6 / m), flush=True)
