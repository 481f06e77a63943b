"""Wall time per Adam step of the resident training kernel, host overhead of
the first chunk (shuffles, upload, launch) removed by differencing a long and
a short run of the same networks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nautilus_amd import emulator


def run(X, y, e, ne):
    torch.cuda.synchronize()
    t = time.perf_counter()
    emulator.train_networks(X, y, list(range(e)), max_epochs=ne,
                            hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize()
    return time.perf_counter() - t


for d, nrow, e in [(50, 24000, 4), (20, 8000, 4), (100, 24000, 8),
                   (30, 30000, 4), (50, 2000, 4)]:
    X = torch.randn((nrow, d), dtype=torch.float64, device='cuda')
    y = torch.rand(nrow, dtype=torch.float64, device='cuda')
    run(X, y, e, 2)
    short, long_ = 32, 160
    t0 = min(run(X, y, e, short) for _ in range(2))
    t1 = min(run(X, y, e, long_) for _ in range(2))
    per_epoch = (nrow + 199) // 200
    print('D=%d E=%d n=%d: %.2f us/step (differenced), %.2f us/step incl. '
          'host start-up over %d epochs' % (
              d, e, nrow, (t1 - t0) / ((long_ - short) * per_epoch) * 1e6,
              t1 / (long_ * per_epoch) * 1e6, long_), flush=True)
