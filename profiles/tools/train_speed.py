import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nautilus_amd import emulator
for d, nrow, e in [(50, 24000, 4), (20, 8000, 4), (100, 24000, 8), (30, 30000, 4)]:
    X = torch.randn((nrow, d), dtype=torch.float64, device='cuda')
    y = torch.rand(nrow, dtype=torch.float64, device='cuda')
    emulator.train_networks(X, y, list(range(e)), max_epochs=2)
    ne = 48
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(X, y, list(range(e)), max_epochs=ne, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    steps = ne * ((nrow + 199) // 200)
    print('D=%d E=%d n=%d: %.2f us/step' % (d, e, nrow, dt / steps * 1e6), flush=True)
