import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from nautilus_amd import emulator
d, n_row, e = int(sys.argv[1]), 24000, int(sys.argv[2])
def fit(x, y, e, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t
x = torch.randn((n_row, d), dtype=torch.float64, device='cuda'); y = torch.rand(n_row, dtype=torch.float64, device='cuda')
fit(x, y, e, 2)
steps = 120
b = min((fit(x, y, e, 96) - fit(x, y, e, 32)) / (64 * steps) * 1e6 for r in range(2))
print('D=%d E=%d late_only=%s shape=%s: %.2f us/step' % (d, e, os.environ.get('NB_TRAIN_LATE_ONLY'), os.environ.get('NB_TRAIN_LATE_SHAPE'), b), flush=True)
