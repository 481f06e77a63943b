import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from nautilus_amd import emulator
def fit(x, y, e, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t
for d, e in [(50, 16), (50, 8), (50, 12)]:
    x = torch.randn((24000, d), dtype=torch.float64, device='cuda'); y = torch.rand(24000, dtype=torch.float64, device='cuda')
    fit(x, y, e, 2); fit(x, y, e, 32)
    b = min((fit(x, y, e, 96) - fit(x, y, e, 32)) / (64 * 120) * 1e6 for r in range(3))
    print('D=%d E=%d: %.2f us per step' % (d, e, b), flush=True)
