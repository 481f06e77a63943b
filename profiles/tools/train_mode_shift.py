"""Does the step time of the resident trainer depend on where its pool lands?
One process; between fits a dummy allocation of a few sizes is made and held,
which shifts the blocks hipMalloc hands to the next trainer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nautilus_amd import emulator

def fit(x, y, e, n_epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_epochs, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t

d, n_row, e = 50, 24000, 4
x = torch.randn((n_row, d), dtype=torch.float64, device='cuda'); y = torch.rand(n_row, dtype=torch.float64, device='cuda')
fit(x, y, e, 2)
hold = []
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
for size in [0, 4096, 65536, 1 << 20, 3 << 20, 1 << 24, 12345678, 1 << 26, 0, 0]:
    if size:
        p = ctypes.c_void_p()
        hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size))
        hold.append(p)
    t_s, t_l = fit(x, y, e, 32), fit(x, y, e, 96)
    print('dummy %10d bytes held: %.2f us/step' % (size, (t_l - t_s) / (64 * 120) * 1e6), flush=True)
