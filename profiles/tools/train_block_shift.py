"""Step time of the resident trainer against the address of its block
(NB_TRAIN_BLOCK_SHIFT moves it inside its allocation, KB): which address bits
decide between the fast and the slow placement?"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nautilus_amd import emulator
os.environ['NB_TRAIN_DEBUG_BLOCK'] = '1'

def fit(x, y, e, n_epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_epochs, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t

d, n_row, e = (int(sys.argv[1]), 24000, int(sys.argv[2])) if len(sys.argv) > 2 else (50, 24000, 4)
x = torch.randn((n_row, d), dtype=torch.float64, device='cuda'); y = torch.rand(n_row, dtype=torch.float64, device='cuda')
os.environ['NB_TRAIN_BLOCK_SHIFT'] = '0'
fit(x, y, e, 2)
steps = (n_row + 199) // 200
for kb in [0, 4, 8, 12, 16, 24, 32, 40, 48, 56, 64, 72, 128, 136]:
    os.environ['NB_TRAIN_BLOCK_SHIFT'] = str(kb)
    t_s, t_l = fit(x, y, e, 32), fit(x, y, e, 96)
    print('shift %4d KB: %.2f us/step' % (kb, (t_l - t_s) / (64 * steps) * 1e6), flush=True)
