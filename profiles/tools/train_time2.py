import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from nautilus_amd import emulator
for d, nrow in [(50, 24000)]:
    X = torch.randn((nrow,d), dtype=torch.float64, device='cuda'); y = torch.rand(nrow, dtype=torch.float64, device='cuda')
    emulator.train_networks(X, y, [0,1,2,3], max_epochs=2)
    for ne in (16, 144):
        torch.cuda.synchronize(); t=time.perf_counter()
        nets_t, st = emulator.train_networks(X, y, [0,1,2,3], max_epochs=ne, hparams=dict(n_iter_no_change=100000))
        torch.cuda.synchronize(); dt=time.perf_counter()-t
        steps = ne*((nrow+199)//200)
        print('D=%d epochs %d (%s): %.3f s -> %.2f us/step' % (d, ne, st['n_iter'], dt, dt/steps*1e6))
