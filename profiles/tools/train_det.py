import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
from nautilus_amd import emulator
def run():
    torch.manual_seed(0)
    g = torch.Generator(device='cuda').manual_seed(1)
    X = torch.randn((24000,50), dtype=torch.float64, device='cuda', generator=g); y = torch.rand(24000, dtype=torch.float64, device='cuda', generator=g)
    nets, st = emulator.train_networks(X, y, [0,1,2,3], max_epochs=40, hparams=dict(n_iter_no_change=100000))
    return np.concatenate([np.concatenate([c.ravel() for c in n.coefs_] + [np.asarray(n.loss_curve_)]) for n in nets])
a = run(); b = run()
print('resident run1 == run2 bitwise:', np.array_equal(a, b), np.abs(a-b).max())
np.save('/tmp/res.npy', a)
os.environ['NB_TRAIN_TWO_LAUNCH'] = '1'
c = run()
print('resident == two-launch bitwise:', np.array_equal(a, c), np.abs(a-c).max())
