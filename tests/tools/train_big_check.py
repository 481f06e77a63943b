import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from nautilus_amd import emulator
from oracle import mlp_oracle as mo
rng = np.random.default_rng(3)
for n, d, ne in [(70001, 50, 2), (20017, 50, 2), (5199, 20, 3), (40000, 50, 2)]:
    x = rng.normal(size=(n, d)); y = rng.random(n)
    nets, _ = emulator.train_networks(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), [0, 1, 2, 3], max_epochs=ne)
    ref = mo.fit_network(x, y, 2, max_iter=ne)
    net = nets[2]
    print(n, d, 'loss rel', np.max(np.abs(np.array(net.loss_curve_) / np.array(ref.loss_curve) - 1)),
          'w', max(np.max(np.abs(net.coefs_[k] - ref.coefs[k])) for k in range(4)), flush=True)
