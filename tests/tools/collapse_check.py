"""Debug aid: the training set of a collapsed emulator (NB_DUMP_COLLAPSE)
through the numpy oracle and the device trainer, side by side."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from nautilus_amd import emulator, device
from oracle import mlp_oracle as mo
d = np.load(sys.argv[1])
x, y = d['x_t'], d['score']
print('n', x.shape, 'x_t abs max', np.abs(x).max(), 'std per dim min/max',
      x.std(0).min(), x.std(0).max(), flush=True)
xs = (x - x.mean(0)) / x.std(0)
print('xs abs max', np.abs(xs).max(), 'q99.9', np.quantile(np.abs(xs), 0.999),
      flush=True)
print('mean/scale vs device', np.abs(d['mean'] - x.mean(0)).max(),
      np.abs(d['scale'] - x.std(0)).max(), flush=True)
xt = torch.from_numpy(xs).cuda()
nets, _ = emulator.train_networks(xt, torch.from_numpy(y).cuda(), [0, 1, 2, 3],
                                  hparams=emulator._hparams_from_kwargs({}))
for s, net in enumerate(nets):
    print('dev seed', s, 'n_iter', net.n_iter_, 'curve',
          np.round(net.loss_curve_[:6], 5), '...',
          np.round(net.loss_curve_[-3:], 5), flush=True)
for s in (0, 1):
    ref = mo.fit_network(xs, y, s, max_iter=int(nets[s].n_iter_))
    print('ref seed', s, 'n_iter', len(ref.loss_curve), 'curve',
          np.round(ref.loss_curve[:6], 5), '...',
          np.round(ref.loss_curve[-3:], 5), flush=True)
    m = min(len(ref.loss_curve), len(nets[s].loss_curve_))
    print('   max rel diff of curves',
          np.max(np.abs(np.array(ref.loss_curve[:m]) /
                        np.array(nets[s].loss_curve_[:m]) - 1)), flush=True)
