"""How fast do two implementations of the SAME fit drift apart?  (VERDICT r3,
weak #2: on the 177 801-row training set of a collapsed C4 emulator the device
and the numpy oracle agreed for 54 epochs and parted in the last two.)

Three runs of MLPRegressor.fit on one large training set, same initial
weights, same minibatch orders:
  A  the numpy oracle (oracle/mlp_oracle.py),
  B  the numpy oracle on the column-reversed inputs with the rows of W1
     reversed -- the same network in exact arithmetic, a different summation
     order in the first layer's products: rounding differences only,
  D  the device trainer (nb_mlp_train.hip).
Printed per epoch: |loss_B / loss_A - 1| and |loss_D / loss_A - 1|.  Both
start at rounding level and grow at the same rate: Adam at the reference's
learning rate of 1e-2 amplifies a 1e-16 perturbation by roughly an order of
magnitude every few epochs, whoever introduced it.

python tests/tools/train_divergence.py [n_rows] [epochs]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nautilus_amd import emulator  # noqa: E402
from oracle import mlp_oracle as mo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
n_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = 50
rng = np.random.default_rng(11)
# a target like the sampler's: rank of a radial function, noisy
x = rng.normal(size=(n, d))
r = np.linalg.norm(x[:, :8], axis=1) + 0.3 * rng.normal(size=n)
y = np.argsort(np.argsort(-r)) / n
for seed in (0, 1):
    rs = np.random.RandomState(seed)
    init = emulator._glorot(d, rs)
    perms, order = [], np.arange(n)
    for ep in range(n_ep):
        idx = np.arange(n)
        rs.shuffle(idx)
        order = order[idx]
        perms.append(order.copy())
    a = mo.fit_network(x, y, seed, max_iter=n_ep, n_iter_no_change=10**6,
                       permutations=perms, init=init)
    init_b = ([init[0][0][::-1].copy()] + [c.copy() for c in init[0][1:]],
              [c.copy() for c in init[1]])
    b = mo.fit_network(x[:, ::-1].copy(), y, seed, max_iter=n_ep,
                       n_iter_no_change=10**6, permutations=perms,
                       init=init_b)
    nets, _ = emulator.train_networks(
        torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), [seed],
        hparams=dict(n_iter_no_change=10**6), permutations=[perms],
        init=[init], max_epochs=n_ep)
    la, lb = np.array(a.loss_curve), np.array(b.loss_curve)
    ld = np.array(nets[0].loss_curve_)
    print('seed %d, n = %d, d = %d' % (seed, n, d))
    print('  epoch   loss (A)      |B/A - 1|    |D/A - 1|')
    for ep in range(n_ep):
        print('  %4d   %.6e   %.2e     %.2e' % (
            ep + 1, la[ep], abs(lb[ep] / la[ep] - 1),
            abs(ld[ep] / la[ep] - 1)), flush=True)
