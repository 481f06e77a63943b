"""A/B of host-side choices of the sampling-phase step on the headline run, in
ONE process and on one sampler (blocks of 20 steps in turn, so that the drift
of the growing shells hits both sides alike):
  merged   -- add_samples fetches log L and the shell statistics in one
              transfer (sampler.DEFER_FETCH) or in two
  max_draw -- proposals per refill launch capped at 2^22 or 2^23
python profiles/tools/step_ab.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
from nautilus_amd import GaussianLikelihood, Sampler, unit_prior  # noqa: E402
from nautilus_amd import bounds, sampler  # noqa: E402

d = 50
like = GaussianLikelihood(np.full(d, 0.5), np.eye(d) * 0.05**2)
s = Sampler(unit_prior, like, n_dim=d, n_live=2000, n_networks=4,
            n_batch=16384, vectorized=True, seed=0)
s.run(n_eff=0, n_shell=0, discard_exploration=True, timeout=300)
s.n_batch = 65536
for _ in range(8):
    s.add_samples(s._next_shell())


def block(k=20):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(k):
        s.add_samples(s._next_shell())
    torch.cuda.synchronize()
    return (time.time() - t0) / k * 1e3


settings = [('split waits, 2^22', False, 1 << 22),
            ('merged waits, 2^22', True, 1 << 22),
            ('split waits, 2^23', False, 1 << 23),
            ('merged waits, 2^23', True, 1 << 23)]
res = {name: [] for name, _, _ in settings}
for rnd in range(5):
    for name, merged, cap in settings:
        sampler.DEFER_FETCH = merged
        bounds.MAX_DRAW = cap
        res[name].append(block())
for name, _, _ in settings:
    v = res[name]
    print('%-22s ms per step: %s   median %.3f' % (
        name, ' '.join('%.2f' % x for x in v), float(np.median(v))))
print('log Z %.4f  n_eff %.0f' % (s.log_z, s.n_eff))
