"""cProfile of the host side of 20 sampling-phase steps of the headline run
(where do the ~0.75 ms of idle queue per step come from?)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
from nautilus_amd import GaussianLikelihood, Sampler, unit_prior  # noqa: E402

d = 50
like = GaussianLikelihood(np.full(d, 0.5), np.eye(d) * 0.05**2)
s = Sampler(unit_prior, like, n_dim=d, n_live=2000, n_networks=4,
            n_batch=16384, vectorized=True, seed=0)
s.run(n_eff=0, n_shell=0, discard_exploration=True, timeout=300)
s.n_batch = 65536
for _ in range(6):
    s.add_samples(s._next_shell())
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(20):
    s.add_samples(s._next_shell())
torch.cuda.synchronize()
prof.disable()
st = pstats.Stats(prof)
st.sort_stats('tottime').print_stats(28)
