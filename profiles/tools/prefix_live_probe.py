import sys, numpy as np, torch
sys.path.insert(0, '.')
from nautilus_amd import Sampler, unit_prior
from nautilus_amd.configs import baseline_config
c = baseline_config('C5')
for seed in (0, 1):
    s = Sampler(unit_prior, c['likelihood'], n_dim=100, n_live=2000, n_networks=4, n_batch=100, vectorized=True, seed=seed)
    for n in (140000, 150000, 160000, 164200, 170000):
        s.run(n_like_max=n, discard_exploration=True)
        ll = np.concatenate(s.log_l); pts = np.concatenate(s.points)
        live = pts[np.argsort(ll)[-s.n_live:]]
        fr = [float(np.mean(b.contains(live))) for b in s.bounds[-3:]]
        print(seed, 'n_like', s.n_like, 'bounds', len(s.bounds), 'shell_n last', s.shell_n[-3:], 'inside', np.round(fr, 3), flush=True)
