"""Exploration phase of the headline run (50-D Gaussian): wall-time breakdown."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nautilus_amd import GaussianLikelihood, Sampler, unit_prior, emulator
d = 50
like = GaussianLikelihood(np.full(d, 0.5), np.eye(d) * 0.05**2)
s = Sampler(unit_prior, like, n_dim=d, n_live=2000, n_networks=4, n_batch=16384, vectorized=True, seed=0)
stats = []
orig = emulator.train_ensembles
def wrapped(jobs, **kw):
    torch.cuda.synchronize(); t0 = time.time()
    out = orig(jobs, **kw)
    torch.cuda.synchronize(); dt = time.time() - t0
    for j, (nets, st) in zip(jobs, out):
        stats.append((st['n_rows'], st['n_iter'], dt))
    return out
emulator.train_ensembles = wrapped
t0 = time.time()
s.run(n_eff=0, n_shell=0, discard_exploration=True, timeout=300)
torch.cuda.synchronize()
print('wall', time.time() - t0, s.timing)
tot_steps = 0
for n, it, dt in stats:
    steps = max(it) * ((n + 199) // 200)
    tot_steps += steps
print('bounds', len(stats), 'train wall', sum(x[2] for x in stats), 'critical-path steps', tot_steps, 'us/step', 1e6 * sum(x[2] for x in stats) / tot_steps)
for n, it, dt in stats[::6]:
    steps = max(it) * ((n + 199) // 200)
    print(' rows %6d  epochs %s  wall %.3f s  %.1f us/step' % (n, it, dt, 1e6 * dt / steps))
