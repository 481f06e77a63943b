"""Where a run at the reference's default n_batch = 100 spends its host
time: the funnel at n_dim 20, reduced settings (n_live 2000, 4 networks),
under cProfile; exploration and sampling phase timed apart.
    python profiles/tools/small_batch_profile.py [n_dim] > gpurun_out/small_batch_profile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np           # noqa: E402
import torch                 # noqa: E402
from nautilus_amd import Sampler, unit_prior             # noqa: E402
from nautilus_amd.configs import baseline_config         # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
limit = float(sys.argv[2]) if len(sys.argv) > 2 else float('inf')
c = baseline_config('C5-D%d' % d)
s = Sampler(unit_prior, c['likelihood'], n_dim=d, n_live=2000, n_networks=4,
            n_batch=100, vectorized=True, seed=0)
t0 = time.time()
pr = cProfile.Profile()
pr.enable()
s.run(n_eff=0, n_shell=0, discard_exploration=True, timeout=limit)
torch.cuda.synchronize()
pr.disable()
t1 = time.time()
batches = s.n_like / 100
print('exploration: %.1f s, %d bounds, %d calls, %.2f ms per batch, timing %s'
      % (t1 - t0, len(s.bounds), s.n_like, (t1 - t0) / batches * 1e3,
         {k: round(v, 2) for k, v in s.timing.items()}))
for key in ('cumulative', 'tottime'):
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats(key).print_stats(30)
    print(out.getvalue())
n0 = s.n_like
tim0 = dict(s.timing)
pr = cProfile.Profile()
pr.enable()
s.run(n_eff=10000, discard_exploration=True, timeout=limit)
torch.cuda.synchronize()
pr.disable()
t2 = time.time()
batches = max(1.0, (s.n_like - n0) / 100)
print('sampling phase: %.1f s, %d batches, %.2f ms per batch, timing %s' % (
    t2 - t1, batches, (t2 - t1) / batches * 1e3,
    {k: round(v - tim0.get(k, 0.0), 2) for k, v in s.timing.items()}))
print('log Z - analytic = %.4f, N_eff %.0f' % (s.log_z - c['analytic_log_z'],
                                             s.n_eff))
for key in ('cumulative', 'tottime'):
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats(key).print_stats(30)
    print(out.getvalue())
