"""This build's side of the round-6 reference anchors (one JSON line per run
into gpurun_out/r06_anchors.jsonl):

  funnel   reduced settings (n_live 2000, 4 networks, n_batch 100) at D = 50,
           the runs tests/golden/make_golden_funnel.py does in the reference
  mixture  config 4's problem at D = 30 (make_golden_mixture.py)
  prefix   config 5 at D = 100, reduced settings, run(n_like_max=N) ladder of
           make_golden_c5_prefix.py

    python profiles/tools/r06_anchor_runs.py funnel 50 0 1 2
    python profiles/tools/r06_anchor_runs.py mixture 30 0 1
    python profiles/tools/r06_anchor_runs.py prefix 300 0 1   # 300 s each
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'gpurun_out', 'r06_anchors.jsonl')


def emit(row):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, 'a') as f:
        f.write(json.dumps(row) + '\n')
    print(json.dumps({k: v for k, v in row.items()
                      if not isinstance(v, list) or len(v) < 8}), flush=True)


def sampler(name, seed, **kw):
    from nautilus_amd import Sampler, unit_prior
    from nautilus_amd.configs import baseline_config
    c = baseline_config(name)
    s = Sampler(unit_prior, c['likelihood'], n_dim=c['n_dim'],
                n_live=kw.get('n_live', c['n_live']),
                n_networks=kw.get('n_networks', c['n_networks']),
                n_batch=kw.get('n_batch', c['n_batch']), vectorized=True,
                seed=seed)
    return c, s


def moments(s):
    pts, log_w, log_l = s.posterior()
    w = np.exp(log_w - np.max(log_w))
    w /= w.sum()
    mean = pts.T @ w
    var = ((pts - mean)**2).T @ w
    return pts, w, mean, var


def funnel(d, seeds):
    for seed in seeds:
        t0 = time.time()
        c, s = sampler('C5-D%d' % d, seed, n_live=2000, n_networks=4,
                       n_batch=100)
        s.run(discard_exploration=True)
        _, _, mean, var = moments(s)
        emit(dict(kind='funnel', n_dim=d, seed=seed, log_z=float(s.log_z),
                  analytic=c['analytic_log_z'], n_eff=float(s.n_eff),
                  n_like=int(s.n_like), n_bounds=len(s.bounds),
                  mean_x0=float(mean[0]), var_x0=float(var[0]),
                  mean_x1=float(mean[1]), var_x1=float(var[1]),
                  wall_s=time.time() - t0, timing=dict(s.timing)))


def mixture(d, seeds):
    for seed in seeds:
        t0 = time.time()
        c, s = sampler('C4-D%d' % d, seed)
        s.run(discard_exploration=True)
        pts, w, _, _ = moments(s)
        means = c['means']
        owner = np.argmin(((pts[:, None, :] - means[None])**2).sum(-1),
                          axis=1)
        emit(dict(kind='mixture', n_dim=d, seed=seed, log_z=float(s.log_z),
                  n_eff=float(s.n_eff), n_like=int(s.n_like),
                  n_bounds=len(s.bounds),
                  n_neural_max=max(len(b.neural_bounds)
                                   for b in s.bounds[1:]),
                  mode_share=[float(w[owner == k].sum())
                              for k in range(len(means))],
                  wall_s=time.time() - t0, timing=dict(s.timing)))


def prefix(budget_s, seeds, step=20000):
    from nautilus_amd.emulator import NeuralNetworkEmulator
    rows = []
    inner = NeuralNetworkEmulator.train_many.__func__

    def counting(cls, data, *a, **k):
        rows.extend(int(x.shape[0]) for x, _ in data)
        return inner(cls, data, *a, **k)
    NeuralNetworkEmulator.train_many = classmethod(counting)
    for seed in seeds:
        del rows[:]
        c, s = sampler('C5', seed, n_live=2000, n_networks=4, n_batch=100)
        t0 = time.time()
        n_max = 0
        while time.time() - t0 < budget_s and not s.explored:
            n_max += step
            s.run(n_like_max=n_max, discard_exploration=True)
            last = s.bounds[-1]
            emit(dict(kind='prefix', n_dim=100, seed=seed, n_like_max=n_max,
                      n_like=int(s.n_like), n_bounds=len(s.bounds),
                      log_v=[float(b.log_v) for b in s.bounds],
                      shell_n=[int(n) for n in s.shell_n],
                      shell_log_l_min=[float(x) for x in s.shell_log_l_min],
                      f_live=float(s.f_live), log_z=float(s.log_z),
                      log_v_live=float(s.log_v_live),
                      train_rows=list(rows),
                      n_neural_last=len(getattr(last, 'neural_bounds', [])),
                      wall_s=time.time() - t0, timing=dict(s.timing)))


if __name__ == '__main__':
    kind = sys.argv[1]
    arg = int(sys.argv[2])
    seeds = [int(a) for a in sys.argv[3:]]
    dict(funnel=funnel, mixture=mixture, prefix=prefix)[kind](arg, seeds)
