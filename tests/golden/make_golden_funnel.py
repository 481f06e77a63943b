"""REFERENCE runs of the config-5 problem family (Neal funnel on the unit
cube, the n_dim form of /root/reference/tests/test_sampler.py:311-326) at the
dimensions the reference finishes on the build container's 8 CPU cores.

Two settings, the same on both sides of the comparison
(tests/test_configs_gpu.py, profiles/r05/funnel_bias.json):

  reduced  n_live = 2000, n_networks = 4  -- D = 10, 20, 30, 50
  full     n_live = 10000, n_networks = 8 -- D = 10 (config 5's own settings)

everything else the reference's defaults (n_batch = 100, n_eff = 10000,
f_live = 0.01, discard_exploration as given).  One process per (setting, D,
seed), one BLAS thread each; every job writes
tests/golden/funnel_parts/<tag>.json as it finishes and ``merge`` folds them
into tests/golden/e2e_funnel.json (data only: log Z, N_eff, n_like, bounds,
posterior mean / variance of x_0 and of x_1, wall seconds).

    nohup python tests/golden/make_golden_funnel.py run 6 &   # 6 workers
    python tests/golden/make_golden_funnel.py merge
"""
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')

import numpy as np  # noqa: E402

sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
PARTS = os.path.join(HERE, 'funnel_parts')

MU, SIGMA0, K, C = 0.5, 0.1, 20.0, 100.0
LOG_2PI = float(np.log(2.0 * np.pi))

# (setting, n_dim, seed, discard_exploration), in the order they are started
JOBS = ([('reduced', 10, s, True) for s in range(3)] +
        [('reduced', 20, s, True) for s in range(3)] +
        [('reduced', 30, 0, True), ('reduced', 30, 1, True),
         ('reduced', 50, 0, True), ('full', 10, 0, True),
         ('reduced', 10, 3, False), ('reduced', 20, 3, False),
         ('reduced', 30, 2, True), ('reduced', 50, 1, True)] +
        # (more seeds where a run is cheap: sigma(log Z) ~ 0.01 per run)
        [('reduced', 20, s, True) for s in range(4, 10)] +
        [('reduced', 10, s, True) for s in range(4, 10)] +
        [('reduced', 50, 2, True)])
SETTINGS = dict(reduced=dict(n_live=2000, n_networks=4),
                full=dict(n_live=10000, n_networks=8))


def funnel(u):
    """log density of the funnel, vectorised over rows of u."""
    u = np.atleast_2d(u)
    x0 = u[:, 0]
    log_s = K * (x0 - MU) - np.log(C)
    d = u.shape[1]
    z0 = (x0 - MU) / SIGMA0
    zi = (u[:, 1:] - MU) * np.exp(-log_s)[:, None]
    return (-0.5 * z0 * z0 - np.log(SIGMA0) - 0.5 * LOG_2PI -
            0.5 * np.sum(zi * zi, axis=1) - (d - 1) * (log_s + 0.5 * LOG_2PI))


def tag(job):
    setting, d, seed, discard = job
    return '%s_D%d_seed%d%s' % (setting, d, seed, '' if discard else '_keep')


def identity(u):
    return u


CKPT = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out',
                    'refjobs')


def run_job(job):
    """One reference run.  The run advances in slices of 15 minutes
    (``run(timeout=...)``: every piece of loop state lives on the sampler,
    so the slices are the uninterrupted run) and the sampler is pickled
    between them -- a job that is killed resumes from its last slice."""
    import pickle
    import nautilus
    setting, d, seed, discard = job
    path = os.path.join(PARTS, tag(job) + '.json')
    if os.path.exists(path):
        return path
    os.makedirs(CKPT, exist_ok=True)
    ckpt = os.path.join(CKPT, 'funnel_' + tag(job) + '.pkl')
    if os.path.exists(ckpt):
        with open(ckpt, 'rb') as f:
            s, spent = pickle.load(f)
    else:
        s = nautilus.Sampler(identity, funnel, n_dim=d, vectorized=True,
                             seed=seed, pool=None, **SETTINGS[setting])
        spent = 0.0
    done = False
    while not done:
        t0 = time.time()
        done = s.run(discard_exploration=discard, verbose=False,
                     timeout=900.0)
        spent += time.time() - t0
        with open(ckpt + '.tmp', 'wb') as f:
            pickle.dump((s, spent), f, protocol=4)
        os.replace(ckpt + '.tmp', ckpt)
    pts, log_w, log_l = s.posterior()
    w = np.exp(log_w - np.max(log_w))
    w /= w.sum()
    mean = pts.T @ w
    var = ((pts - mean)**2).T @ w
    out = dict(setting=setting, n_dim=d, seed=seed, discard_exploration=discard,
               log_z=float(s.log_z), n_eff=float(s.n_eff),
               n_like=int(s.n_like), n_bounds=len(s.bounds),
               eta=float(s.eta), wall_s=spent,
               mean_x0=float(mean[0]), var_x0=float(var[0]),
               mean_x1=float(mean[1]), var_x1=float(var[1]),
               mean_log_l=float(log_l @ w), **SETTINGS[setting])
    os.makedirs(PARTS, exist_ok=True)
    with open(path + '.tmp', 'w') as f:
        json.dump(out, f, indent=1)
    os.replace(path + '.tmp', path)
    return path


def snapshot(job):
    """State of a run in progress, read from its checkpoint: what
    ``run(n_like_max=<its n_like>)`` of the reference leaves behind (the
    slices end between batches, so the state is the one the reference has
    when it stops at that number of calls).  Written to
    tests/golden/funnel_parts/<tag>_snapshot.json -- a like-for-like anchor
    for runs that do not finish inside a round."""
    import pickle
    ckpt = os.path.join(CKPT, 'funnel_' + tag(job) + '.pkl')
    with open(ckpt, 'rb') as f:
        s, spent = pickle.load(f)
    out = dict(setting=job[0], n_dim=job[1], seed=job[2],
               n_like=int(s.n_like), n_bounds=len(s.bounds),
               explored=bool(s.explored),
               log_v=[float(b.log_v) for b in s.bounds],
               shell_n=[int(n) for n in s.shell_n],
               shell_log_l_min=[float(x) for x in s.shell_log_l_min],
               f_live=None if s.explored else float(s.f_live),
               log_z=float(s.log_z), cpu_s=float(spent),
               **SETTINGS[job[0]])
    path = os.path.join(PARTS, tag(job) + '_snapshot.json')
    with open(path, 'w') as f:
        json.dump(out, f)
    return path


def merge():
    runs = []
    for job in JOBS:
        path = os.path.join(PARTS, tag(job) + '.json')
        if os.path.exists(path):
            with open(path) as f:
                runs.append(json.load(f))
    out = dict(problem='Neal funnel on the unit cube: x_0 ~ N(0.5, 0.1^2), '
                       'x_i ~ N(0.5, (exp(20 (x_0 - 0.5)) / 100)^2), identity '
                       'prior; reference defaults except n_live / n_networks '
                       'as listed per run (n_batch 100, n_eff 10000), '
                       'pool=None, nautilus 1.0.6 of /root/reference',
               runs=runs)
    with open(os.path.join(HERE, 'e2e_funnel.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('merged %d of %d runs' % (len(runs), len(JOBS)))


if __name__ == '__main__':
    if sys.argv[1] == 'merge':
        merge()
    elif sys.argv[1] == 'snapshot':
        one = [j for j in JOBS if tag(j) == sys.argv[2]]
        print(snapshot(one[0]))
    elif sys.argv[1] == 'run':
        import multiprocessing as mp
        workers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
        with mp.get_context('fork').Pool(workers, maxtasksperchild=1) as pool:
            for path in pool.imap_unordered(run_job, JOBS, chunksize=1):
                print('done', path, flush=True)
        merge()
    else:
        one = [j for j in JOBS if tag(j) == sys.argv[1]]
        print(run_job(one[0]))
