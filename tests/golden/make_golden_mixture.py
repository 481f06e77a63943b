"""REFERENCE runs of the config-4 problem family (equal-weight mixture of four
isotropic Gaussians on the unit cube, BASELINE config 4 at the dimensions the
reference finishes on the build container's CPU cores): the multi-ellipsoid
Union with several members and several neural bounds per NautilusBound, end
to end, in the reference itself (nautilus 1.0.6 of /root/reference).

Settings, the same on both sides of the comparison
(tests/test_configs_gpu.py::test_mixture_against_reference_runs):
n_live = 2000, n_networks = 4, everything else the reference's defaults
(n_batch = 100, n_eff = 10000, f_live = 0.01, discard_exploration = True).
The means are those of ``nautilus_amd.configs.baseline_config('C4-D<d>')``:
0.25 + 0.5 * numpy.random.default_rng(3).random((4, d)), sigma = 0.02;
analytic log Z = 0.

One process per (n_dim, seed), one BLAS thread each; every job writes
tests/golden/mixture_parts/<tag>.json as it finishes and ``merge`` folds them
into tests/golden/e2e_mixture.json (data only: log Z, N_eff, n_like, bounds,
neural bounds of the last bound, the posterior weight of every mode, wall
seconds).

    nohup python tests/golden/make_golden_mixture.py run 5 &
    python tests/golden/make_golden_mixture.py merge
"""
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')

import numpy as np  # noqa: E402
from scipy.special import logsumexp  # noqa: E402

sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
PARTS = os.path.join(HERE, 'mixture_parts')
SIGMA = 0.02
SETTINGS = dict(n_live=2000, n_networks=4)

# (n_dim, seed), in the order they are started
JOBS = [(10, 0), (10, 1), (10, 2), (20, 0), (20, 1), (10, 3), (20, 2),
        (30, 0), (30, 1), (30, 2)]


def means_of(d):
    return 0.25 + 0.5 * np.random.default_rng(3).random((4, d))


def tag(job):
    return 'D%d_seed%d' % job


def identity(u):
    return u


class Mixture:
    """The likelihood as a picklable object (the run is checkpointed)."""

    def __init__(self, d):
        self.means = means_of(d)
        self.log_norm = (-d * np.log(SIGMA * np.sqrt(2 * np.pi)) -
                         np.log(len(self.means)))

    def __call__(self, u):
        u = np.atleast_2d(u)
        r2 = np.sum((u[:, None, :] - self.means[None])**2, axis=2)
        return logsumexp(-0.5 * r2 / SIGMA**2, axis=1) + self.log_norm


CKPT = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out',
                    'refjobs')


def run_job(job):
    """One reference run, advanced in slices of 15 minutes
    (``run(timeout=...)``: all loop state lives on the sampler, the slices
    are the uninterrupted run) with the sampler pickled between them, so
    that a job that is killed resumes from its last slice."""
    import pickle
    import nautilus
    d, seed = job
    path = os.path.join(PARTS, tag(job) + '.json')
    if os.path.exists(path):
        return path
    means = means_of(d)
    os.makedirs(CKPT, exist_ok=True)
    ckpt = os.path.join(CKPT, 'mixture_' + tag(job) + '.pkl')
    if os.path.exists(ckpt):
        with open(ckpt, 'rb') as f:
            s, spent = pickle.load(f)
    else:
        s = nautilus.Sampler(identity, Mixture(d), n_dim=d, vectorized=True,
                             seed=seed, pool=None, **SETTINGS)
        spent = 0.0
    done = False
    while not done:
        t0 = time.time()
        done = s.run(discard_exploration=True, verbose=False, timeout=900.0)
        spent += time.time() - t0
        with open(ckpt + '.tmp', 'wb') as f:
            pickle.dump((s, spent), f, protocol=4)
        os.replace(ckpt + '.tmp', ckpt)
    pts, log_w, log_l = s.posterior()
    w = np.exp(log_w - np.max(log_w))
    w /= w.sum()
    owner = np.argmin(np.sum((pts[:, None, :] - means[None])**2, axis=2),
                      axis=1)
    share = [float(w[owner == k].sum()) for k in range(len(means))]
    last = s.bounds[-1]
    out = dict(n_dim=d, seed=seed, log_z=float(s.log_z),
               n_eff=float(s.n_eff), n_like=int(s.n_like),
               n_bounds=len(s.bounds), eta=float(s.eta),
               n_neural_last=len(getattr(last, 'neural_bounds', [])),
               n_neural_max=max(len(getattr(b, 'neural_bounds', []))
                                for b in s.bounds),
               mode_share=share, wall_s=spent, **SETTINGS)
    os.makedirs(PARTS, exist_ok=True)
    with open(path + '.tmp', 'w') as f:
        json.dump(out, f, indent=1)
    os.replace(path + '.tmp', path)
    return path


def merge():
    runs = []
    for job in JOBS:
        path = os.path.join(PARTS, tag(job) + '.json')
        if os.path.exists(path):
            with open(path) as f:
                runs.append(json.load(f))
    out = dict(problem='equal-weight mixture of four isotropic Gaussians '
                       '(sigma 0.02, means 0.25 + 0.5 * default_rng(3)'
                       '.random((4, d))) on the unit cube, identity prior; '
                       'reference defaults except n_live 2000 / n_networks 4 '
                       '(n_batch 100, n_eff 10000), pool=None, '
                       'discard_exploration=True, nautilus 1.0.6 of '
                       '/root/reference',
               runs=runs)
    with open(os.path.join(HERE, 'e2e_mixture.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('merged %d of %d runs' % (len(runs), len(JOBS)))


if __name__ == '__main__':
    if sys.argv[1] == 'merge':
        merge()
    elif sys.argv[1] == 'run':
        import multiprocessing as mp
        workers = int(sys.argv[2]) if len(sys.argv) > 2 else 5
        with mp.get_context('fork').Pool(workers, maxtasksperchild=1) as pool:
            for path in pool.imap_unordered(run_job, JOBS, chunksize=1):
                print('done', path, flush=True)
        merge()
    else:
        one = [j for j in JOBS if tag(j) == sys.argv[1]]
        print(run_job(one[0]))
