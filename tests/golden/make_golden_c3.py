"""One full REFERENCE run of BASELINE config 3 (30-D Rosenbrock, n_live 3000,
n_networks 4, pool of 4 processes) -- hours on the build container's CPUs.
Writes tests/golden/e2e_C3.json (data only).

    OMP_NUM_THREADS=1 nohup python tests/golden/make_golden_c3.py &
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))

import nautilus  # noqa: E402


def rosenbrock(u):
    x = 10.0 * np.atleast_2d(u) - 5.0
    return -np.sum(100.0 * (x[:, 1:] - x[:, :-1]**2)**2 +
                   (1.0 - x[:, :-1])**2, axis=1)


def identity(u):
    return u


def run_single(seed):
    """A further reference run on ONE core (pool=None), advanced in slices of
    15 minutes with the sampler pickled between them (it resumes after a
    kill); appended to tests/golden/e2e_C3.json when it ends."""
    import pickle
    ckpt_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)),
                            'gpurun_out', 'refjobs')
    os.makedirs(ckpt_dir, exist_ok=True)
    ckpt = os.path.join(ckpt_dir, 'c3_seed%d.pkl' % seed)
    if os.path.exists(ckpt):
        with open(ckpt, 'rb') as f:
            s, spent = pickle.load(f)
    else:
        s = nautilus.Sampler(identity, rosenbrock, n_dim=30, n_live=3000,
                             vectorized=True, seed=seed, pool=None)
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
    w = np.exp(log_w)
    run = dict(seed=seed, discard_exploration=True, pool=None,
               log_z=float(s.log_z), n_eff=float(s.n_eff),
               n_like=int(s.n_like), n_bounds=len(s.bounds),
               eta=float(s.eta), wall_s=spent,
               mean=np.average(pts, weights=w, axis=0).tolist())
    path = os.path.join(HERE, 'e2e_C3.json')
    with open(path) as f:
        out = json.load(f)
    out['runs'] = [r for r in out['runs'] if r.get('seed') != seed] + [run]
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('done', run['log_z'], run['wall_s'])


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == 'single':
        run_single(int(sys.argv[1]))
        sys.exit(0)
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    t0 = time.time()
    s = nautilus.Sampler(lambda u: u, rosenbrock, n_dim=30, n_live=3000,
                         vectorized=True, seed=seed, pool=4)
    s.run(discard_exploration=True, verbose=True)
    pts, log_w, log_l = s.posterior()
    w = np.exp(log_w)
    out = dict(problem='30-D Rosenbrock on x = 10 u - 5, identity prior, '
                       'n_live=3000, n_networks=4, n_eff=10000, pool=4',
               runs=[dict(seed=seed, discard_exploration=True,
                          log_z=float(s.log_z), n_eff=float(s.n_eff),
                          n_like=int(s.n_like), n_bounds=len(s.bounds),
                          eta=float(s.eta), wall_s=time.time() - t0,
                          mean=np.average(pts, weights=w, axis=0).tolist())])
    with open(os.path.join(HERE, 'e2e_C3.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('done', out['runs'][0]['log_z'], out['runs'][0]['wall_s'])
