"""A like-for-like PREFIX of BASELINE configuration 5 in the REFERENCE: the
100-dimensional Neal funnel (the n_dim form of
/root/reference/tests/test_sampler.py:311-326) at the reduced settings of
make_golden_funnel.py (n_live 2000, 4 networks, n_batch 100), driven through
``run(n_like_max=N)`` (/root/reference/nautilus/sampler.py:373-374, 433) for
a ladder of N.  ``n_like_max`` counts calls across runs and every piece of
loop state lives on the sampler, so ``run(n_like_max=N1); run(n_like_max=N2)``
is the run ``run(n_like_max=N2)`` -- each rung of the ladder is therefore the
state the reference has when it stops at that N, and the build can be run to
the same N and compared (tests/test_configs_gpu.py::
test_C5_prefix_against_the_reference).

The run does not end inside any budget (about 830 bounds); the ladder is
written after every rung, so whatever rung was reached when the job is
stopped is a usable anchor.  Recorded per rung (data only): likelihood calls,
number of bounds, log_v of every bound, shell occupation, f_live, log Z so
far, log_l_min of the last bound, rows handed to the emulator's ``train`` for
every bound built, members of the outer union / neural bounds of the last
bound, CPU seconds.

    nohup python tests/golden/make_golden_c5_prefix.py 0 &    # seed 0
    python tests/golden/make_golden_c5_prefix.py merge
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
PARTS = os.path.join(HERE, 'c5_prefix_parts')

N_DIM = 100
SETTINGS = dict(n_live=2000, n_networks=4)
STEP = 20000                      # likelihood calls between rungs
MU, SIGMA0, K, C = 0.5, 0.1, 20.0, 100.0
LOG_2PI = float(np.log(2.0 * np.pi))


def funnel(u):
    """log density of the funnel, vectorised over rows of u (the one of
    make_golden_funnel.py)."""
    u = np.atleast_2d(u)
    x0 = u[:, 0]
    log_s = K * (x0 - MU) - np.log(C)
    d = u.shape[1]
    z0 = (x0 - MU) / SIGMA0
    zi = (u[:, 1:] - MU) * np.exp(-log_s)[:, None]
    return (-0.5 * z0 * z0 - np.log(SIGMA0) - 0.5 * LOG_2PI -
            0.5 * np.sum(zi * zi, axis=1) - (d - 1) * (log_s + 0.5 * LOG_2PI))


def identity(u):
    return u


CKPT = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out',
                    'refjobs')


def run(seed, cpu_budget_s):
    """Climb the ladder; the sampler and the ladder so far are pickled after
    every rung, so that a job that is killed resumes at its last rung (the
    committed part file is only ever replaced by a longer ladder)."""
    import pickle
    import nautilus
    from nautilus.bounds import neural as ref_neural

    train_rows = []
    inner = ref_neural.NeuralNetworkEmulator.train.__func__

    def counting_train(cls, x, y, **kwargs):
        train_rows.append(int(len(x)))
        return inner(cls, x, y, **kwargs)

    path = os.path.join(PARTS, 'D%d_seed%d.json' % (N_DIM, seed))
    os.makedirs(PARTS, exist_ok=True)
    os.makedirs(CKPT, exist_ok=True)
    ckpt = os.path.join(CKPT, 'c5_prefix_seed%d.pkl' % seed)
    have = 0
    if os.path.exists(path):
        with open(path) as f:
            have = len(json.load(f)['rungs'])
    if os.path.exists(ckpt):
        with open(ckpt, 'rb') as f:
            s, rungs, rows_so_far, spent = pickle.load(f)
        train_rows.extend(rows_so_far)
    else:
        s = nautilus.Sampler(identity, funnel, n_dim=N_DIM, vectorized=True,
                             seed=seed, pool=None, **SETTINGS)
        rungs, spent = [], 0.0
    ref_neural.NeuralNetworkEmulator.train = classmethod(counting_train)
    t0 = time.process_time() - spent
    n_max = rungs[-1]['n_like_max'] if rungs else 0
    if s.n_like > n_max:             # resumed inside a rung
        n_max = (s.n_like - 1) // STEP * STEP
    while time.process_time() - t0 < cpu_budget_s and not s.explored:
        n_max += STEP
        # (in slices of 15 minutes with a checkpoint after each: a rung takes
        # more than an hour from the eighth on, and a killed job should not
        # lose it -- the slices are the uninterrupted run, see the docstring)
        while s.n_like < n_max and not s.explored:
            s.run(n_like_max=n_max, discard_exploration=True, verbose=False,
                  timeout=900.0)
            ref_neural.NeuralNetworkEmulator.train = classmethod(inner)
            with open(ckpt + '.tmp', 'wb') as f:
                pickle.dump((s, rungs, list(train_rows),
                             time.process_time() - t0), f, protocol=4)
            os.replace(ckpt + '.tmp', ckpt)
            ref_neural.NeuralNetworkEmulator.train = \
                classmethod(counting_train)
        last = s.bounds[-1]
        rungs.append(dict(
            n_like_max=n_max, n_like=int(s.n_like), n_bounds=len(s.bounds),
            explored=bool(s.explored),
            log_v=[float(b.log_v) for b in s.bounds],
            shell_n=[int(n) for n in s.shell_n],
            shell_log_l_min=[float(x) for x in s.shell_log_l_min],
            f_live=float(s.f_live), log_z=float(s.log_z),
            n_eff=float(s.n_eff), log_v_live=float(s.log_v_live),
            train_rows=list(train_rows),
            n_neural_last=len(getattr(last, 'neural_bounds', [])),
            n_outer_last=len(getattr(getattr(last, 'outer_bound', None),
                                     'bounds', [])),
            cpu_s=time.process_time() - t0))
        ref_neural.NeuralNetworkEmulator.train = classmethod(inner)
        with open(ckpt + '.tmp', 'wb') as f:
            pickle.dump((s, rungs, list(train_rows),
                         time.process_time() - t0), f, protocol=4)
        os.replace(ckpt + '.tmp', ckpt)
        ref_neural.NeuralNetworkEmulator.train = classmethod(counting_train)
        if len(rungs) > have:
            out = dict(n_dim=N_DIM, seed=seed, step=STEP, rungs=rungs,
                       **SETTINGS)
            with open(path + '.tmp', 'w') as f:
                json.dump(out, f)
            os.replace(path + '.tmp', path)
    return path


def merge():
    runs = []
    for name in sorted(os.listdir(PARTS)):
        if name.endswith('.json'):
            with open(os.path.join(PARTS, name)) as f:
                runs.append(json.load(f))
    out = dict(problem='100-D Neal funnel on the unit cube (x_0 ~ N(0.5, '
                       '0.1^2), x_i ~ N(0.5, (exp(20 (x_0 - 0.5)) / 100)^2)),'
                       ' identity prior; n_live 2000, 4 networks, n_batch '
                       '100, pool=None; the state of the reference '
                       '(nautilus 1.0.6 of /root/reference) every 20000 '
                       'likelihood calls of run(n_like_max=...)',
               runs=runs)
    with open(os.path.join(HERE, 'e2e_C5_prefix.json'), 'w') as f:
        json.dump(out, f)
    print('merged %d runs, rungs: %s' % (
        len(runs), [len(r['rungs']) for r in runs]))


if __name__ == '__main__':
    if sys.argv[1] == 'merge':
        merge()
    else:
        budget = float(sys.argv[2]) if len(sys.argv) > 2 else 6 * 3600.0
        print(run(int(sys.argv[1]), budget))
