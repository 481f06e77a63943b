"""The five BASELINE.json configurations at their REAL sizes (n_dim, n_live,
n_networks, device likelihood) on one GPU.

C1, C2 and C3 run to completion and are held against the analytic evidence
and the reference's own runs of the same problem (tests/golden/e2e_C1.json,
e2e_C2.json, e2e_C3.json, written by make_golden.py / make_golden_c3.py).  C4
runs its whole exploration and a sampling phase to a reduced N_eff against the
analytic evidence.  C5 does not finish at its full dimension anywhere (~830
bounds with training sets beyond 10^6 rows, docs/history/round4.md; the
reference's FAQ stops at ~60 dimensions), so it runs for
a bounded wall time and the test asserts what must hold at any point of a run
-- every bound built on the device, volumes shrinking, evidence finite and
consistent with its shells; with ``NB_FULL_CONFIGS=1`` C4 runs to the
reference's default N_eff = 10 000 (committed full runs: the newest
profiles/r*/configs.json).  Configuration 5's problem family (the funnel) is
held against the REFERENCE'S OWN RUNS of the same problem at the settings the
reference finishes on CPUs (tests/golden/e2e_funnel.json, written by
make_golden_funnel.py) and against quadrature values at config 5's own
settings at 10 and 30 dimensions."""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

FULL = os.environ.get('NB_FULL_CONFIGS', '') not in ('', '0')


@pytest.fixture(autouse=True)
def gpu_only():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _run(name, seed=0, timeout=np.inf, n_batch=None, n_eff=10000,
         discard_exploration=True, n_live=None, n_networks=None):
    import torch
    from nautilus_amd import Sampler, unit_prior
    from nautilus_amd.configs import baseline_config
    c = baseline_config(name)
    # (the construction has no host path at any dimension: the mixture fit
    # raises geometry.DegenerateMixture where every device restart fails)
    s = Sampler(unit_prior, c['likelihood'], n_dim=c['n_dim'],
                n_live=n_live or c['n_live'],
                n_networks=n_networks or c['n_networks'],
                n_batch=n_batch or c['n_batch'], vectorized=True,
                seed=seed)
    done = s.run(n_eff=n_eff, discard_exploration=discard_exploration,
                 timeout=timeout)
    torch.cuda.synchronize()
    return c, s, done


def _invariants(c, s):
    """What must hold at any point of a run."""
    assert len(s.bounds) >= 2
    vols = np.array([b.log_v for b in s.bounds])
    # sampler.py:1035-1038 accepts a bound whose volume estimate is smaller
    # than its predecessor's AT THAT MOMENT; both are Monte-Carlo estimates
    # that keep moving as their bounds are sampled (union.py:329-343,
    # nautilus.py:246-261), so two neighbours of nearly equal volume may
    # change places later -- by the scatter of the estimates, not more
    steps = np.diff(vols)
    assert np.all(steps < 0.1), steps[steps >= 0]
    assert np.mean(steps < 0) > 0.97
    assert np.isfinite(s.log_z)
    used = s.shell_n > 0
    assert np.all(s.shell_n_sample[used] >= s.shell_n[used])
    assert np.all(np.isfinite(s.shell_log_v[used]))
    assert int(np.sum(s.shell_n)) <= s.n_like
    for b in s.bounds[1:]:
        assert b.n_dim == c['n_dim']
        for nb in b.neural_bounds:
            assert len(nb.emulator.neural_networks) == s.n_networks
    # the newest bound encloses most of the current live points (not all: the
    # emulator's threshold sits at the predicted score of the lowest live
    # point, neural.py:97, so points near the likelihood threshold may fall
    # outside -- 80-90 % early in the Rosenbrock run).  Only once the bound has
    # been sampled for a while: right after its creation the live set is
    # still the one it was fitted to, and on the 100-D funnel a third of that
    # lies inside (0.335 with 2128 points in the newest shell, 0.80 with 7843
    # -- the reference at that point of the same run: 0.80 with 7770,
    # profiles/tools/prefix_live_probe.py against the reference's checkpoint)
    ll = np.concatenate(s.log_l)
    if not s.explored and len(ll) > s.n_live and \
            s.shell_n[-1] >= 3 * s.n_live:
        pts = np.concatenate(s.points)
        live = pts[np.argsort(ll)[-s.n_live:]]
        assert np.mean(s.bounds[-1].contains(live)) > 0.6


def _reference_band(name):
    with open(os.path.join(GOLDEN, 'e2e_%s.json' % name)) as f:
        data = json.load(f)
    runs = [r for r in data['runs'] if r['discard_exploration']]
    return data, runs


@pytest.mark.parametrize('name,seeds', [('C1', (0, 1, 2)), ('C2', (0, 1))])
def test_gaussian_configs_against_reference_runs(name, seeds):
    """C1 (3-D Gaussian, n_live 1000) and C2 (20-D correlated Gaussian,
    n_live 2000), full runs with the reference's defaults (n_eff 10000,
    n_networks 4): evidence within the north star's 0.01 of the analytic
    value on average and inside the band of the reference's own seed sweep
    for every seed; likelihood calls, number of bounds and posterior moments
    in the reference's range."""
    data, ref = _reference_band(name)
    analytic = data['analytic_log_z']
    ref_z = np.array([r['log_z'] for r in ref])
    ref_like = np.array([r['n_like'] for r in ref])
    ref_bounds = np.array([r['n_bounds'] for r in ref])
    sigma = max(np.std(ref_z - analytic), 1.0 / np.sqrt(10000))
    zs = []
    for seed in seeds:
        c, s, done = _run(name, seed=seed)
        assert done
        _invariants(c, s)
        assert s.n_eff >= 10000
        assert abs(s.log_z - analytic) < 4 * sigma + 0.005
        assert ref_z.min() - 4 * sigma < s.log_z < ref_z.max() + 4 * sigma
        assert 0.5 * ref_like.min() < s.n_like < 2.0 * ref_like.max()
        assert ref_bounds.min() - 3 <= len(s.bounds) <= ref_bounds.max() + 3
        pts, log_w, _ = s.posterior()
        w = np.exp(log_w)
        mean = np.average(pts, weights=w, axis=0)
        ref_mean = np.array([r['mean'] for r in ref])
        spread = np.maximum(ref_mean.std(axis=0), 1e-3)
        assert np.all(np.abs(mean - ref_mean.mean(axis=0)) < 6 * spread)
        zs.append(s.log_z)
    assert abs(np.mean(zs) - analytic) < 0.01 + 2 * sigma / np.sqrt(len(zs))


def _committed(name):
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*',
                                              'configs.json')))[::-1]:
        if os.path.exists(path):
            with open(path) as f:
                row = json.load(f).get(name)
            if row is not None and row.get('finished'):
                return row
    return None


def test_C3_rosenbrock_full_run_against_the_reference():
    """C3 (30-D Rosenbrock, n_live 3000, NeuralBound active) to completion
    with the reference's defaults, held against the reference's OWN full run
    of the same problem (tests/golden/e2e_C3.json: 3.2 h on four cores;
    make_golden_c3.py).  Both samplers miss the exact evidence (transfer
    quadrature, helpers.rosenbrock_log_z_exact) by ~0.7: what an emulator
    cuts off is lost to all later shells."""
    c, s, done = _run('C3')
    _invariants(c, s)
    assert done and s.explored and s.n_eff >= 10000
    _, ref = _reference_band('C3')
    assert abs(s.log_z - ref[0]['log_z']) < 0.3
    assert 0.8 * ref[0]['n_like'] < s.n_like < 1.25 * ref[0]['n_like']
    assert abs(len(s.bounds) - ref[0]['n_bounds']) <= 10
    assert -1.5 < s.log_z - c['analytic_log_z'] < 0.1
    ours = _committed('C3')
    if ours is not None:
        assert abs(s.log_z - ours['log_z']) < 0.2


def test_C4_mixture_explored_and_evidence():
    """C4 (50-D four-mode mixture, n_live 5000, multi-ellipsoid Union bound):
    the whole exploration plus a sampling phase to N_eff = 2000 (10 000 with
    NB_FULL_CONFIGS=1; the exploration is ~95 % of either).  The evidence is
    analytic (0), the posterior mass must split evenly over the four modes,
    and the decomposition must take the modes apart: bounds with three or
    four neural bounds exist (the non-overlapping split of the closest pair
    of modes depends on the realisation: four in two of the three runs of
    round 3, three in the third), the sampling envelope -- which may overlap
    and only splits while it is split_threshold times too large -- has two
    members or more."""
    c, s, done = _run('C4', n_eff=10000 if FULL else 2000)
    _invariants(c, s)
    assert done and s.explored
    assert max(len(b.neural_bounds) for b in s.bounds[1:]) >= 3
    assert max(b.n_ell for b in s.bounds[1:]) >= 2
    # The target is nominal: with batches of 16384 the first batch of every
    # shell (sampler.py:482-486) already brings N_eff to ~70 000 (committed
    # run: 71 081, log Z +0.004), sigma(log Z) ~ 1 / sqrt(N_eff) = 0.004 and
    # sigma of a mode's weight = sqrt(0.25 * 0.75 / N_eff) = 0.002.
    assert s.n_eff > 20000
    assert abs(s.log_z - c['analytic_log_z']) < 0.03
    # posterior mass splits evenly over the four modes
    pts, log_w, _ = s.posterior()
    w = np.exp(log_w)
    means = c['means']
    mode = np.argmin(((pts[:, None, :] - means[None]) ** 2).sum(-1), axis=1)
    share = np.array([w[mode == k].sum() for k in range(len(means))])
    assert np.all(np.abs(share / share.sum() - 0.25) < 0.025)


def _mixture_reference(n_dim):
    path = os.path.join(GOLDEN, 'e2e_mixture.json')
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return [r for r in json.load(f)['runs'] if r['n_dim'] == n_dim]


@pytest.mark.parametrize('n_dim', [10, 20, 30])
def test_mixture_against_reference_runs(n_dim):
    """Configuration 4's problem (four-mode mixture: a Union with several
    members, several neural bounds per NautilusBound) at the dimensions the
    REFERENCE finishes, with the settings of its runs in
    tests/golden/e2e_mixture.json (make_golden_mixture.py: n_live 2000, 4
    networks, n_batch 100, n_eff 10000, exploration discarded): evidence in
    the reference's band and within 0.05 of the analytic value 0, likelihood
    calls, number of bounds, the largest number of neural bounds of a bound and
    the posterior weight of every mode as the reference has them."""
    ref = _mixture_reference(n_dim)
    if len(ref) < 2:
        pytest.skip('fewer than two reference runs at n_dim %d in '
                    'tests/golden/e2e_mixture.json' % n_dim)
    ref_z = np.array([r['log_z'] for r in ref])
    ref_like = np.mean([r['n_like'] for r in ref])
    ref_bounds = np.mean([r['n_bounds'] for r in ref])
    ref_neural = max(r['n_neural_max'] for r in ref)
    c, s, done = _run('C4-D%d' % n_dim, seed=0)
    assert done and s.n_eff >= 10000
    _invariants(c, s)
    sigma = 0.01                     # of log Z of ONE run at N_eff 10 000
    assert abs(s.log_z - ref_z.mean()) < 4 * sigma * np.sqrt(1 + 1 / len(ref))
    assert abs(s.log_z) < 0.05
    assert abs(s.n_like / ref_like - 1) < 0.15
    assert abs(len(s.bounds) - ref_bounds) <= 0.15 * ref_bounds + 2
    ours_neural = max(len(b.neural_bounds) for b in s.bounds[1:])
    assert ours_neural >= 2 and abs(ours_neural - ref_neural) <= 1
    pts, log_w, _ = s.posterior()
    w = np.exp(log_w)
    means = c['means']
    mode = np.argmin(((pts[:, None, :] - means[None]) ** 2).sum(-1), axis=1)
    share = np.array([w[mode == k].sum() for k in range(len(means))])
    ref_share = np.mean([r['mode_share'] for r in ref], axis=0)
    # sigma of a mode's weight at N_eff 10 000: sqrt(0.25 * 0.75 / 1e4)
    assert np.all(np.abs(share / share.sum() - ref_share) < 0.03)


def test_C5_funnel_real_size():
    """C5 (100-D funnel, n_live 10000, 8 networks): the n_dim > 64 kernels and
    the device MVEE / mixture fit at 100 dimensions -- 15 s of the run and
    the invariants of a run in progress.  The run itself does not end inside
    any budget this project has (docs/history/round4.md: the exploration front
    has to walk down the funnel to x_0 ~ 0.27, ~830 bounds at the measured
    8.3 bounds per dimension, with training sets beyond 10^6 rows from bound
    50 on; a GPU lease lasts one hour and a checkpoint of ~10 GB cannot
    travel between leases).  What CAN be verified end to end is verified in
    ``test_C5_family_finishes`` and ``test_funnel_*_against_reference_runs``:
    the same problem at the dimensions whose runs finish, and like for like
    with the reference at this dimension in
    ``test_C5_prefix_against_the_reference``."""
    c, s, done = _run('C5', timeout=np.inf if FULL else 15.0)
    _invariants(c, s)
    assert len(s.bounds) >= 2
    if FULL:
        assert done and s.n_eff >= 10000
        assert abs(s.log_z - c['analytic_log_z']) < 0.1


def _prefix_reference():
    path = os.path.join(GOLDEN, 'e2e_C5_prefix.json')
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return json.load(f)['runs']


def test_C5_prefix_against_the_reference():
    """Configuration 5 at its REAL dimension, like for like with the
    reference: the 100-D funnel at the reduced settings (n_live 2000, 4
    networks, n_batch 100) stopped by ``run(n_like_max=N)``
    (/root/reference/nautilus/sampler.py:373-374, 433) -- the reference's
    state at that N is in tests/golden/e2e_C5_prefix.json
    (make_golden_c5_prefix.py: one rung per 20 000 likelihood calls, two
    seeds, as far as the CPU budget of the round carried them).  Held at the
    last rung both reference runs reached: number of bounds, log volume of
    the newest bound, likelihood calls per bound, rows of the newest
    emulator's training set, shell occupation and the live-set volume.  (The
    evidence so far is the sum of a few shells with f_live ~ 1: it moves by
    nats from rung to rung in the reference as well and is only required to
    be finite.)"""
    import torch
    from nautilus_amd.emulator import NeuralNetworkEmulator
    ref = _prefix_reference()
    assert len(ref) >= 2, 'tests/golden/e2e_C5_prefix.json: two runs needed'
    n_rungs = min(len(r['rungs']) for r in ref)
    assert n_rungs >= 3
    rungs = [r['rungs'][n_rungs - 1] for r in ref]
    n_max = rungs[0]['n_like_max']
    assert all(r['n_like_max'] == n_max for r in rungs)
    rows = []
    inner = NeuralNetworkEmulator.train_many.__func__

    def counting(cls, data, *a, **k):
        rows.extend(int(x.shape[0]) for x, _ in data)
        return inner(cls, data, *a, **k)
    NeuralNetworkEmulator.train_many = classmethod(counting)
    try:
        from nautilus_amd import Sampler, unit_prior
        from nautilus_amd.configs import baseline_config
        c = baseline_config('C5')
        s = Sampler(unit_prior, c['likelihood'], n_dim=100, n_live=2000,
                    n_networks=4, n_batch=100, vectorized=True, seed=0)
        done = s.run(n_like_max=n_max, discard_exploration=True)
        torch.cuda.synchronize()
    finally:
        NeuralNetworkEmulator.train_many = classmethod(inner)
    assert not done and not s.explored
    _invariants(c, s)
    # stopped where the reference stops: at the first batch boundary at or
    # past N
    assert s.n_like == rungs[0]['n_like'] == n_max
    ref_bounds = np.mean([r['n_bounds'] for r in rungs])
    assert abs(len(s.bounds) - ref_bounds) <= 0.1 * ref_bounds + 1
    ref_log_v = np.mean([r['log_v'][-1] for r in rungs])
    assert abs(s.bounds[-1].log_v - ref_log_v) < 1.0
    # ... and bound by bound: the volumes shrink at the reference's rate
    k = int(min(len(s.bounds), min(r['n_bounds'] for r in rungs)))
    ours = np.array([b.log_v for b in s.bounds[:k]])
    theirs = np.mean([r['log_v'][:k] for r in rungs], axis=0)
    assert np.max(np.abs(ours - theirs)) < 1.0
    ref_rows = np.mean([r['train_rows'][-1] for r in rungs])
    assert abs(rows[-1] / ref_rows - 1) < 0.15
    ref_built = np.mean([len(r['train_rows']) for r in rungs])
    assert abs(len(rows) - ref_built) <= 0.1 * ref_built + 1
    # shell occupation: every finished shell holds what the reference's does
    ref_shell = np.mean([r['shell_n'][:k - 1] for r in rungs], axis=0)
    assert np.all(np.abs(s.shell_n[:k - 1] / ref_shell - 1) < 0.25)
    ref_live = np.mean([r['log_v_live'] for r in rungs])
    assert abs(s.log_v_live - ref_live) < 1.0
    assert np.isfinite(s.log_z)


def _x0_moments(s):
    pts, log_w, _ = s.posterior()
    w = np.exp(log_w - log_w.max())
    w /= w.sum()
    mean = pts.T @ w
    return mean, float(((pts[:, 0] - mean[0])**2) @ w)


def _funnel_reference(n_dim, discard):
    with open(os.path.join(GOLDEN, 'e2e_funnel.json')) as f:
        runs = json.load(f)['runs']
    return [r for r in runs if r['setting'] == 'reduced' and
            r['n_dim'] == n_dim and r['discard_exploration'] == discard]


@pytest.mark.parametrize('n_dim,seeds', [(10, (0, 1) if FULL else (0,)),
                                         (20, (0,))])
def test_funnel_against_reference_runs(n_dim, seeds):
    """Configuration 5's problem at 10 / 20 dimensions with the settings of
    the reference runs in tests/golden/e2e_funnel.json (n_live 2000, 4
    networks, n_batch 100, n_eff 10000, exploration discarded; the reference
    needs 12 / 42 minutes per run on a CPU core): the evidence in the
    reference's band -- sigma(log Z) of ONE run of either sampler is ~0.01 at
    this N_eff --, likelihood calls, number of bounds and the posterior of
    x_0 as the reference has them.  Both samplers put E[x_0] ABOVE the
    quadrature value (0.4895 / 0.4879) by 0.002-0.004 and Var[x_0] below it:
    the bounds lose mass at the narrow end of the funnel, in the reference
    exactly as here (profiles/r05/funnel_bias.json)."""
    from nautilus_amd.configs import funnel_log_z, funnel_moments
    ref = _funnel_reference(n_dim, True)
    assert len(ref) >= 3
    ref_z = np.array([r['log_z'] for r in ref])
    ref_like = np.mean([r['n_like'] for r in ref])
    ref_bounds = np.mean([r['n_bounds'] for r in ref])
    ref_mean = np.mean([r['mean_x0'] for r in ref])
    ref_var = np.mean([r['var_x0'] for r in ref])
    sigma = 0.01
    zs = []
    for seed in seeds:
        c, s, done = _run('C5-D%d' % n_dim, seed=seed, n_batch=100,
                                      n_live=2000, n_networks=4)
        assert done and s.n_eff >= 10000
        _invariants(c, s)
        assert abs(s.log_z - ref_z.mean()) < 4 * sigma * \
            np.sqrt(1 + 1 / len(ref))
        # the reference's own assertion (tests/test_sampler.py:326)
        assert abs(s.log_z - funnel_log_z(n_dim)) < 0.1
        assert abs(s.n_like / ref_like - 1) < 0.06
        assert abs(len(s.bounds) - ref_bounds) <= 8
        mean, var = _x0_moments(s)
        se = np.sqrt(ref_var / 10000)
        assert abs(mean[0] - ref_mean) < 4.5 * se * np.sqrt(1 + 1 / len(ref))
        assert abs(var / ref_var - 1) < 0.12
        assert np.all(np.abs(mean[1:] - 0.5) < 0.01)
        # ... and neither sits on the quadrature value
        assert abs(mean[0] - funnel_moments(n_dim)[0]) < 0.012
        zs.append(s.log_z)
    assert abs(np.mean(zs) - ref_z.mean()) < 3.5 * sigma * \
        np.sqrt(1 / len(zs) + 1 / len(ref))


def test_funnel_exploration_kept_shares_the_reference_bias():
    """``discard_exploration=False`` (the reference's default, and what its
    funnel test runs with): the estimate that keeps the exploration points
    is biased LOW on the funnel, in the reference as here -- 10 dimensions,
    n_live 2000: log Z - analytic = -0.044 in the reference's run
    (e2e_funnel.json), -0.043 / -0.046 / -0.047 here (profiles/r05/
    funnel_b.jsonl); at 20 dimensions -0.118 against -0.114, beyond the
    reference's own tolerance of 0.1.  The bias of configuration 5's runs
    in round 4 (-0.054 at 30, -0.179 at 50 dimensions with n_live 10000) is
    this."""
    from nautilus_amd.configs import funnel_log_z
    ref = _funnel_reference(10, False)
    assert len(ref) >= 1
    c, s, done = _run('C5-D10', seed=3, n_batch=100, n_live=2000,
                                  n_networks=4, discard_exploration=False)
    assert done
    ref_z = np.mean([r['log_z'] for r in ref])
    assert abs(s.log_z - ref_z) < 0.02
    assert s.log_z - funnel_log_z(10) < -0.02          # the shared bias
    assert abs(s.n_like / np.mean([r['n_like'] for r in ref]) - 1) < 0.06


def _funnel_snapshots(n_dim):
    import glob
    out = []
    for path in sorted(glob.glob(os.path.join(
            GOLDEN, 'funnel_parts', 'reduced_D%d_seed*_snapshot.json' % n_dim))):
        with open(path) as f:
            out.append(json.load(f))
    return out


def test_funnel_D50_prefix_against_the_reference():
    """Stand-in for ``test_funnel_D50_against_reference_runs`` while the
    reference's runs at 50 dimensions are still on their way (7-8 CPU-hours
    each next to the other jobs of a round): the reference's state at the end
    of its last 15-minute slice (``make_golden_funnel.py snapshot``: the
    state ``run(n_like_max=<its n_like>)`` leaves behind) against this build
    stopped at the same number of likelihood calls -- number of bounds, log
    volume bound by bound, occupation of the finished shells.  Skipped once
    two finished runs are in tests/golden/e2e_funnel.json."""
    if len(_funnel_reference(50, True)) >= 2:
        pytest.skip('finished reference runs exist: '
                    'test_funnel_D50_against_reference_runs holds them')
    snaps = [sn for sn in _funnel_snapshots(50) if not sn['explored']]
    assert len(snaps) >= 1, 'no snapshot of a reference run at n_dim 50'
    ref = snaps[0]
    import torch
    from nautilus_amd import Sampler, unit_prior
    from nautilus_amd.configs import baseline_config
    c = baseline_config('C5-D50')
    s = Sampler(unit_prior, c['likelihood'], n_dim=50, n_live=2000,
                n_networks=4, n_batch=100, vectorized=True, seed=0)
    done = s.run(n_like_max=ref['n_like'], discard_exploration=True)
    torch.cuda.synchronize()
    assert not done and not s.explored
    _invariants(c, s)
    assert s.n_like == ref['n_like']
    assert abs(len(s.bounds) - ref['n_bounds']) <= 0.05 * ref['n_bounds'] + 2
    k = min(len(s.bounds), ref['n_bounds'])
    ours = np.array([b.log_v for b in s.bounds[:k]])
    theirs = np.array(ref['log_v'][:k])
    # the volumes shrink at the reference's rate: 0.8 nats per bound, the
    # reference's two runs within 0.42 of each other at every bound
    assert np.max(np.abs(ours - theirs)) < 1.5
    assert abs(ours[k // 2] - theirs[k // 2]) < 1.0
    shell = np.array(ref['shell_n'][:k - 1], dtype=float)
    assert np.median(np.abs(s.shell_n[:k - 1] / shell - 1)) < 0.1


def test_funnel_D50_against_reference_runs():
    """Configuration 5's problem at FIFTY dimensions -- the dimension of the
    headline metric -- against the reference's own runs at the settings the
    reference finishes (tests/golden/e2e_funnel.json, ``reduced``: n_live
    2000, 4 networks, n_batch 100, n_eff 10000, exploration discarded; 4-5
    hours per run on a CPU core, ~2 minutes here).  At this dimension BOTH
    samplers miss the quadrature evidence by more than the north star's
    0.01 and put E[x_0] ~0.015 above the quadrature value: the bounds lose
    mass at the narrow end of the funnel.  What is
    held: this build's evidence, likelihood calls, number of bounds and the
    posterior of x_0 against the REFERENCE'S, and the reference's own
    assertion |log Z - log Z_true| < 0.1 (tests/test_sampler.py:326) with
    the margin the reference's runs themselves need."""
    from nautilus_amd.configs import funnel_log_z
    ref = _funnel_reference(50, True)
    if len(ref) < 2:
        pytest.skip('fewer than two finished reference runs at n_dim 50 '
                    '(test_funnel_D50_prefix_against_the_reference holds '
                    'the runs in progress)')
    ref_z = np.array([r['log_z'] for r in ref])
    ref_like = np.mean([r['n_like'] for r in ref])
    ref_bounds = np.mean([r['n_bounds'] for r in ref])
    ref_mean = np.mean([r['mean_x0'] for r in ref])
    ref_var = np.mean([r['var_x0'] for r in ref])
    c, s, done = _run('C5-D50', seed=0, n_batch=100, n_live=2000,
                      n_networks=4)
    assert done and s.n_eff >= 10000
    _invariants(c, s)
    # sigma(log Z) of ONE run at this dimension: 0.017 over this build's
    # seeds (profiles/r06/anchors_here.jsonl), the reference's two runs
    # differ by as much
    sigma = 0.017
    assert abs(s.log_z - ref_z.mean()) < 4 * sigma * np.sqrt(1 + 1 / len(ref))
    worst_ref = np.max(np.abs(ref_z - funnel_log_z(50)))
    assert abs(s.log_z - funnel_log_z(50)) < max(0.1, worst_ref + 2 * sigma)
    assert abs(s.n_like / ref_like - 1) < 0.08
    assert abs(len(s.bounds) - ref_bounds) <= 0.05 * ref_bounds + 4
    mean, var = _x0_moments(s)
    se = np.sqrt(ref_var / 10000)
    assert abs(mean[0] - ref_mean) < 5 * se * np.sqrt(1 + 1 / len(ref))
    assert abs(var / ref_var - 1) < 0.15
    assert np.all(np.abs(mean[1:] - 0.5) < 0.01)


@pytest.mark.parametrize('name,discard', [('C5-D10', False)]
                         + ([('C5-D30', True), ('C5-D20', False)]
                            if FULL else []))
def test_C5_family_finishes(name, discard):
    """Configuration 5's own settings (n_live 10000, 8 networks) at the
    dimensions where a run ends.  10 (and 20) dimensions with the
    reference's defaults (discard_exploration=False), the REFERENCE'S OWN
    ASSERTION |log Z - log Z_true| < 0.1 (tests/test_sampler.py:326;
    measured -0.004 / -0.003) and E[x_0] within 0.002 of the quadrature value
    0.4895 (measured +0.0004; five standard errors of this run are 0.0006,
    the exploration bias of test_funnel_exploration_kept_... adds to them).
    30 dimensions with the exploration discarded (260 bounds, ~200 s):
    log Z - analytic = -0.014 and E[x_0] 0.0037 above the quadrature value --
    the loss of mass at the narrow end that the reference shows at its own
    sizes (test_funnel_against_reference_runs); with the exploration kept
    the same run gives -0.048 (round 4: -0.054)."""
    from nautilus_amd.configs import funnel_moments
    c, s, done = _run(name, discard_exploration=discard)
    assert done and s.explored and s.n_eff >= 10000
    assert s.n_dead_bounds <= 2
    mean, var = _x0_moments(s)
    want_mean, want_var = funnel_moments(c['n_dim'])
    if name == 'C5-D30':
        assert abs(s.log_z - c['analytic_log_z']) < 0.05
        assert 0.0 < mean[0] - want_mean < 0.008
    else:
        assert abs(s.log_z - c['analytic_log_z']) < 0.1
        # (at 20 dimensions E[x_0] sits 0.003-0.004 above the quadrature
        # value in the reference's runs as in this build's: the mass lost at
        # the narrow end, test_funnel_against_reference_runs[20])
        assert abs(mean[0] - want_mean) < (0.002 if c['n_dim'] <= 10
                                           else 0.007)
    assert abs(var / want_var - 1) < 0.2
    assert np.all(np.abs(mean[1:] - 0.5) < 0.01)
