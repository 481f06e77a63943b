"""The scenarios of the reference's tests/test_sampler.py that are not already
in test_sampler_gpu.py, with the reference's parameters and thresholds (each
test names the lines it mirrors).  Host likelihoods throughout, as in the
reference: these runs exercise the sampler logic around the device path
(stopping rules, shells, plateaus, exploration switch), not its speed."""

import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _identity(u):
    return u


def _flat_bowl(x):
    return -np.linalg.norm(x - 0.5, axis=-1)**2 * 0.001


@pytest.mark.parametrize('start,end', [(True, True), (True, False),
                                       (False, True), (False, False),
                                       (True, 'yes')])
def test_switch_exploration(start, end):
    """tests/test_sampler.py:97-131: ``discard_exploration`` can be switched
    after the run; points and evidence change exactly when the flag does, a
    non-boolean raises ValueError."""
    from nautilus_amd import Sampler

    def like(x):
        return -np.linalg.norm(x - 0.5, axis=-1) * 0.001
    s = Sampler(_identity, like, n_dim=2, n_networks=1, vectorized=True,
                n_live=500, seed=0)
    s.run(f_live=0.45, n_eff=10000, discard_exploration=start)
    assert s.discard_exploration == start
    n_start, z_start = len(s.posterior()[0]), s.log_z
    if not isinstance(end, bool):
        with pytest.raises(ValueError):
            s.discard_exploration = end
        return
    s.discard_exploration = end
    n_end, z_end = len(s.posterior()[0]), s.log_z
    assert (start == end) == (n_start == n_end)
    assert (start == end) == (z_start == z_end)


def test_enlarge_per_dim_keeps_one_bound():
    """tests/test_sampler.py:218-241 with the reference's exact settings
    (defaults, n_networks 0, enlarge_per_dim 100, f_live 0.1, n_eff 0)."""
    from nautilus_amd import Sampler
    s = Sampler(_identity, _flat_bowl, n_dim=2, enlarge_per_dim=100,
                n_networks=0, seed=0)
    s.run(f_live=0.1, n_eff=0)
    assert np.isclose(s.n_like, s.n_eff, rtol=0, atol=1)
    assert len(s.bounds) == 1
    assert np.isclose(s.log_z, -4 * 0.5**3 / 3 * 0.001, rtol=0, atol=1e-4)


def test_empty_shells():
    """tests/test_sampler.py:244-258: one sample per shell on average
    (n_live 10, n_batch 1, n_update 1) leaves shells empty at the end; the
    run has to get through it."""
    from nautilus_amd import Sampler
    s = Sampler(_identity, _flat_bowl, n_dim=2, n_networks=0, seed=0,
                n_update=1, n_live=10, n_batch=1)
    assert s.run(f_live=1e-3, n_eff=0) is True
    assert np.isfinite(s.log_z)
    assert np.any(np.asarray(s.shell_n) == 0) or len(s.bounds) > 3


def test_n_like_max_stops_and_resumes():
    """tests/test_sampler.py:261-281: stepping ``n_like_max`` up to the
    length of an uninterrupted run stops early every time, never overshoots
    by more than a batch and ends with the identical evidence and effective
    sample size.  (The reference steps by 1; a stride keeps the number of
    ``run`` calls in the hundreds.)"""
    from nautilus_amd import Sampler
    a = Sampler(_identity, _flat_bowl, n_dim=2, n_networks=0, seed=0)
    b = Sampler(_identity, _flat_bowl, n_dim=2, n_networks=0, seed=0)
    a.run()
    limits = list(range(0, a.n_like + 1, 23)) + [a.n_like]
    for n_like_max in limits:
        success = b.run(n_like_max=n_like_max)
        assert b.n_like <= n_like_max + b.n_batch
        assert success == (a.n_like == b.n_like)
    assert a.log_z == b.log_z
    assert a.n_eff == b.n_eff


def test_timeout_stops_and_resumes():
    """tests/test_sampler.py:284-299.  The reference's 10-D run "shouldn't
    finish within 1 second"; here it takes 0.4 s, so the first limit is 50 ms."""
    from nautilus_amd import Sampler
    s = Sampler(_identity, _flat_bowl, n_dim=10, n_networks=0, seed=0)
    assert s.run(timeout=0.05) is False
    n_like = s.n_like
    assert s.run(timeout=5) is True
    assert s.n_like > n_like


def test_funnel():
    """tests/test_sampler.py:302-331: 2-D funnel, evidence within 0.1 of a
    10^6-sample Monte Carlo estimate; the bounds are usually not nested."""
    from scipy.stats import norm
    from nautilus_amd import Sampler

    def like(x):
        return (norm.logpdf(x[0], loc=0.5, scale=0.1) +
                norm.logpdf(x[1], loc=0.5,
                            scale=np.exp(20 * (x[0] - 0.5)) / 100))
    rng = np.random.RandomState(0)
    x_0 = rng.normal(loc=0.5, scale=0.1, size=1000000)
    x_1 = rng.normal(loc=0.5, scale=np.exp(20 * (x_0 - 0.5)) / 100)
    log_z_true = np.log(np.mean((x_0 > 0) & (x_0 < 1) & (x_1 > 0) &
                                (x_1 < 1)))
    s = Sampler(_identity, like, n_dim=2, n_networks=1, seed=0)
    s.run()
    assert np.isclose(log_z_true, s.log_z, rtol=0, atol=0.1)
    occ = s.shell_bound_occupation()
    if np.all(occ == np.tril(np.ones_like(occ))):
        warnings.warn('The funnel distribution was too easy.', RuntimeWarning)


def test_constant_likelihood():
    """tests/test_sampler.py:334-348: log Z = 0 and no bound is ever built."""
    from nautilus_amd import Sampler
    s = Sampler(_identity, lambda x: 0, 2, n_live=500, seed=0)
    s.run(f_live=0.1, n_eff=0)
    assert np.isclose(s.log_z, 0)
    assert len(s.bounds) == 1


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_plateau_with_excluded_region(seed):
    """tests/test_sampler.py:351-368: log L = -inf on 90 % of the prior
    (seeds 0-2 of the reference's ten)."""
    from nautilus_amd import Sampler

    def like(x):
        return -np.inf if x[0] < 0.9 else np.log(x[0] - 0.9)
    s = Sampler(_identity, like, 2, n_live=1000, n_networks=1, seed=seed)
    s.run(f_live=0.1)
    assert np.isclose(s.log_z, np.log(0.5 * 0.1**2), rtol=0, atol=0.1)


def test_plateau_staircase():
    """tests/test_sampler.py:371-392: a staircase likelihood; the shell
    thresholds land on the plateaus."""
    from nautilus_amd import Sampler

    def like(x):
        return np.ceil(-np.log10(1 - x[0]))
    log_z_true = np.log(np.sum(0.9 * 0.1**np.arange(100) *
                               np.exp(1 + np.arange(100))))
    s = Sampler(_identity, like, 2, n_live=2000, n_networks=1, seed=0)
    s.run(f_live=1e-6)
    assert np.isclose(s.log_z, log_z_true, atol=0.1)
    assert np.all(np.isclose(s.shell_log_l_min[1:],
                             np.arange(len(s.bounds) - 1) + 2, rtol=0))


# ---- blobs (the reference's tests/test_blobs.py) ---------------------------

def _blob_run(like, vectorized, discard, **kwargs):
    from nautilus_amd import Sampler
    s = Sampler(_identity, like, n_dim=2, n_live=200, vectorized=vectorized,
                n_networks=0, **kwargs)
    s.run(f_live=0.2, n_like_max=2000, discard_exploration=discard)
    return s


def _ll(x):
    return -np.linalg.norm(x - 0.5, axis=-1) * 0.001


@pytest.mark.parametrize('dtype', [np.float64, np.int64])
@pytest.mark.parametrize('vectorized', [True, False])
@pytest.mark.parametrize('discard', [True, False])
def test_blobs_single(dtype, vectorized, discard):
    """tests/test_blobs.py:14-39: one blob keeps its dtype and follows its
    point, also through equal-weight resampling."""
    def like(x):
        if vectorized:
            return _ll(x), (10 * x[:, 0]).astype(dtype)
        return _ll(x), dtype(10 * x[0])
    s = _blob_run(like, vectorized, discard)
    s.posterior(return_blobs=True)
    points, _, _, blobs = s.posterior(return_blobs=True, equal_weight=True)
    assert len(points) == len(blobs)
    assert blobs.dtype == dtype
    assert np.all((10 * points[:, 0]).astype(dtype) == blobs)


@pytest.mark.parametrize('vectorized', [True, False])
@pytest.mark.parametrize('discard', [True, False])
def test_blobs_multi(vectorized, discard):
    """tests/test_blobs.py:42-65: several blobs become the fields blob_0,
    blob_1 with their own dtypes."""
    def like(x):
        if vectorized:
            return (_ll(x), x[:, 0].astype(np.float32),
                    x[:, 1].astype(np.float32))
        return _ll(x), np.float32(x[0]), np.float32(x[1])
    s = _blob_run(like, vectorized, discard)
    points, _, _, blobs = s.posterior(return_blobs=True)
    assert len(points) == len(blobs)
    assert blobs['blob_0'].dtype == np.float32
    assert blobs['blob_1'].dtype == np.float32
    assert np.all(points[:, 0].astype(np.float32) == blobs['blob_0'])
    assert np.all(points[:, 1].astype(np.float32) == blobs['blob_1'])


def _two_blobs(vectorized):
    def like(x):
        if vectorized:
            return _ll(x), x[:, 0], x[:, 1]
        return _ll(x), x[0], x[1]
    return like


@pytest.mark.parametrize('vectorized', [True, False])
@pytest.mark.parametrize('discard', [True, False])
def test_blobs_structured_dtype(vectorized, discard):
    """tests/test_blobs.py:68-91: a structured ``blobs_dtype`` (bytes and
    int16 fields)."""
    dt = [('a', '|S10'), ('b', np.int16)]
    s = _blob_run(_two_blobs(vectorized), vectorized, discard, blobs_dtype=dt)
    points, _, _, blobs = s.posterior(return_blobs=True)
    assert len(points) == len(blobs)
    assert blobs['a'].dtype == dt[0][1] and blobs['b'].dtype == dt[1][1]
    assert np.all(points[:, 0].astype(dt[0][1]) == blobs['a'])
    assert np.all(points[:, 1].astype(dt[1][1]) == blobs['b'])


@pytest.mark.parametrize('vectorized', [True, False])
@pytest.mark.parametrize('discard', [True, False])
@pytest.mark.parametrize('as_array', [False, True])
def test_blobs_single_dtype(vectorized, discard, as_array):
    """tests/test_blobs.py:94-135: one dtype for all blobs gives a 2-D array,
    whether the likelihood returns the blobs one by one or as one array."""
    like = (lambda x: (_ll(x), x[..., :2])) if as_array else \
        _two_blobs(vectorized)
    s = _blob_run(like, vectorized, discard, blobs_dtype=np.float32)
    points, _, _, blobs = s.posterior(return_blobs=True)
    assert len(points) == len(blobs)
    assert np.all(points[:, 0].astype(np.float32) == blobs[:, 0])
    assert np.all(points[:, 1].astype(np.float32) == blobs[:, 1])
