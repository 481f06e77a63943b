"""Driver behaviour of the product Sampler on the GPU, following the
situations the reference's tests/test_sampler.py and tests/test_blobs.py
exercise: switching discard_exploration after the run, huge enlargement
factors, empty shells, n_like_max / timeout interruption and resumption,
non-nested bounds (funnel), likelihood plateaus and -inf regions, and every
blob dtype convention."""

import warnings

import numpy as np
import pytest
from scipy.stats import norm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def flat(x):
    return -np.linalg.norm(x - 0.5, axis=-1) * 0.001


def bowl(x):
    return -np.linalg.norm(x - 0.5)**2 * 0.001


def ident(x):
    return x


@pytest.mark.parametrize('start,end', [(True, True), (True, False),
                                       (False, True), (False, False),
                                       (True, 1)])
def test_switch_discard_exploration(start, end):
    from nautilus_amd import Sampler
    s = Sampler(ident, flat, n_dim=2, n_networks=1, vectorized=True,
                n_live=500, seed=3)
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


def test_enlarge_per_dim_leaves_one_bound():
    from nautilus_amd import Sampler
    s = Sampler(ident, bowl, n_dim=2, enlarge_per_dim=100, n_networks=0,
                seed=0)
    s.run(f_live=0.1, n_eff=0)
    assert np.isclose(s.n_like, s.n_eff, rtol=0, atol=1)
    assert len(s.bounds) == 1
    assert np.isclose(s.log_z, -4 * 0.5**3 / 3 * 0.001, rtol=0, atol=1e-4)


def test_constant_likelihood_builds_no_bound():
    """tests/test_sampler.py:334-348: log Z = 0 and only the unit cube."""
    from nautilus_amd import Sampler
    s = Sampler(ident, lambda x: 0, 2, n_live=500, seed=0)
    s.run(f_live=0.1, n_eff=0)
    assert np.isclose(s.log_z, 0)
    assert len(s.bounds) == 1


def test_empty_shells_at_the_end():
    from nautilus_amd import Sampler
    s = Sampler(ident, bowl, n_dim=2, n_networks=0, seed=0, n_update=1,
                n_live=10, n_batch=1)
    s.run(f_live=1e-3, n_eff=0)
    assert np.isfinite(s.log_z)


def test_n_like_max_stops_and_resumes():
    from nautilus_amd import Sampler
    a = Sampler(ident, bowl, n_dim=2, n_networks=0, seed=0)
    b = Sampler(ident, bowl, n_dim=2, n_networks=0, seed=0)
    assert a.run()
    limits = list(range(0, a.n_like, 7 * a.n_batch)) + [a.n_like]
    for n_like_max in limits:
        ok = b.run(n_like_max=n_like_max)
        assert b.n_like <= n_like_max + b.n_batch
        assert ok == (a.n_like == b.n_like)
    assert a.log_z == b.log_z
    assert a.n_eff == b.n_eff


def test_timeout_and_continue():
    from nautilus_amd import Sampler
    s = Sampler(ident, bowl, n_dim=10, n_networks=0, seed=0)
    # (the reference needs more than a second for this problem; the device
    # path does not, hence the shorter limit)
    assert not s.run(timeout=0.05)
    n_like = s.n_like
    assert s.run(timeout=60)
    assert s.n_like > n_like


def funnel2(x):
    return (norm.logpdf(x[0], loc=0.5, scale=0.1) +
            norm.logpdf(x[1], loc=0.5, scale=np.exp(20 * (x[0] - 0.5)) / 100))


def test_funnel_non_nested_bounds():
    from nautilus_amd import Sampler
    rng = np.random.default_rng(0)
    x0 = rng.normal(0.5, 0.1, 1000000)
    x1 = rng.normal(0.5, np.exp(20 * (x0 - 0.5)) / 100)
    truth = np.log(np.mean((x0 > 0) & (x0 < 1) & (x1 > 0) & (x1 < 1)))
    s = Sampler(ident, funnel2, n_dim=2, n_networks=1, seed=0)
    s.run()
    assert np.isclose(truth, s.log_z, rtol=0, atol=0.1)
    occ = s.shell_bound_occupation()
    if np.all(occ == np.tril(np.ones_like(occ))):
        warnings.warn('The funnel distribution was too easy.', RuntimeWarning)


def wall(x):
    return -np.inf if x[0] < 0.9 else np.log(x[0] - 0.9)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_plateau_with_minus_infinity(seed):
    from nautilus_amd import Sampler
    s = Sampler(ident, wall, 2, n_live=1000, n_networks=1, seed=seed)
    s.run(f_live=0.1)
    assert np.isclose(s.log_z, np.log(0.5 * 0.1**2), rtol=0, atol=0.1)


def stairs(x):
    return np.ceil(-np.log10(1 - x[0]))


def test_plateau_staircase():
    from nautilus_amd import Sampler
    truth = np.log(np.sum(0.9 * 0.1**np.arange(100) *
                          np.exp(1 + np.arange(100))))
    s = Sampler(ident, stairs, 2, n_live=2000, n_networks=1, seed=0)
    s.run(f_live=1e-6)
    assert np.isclose(s.log_z, truth, atol=0.1)
    assert np.all(np.isclose(s.shell_log_l_min[1:],
                             np.arange(len(s.bounds) - 1) + 2, rtol=0))


@pytest.mark.parametrize('vectorized', [True, False])
@pytest.mark.parametrize('discard', [True, False])
@pytest.mark.parametrize('case', ['f64', 'i64', 'two_f32', 'named', 'one_dtype',
                                  'array'])
def test_blob_conventions(case, vectorized, discard):
    from nautilus_amd import Sampler
    f32 = np.float32

    def like(x):
        ll = -np.linalg.norm(x - 0.5, axis=-1) * 0.001
        a, b = (x[:, 0], x[:, 1]) if vectorized else (x[0], x[1])
        if case in ('f64', 'i64'):
            t = np.float64 if case == 'f64' else np.int64
            return ll, ((10 * a).astype(t) if vectorized else t(10 * a))
        if case == 'two_f32':
            return (ll, a.astype(f32), b.astype(f32)) if vectorized else \
                (ll, f32(a), f32(b))
        if case == 'array':
            return ll, x[..., :2]
        return ll, a, b

    dtype = {'named': [('a', '|S10'), ('b', np.int16)], 'one_dtype': f32,
             'array': f32}.get(case)
    s = Sampler(ident, like, n_dim=2, n_live=200, vectorized=vectorized,
                n_networks=0, blobs_dtype=dtype, seed=5)
    s.run(f_live=0.2, n_like_max=2000, discard_exploration=discard)
    pts, log_w, log_l, blobs = s.posterior(return_blobs=True)
    assert len(pts) == len(blobs)
    if case in ('f64', 'i64'):
        t = np.float64 if case == 'f64' else np.int64
        assert blobs.dtype == t
        assert np.all((10 * pts[:, 0]).astype(t) == blobs)
        pe, _, _, be = s.posterior(return_blobs=True, equal_weight=True)
        assert np.all((10 * pe[:, 0]).astype(t) == be)
    elif case == 'two_f32':
        assert blobs['blob_0'].dtype == f32 and blobs['blob_1'].dtype == f32
        assert np.all(pts[:, 0].astype(f32) == blobs['blob_0'])
        assert np.all(pts[:, 1].astype(f32) == blobs['blob_1'])
    elif case == 'named':
        assert blobs['a'].dtype == np.dtype('|S10')
        assert blobs['b'].dtype == np.int16
        assert np.all(pts[:, 0].astype('|S10') == blobs['a'])
        assert np.all(pts[:, 1].astype(np.int16) == blobs['b'])
    else:
        assert np.all(pts[:, 0].astype(f32) == blobs[:, 0])
        assert np.all(pts[:, 1].astype(f32) == blobs[:, 1])


@pytest.mark.parametrize('case', ['one', 'two', 'named'])
def test_blobs_of_a_device_likelihood(case):
    """A likelihood that runs on the device may return blobs as well (cuda
    tensors, one row per point): they are kept like a vectorized host
    likelihood's (reference sampler.py:875-904, tests/test_blobs.py)."""
    import torch
    from nautilus_amd import Sampler, unit_prior

    def like(x):
        assert x.is_cuda
        ll = -torch.linalg.norm(x - 0.5, dim=-1) * 0.001
        if case == 'one':
            return ll, (10 * x[:, 0]).to(torch.int64)
        return ll, x[:, 0].to(torch.float32), (1000 * x[:, 1]).to(torch.int16)
    like.device = True
    dtype = [('a', np.float32), ('b', np.int16)] if case == 'named' else None
    s = Sampler(unit_prior, like, n_dim=2, n_live=200, vectorized=True,
                n_networks=1, blobs_dtype=dtype, seed=5)
    s.run(f_live=0.2, n_like_max=3000, discard_exploration=True)
    pts, log_w, log_l, blobs = s.posterior(return_blobs=True)
    assert len(pts) == len(blobs) > 0
    if case == 'one':
        assert blobs.dtype == np.int64
        assert np.all((10 * pts[:, 0]).astype(np.int64) == blobs)
    else:
        a, b = ('a', 'b') if case == 'named' else ('blob_0', 'blob_1')
        assert blobs[a].dtype == np.float32 and blobs[b].dtype == np.int16
        assert np.all(pts[:, 0].astype(np.float32) == blobs[a])
        assert np.all((1000 * pts[:, 1]).astype(np.int16) == blobs[b])


def test_run_with_a_narrower_emulator_architecture():
    """``neural_network_kwargs=dict(hidden_layer_sizes=(64, 32, 16))`` (the
    reference passes it to MLPRegressor, neural.py:79-83): the run's emulators
    have that architecture and the evidence of the README Gaussian comes out
    as with the default one."""
    from nautilus_amd import GaussianLikelihood, Sampler, unit_prior
    like = GaussianLikelihood([0.4, 0.5, 0.6], 0.01 * np.eye(3))
    s = Sampler(unit_prior, like, n_dim=3, n_live=500, seed=1,
                neural_network_kwargs=dict(hidden_layer_sizes=(64, 32, 16)))
    assert s.run(n_eff=2000, discard_exploration=True) is True
    nets = [net for b in s.bounds[1:] for nb in b.neural_bounds
            if nb.emulator is not None for net in nb.emulator.neural_networks]
    assert len(nets) >= 4
    assert all([c.shape for c in net.coefs_] ==
               [(3, 64), (64, 32), (32, 16), (16, 1)] for net in nets)
    assert abs(s.log_z - (-6.4e-5)) < 0.05
    # no emulators: the options are not looked at
    Sampler(unit_prior, like, n_dim=3, n_networks=0,
            neural_network_kwargs=dict(solver='lbfgs'))
