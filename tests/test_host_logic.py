"""Host-side logic of the product (no GPU): construction numerics against the
oracle / golden vectors, Prior, pool wrapper, argument checking."""

import numpy as np
import pytest
from scipy.stats import norm

from conftest import load_golden


@pytest.fixture(autouse=True)
def host_khachiyan(monkeypatch):
    """No GPU here: the host logic around the MVEE (enlargement, Cholesky
    factors, greedy cube/ellipsoid choice, overlap test, batching of the
    construction coroutines) is exercised with a numpy stand-in for the device
    fit; the device kernels themselves are pinned under -m gpu
    (tests/test_hip_parity.py)."""
    from nautilus_amd import geometry
    import helpers
    monkeypatch.setattr(geometry, 'mvee_batch', helpers.mvee_numpy_batch)
    geometry._ELL_CACHE.clear()


@pytest.mark.parametrize('d', [3, 20, 50])
def test_ellipsoid_construction_matches_reference(d):
    from nautilus_amd import geometry
    g = load_golden('ellipsoid_D%d' % d)
    p = geometry.ellipsoid_params(g['points'], float(g['enlarge']))
    assert np.allclose(p['c'], g['c'], rtol=0, atol=1e-9)
    assert np.allclose(p['B'], g['B'], rtol=0, atol=1e-9)
    assert np.allclose(p['B_inv'], g['B_inv'], rtol=1e-8, atol=1e-8)
    assert np.all(np.triu(p['B_inv'], 1) == 0)
    assert abs(geometry.ellipsoid_log_volume(p['B']) - float(g['log_v'])) < \
        1e-8
    # every input point is enclosed (basic.py:237-239)
    y = (g['points'] - p['c']) @ p['B_inv'].T
    assert np.all(np.sum(y**2, axis=1) < 1)


def test_ellipsoid_volume_is_analytic():
    # reference tests/test_bounds.py:136-145
    from nautilus_amd import geometry
    from scipy.special import gamma
    d = 4
    b = np.eye(d) * 0.3
    assert np.isclose(geometry.ellipsoid_log_volume(b),
                      np.log(0.3**d * np.pi**(d / 2) / gamma(d / 2 + 1)))


def test_ellipsoid_errors():
    from nautilus_amd import geometry
    with pytest.raises(ValueError):
        geometry.ellipsoid_params(np.random.random((3, 3)))
    with pytest.raises(ValueError):
        geometry.ellipsoid_params(np.random.random((30, 3)), 0.9)


def test_mixture_construction_matches_reference():
    from nautilus_amd import geometry
    g = load_golden('mixture_D6')
    dim_cube, ell = geometry.mixture_params(g['points'], 1.1)
    assert np.array_equal(dim_cube, g['dim_cube'])
    assert np.allclose(ell['B'], g['B'], rtol=0, atol=1e-9)
    assert np.allclose(ell['c'], g['c'], rtol=0, atol=1e-9)


@pytest.mark.parametrize('seed,d,n_flat', [(0, 8, 2), (1, 12, 5), (2, 20, 0),
                                           (3, 5, 5)])
def test_mixture_greedy_choice_matches_oracle(seed, d, n_flat):
    """The Schur-complement form of the drop-one-dimension volumes
    (geometry.mixture_params) picks the same cube dimensions and the same
    ellipsoid as the reference's loop restated in the oracle."""
    from nautilus_amd import geometry
    from oracle import bounds_oracle as bo
    rng = np.random.default_rng(seed)
    pts = np.clip(0.5 + 0.07 * rng.normal(size=(400, d)), 0, 1)
    pts[:, :n_flat] = rng.random((400, n_flat))
    dim_cube, ell = geometry.mixture_params(pts, 1.1)
    ref = bo.OMixture.build(pts, enlarge_per_dim=1.1)
    assert np.array_equal(dim_cube, ref.dim_cube)
    if ref.ellipsoid is None:
        assert ell is None
    else:
        assert np.allclose(ell['B'], ref.ellipsoid.B, rtol=1e-8, atol=1e-10)
        assert np.allclose(ell['c'], ref.ellipsoid.c, rtol=0, atol=1e-9)


def test_overlap_test_matches_oracle():
    from nautilus_amd import geometry
    from oracle import bounds_oracle as bo
    rng = np.random.default_rng(0)
    for shift in (0.05, 0.3, 3.0):
        a = rng.normal(size=(200, 3)) * 0.1
        b = rng.normal(size=(200, 3)) * 0.1 + shift
        pa, pb = (geometry.ellipsoid_params(x, 1.0) for x in (a, b))
        ea, eb = (bo.OEllipsoid.build(x, 1.0) for x in (a, b))
        assert geometry.ellipsoids_overlap([pa, pb]) == \
            bo.ellipsoids_overlap([ea, eb])
    assert geometry.ellipsoids_overlap([pa, pb]) is False


def test_glorot_init_is_sklearns():
    from nautilus_amd import emulator
    from oracle import mlp_oracle as mo
    for seed in (0, 3):
        c1, i1 = emulator._glorot(7, np.random.RandomState(seed))
        c2, i2, _ = mo.glorot_init(7, seed)
        for a, b in zip(c1 + i1, c2 + i2):
            assert np.array_equal(a, b)


def test_emulator_kwargs_translation():
    from nautilus_amd import emulator
    hp = emulator._hparams_from_kwargs(dict(learning_rate_init=1e-3,
                                            max_iter=50, tol=1e-4))
    assert hp == dict(lr=1e-3, max_iter=50, tol=1e-4)
    assert emulator._hparams_from_kwargs(
        dict(hidden_layer_sizes=(100, 50, 20), alpha=0)) == \
        dict(hidden=(100, 50, 20))
    # narrower three-layer networks ride in the default's tiles (zero padded)
    assert emulator._hparams_from_kwargs(
        dict(hidden_layer_sizes=[64, 32, 16])) == dict(hidden=(64, 32, 16))
    for bad in ((10, 10), (128, 50, 20), (100, 50, 20, 5), (100, 50, 0)):
        with pytest.raises(ValueError):
            emulator._hparams_from_kwargs(dict(hidden_layer_sizes=bad))
    with pytest.raises(ValueError):
        emulator._hparams_from_kwargs(dict(momentum=0.5))
    with pytest.raises(ValueError):
        emulator._hparams_from_kwargs(dict(activation='tanh'))
    # the Glorot draw of a narrower network consumes the stream as
    # scikit-learn does for that architecture; padding adds exact zeros
    from oracle import mlp_oracle as mo
    c1, i1 = emulator._glorot(7, np.random.RandomState(2), (64, 32, 16))
    c2, i2, _ = mo.glorot_init(7, 2, (64, 32, 16))
    for a, b in zip(c1 + i1, c2 + i2):
        assert np.array_equal(a, b)
    pc, pi = emulator.pad_network(c1, i1, 7)
    assert [w.shape for w in pc] == [(7, 100), (100, 50), (50, 20), (20, 1)]
    assert np.array_equal(pc[1][:64, :32], c1[1]) and pc[1][64:].max() == 0 \
        and np.abs(pc[1][:, 32:]).max() == 0 and pi[2][16:].max() == 0
    with pytest.warns(Warning):
        emulator._hparams_from_kwargs(dict(random_state=1))


def test_prior_matches_reference_semantics():
    # reference tests/test_prior.py
    from nautilus_amd import Prior
    prior = Prior()
    prior.add_parameter('a')
    prior.add_parameter('b', dist=(1, 3))
    prior.add_parameter('c', dist=norm(loc=2.0, scale=0.5))
    prior.add_parameter('d', dist=1.5)
    prior.add_parameter('e', dist='a')
    prior.add_parameter()
    assert prior.keys[-1] == 'x_5'
    assert prior.dimensionality() == 4
    u = np.random.default_rng(0).random((10, 4))
    phys = prior.unit_to_physical(u)
    assert np.allclose(phys[:, 0], u[:, 0])
    assert np.allclose(phys[:, 1], 1 + 2 * u[:, 1])
    assert np.allclose(phys[:, 2], norm(loc=2.0, scale=0.5).ppf(u[:, 2]))
    as_dict = prior.unit_to_dictionary(u)
    assert np.all(as_dict['d'] == 1.5)
    assert np.array_equal(as_dict['e'], as_dict['a'])
    with pytest.raises(ValueError):
        prior.unit_to_physical(u[:, :3])
    with pytest.raises(ValueError):
        prior.add_parameter('a')
    with pytest.raises(TypeError):
        prior.add_parameter(3)
    with pytest.raises(ValueError):
        prior.add_parameter('z', dist='nope')
    with pytest.raises(TypeError):
        prior.add_parameter('y', dist=[1, 2])


def _square(x):
    return x * x


def test_pool_wrapper():
    # reference nautilus/pool.py:65-107
    from multiprocessing import Pool
    from nautilus_amd import NautilusPool
    p = NautilusPool(2)
    try:
        assert p.size == 2
        assert p.map(_square, [1, 2, 3]) == [1, 4, 9]
    finally:
        p.pool.close()
    with Pool(3) as raw:
        assert NautilusPool(raw).size == 3

    class Odd:
        def map(self, f, it):
            return map(f, it)
    with pytest.raises(ValueError):
        NautilusPool(Odd()).size


def test_sampler_argument_errors():
    from nautilus_amd import GaussianLikelihood, Prior, Sampler, unit_prior

    def like(x):
        return 0.0
    with pytest.raises(ValueError):
        Sampler(lambda x: x, like)                       # n_dim missing
    with pytest.raises(ValueError):
        Sampler(lambda x: x, like, n_dim=1)
    with pytest.raises(ImportError):                     # no h5py here
        Sampler(lambda x: x, like, n_dim=2, filepath='x.hdf5')
    dev_like = GaussianLikelihood(np.zeros(2) + 0.5, np.eye(2) * 0.01)
    with pytest.raises(ValueError):
        Sampler(lambda x: x, dev_like, n_dim=2)          # host prior
    assert unit_prior.device is True
    assert isinstance(Prior(), Prior)
    # emulator options the device trainer does not hold: at construction, not
    # at the first add_bound (neural.py:79-83 passes them to MLPRegressor)
    for kw in (dict(hidden_layer_sizes=(10, 10)), dict(solver='lbfgs'),
               dict(hidden_layer_sizes=(200, 50, 20))):
        with pytest.raises(ValueError):
            Sampler(unit_prior, dev_like, n_dim=2, neural_network_kwargs=kw)
    # (accepted options: tests/test_sampler_behaviour_gpu.py)


def test_shell_batch_prefix_rule_is_negative_binomial():
    """The device driver stops at the n-th in-shell point of a block of
    in-bound points; the reference loops with a shrinking deficit
    (sampler.py:790-823).  Both must examine exactly the points up to and
    including the n-th success."""
    rng = np.random.default_rng(2)
    for _ in range(50):
        keep = rng.random(500) < 0.3
        need = 20
        # reference loop
        pos, have, n_bound = 0, 0, 0
        while have < need:
            req = need - have
            n_bound += req
            have += int(keep[pos:pos + req].sum())
            pos += req
        # prefix rule
        csum = np.cumsum(keep)
        used = int(np.searchsorted(csum, need)) + 1
        assert used == n_bound


def test_prior_device_spec():
    """Which priors can be transformed on the GPU (row f4): uniform / normal
    parameters, fixed and tied parameters are passed through."""
    from scipy.stats import expon
    from nautilus_amd import Prior
    p = Prior()
    p.add_parameter('a', dist=(-3, 5))
    p.add_parameter('b', dist=norm(loc=2.0, scale=0.5))
    p.add_parameter('c', dist=1.5)
    p.add_parameter('d', dist='a')
    kind, loc, scale = p.device_spec()
    assert kind.tolist() == [0, 1]
    assert np.allclose(loc, [-3, 2.0]) and np.allclose(scale, [8, 0.5])
    assert p.device
    p.add_parameter('e', dist=expon())
    assert p.device_spec() is None and not p.device


def test_rosenbrock_exact_evidence_helper():
    """The transfer quadrature behind BASELINE config 3's evidence: equal to
    brute-force quadrature in two dimensions, and the value quoted in
    nautilus_amd/configs.py in thirty."""
    from helpers import rosenbrock_log_z_exact
    g = (np.arange(1000) + 0.5) / 1000
    x = 10 * np.stack(np.meshgrid(g, g, indexing='ij'), axis=-1) - 5
    brute = np.log(np.mean(np.exp(-(100 * (x[..., 1] - x[..., 0]**2)**2 +
                                    (1 - x[..., 0])**2))))
    assert abs(rosenbrock_log_z_exact(2, 1000) - brute) < 1e-10
    assert abs(rosenbrock_log_z_exact(30, 1000) - (-137.4875)) < 1e-4


def test_native_shuffle_streams_equal_numpy():
    """``nb_host_shuffle_epochs`` (the minibatch orders of the emulator fits,
    one native thread per network) against the calls it replaces: numpy's
    legacy ``RandomState.shuffle`` composed epoch after epoch, as
    sklearn.utils.shuffle does inside MLPRegressor.fit
    (_multilayer_perceptron.py:700-704) -- bit for bit, across chunks, with
    an inactive stream left where it is."""
    from nautilus_amd.emulator import _ShuffleStreams
    ns = [1000, 37, 70001, 1, 2]
    states = [np.random.RandomState(i) for i in range(len(ns))]
    ref = [np.random.RandomState(i) for i in range(len(ns))]
    for rs in states + ref:
        rs.uniform(-1, 1, (50, 100))         # the Glorot draw comes first
    orders = [np.arange(n) for n in ns]
    sh = _ShuffleStreams(states, ns)
    n_ep = 3
    for rnd in range(3):
        active = [True] * len(ns)
        if rnd == 1:
            active[2] = False                # stopped network
        out = np.full(n_ep * sum(ns), -1, dtype=np.int32)
        sh.fill(out, n_ep, active)
        for i, n in enumerate(ns):
            at = int(sh.offsets[i]) * n_ep
            rows = out[at:at + n_ep * n].reshape(n_ep, n)
            for ep in range(n_ep):
                if active[i]:
                    idx = np.arange(n)
                    ref[i].shuffle(idx)
                    orders[i] = orders[i][idx]
                assert np.array_equal(rows[ep], orders[i]), (rnd, i, ep)


def test_amortised_host_append_equals_np_append():
    """``sampler._grow`` (the shells' log L on the host grow batch by batch,
    sampler.py:1135-1136): the same values as ``np.append`` whatever happens
    to the array in between (boolean masks as in add_bound, other views), and
    views handed out earlier keep their values."""
    from nautilus_amd.sampler import _grow
    rng = np.random.default_rng(0)
    a = np.zeros(0)
    ref = np.zeros(0)
    held = []
    for i in range(300):
        new = rng.random(int(rng.integers(0, 40)))
        a = _grow(a, new)
        ref = np.append(ref, new)
        assert np.array_equal(a, ref)
        if i % 37 == 0:
            held.append((a, a.copy()))
        if i % 53 == 0:
            keep = rng.random(len(a)) < 0.5
            a, ref = a[keep], ref[keep]
    for view, copy in held:
        assert np.array_equal(view, copy)
    # not the leading view of a buffer: copied, never written behind
    base = np.arange(10.0)
    tail = base[2:6]
    out = _grow(tail, [7.0])
    assert np.array_equal(out, [2, 3, 4, 5, 7]) and base[6] == 6.0
    assert np.array_equal(_grow(np.arange(5.0)[::2], [1.0]), [0, 2, 4, 1])
