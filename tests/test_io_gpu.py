"""Checkpoint / resume, following the reference's
tests/test_io.py: bounds and emulators survive write + read, a resumed
sampler continues exactly like the one that was never interrupted."""

import sys

import numpy as np
import pytest

import fake_h5py

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def h5(monkeypatch):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    monkeypatch.setitem(sys.modules, 'h5py', fake_h5py)


def _cloud(n=600, d=3, seed=0):
    return np.random.default_rng(seed).random((n, d))


@pytest.mark.parametrize('name', ['UnitCube', 'Ellipsoid',
                                  'UnitCubeEllipsoidMixture', 'Union',
                                  'NeuralBound', 'NautilusBound',
                                  'PhaseShift'])
@pytest.mark.parametrize('sync', [True, False])
def test_bounds_round_trip(name, sync):
    import nautilus_amd.bounds as nb
    pts = _cloud()
    log_l = -np.linalg.norm(pts - 0.5, axis=1)
    rng = np.random.default_rng(0)
    cls = getattr(nb, name)
    if name == 'UnitCube':
        b = cls.compute(3, rng=rng)
    elif name in ('NeuralBound', 'NautilusBound'):
        args = (np.log(0.5),) if name == 'NautilusBound' else ()
        b = cls.compute(pts, log_l, np.median(log_l), *args, n_networks=1,
                        rng=rng)
    elif name == 'PhaseShift':
        b = cls.compute(pts, np.arange(2))
    else:
        b = cls.compute(pts[log_l > np.median(log_l)], rng=rng)
    if name in ('Union', 'NautilusBound'):
        b.sample(50)                     # non-trivial counters and queue
    group = fake_h5py.Group()
    b.write(group)
    if name == 'PhaseShift':
        r = cls.read(group)
        assert np.array_equal(r.centers, b.centers)
        assert np.array_equal(r.transform(pts), b.transform(pts))
        return
    other = np.random.default_rng(1)
    if not sync:                         # a file written by the reference
        group.attrs.pop('amd_philox_seed', None)
        for g in (group.items_.get('outer_bound'),):
            if g is not None:
                g.attrs.pop('amd_philox_seed', None)
    r = cls.read(group, rng=other)
    probe = _cloud(2000, 3, 9)
    assert np.array_equal(r.contains(probe), b.contains(probe))
    if name == 'NeuralBound':
        return
    # more than the stored queue holds, so fresh proposals are drawn
    assert (np.array_equal(b.sample(60000), r.sample(60000))) == sync
    if sync:
        assert b.log_v == r.log_v


def _flat(x):
    return -np.linalg.norm(x - 0.5) * 0.001


def _flat_blob(x):
    return -np.linalg.norm(x - 0.5) * 0.001, x[0]


@pytest.mark.parametrize('blobs,n_like_max,discard,n_networks,periodic', [
    (False, np.inf, False, 0, None), (True, 500, True, 0, None),
    (False, 500, True, 1, np.arange(1)), (True, np.inf, False, 1, None)])
def test_sampler_resume_is_exact(tmp_path, blobs, n_like_max, discard,
                                 n_networks, periodic):
    from nautilus_amd import Sampler
    path = str(tmp_path / 'run.hdf5')
    like = _flat_blob if blobs else _flat
    kw = dict(n_dim=2, n_live=100, n_networks=n_networks, periodic=periodic,
              filepath=path)
    a = Sampler(lambda u: u, like, resume=False, seed=0, **kw)
    a.run(f_live=0.45, n_eff=1000, n_like_max=n_like_max,
          discard_exploration=discard)
    b = Sampler(lambda u: u, like, resume=True, **kw)
    assert a.log_z == b.log_z and a.n_like == b.n_like
    a.run(f_live=0.45, n_eff=5000, discard_exploration=discard)
    b.run(f_live=0.45, n_eff=5000, discard_exploration=discard)
    for x, y in zip(a.posterior(return_blobs=blobs),
                    b.posterior(return_blobs=blobs)):
        assert np.array_equal(x, y)
    assert a.log_z == b.log_z
    with pytest.raises(ValueError):
        a.write(str(tmp_path / 'run.txt'))
    with pytest.raises(RuntimeError):
        a.write(path)
