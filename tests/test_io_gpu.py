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


def test_periodic_bound_is_stored_in_the_reference_frame():
    """bounds/nautilus.py:239-243: the reference keeps the queued points of a
    NautilusBound in the SHIFTED frame and undoes the shift when it hands
    them out; the file must hold that frame, whatever the device queue
    holds, and a file written by the reference (no amd_* entries) must come
    back in the sampler frame."""
    import nautilus_amd.bounds as nb
    rng = np.random.default_rng(0)
    pts = np.random.default_rng(3).random((1500, 2))
    pts[:, 0] = (0.95 + 0.04 * np.random.default_rng(4).normal(size=1500)) % 1
    log_l = -((pts[:, 0] - 0.95 + 0.5) % 1 - 0.5)**2 - (pts[:, 1] - 0.5)**2
    b = nb.NautilusBound.compute(pts, log_l, np.median(log_l), np.log(0.5),
                                 n_networks=0, periodic=np.arange(1), rng=rng)
    assert b.shift is not None
    b.sample(10)
    queue = b.points
    assert len(queue) > 0
    group = fake_h5py.Group()
    b.write(group)
    stored = np.array(group['points'])
    assert np.array_equal(stored, b.shift.transform(queue))
    # the stored rows are inside the (shifted-frame) envelope, as in the
    # reference, where contains() of the outer bound sees shifted points
    assert np.all(b.outer_bound.contains(stored))
    # a reference file: no implementation-specific entries
    del group.items_['amd_points']
    group.attrs.pop('amd_philox_seed', None)
    r = nb.NautilusBound.read(group, rng=np.random.default_rng(1))
    assert np.allclose(r.points, queue, rtol=0, atol=1e-15)
    assert np.all(r.contains(r.points))


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
