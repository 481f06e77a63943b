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


# ---------------------------------------------------------------------------
# the layout against the reference's own writers (tests/golden/h5_layout.json
# and ref_checkpoint_*.pkl, written by the REFERENCE through the same shim:
# tests/golden/make_golden_h5.py)
# ---------------------------------------------------------------------------

def _ring(x):
    d = (x[0] - 0.97 + 0.5) % 1.0 - 0.5
    ll = -0.5 * (d / 0.05)**2 - 0.5 * np.sum(((x[1:] - 0.5) / 0.1)**2)
    return ll, x[0] + x[1], int(1000 * x[2])


def _bowl(x):
    return -0.5 * np.sum(((x - 0.5) / 0.15)**2)


_LAYOUT_CASES = {
    'periodic_blobs': (_ring, dict(n_dim=3, n_live=200, n_networks=1,
                                   periodic=np.arange(1), n_batch=50)),
    'plain': (_bowl, dict(n_dim=2, n_live=150, n_networks=0, n_batch=50))}


def _layout():
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, 'h5_layout.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('name', ['periodic_blobs', 'plain'])
def test_written_tree_equals_the_reference_layout(tmp_path, name):
    """Every group, dataset and attribute the reference's ``Sampler.write`` /
    ``write_shell_update`` / bound ``write`` + ``update`` / emulator ``write``
    emit for this problem exists here with the same kind, rank and
    ``maxshape`` -- and nothing else except the documented ``amd_*``
    entries (sampler.py:1253-1377, bounds/*.py, neural.py:118-146)."""
    from nautilus_amd import Sampler
    like, kw = _LAYOUT_CASES[name]
    path = str(tmp_path / 'run.hdf5')
    s = Sampler(lambda u: u, like, filepath=path, seed=3, **kw)
    s.run(n_eff=400, f_live=0.05, discard_exploration=False)
    with fake_h5py.File(path, 'r') as f:
        ours = fake_h5py.tree_schema(f)
    ref = _layout()['layout'][name]
    extra = sorted(k for k in ours if k not in ref)
    assert all('amd_' in k for k in extra), extra
    assert sorted(k for k in ref if k not in ours) == []
    for key, want in ref.items():
        assert ours[key] == want, (key, ours[key], want)


@pytest.mark.parametrize('name', ['periodic_blobs', 'plain'])
def test_resume_from_a_file_written_by_the_reference(tmp_path, name):
    """``io.read_sampler`` / ``read_bound`` / ``read_emulator`` on the
    reference's own checkpoint of a finished run (periodic parameter, two
    blobs, one network; and a plain one): the restored sampler reports the
    reference's evidence, counts and posterior, its bounds the reference's
    volumes and queues, and it keeps running and updating the file."""
    import os
    import shutil
    from conftest import GOLDEN
    from nautilus_amd import Sampler
    like, kw = _LAYOUT_CASES[name]
    ref = _layout()['reference'][name]
    path = str(tmp_path / 'ref.hdf5')
    shutil.copy(os.path.join(GOLDEN, 'ref_checkpoint_%s.pkl' % name), path)
    s = Sampler(lambda u: u, like, filepath=path, resume=True, **kw)
    assert s.explored == ref['explored'] and s.n_like == ref['n_like']
    assert len(s.bounds) == ref['n_bounds']
    assert np.array_equal(s.shell_n, ref['shell_n'])
    assert np.array_equal(s.shell_n_sample, ref['shell_n_sample'])
    assert abs(s.log_z - ref['log_z']) < 1e-12
    assert abs(s.n_eff - ref['n_eff']) < 1e-9 * ref['n_eff']
    assert (s.blobs is not None) == ref['blobs']
    for b, log_v, n_queue in zip(s.bounds, ref['bound_log_v'],
                                 ref['bound_queue']):
        assert abs(b.log_v - log_v) < 1e-12
        if hasattr(b, 'points'):
            assert len(b.points) == n_queue
    out = s.posterior(return_blobs=ref['blobs'])
    pts, log_w = out[0], out[1]
    assert len(pts) == ref['n_points']
    mean = np.average(pts, weights=np.exp(log_w), axis=0)
    assert np.allclose(mean, ref['posterior_mean'], rtol=0, atol=1e-12)
    # every restored bound answers like its stored points say: what the
    # reference kept in a shell lies inside that shell's bound
    for b, p in zip(s.bounds[1:], s.points[1:]):
        if len(p):
            assert np.all(b.contains(p))
    # ... and the run continues from there, updating the reference's file
    assert s.run(n_eff=2 * ref['n_eff'], f_live=0.05)
    assert s.n_eff >= 2 * ref['n_eff']
    assert abs(s.log_z - ref['log_z']) < 0.25
    with fake_h5py.File(path, 'r') as f:
        assert int(f['sampler'].attrs['n_like']) == s.n_like
