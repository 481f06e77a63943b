"""Randomised parity sweep: unions of cube/ellipsoid mixtures of random shape
(1 <= D <= 128, 1 <= K <= 5, random cube masks, with and without the unit-cube
clip, with and without neural bounds and a phase shift) -- contains, overlap
counts, proposals, acceptance flags and shell association of the HIP kernels
against the oracle on the same Philox streams."""

import numpy as np
import pytest

from helpers import near_boundary, upload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _random_union(rng, d, k, unit):
    from oracle import bounds_oracle as bo
    members = []
    for _ in range(k):
        mask = rng.random(d) < rng.choice([0.0, 0.3, 0.7])
        if mask.all() and rng.random() < 0.7:
            mask[rng.integers(d)] = False
        ell = None
        de = int((~mask).sum())
        if de > 0:
            b_mat = (np.tril(rng.normal(size=(de, de))) * 0.04 / np.sqrt(de) +
                     np.eye(de) * rng.uniform(0.15, 0.45))
            ell = bo.OEllipsoid.from_params(rng.uniform(0.35, 0.65, de), b_mat)
        members.append(bo.OMixture.from_params(mask, ell))
    return bo.OUnion.from_members(members, unit=unit)


CASES = [(seed, d, k, unit) for seed, (d, k, unit) in enumerate([
    (1, 1, True), (2, 3, True), (3, 2, False), (5, 5, True), (8, 4, True),
    (15, 2, True), (16, 3, False), (17, 1, True), (31, 5, True),
    (32, 2, True), (33, 3, True), (48, 1, False), (50, 4, True),
    (63, 2, True), (64, 3, True), (65, 2, True), (90, 3, False),
    (100, 2, True), (127, 2, True), (128, 3, True), (120, 1, True),
    (128, 1, False), (72, 1, True)])]


@pytest.mark.parametrize('seed,d,k,unit', CASES)
def test_union_of_mixtures(seed, d, k, unit):
    from oracle import philox
    rng = np.random.default_rng(1000 + seed)
    u = _random_union(rng, d, k, unit)
    b = upload(u)
    # probe points: proposals of the union itself plus uniform noise
    x_o, keep_o, k_o = philox.union_propose(u, 7 + seed, 10**9 + seed, 6000)
    noise = rng.random((2000, d)) * 1.2 - 0.1
    probe = np.vstack([x_o, noise])
    counts = u.member_count(probe)
    got_counts = b.member_count(probe).cpu().numpy()
    # exclude points within rounding distance of a member's surface
    edge = np.zeros(len(probe), dtype=bool)
    for m in u.bounds:
        if m.ellipsoid is not None:
            y = m.ellipsoid.transform(probe[:, ~m.dim_cube])
            edge |= near_boundary(np.sum(y**2, axis=1), 1.0, 1e-9)
    edge |= np.any(near_boundary(probe, 0.0, 1e-12) |
                   near_boundary(probe, 1.0, 1e-12), axis=1)
    assert np.array_equal(got_counts[~edge], counts[~edge])
    assert np.array_equal(b.contains(probe).cpu().numpy()[~edge],
                          u.contains(probe)[~edge])
    # proposals and acceptance flags
    x = b.propose(7 + seed, 10**9 + seed, 6000)
    assert np.allclose(x.cpu().numpy(), x_o, rtol=0, atol=1e-11)
    flags = b.accept(7 + seed, 10**9 + seed, x).cpu().numpy()
    e6 = edge[:6000]
    assert np.array_equal((flags & 1).astype(bool)[~e6], keep_o[~e6])


@pytest.mark.parametrize('d,e,periodic', [(6, 1, False), (20, 2, True),
                                          (40, 4, False), (70, 1, True)])
def test_nested_nautilus_bounds(d, e, periodic):
    """Lists of NautilusBounds (shell exclusion / association) with random
    networks, optional phase shift."""
    from nautilus_amd import device
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    rng = np.random.default_rng(d)
    obs = []
    for level in range(3):
        width = 0.3 / (1 + level)
        ell = bo.OEllipsoid.from_params(
            np.full(d, 0.5) + rng.normal(size=d) * 0.01,
            np.eye(d) * width + np.tril(rng.normal(size=(d, d))) * 0.01)
        outer = bo.OUnion.from_members(
            [bo.OMixture.from_params(np.zeros(d, bool), ell)], unit=True)
        nb = bo.ONeural()
        nb.outer_bound, nb.n_dim = ell, d
        nb.emulator = mo.Emulator.from_weights(
            rng.normal(size=d) * 0.05, rng.uniform(0.7, 1.3, d),
            [mo.glorot_init(d, 10 * level + i)[:2] for i in range(e)])
        probe = 0.5 + rng.normal(size=(2000, d)) * width * 0.5
        nb.score_predict_min = float(np.median(
            nb.emulator.predict(ell.transform(probe))))
        shift = None
        if periodic:
            shift = bo.OPhaseShift.from_params(
                np.arange(2), rng.uniform(0.3, 0.7, 2))
        obs.append(bo.ONautilus.from_parts(outer, [nb], shift=shift))
    # points spread over the three nested ellipsoids (and beyond)
    scale = rng.choice([0.3, 0.6, 1.2], size=6000) / np.sqrt(d)
    which = rng.integers(0, 3, 6000)
    x = np.empty((6000, d))
    for i, o in enumerate(obs):
        e_i = o.neural_bounds[0].outer_bound
        sel = which == i
        x[sel] = e_i.c + (rng.normal(size=(sel.sum(), d)) @ e_i.B.T) * \
            scale[sel, None]
    x = np.clip(x, 0, 1 - 1e-9)
    if periodic:            # undo the (last) shift so that points land inside
        x = obs[-1].shift.transform(x, inverse=True)
    inside = np.array([o.contains(x) for o in obs])
    edge = np.zeros(len(x), dtype=bool)
    for o in obs:
        xs = x if o.shift is None else o.shift.transform(x)
        nbd = o.neural_bounds[0]
        y = nbd.outer_bound.transform(xs)
        edge |= near_boundary(np.sum(y**2, axis=1), 1.0, 1e-9)
        edge |= near_boundary(nbd.emulator.predict(y),
                              nbd.score_predict_min - 1e-9, 1e-9)
    devs = [upload(o) for o in obs]
    for o, b, want in zip(obs, devs, inside):
        assert np.array_equal(b.contains(x).cpu().numpy()[~edge], want[~edge])
    lst = device.DeviceBoundList(devs)
    assert np.array_equal(lst.contains_any(x).cpu().numpy()[~edge],
                          inside.any(axis=0)[~edge])
    first = lst.first_containing(x).cpu().numpy()
    want = np.full(len(x), -1)
    for i in (2, 1, 0):
        want[inside[i]] = i
    assert np.array_equal(first[~edge], want[~edge])
    assert 0.02 < inside.mean() < 0.98


@pytest.mark.parametrize('d,e', [(12, 2), (33, 1), (40, 4), (48, 2), (50, 4),
                                 (64, 1), (70, 2), (100, 1)])
def test_nautilus_sample_dims(d, e):
    """NautilusBound.sample (nautilus.py:199-224) in every size class of the
    proposal path of the evaluation kernel (two / one tile per wavefront, the
    pre-issued copies up to n_dim = 48): accepted points and counters against
    the oracle, plus the emulator scores of the proposals."""
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    from oracle import philox
    rng = np.random.default_rng(77 * d + e)
    ell = bo.OEllipsoid.from_params(
        np.full(d, 0.5) + rng.normal(size=d) * 0.01,
        np.eye(d) * 0.2 + np.tril(rng.normal(size=(d, d))) * 0.01)
    outer = bo.OUnion.from_members(
        [bo.OMixture.from_params(np.zeros(d, bool), ell)], unit=True)
    nb = bo.ONeural()
    nb.outer_bound, nb.n_dim = ell, d
    nb.emulator = mo.Emulator.from_weights(
        rng.normal(size=d) * 0.05, rng.uniform(0.7, 1.3, d),
        [mo.glorot_init(d, 3 * d + i)[:2] for i in range(e)])
    seed, offset, n = 11 + d, 10**10 + d, 4000
    x_all, _, _ = philox.union_sample(outer, seed, offset, n)
    score_all = nb.emulator.predict(ell.transform(x_all))
    nb.score_predict_min = float(np.median(score_all))
    ob = bo.ONautilus.from_parts(outer, [nb])
    b = upload(ob)
    pts, counts = b.sample_launch(seed, offset, n)
    c = counts.cpu().numpy()
    pts_o, cnt_o = philox.nautilus_sample(ob, seed, offset, n)
    n_edge = int(near_boundary(score_all, nb.score_predict_min - 1e-9,
                               1e-9).sum())
    assert int(c[0]) == int(cnt_o[2])                # proposals in the cube
    assert abs(int(c[1]) - len(pts_o)) <= n_edge
    assert 0.2 * len(x_all) < len(pts_o) < 0.8 * len(x_all)
    if int(c[1]) == len(pts_o):
        assert np.allclose(pts[:c[1]].cpu().numpy(), pts_o, rtol=0,
                           atol=1e-11)
    _, score = b.neural_score(x_all)
    assert np.allclose(score.cpu().numpy(), score_all, rtol=0, atol=1e-10)
