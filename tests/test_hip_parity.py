"""Parity of the HIP path (through the C ABI) with the CPU oracle and the
reference's golden vectors.  GPU only.

Tolerances: index / counter / mask work is bit-exact except for points whose
decision value lies within 1e-11 of the decision threshold (fp64 sums are
associated differently on the matrix cores than in numpy/BLAS); floating
point outputs agree to 1e-11 relative."""

import os

import numpy as np
import torch
import pytest

from conftest import load_golden
from helpers import (upload, near_boundary, nautilus_from_golden,
                     khachiyan_weights_numpy)

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.fixture(scope='module')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nautilus_amd import device
    return device


def _ell_from_golden(g):
    from oracle import bounds_oracle as bo
    return bo.OEllipsoid.from_params(g['c'], g['B'], g['B_inv'], g['A'])


def test_philox_matches_oracle(dev):
    from oracle import philox
    for seed, off, block, tag in [(0, 0, 0, 0), (2**40 + 17, 2**33 + 5, 3, 1),
                                  (12345, 999, 7, 2)]:
        u = dev.philox_uniform(seed, off, block, tag, 4096).cpu().numpy()
        g = np.uint64(off) + np.arange(4096, dtype=np.uint64)
        u0, u1 = philox.uniform_pair(seed, g, block, tag)
        assert np.array_equal(u[:, 0], u0) and np.array_equal(u[:, 1], u1)


@pytest.mark.parametrize('d', [3, 20, 50])
def test_ellipsoid_contains_golden(dev, d):
    g = load_golden('ellipsoid_D%d' % d)
    ell = _ell_from_golden(g)
    b = upload(ell)
    for fn in (b.contains, b.contains_stream):
        mask = fn(g['test']).cpu().numpy()
        diff = mask != g['contains']
        assert not np.any(diff & ~near_boundary(g['r2'], 1.0, TOL))
    assert 0 < g['contains'].sum() < len(g['contains'])
    # the ellipsoid-frame radius through the neural-bound path
    from nautilus_amd import device
    nb = device.DeviceBound(d, [], None, False, [dict(
        ellipsoid=device.member(g['c'], g['B'], g['B_inv']))])
    r2, _ = nb.neural_score(g['test'])
    assert np.allclose(r2.cpu().numpy(), g['r2'], rtol=TOL, atol=0)


def test_ellipsoid_stream_large(dev):
    """Full-size streaming run: 2^22 points at D = 50 agree with the
    matrix-core path and with the oracle on a subsample."""
    import torch
    g = load_golden('ellipsoid_D50')
    ell = _ell_from_golden(g)
    b = upload(ell)
    n = 1 << 22
    gen = torch.Generator(device='cuda').manual_seed(1)
    x = torch.rand((n, 50), dtype=torch.float64, device='cuda',
                   generator=gen)
    x[::3] = torch.from_numpy(ell.c).cuda() + 0.12 * (x[::3] - 0.5)
    m1 = b.contains_stream(x)
    m2 = b.contains(x)
    assert int((m1 != m2).sum()) <= 2
    sub = x[:20000].cpu().numpy()
    r2 = np.sum(ell.transform(sub)**2, axis=-1)
    bad = (m1[:20000].cpu().numpy() != (r2 < 1)) & ~near_boundary(r2, 1, TOL)
    assert not np.any(bad)
    assert 0.05 < float(m1.double().mean()) < 0.95
    # ragged sizes
    for k in (1, 63, 64, 65, 1000):
        assert torch.equal(b.contains_stream(x[:k]), m1[:k])


@pytest.mark.parametrize('d', [2, 16, 18, 20, 36, 52, 56, 64, 66, 68, 80, 84,
                               90, 96, 100, 112, 116, 128])
def test_ellipsoid_stream_every_variant(dev, d):
    """Even n_dim through every instantiation of the streaming kernels: the
    plain kernel (n_dim <= 16), the pipelined one (<= 64), two and one tile
    per wavefront with the operands read ahead beyond that, each with a full
    last row tile, a last row tile of at most four rows (4x4x4 matrix
    instructions: 20, 36, 52, 68, 84, 100, 116) and a trimmed K range --
    against the oracle (basic.py:340, 360), ragged sizes included."""
    import torch
    from oracle import bounds_oracle as bo
    rng = np.random.default_rng(1000 + d)
    b_mat = np.tril(rng.normal(size=(d, d)) * 0.05) + np.eye(d) * 0.5
    ell = bo.OEllipsoid.from_params(np.full(d, 0.5), b_mat)
    b = upload(ell)
    n = 40003
    x = 0.5 + (rng.normal(size=(n, d)) @ b_mat.T) / np.sqrt(d + 2.0)
    r2 = np.sum(ell.transform(x)**2, axis=-1)
    want = r2 < 1
    xt = torch.from_numpy(x).cuda()
    full = b.contains_stream(xt).cpu().numpy()
    edge = near_boundary(r2, 1.0, TOL)
    assert np.array_equal(full[~edge], want[~edge])
    assert 0.05 < want.mean() < 0.98
    for k in (1, 17, 32, 33, 64, 65, 4097, 39999):
        part = b.contains_stream(xt[:k]).cpu().numpy()
        assert np.array_equal(part, full[:k]), k


@pytest.mark.parametrize('d', [1, 3, 5, 7, 33, 49, 63, 65, 67, 99, 115, 127])
def test_ellipsoid_stream_odd_dims(dev, d):
    """Odd n_dim (rows only 8-byte aligned): 16-byte pair loads from
    8-byte-aligned addresses (n_dim >= 3; the pair with a row's last feature
    is read one element earlier) and the one-element path of n_dim = 1
    against the oracle, for sizes around the point groups of a wavefront and
    the end of the array."""
    import torch
    from oracle import bounds_oracle as bo
    rng = np.random.default_rng(d)
    b_mat = np.tril(rng.normal(size=(d, d)) * 0.05) + np.eye(d) * 0.5
    ell = bo.OEllipsoid.from_params(np.full(d, 0.5), b_mat)
    b = upload(ell)
    n = 70001
    x = 0.5 + (rng.normal(size=(n, d)) @ b_mat.T) / np.sqrt(d + 2.0)
    r2 = np.sum(ell.transform(x)**2, axis=-1)
    want = r2 < 1
    xt = torch.from_numpy(x).cuda()
    full = b.contains_stream(xt).cpu().numpy()
    edge = near_boundary(r2, 1.0, TOL)
    assert np.array_equal(full[~edge], want[~edge])
    assert 0.05 < want.mean() < 0.98
    for k in (1, 31, 32, 33, 63, 64, 65, 127, 4096, 4097, 69999):
        part = b.contains_stream(xt[:k]).cpu().numpy()
        assert np.array_equal(part, full[:k]), k


def test_mixture_contains_golden(dev):
    from oracle import bounds_oracle as bo
    g = load_golden('mixture_D6')
    ell = bo.OEllipsoid.from_params(g['c'], g['B'], g['B_inv'], g['A'])
    mix = bo.OMixture.from_params(g['dim_cube'], ell)
    mask = upload(mix).contains(g['test']).cpu().numpy()
    assert np.array_equal(mask, g['contains'])
    assert np.array_equal(mix.contains(g['test']), g['contains'])


def _union_from_golden(g, mixture):
    from helpers import union_from_golden
    return union_from_golden(g, mixture)


@pytest.mark.parametrize('name,mixture', [('union_K2_D3', False),
                                          ('union_K4_D8', True),
                                          ('union_K4_D50', True)])
def test_union_golden(dev, name, mixture):
    g = load_golden(name)
    u = _union_from_golden(g, mixture)
    b = upload(u)
    assert np.array_equal(b.member_count(g['test']).cpu().numpy(),
                          g['counts'])
    assert np.array_equal(b.contains(g['test']).cpu().numpy(), g['contains'])
    # the reference's own accepted samples lie inside (union.py:291-327)
    assert bool(b.contains(g['sample']).all())


@pytest.mark.parametrize('name,mixture', [('union_K2_D3', False),
                                          ('union_K4_D8', True),
                                          ('union_K4_D50', True)])
def test_union_proposals_match_oracle(dev, name, mixture):
    """Device Union.sample == oracle Philox tier, proposal by proposal."""
    from oracle import philox
    g = load_golden(name)
    u = _union_from_golden(g, mixture)
    b = upload(u)
    seed, offset, n = 77, 123456789, 20000
    x = b.propose(seed, offset, n)
    flags = b.accept(seed, offset, x).cpu().numpy()
    x_o, keep_o, k_o = philox.union_propose(u, seed, offset, n)
    assert np.allclose(x.cpu().numpy(), x_o, rtol=0, atol=1e-12)
    assert np.array_equal((flags & 1).astype(bool), keep_o)
    assert 0 < keep_o.sum() < n
    # compaction keeps proposal order; counters are exact
    pts, counts, src = dev.compact_rows(x, __import__('torch').from_numpy(
        flags).cuda(), 1, want_index=True)
    c = counts.cpu().numpy()
    assert c[0] == c[1] == keep_o.sum()
    assert np.array_equal(src[:c[1]].cpu().numpy(), np.flatnonzero(keep_o))
    assert np.allclose(pts[:c[1]].cpu().numpy(), x_o[keep_o], rtol=0,
                       atol=1e-12)
    # statistical: survivors are uniform over the union -> volume estimate
    log_v = np.logaddexp.reduce(u.log_v_all) + np.log(keep_o.mean())
    assert abs(log_v - float(g['log_v'])) < 0.1


def test_unit_cube(dev):
    from oracle import bounds_oracle as bo
    from oracle import philox
    cube = bo.OCube(5)
    b = upload(cube)
    x = b.propose(3, 0, 5000)
    xo, keep, _ = philox.union_propose(
        bo.OUnion.from_members([bo.OMixture.from_params(np.ones(5, bool),
                                                        None)]), 3, 0, 5000)
    assert np.array_equal(x.cpu().numpy(), xo)
    assert bool(b.contains(x).all())
    pts = np.array([[0.5] * 5, [1.0] + [0.5] * 4, [-1e-9] + [0.5] * 4,
                    [0.0] * 5])
    assert b.contains(pts).cpu().numpy().tolist() == [True, False, False,
                                                     True]


@pytest.fixture(scope='module')
def neural_d4():
    from helpers import neural_from_golden
    g = load_golden('neuralbound_D4')
    return g, neural_from_golden(g, 'nb_')


def test_neural_bound_golden(dev, neural_d4):
    g, nb = neural_d4
    b = upload(nb)
    r2, score = b.neural_score(g['test'])
    r2_o = np.sum(nb.outer_bound.transform(g['test'])**2, axis=-1)
    assert np.allclose(r2.cpu().numpy(), r2_o, rtol=TOL)
    assert np.allclose(score.cpu().numpy(), g['score'], rtol=0, atol=1e-10)
    mask = b.contains(g['test']).cpu().numpy()
    edge = near_boundary(g['score'], nb.score_predict_min - 1e-9, 1e-10) | \
        near_boundary(r2_o, 1.0, TOL)
    assert not np.any((mask != g['contains']) & ~edge)
    assert 0 < g['contains'].sum() < len(g['contains'])


@pytest.mark.parametrize('d,e', [(5, 1), (20, 2), (50, 4)])
def test_emulator_predict_golden(dev, d, e):
    """NeuralNetworkEmulator.predict (neural.py:100-116) on the matrix cores
    against sklearn's own predictions."""
    from nautilus_amd import device
    g = load_golden('emulator_D%d_E%d' % (d, e))
    nets = [([g['coef_%d_%d' % (i, k)] for k in range(4)],
             [g['intercept_%d_%d' % (i, k)] for k in range(4)])
            for i in range(e)]
    ident = device.member(np.zeros(d), np.eye(d), np.eye(d))
    b = device.DeviceBound(d, [], None, False, [dict(
        ellipsoid=ident, score_predict_min=0.0,
        mlp=dict(mean=g['mean'], scale=g['scale'], nets=nets))])
    _, score = b.neural_score(g['test'])
    assert np.allclose(score.cpu().numpy(), g['predict'], rtol=0, atol=1e-11)


@pytest.mark.parametrize('d,e', [(49, 3), (64, 2), (65, 2), (79, 1), (80, 2),
                                 (96, 1), (100, 8), (127, 2), (128, 1)])
def test_neural_bound_large_dims(dev, d, e):
    """Every kernel variant of the emulator evaluation (one / two tiles per
    wavefront, layer 1 streamed in one or two K chunks): NeuralBound.contains,
    the score and shell exclusion against the oracle with random networks."""
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    rng = np.random.default_rng(1000 * d + e)
    nets = [mo.glorot_init(d, i)[:2] for i in range(e)]
    mean, scale = rng.normal(size=d) * 0.1, rng.uniform(0.5, 1.5, d)
    emu = mo.Emulator.from_weights(mean, scale, nets)
    b_mat = np.tril(rng.normal(size=(d, d)) * 0.02) + np.eye(d) * 0.4
    ell = bo.OEllipsoid.from_params(np.full(d, 0.5), b_mat)
    nb = bo.ONeural()
    nb.outer_bound, nb.n_dim, nb.emulator = ell, d, emu
    x = 0.5 + (rng.normal(size=(3000, d)) @ b_mat.T) * (0.6 / np.sqrt(d))
    score_o = emu.predict(ell.transform(x))
    nb.score_predict_min = float(np.median(score_o))
    b = upload(nb)
    r2, score = b.neural_score(x)
    assert np.allclose(score.cpu().numpy(), score_o, rtol=0, atol=1e-10)
    want = nb.contains(x)
    got = b.contains(x).cpu().numpy()
    edge = near_boundary(score_o, nb.score_predict_min - 1e-9, 1e-9)
    assert np.array_equal(got[~edge], want[~edge])
    assert 0.2 < want.mean() < 0.8
    lst = dev.DeviceBoundList([b, upload(ell)])
    any_o = want | ell.contains(x)
    assert np.array_equal(lst.contains_any(x).cpu().numpy()[~edge],
                          any_o[~edge])


@pytest.mark.parametrize('d,e,k_outer', [
    (1, 1, 1), (3, 2, 1), (15, 1, 1), (16, 2, 1), (17, 1, 0), (31, 2, 1),
    (32, 1, 1), (33, 3, 1), (47, 1, 1), (48, 2, 0), (49, 1, 1), (50, 4, 1),
    (62, 2, 1), (63, 1, 1), (64, 2, 1), (65, 1, 1), (79, 2, 1), (80, 1, 0),
    (81, 2, 1), (96, 1, 1), (100, 8, 1), (112, 2, 1), (127, 1, 1),
    (128, 2, 1)])
def test_pipelined_accept_and_score(dev, d, e, k_outer):
    """nb_accept / nb_neural_score of a bound with ONE neural bound and at
    most one outer member run through the pipelined kernel (nb_eval_fast.hip)
    for every n_dim <= 128 -- every (DT, KT1) instantiation, n_dim = 16 DT
    included, where layer 1 needs one more k-tile, layer 1 in one stage and
    in two K chunks (n_dim >= 80).  Launch sizes: below one pass, ragged
    tails, and more 128-point passes than workgroups (the points of the next
    pass are prefetched during the last stage).  Oracle: union.py:313-319 +
    neural.py:115-126 on the same Philox stream."""
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    from oracle import philox
    rng = np.random.default_rng(77 * d + e)
    nets = [mo.glorot_init(d, i)[:2] for i in range(e)]
    mean, scale = rng.normal(size=d) * 0.1, rng.uniform(0.5, 1.5, d)
    emu = mo.Emulator.from_weights(mean, scale, nets)
    b_mat = np.tril(rng.normal(size=(d, d)) * 0.02) + np.eye(d) * 0.4
    centre = np.full(d, 0.5)
    centre[0] = 0.8                         # part of the ellipsoid leaves the cube
    ell = bo.OEllipsoid.from_params(centre, b_mat)
    nb = bo.ONeural()
    nb.outer_bound, nb.n_dim, nb.emulator = ell, d, emu
    outer = None
    if k_outer:
        outer = bo.OUnion.from_members(
            [bo.OEllipsoid.from_params(centre, 1.05 * b_mat)], unit=True)
    probe = centre + (rng.normal(size=(4000, d)) @ b_mat.T) * (
        0.7 / np.sqrt(d))
    nb.score_predict_min = float(np.median(emu.predict(ell.transform(probe))))
    if k_outer:
        b = upload(bo.ONautilus.from_parts(outer, [nb]))
    else:       # unit cube as the only outer bound
        from helpers import neural_from_oracle
        b = dev.DeviceBound(d, [], None, True, [neural_from_oracle(nb)])
    seed, offset = 11 + d, 10**11 + 3
    for n in (1, 127, 128, 129, 5000, 40000):
        if k_outer:
            x, _, _ = philox.union_propose(outer, seed, offset, n)
            assert np.allclose(b.propose(seed, offset, n).cpu().numpy(), x,
                               rtol=0, atol=1e-12)
        else:
            x = centre + (rng.normal(size=(n, d)) @ b_mat.T) * (
                0.9 / np.sqrt(d))
        xd = torch.as_tensor(x, device='cuda')
        y = ell.transform(x)
        r2_o, score_o = np.sum(y**2, axis=1), emu.predict(y)
        r2, score = b.neural_score(xd)
        assert np.allclose(r2.cpu().numpy(), r2_o, rtol=1e-12, atol=1e-13)
        assert np.allclose(score.cpu().numpy(), score_o, rtol=0, atol=1e-10)
        flags = b.accept(seed, offset, xd).cpu().numpy()
        g = np.uint64(offset) + np.arange(n, dtype=np.uint64)
        _, u_acc = philox.uniform_pair(seed, g, 0, philox.TAG_CTRL)
        in_cube = np.all((x >= 0) & (x < 1), axis=1)
        keep = in_cube & (u_acc > 0 if k_outer else True)
        inside = (r2_o < 1) & (score_o > nb.score_predict_min - 1e-9)
        edge = (near_boundary(score_o, nb.score_predict_min - 1e-9, 1e-9) |
                near_boundary(r2_o, 1.0, 1e-12))
        assert np.array_equal(flags & 1, keep.astype(np.uint8))
        want = (keep & inside).astype(np.uint8)
        assert np.array_equal((flags >> 1)[~edge], want[~edge])
        if n >= 5000 and d <= 64:         # (the test data decides both ways)
            assert 0 < want.mean() < 1


# (n_dim mod 16 in 1..4 -- 20, 33, 50, 67, 100: the last row tile of every
# ellipsoid test runs on the 4-row matrix instruction, nb_cand.hip StepTable)
@pytest.mark.parametrize('d,k,m', [(20, 3, 2), (50, 2, 3), (33, 2, 2),
                                   (67, 2, 2), (100, 2, 2), (53, 2, 2)])
def test_two_stage_large_launch(dev, d, k, m):
    """Bounds with several outer members and several neural bounds go through
    the two device-side stages (nb_cand.hip: geometric tests + candidate
    lists, then ONE batched emulator launch of nb_eval_fast.hip).  The launch
    size decides how the points are dealt out over the wavefronts and how the
    candidate lists are laid out: the flags of the same proposals must not
    depend on it, and both agree with the oracle (union.py:305-327,
    nautilus.py:146-169 on the same Philox stream)."""
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    from oracle import philox
    rng = np.random.default_rng(5 * d + k)
    centres = 0.5 + 0.04 * rng.normal(size=(max(k, m), d))
    members, neural = [], []
    for j in range(max(k, m)):
        b_mat = np.tril(rng.normal(size=(d, d)) * 0.01) + np.eye(d) * 0.2
        ell = bo.OEllipsoid.from_params(centres[j], b_mat)
        if j < k:
            members.append(bo.OEllipsoid.from_params(centres[j],
                                                     1.05 * b_mat))
        if j < m:
            nb = bo.ONeural()
            nb.outer_bound, nb.n_dim = ell, d
            nb.emulator = mo.Emulator.from_weights(
                rng.normal(size=d) * 0.1, rng.uniform(0.5, 1.5, d),
                [mo.glorot_init(d, 3 * j + i)[:2] for i in range(2)])
            probe = centres[j] + (rng.normal(size=(2000, d)) @ b_mat.T) * (
                0.7 / np.sqrt(d))
            nb.score_predict_min = float(np.median(
                nb.emulator.predict(ell.transform(probe))))
            neural.append(nb)
    outer = bo.OUnion.from_members(members, unit=True)
    ob = bo.ONautilus.from_parts(outer, neural)
    b = upload(ob)
    seed, offset, n = 31 + d, 10**10 + 7, 150000
    xd = b.propose(seed, offset, n)
    big = b.accept(seed, offset, xd).cpu().numpy()
    inside_big = b.contains(xd).cpu().numpy()
    # (both outcomes occur: 0.4 % of the proposals at n_dim 100, 600 rows)
    assert 0.002 < (big >> 1).mean() < 0.99
    # the same proposals in launches below the residency threshold
    for lo, hi in ((0, 4000), (70001, 90000), (n - 3000, n)):
        part = b.accept(seed, offset + lo, xd[lo:hi].contiguous())
        assert np.array_equal(part.cpu().numpy(), big[lo:hi])
        assert np.array_equal(
            b.contains(xd[lo:hi].contiguous()).cpu().numpy(),
            inside_big[lo:hi])
    # ... and the oracle on a slice
    lo, hi = 20000, 32000
    x, keep, _ = philox.union_propose(outer, seed, offset + lo, hi - lo)
    assert np.allclose(xd[lo:hi].cpu().numpy(), x, rtol=0, atol=1e-12)
    assert np.array_equal(big[lo:hi] & 1, keep.astype(np.uint8))
    score_edge = np.zeros(hi - lo, dtype=bool)
    for nb in neural:
        y = nb.outer_bound.transform(x)
        score_edge |= near_boundary(nb.emulator.predict(y),
                                    nb.score_predict_min, 1e-9)
        score_edge |= near_boundary(np.sum(y**2, axis=1), 1.0, 1e-12)
    want = keep & ob.neural_contains(x)
    assert np.array_equal((big[lo:hi] >> 1)[~score_edge],
                          want.astype(np.uint8)[~score_edge])
    want_in = ob.contains(x)
    assert np.array_equal(inside_big[lo:hi][~score_edge],
                          want_in[~score_edge])


def test_accept_routes_agree(dev):
    """A bound with one neural bound and at most one outer member may take
    the fused acceptance kernel or the staged route (chosen once per bound
    from its first launch).  Same arithmetic in the same order: the flags of
    the same proposals are identical, bit for bit, on both."""
    from helpers import nautilus_from_golden
    g = load_golden('nautilusbound_D4')
    b = upload(nautilus_from_golden(g))
    assert b.n_neural == 1 and b.n_members <= 1 and b.n_networks >= 1
    seed, offset, n = 77, 123456789, 200000
    x = b.propose(seed, offset, n)
    b.dense_need = 1.0                       # fused kernel
    fused = b.accept(seed, offset, x).cpu().numpy()
    b.dense_need = 0.0                       # staged route
    staged = b.accept(seed, offset, x).cpu().numpy()
    assert 0.001 < (fused >> 1).mean() < 0.999
    assert np.array_equal(fused, staged)
    b.dense_need = None                      # first launch: staged + probe
    first = b.accept(seed, offset, x).cpu().numpy()
    assert np.array_equal(first, fused)
    assert 0.0 <= b.dense_need <= 1.0


def test_list_eval_against_the_oracle(dev):
    """nb_list_eval (geometric stage + candidate lists + one batched emulator
    launch) on a list of SEVEN nested NautilusBounds with emulators against
    the oracle's ``contains`` of every bound: exclusion = any bound
    (sampler.py:797-798), association = first bound of the list
    (sampler.py:1213-1219) -- at once, through the slab path (small work
    space) and through the group-slice path (two groups per slice)."""
    from nautilus_amd import device
    g = load_golden('nautilusbound_D4')
    obs = []
    for k in range(7):
        ob = nautilus_from_golden(g)
        for n_b in ob.neural_bounds:
            # nested: the same bound with rising emulator thresholds
            n_b.score_predict_min = n_b.score_predict_min + 0.03 * k
        obs.append(ob)
    devs = [upload(ob) for ob in obs]
    x = devs[0].propose(5, 0, 60000)
    xh = x.cpu().numpy()
    inside = np.array([ob.contains(xh) for ob in obs])
    # rows whose decision lies within rounding of an edge (fp64 sums are
    # associated differently on the matrix cores)
    edge = np.zeros(len(xh), dtype=bool)
    for ob in obs:
        for n_b in ob.neural_bounds:
            y = n_b.outer_bound.transform(xh)
            edge |= near_boundary(np.sum(y**2, axis=1), 1.0, 1e-12)
            edge |= near_boundary(n_b.emulator.predict(y),
                                  n_b.score_predict_min, 1e-9)
    assert edge.mean() < 1e-3
    # the list in sampler order and in reverse (the innermost bound first)
    for order in (list(range(7)), list(range(6, -1, -1)), [3, 0, 6, 1]):
        lst = device.DeviceBoundList([devs[i] for i in order])
        want_any = inside[order].any(axis=0)
        want_first = np.full(len(xh), -1)
        for pos in range(len(order) - 1, -1, -1):
            want_first[inside[order[pos]]] = pos
        assert 0.01 < want_any.mean() < 0.99
        if order[0] != 0:      # (behind the outermost bound every row is 0)
            assert len(np.unique(want_first)) >= 3

        def check():
            got_any = lst.contains_any(x).cpu().numpy()
            got_first = lst.first_containing(x).cpu().numpy()
            assert np.array_equal(got_any[~edge], want_any[~edge])
            assert np.array_equal(got_first[~edge], want_first[~edge])
        check()
        old = device.WORK_BYTES
        device.WORK_BYTES = 1 << 20
        try:
            check()
        finally:
            device.WORK_BYTES = old
        os.environ['NB_LIST_SLICE_GROUPS'] = '2'
        try:
            check()
        finally:
            del os.environ['NB_LIST_SLICE_GROUPS']


@pytest.mark.parametrize('d,n_bounds', [(20, 70), (50, 9)])
def test_long_list_over_few_and_many_rows(dev, d, n_bounds):
    """A long list over FEW rows (the shell association / exclusion of a run
    at the reference's batch size: hundreds of bounds, a few hundred rows)
    deals its bounds out over block rows of the candidate kernel, whose
    results meet in the status bytes (OR) and first-bound words (MIN); over
    many rows one block row walks the list.  Both against the oracle's
    ``contains`` of every bound (sampler.py:797-798, 1213-1219), on bounds
    scattered over the cube (most (row, bound) pairs end at the bounding
    sphere), every fifth with periodic dimensions recentred, two without
    neural bounds (decided in the first stage: the first-bound MIN must beat
    later bounds' emulators); odd slab offsets (the byte-wise OR) and the
    group-slice path too."""
    from nautilus_amd import device
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    rng = np.random.default_rng(17 * d + n_bounds)
    obs, probes = [], []
    for j in range(n_bounds):
        centre = rng.uniform(0.15, 0.85, size=d)
        b_mat = np.tril(rng.normal(size=(d, d)) * 0.004) + np.eye(d) * 0.09
        shift = None
        if j % 5 == 2:
            shift = bo.OPhaseShift.from_params(
                np.array([0, d - 1]), rng.uniform(0.0, 1.0, size=2))
        ell = bo.OEllipsoid.from_params(centre, b_mat)
        outer = bo.OUnion.from_members(
            [bo.OEllipsoid.from_params(centre, 1.05 * b_mat)], unit=True)
        neural = []
        if j not in (3, n_bounds - 2):
            nb = bo.ONeural()
            nb.outer_bound, nb.n_dim = ell, d
            nb.emulator = mo.Emulator.from_weights(
                rng.normal(size=d) * 0.1, rng.uniform(0.5, 1.5, d),
                [mo.glorot_init(d, 2 * j + i)[:2] for i in range(2)])
            probe = centre + (rng.normal(size=(500, d)) @ b_mat.T) * (
                0.7 / np.sqrt(d))
            nb.score_predict_min = float(np.quantile(
                nb.emulator.predict(ell.transform(probe)), 0.3))
            neural.append(nb)
        ob = bo.ONautilus.from_parts(outer, neural, shift=shift)
        obs.append(ob)
        # points around the bound, in the frame the sampler sees
        pts = centre + (rng.normal(size=(600, d)) @ b_mat.T) * (
            0.8 / np.sqrt(d))
        if shift is not None:
            pts = shift.transform(pts % 1.0, inverse=True)
        probes.append(pts)
    # the sampler's association list ends with the unit cube (bound 0 of a
    # run, sampler.py:1002): it holds every row and is decided at once, in
    # the block row of the LAST bounds -- the first bound must still win
    obs.append(bo.OCube(d))
    x = np.vstack(probes + [rng.random((3000, d))])
    x = x[rng.permutation(len(x))]
    xd = torch.as_tensor(x, device='cuda')
    inside = np.array([ob.contains(x) for ob in obs])
    edge = np.zeros(len(x), dtype=bool)
    for ob in obs[:-1]:
        xs = x if ob.shift is None else ob.shift.transform(x)
        for n_b in ob.neural_bounds:
            y = n_b.outer_bound.transform(xs)
            edge |= near_boundary(np.sum(y**2, axis=1), 1.0, 1e-12)
            edge |= near_boundary(n_b.emulator.predict(y),
                                  n_b.score_predict_min, 1e-9)
        for m in ob.outer_bound.bounds:
            edge |= near_boundary(np.sum(m.transform(xs)**2, axis=1), 1.0,
                                  1e-12)
    assert edge.mean() < 1e-3
    assert 0.1 < inside[:-1].any(axis=0).mean() < 0.9
    assert inside[:-1].sum(axis=0).max() <= 3     # scattered, not nested
    assert inside[-1].all()
    lst = device.DeviceBoundList([upload(ob) for ob in obs])
    want_any = inside.any(axis=0)
    want_first = np.where(want_any, np.argmax(inside, axis=0), -1)

    def check():
        # all rows (one block row per row range), 701 rows from an odd offset
        # (block rows over the bounds), 33 rows
        for lo, hi in ((0, len(x)), (1203, 1904), (5, 38)):
            part = xd[lo:hi].contiguous()
            keep = ~edge[lo:hi]
            got_any = lst.contains_any(part).cpu().numpy()
            got_first = lst.first_containing(part).cpu().numpy()
            assert np.array_equal(got_any[keep], want_any[lo:hi][keep])
            assert np.array_equal(got_first[keep], want_first[lo:hi][keep])
    check()
    old = device.WORK_BYTES
    device.WORK_BYTES = 1 << 18
    try:
        check()
    finally:
        device.WORK_BYTES = old
    os.environ['NB_LIST_SLICE_GROUPS'] = '5'
    try:
        check()
    finally:
        del os.environ['NB_LIST_SLICE_GROUPS']


def test_list_eval_matches_the_one_kernel_form(dev):
    """nb_list_eval (candidate lists + one batched emulator launch; exclusion
    = any bound, association = first bound) against the one-kernel form that
    walks the list inside the kernel (nb_contains_any / nb_first_containing),
    on a list of nested bounds longer than anything the golden fixtures hold,
    in slabs (small work space) and at once."""
    import ctypes as C
    from nautilus_amd import _lib, device
    from helpers import nautilus_from_golden
    g = load_golden('nautilusbound_D4')
    base = nautilus_from_golden(g)
    bounds = [upload(base)]
    # nested copies: the same bound with rising emulator thresholds
    for k in range(1, 7):
        nb = nautilus_from_golden(g)
        for n_b in nb.neural_bounds:
            n_b.score_predict_min = n_b.score_predict_min + 0.03 * k
        bounds.append(upload(nb))
    lst = device.DeviceBoundList(bounds[1:])
    x = bounds[0].propose(5, 0, 100000)
    lib = _lib.load()
    ref_any = torch.empty(x.shape[0], dtype=torch.uint8, device='cuda')
    ref_first = torch.empty(x.shape[0], dtype=torch.int32, device='cuda')
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.nb_contains_any(lst._h, C.c_void_p(x.data_ptr()),
                                   x.shape[0], C.c_void_p(ref_any.data_ptr()),
                                   stream))
    _lib.check(lib.nb_first_containing(
        lst._h, C.c_void_p(x.data_ptr()), x.shape[0],
        C.c_void_p(ref_first.data_ptr()), stream))
    got_any = lst.contains_any(x, as_flags=True)
    got_first = lst.first_containing(x)
    assert 0.01 < float(ref_any.float().mean()) < 0.99
    assert torch.equal(got_any, ref_any)
    assert torch.equal(got_first, ref_first)
    old = device.WORK_BYTES
    device.WORK_BYTES = 1 << 20
    try:
        assert torch.equal(lst.contains_any(x, as_flags=True), ref_any)
        assert torch.equal(lst.first_containing(x), ref_first)
    finally:
        device.WORK_BYTES = old
    # a list with more groups than the second stage's pass table holds is
    # evaluated in slices (rows already inside stay as they are, the first
    # containing bound is kept): forced here with two groups per slice
    os.environ['NB_LIST_SLICE_GROUPS'] = '2'
    try:
        assert torch.equal(lst.contains_any(x, as_flags=True), ref_any)
        assert torch.equal(lst.first_containing(x), ref_first)
    finally:
        del os.environ['NB_LIST_SLICE_GROUPS']


@pytest.fixture(scope='module')
def nautilus_d4():
    from helpers import nautilus_from_golden
    g = load_golden('nautilusbound_D4')
    return g, nautilus_from_golden(g)


def test_nautilus_bound_contains_and_sample(dev, nautilus_d4):
    from oracle import philox
    g, ob = nautilus_d4
    b = upload(ob)
    mask = b.contains(g['test']).cpu().numpy()
    assert (mask != g['contains']).sum() <= 1
    assert bool(b.contains(g['sample']).all())
    seed, offset, n = 5, 10**12, 30000
    pts, counts = b.sample_launch(seed, offset, n)
    c = counts.cpu().numpy()
    pts_o, cnt_o = philox.nautilus_sample(ob, seed, offset, n)
    assert abs(int(c[0]) - int(cnt_o[2])) <= 1
    assert abs(int(c[1]) - len(pts_o)) <= 2
    if int(c[1]) == len(pts_o):
        assert np.allclose(pts[:c[1]].cpu().numpy(), pts_o, rtol=0,
                           atol=1e-12)
    # the MC volume agrees with the reference's estimate (nautilus.py:257-261)
    log_v = (np.logaddexp.reduce(ob.outer_bound.log_v_all) +
             np.log(c[1] / n))
    assert abs(log_v - float(g['log_v'])) < 0.1


def test_sample_launch_scratch_reuse(dev, nautilus_d4):
    """``reuse=True`` (the refill loops of the bounds) runs the same launch in
    the grow-only scratch buffers: same rows and counters, the storage is
    shared by consecutive launches."""
    g, ob = nautilus_d4
    b = upload(ob)
    p1, c1 = b.sample_launch(5, 10**12, 30000)
    p2, c2 = b.sample_launch(5, 10**12, 30000, reuse=True)
    k = int(c1[1])
    assert np.array_equal(c1.cpu().numpy(), c2.cpu().numpy())
    assert torch.equal(p1[:k], p2[:k])
    kept = p2[:k].clone()
    p3, _ = b.sample_launch(6, 0, 20000, reuse=True)
    assert p3.data_ptr() == p2.data_ptr()          # same scratch
    assert torch.equal(p1[:k], kept)
    p4, _ = b.sample_launch(6, 0, 40000, reuse=True)   # grows, still correct
    p5, c5 = b.sample_launch(6, 0, 40000)
    assert torch.equal(p4[:int(c5[1])], p5[:int(c5[1])])


@pytest.mark.parametrize('d', [3, 20, 50])
def test_mvee_kernel_matches_reference(dev, d):
    """nb_mvee_weights (basic.py:175-232 on the device) against the oracle's
    restatement and the reference's own ellipsoids (golden fixtures)."""
    from nautilus_amd import geometry
    from oracle import bounds_oracle as bo
    g = load_golden('ellipsoid_D%d' % d)
    pts = g['points']
    c_o, a_o, _ = bo.mvee(pts)
    u = dev.mvee_weights(pts).cpu().numpy()
    assert u.shape == (len(pts),) and abs(u.sum() - 1.0) < 1e-12
    assert np.all(u >= 0)
    u_h = khachiyan_weights_numpy(pts)
    assert np.allclose(u, u_h, rtol=0, atol=1e-9)
    c, a, a_inv = geometry.mvee(pts)
    # the reference's module-level names (bounds/basic.py:154, 175)
    from nautilus_amd import bounds
    c2, a2, a2_inv = bounds.minimum_volume_enclosing_ellipsoid(pts)
    assert np.array_equal(c, c2) and np.array_equal(a, a2)
    assert np.allclose(
        bounds.invert_symmetric_positive_semidefinite_matrix(a), a_inv,
        rtol=1e-9, atol=1e-12 * np.abs(a_inv).max())
    assert np.allclose(c, c_o, rtol=0, atol=1e-9)
    assert np.allclose(a, a_o, rtol=1e-7, atol=1e-8 * np.abs(a_o).max())
    # every point inside, at least one on the surface (basic.py:236-239)
    r2 = np.einsum('ij,jk,ik->i', pts - c, a, pts - c)
    assert abs(r2.max() - 1.0) < 1e-12
    p = geometry.ellipsoid_params(pts, float(g['enlarge']))
    assert np.allclose(p['c'], g['c'], rtol=0, atol=1e-9)
    assert np.allclose(p['B'], g['B'], rtol=1e-7, atol=1e-12)


def test_mvee_kernel_shapes(dev):
    """Known answers and ragged sizes: points on a sphere (reference
    tests/test_bounds.py), n barely above n_dim, n not a multiple of 16, odd
    and even dimensions up to the kernels' limit of 128, more points than one
    workgroup per 128 handles, point sets beyond 65 536 points (up to 8192
    points per workgroup and candidate list)."""
    from nautilus_amd import geometry
    rng = np.random.default_rng(5)
    for d, n in [(2, 3), (2, 100), (7, 9), (15, 333), (16, 200), (31, 64),
                 (33, 1000), (62, 300), (63, 129), (64, 200), (79, 700),
                 (100, 1500), (127, 400), (128, 300), (20, 6000),
                 (8, 70000), (50, 66000), (5, 250000)]:
        pts = rng.normal(size=(n, d)) * rng.uniform(0.5, 2.0, size=d) + 0.3
        u = dev.mvee_weights(pts).cpu().numpy()
        u_h = khachiyan_weights_numpy(pts)
        assert abs(u.sum() - 1.0) < 1e-12, (d, n)
        assert np.allclose(u, u_h, rtol=0, atol=1e-8), (d, n)
    x = rng.normal(size=(500, 10))
    x /= np.linalg.norm(x, axis=1)[:, None]
    c, a, _ = geometry.mvee(x)
    assert np.allclose(c, 0, atol=0.05)
    assert np.allclose(a, np.eye(10), atol=0.1)
    with pytest.raises(RuntimeError):
        dev.mvee_weights(rng.normal(size=(200, 129)))
    # n_batch other than the default, fewer points than n_batch
    pts = rng.normal(size=(300, 6))
    for nb in (1, 5, 32):
        assert np.allclose(dev.mvee_weights(pts, n_batch=nb).cpu().numpy(),
                           khachiyan_weights_numpy(pts, n_batch=nb),
                           rtol=0, atol=1e-9)
    few = rng.normal(size=(12, 4))
    assert np.allclose(dev.mvee_weights(few).cpu().numpy(),
                       khachiyan_weights_numpy(few), rtol=0, atol=1e-9)


def test_mvee_batch_equals_single_fits(dev):
    """nb_mvee_khachiyan advances several point sets in the same launches
    (the children of Union.split): every set gets exactly the result of a
    fit of its own, whatever the sizes of its neighbours."""
    from nautilus_amd import geometry
    rng = np.random.default_rng(11)
    sets = [rng.normal(size=(n, 24)) * rng.uniform(0.2, 3.0, size=24)
            for n in (40, 1000, 333, 5000)]
    geometry._ELL_CACHE.clear()
    batch = geometry.mvee_batch(sets)
    for pts, (c, a, a_inv) in zip(sets, batch):
        c1, a1, a_inv1 = geometry.mvee(pts)
        assert np.array_equal(c, c1) and np.array_equal(a, a1)
        r2 = np.einsum('ij,jk,ik->i', pts - c, a, pts - c)
        assert abs(r2.max() - 1.0) < 1e-12
        assert np.allclose(a @ a_inv, np.eye(24), atol=1e-9)


def test_weighted_moments_and_quadform(dev):
    """nb_weighted_moments / nb_quadform_max against numpy (basic.py:233-236)."""
    rng = np.random.default_rng(3)
    for n, d in [(5, 2), (1000, 17), (3000, 50), (777, 100), (4500, 128)]:
        x = rng.normal(size=(n, d)) * 0.3 + 0.1
        w = rng.random(n)
        q = np.hstack([x, np.ones((n, 1))])
        s = dev.weighted_moments(x, torch.from_numpy(w).cuda(), 0.5)
        ref = 0.5 * (q * w[:, None]).T @ q
        assert np.allclose(s.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)
        s1 = dev.weighted_moments(x).cpu().numpy()
        assert np.allclose(s1, q.T @ q, rtol=1e-12, atol=1e-12)
        p = np.linalg.inv(q.T @ q / n)
        p = 0.5 * (p + p.T)
        g = np.einsum('ij,jk,ik->i', q, p, q)
        got = float(dev.quadform_max(x, p).cpu()[0])
        assert abs(got - g.max()) < 1e-10 * g.max()


def _sklearn_em_from_labels(x, labels, **kw):
    """scikit-learn's EM started from the M-step of a hard assignment (what
    mixture/_base.py:_initialize_parameters does with the k-means labels)."""
    from sklearn.mixture import GaussianMixture
    resp = np.stack([labels == 0, labels == 1], axis=1).astype(float)
    nk = resp.sum(axis=0) + 10 * np.finfo(float).eps
    means = resp.T @ x / nk[:, None]
    covs = []
    for k in range(2):
        diff = x - means[k]
        cov = (resp[:, k] * diff.T) @ diff / nk[k]
        cov.flat[::x.shape[1] + 1] += 1e-6
        covs.append(cov)
    gmm = GaussianMixture(
        n_components=2, weights_init=nk / nk.sum(), means_init=means,
        precisions_init=np.array([np.linalg.inv(c) for c in covs]), **kw)
    with np.errstate(all='ignore'):
        return gmm.fit(x)


@pytest.mark.parametrize('d,n', [(2, 300), (7, 500), (20, 1500), (50, 2000),
                                 (63, 700), (64, 900), (100, 3000),
                                 (127, 1200), (128, 1000)])
def test_gmm_em_matches_sklearn(dev, d, n):
    """nb_gmm_fit against scikit-learn 1.7 (the reference's dependency for
    Union.split, union.py:185-187): same initial assignment -> same EM
    trajectory (number of iterations, lower bound, parameters)."""
    rng = np.random.default_rng(d)
    shift = np.zeros(d)
    shift[0] = 3.0
    x = np.vstack([rng.normal(size=(n // 2, d)),
                   rng.normal(size=(n - n // 2, d)) * 0.7 + shift])
    x = x[rng.permutation(n)]
    inits = np.array([(x[:, 0] > 1.5).astype(np.int32),
                      (x[:, 1] > 0.0).astype(np.int32),
                      rng.integers(0, 2, n).astype(np.int32)])
    fits = dev.gmm_fit(x, n_init=3, init_labels=inits)
    for lab, fit in zip(inits, fits):
        ref = _sklearn_em_from_labels(x, lab)
        assert not fit['failed']
        assert fit['n_iter'] == ref.n_iter_
        assert fit['converged'] == ref.converged_
        assert abs(fit['lower_bound'] - ref.lower_bound_) < 1e-9
        assert np.allclose(fit['weights'], ref.weights_, rtol=0, atol=1e-10)
        assert np.allclose(fit['means'], ref.means_, rtol=0, atol=1e-8)
        assert np.allclose(fit['covariances'], ref.covariances_, rtol=0,
                           atol=1e-8)


def test_gmm_full_fit(dev):
    """k-means++ / Lloyd seeding + EM: separated clusters are recovered like
    scikit-learn recovers them; on a single Gaussian the best restart reaches
    scikit-learn's likelihood; restarts are reproducible for a given seed."""
    from sklearn.mixture import GaussianMixture
    from nautilus_amd import geometry
    rng = np.random.default_rng(3)
    d, n = 10, 1200
    shift = np.zeros(d)
    shift[:2] = 6.0
    x = np.vstack([rng.normal(size=(n // 3, d)),
                   rng.normal(size=(n - n // 3, d)) + shift])
    truth = np.r_[np.zeros(n // 3, int), np.ones(n - n // 3, int)]
    fits = dev.gmm_fit(x, n_init=10, seed=42)
    again = dev.gmm_fit(x, n_init=10, seed=42)
    assert [f['lower_bound'] for f in fits] == \
        [f['lower_bound'] for f in again]
    best = max(fits, key=lambda f: f['lower_bound'])
    ref = GaussianMixture(n_components=2, n_init=10, random_state=0).fit(x)
    assert abs(best['lower_bound'] - ref.lower_bound_) < 2e-3
    # the log probabilities the fit leaves on the device (what Union.split's
    # hard assignment uses, union.py:188-190) against scipy's logpdf with the
    # returned parameters
    from scipy.stats import multivariate_normal
    want = np.vstack([multivariate_normal.logpdf(
        x, mean=best['means'][k], cov=best['covariances'][k]) +
        np.log(best['weights'][k]) for k in range(2)])
    got = best['logp'].cpu().numpy()
    assert np.allclose(got, want, rtol=1e-9, atol=1e-8)
    lab = geometry.two_component_labels(x, d + 1, 42)
    agree = max(np.mean(lab == truth), np.mean(lab != truth))
    assert agree == 1.0
    # unimodal input (the usual case in Union.split): likelihood on par
    x1 = rng.normal(size=(2000, 30))
    best1 = max(dev.gmm_fit(x1, n_init=10, seed=7),
                key=lambda f: f['lower_bound'])
    ref1 = GaussianMixture(n_components=2, n_init=10, random_state=0).fit(x1)
    assert best1['lower_bound'] > ref1.lower_bound_ - 0.02
    assert np.bincount(geometry.two_component_labels(x1, 31, 7)).min() >= 31


@pytest.mark.parametrize('d', [3, 20, 50])
def test_transform_and_standardize(dev, d):
    """nb_ellipsoid_transform (basic.py:340) and nb_standardize
    (neural.py:74-77) against numpy on the golden ellipsoids."""
    import torch
    from oracle import bounds_oracle as bo
    g = load_golden('ellipsoid_D%d' % d)
    e = bo.OEllipsoid.from_params(g['c'], g['B'], g['B_inv'])
    b = upload(e)
    x = np.random.default_rng(d).random((1003, d))
    y = b.transform(x).cpu().numpy()
    want = e.transform(x)
    assert np.allclose(y, want, rtol=0, atol=1e-11 * np.abs(want).max())
    mean, scale, xs = dev.standardize(torch.from_numpy(want).cuda())
    assert np.allclose(mean.cpu().numpy(), want.mean(axis=0), rtol=0,
                       atol=1e-13 * np.abs(want).max())
    assert np.allclose(scale.cpu().numpy(), want.std(axis=0), rtol=1e-13)
    assert np.allclose(xs.cpu().numpy(),
                       (want - want.mean(axis=0)) / want.std(axis=0),
                       rtol=0, atol=1e-11)


def test_gmm_degenerate_input_is_a_rejected_split(dev):
    """A cloud the mixture cannot divide (every point the same: each device
    restart ends its Lloyd iterations with an empty cluster) makes
    ``two_component_labels`` raise ``geometry.DegenerateMixture`` -- no host
    fit stands behind the device one -- and ``Union.split`` treats that as a
    split that does not pay (union.py:204-207): the member is blocked, the
    union stays as it was and the run goes on.  A cloud with a few displaced
    points still divides."""
    from nautilus_amd import geometry
    from nautilus_amd.bounds import Union
    x = np.zeros((300, 4)) + 0.5
    fits = dev.gmm_fit(x, n_init=4, seed=1)
    assert len(fits) == 4 and all(f['failed'] for f in fits)
    with pytest.raises(geometry.DegenerateMixture):
        geometry.two_component_labels(x, 5, 3)
    # the union of a regular cloud whose fit is made to fail: blocked, intact
    rng = np.random.default_rng(0)
    pts = 0.5 + 0.05 * rng.normal(size=(400, 4))
    u = Union.compute(pts, rng=np.random.default_rng(1))
    real = geometry.two_component_labels

    def failing(*args):
        raise geometry.DegenerateMixture('provoked')
    geometry.two_component_labels = failing
    try:
        assert u.split() is False
    finally:
        geometry.two_component_labels = real
    assert len(u.bounds) == 1 and bool(u.block[0])
    y = np.zeros((300, 4)) + 0.5
    y[:5] += 0.1
    lab = geometry.two_component_labels(y, 5, 3)
    assert lab.shape == (300,) and set(np.unique(lab)) <= {0, 1}


def test_phase_shift_bit_exact(dev):
    """bounds/periodic.py on the device: centres, forward and inverse
    transform are bit-identical to the reference's (golden fixture)."""
    from nautilus_amd.bounds import PhaseShift
    g = load_golden('phaseshift')
    for pts, centers, fwd, back in zip(g['points'], g['centers'],
                                       g['forward'], g['inverse']):
        shift = PhaseShift.compute(pts, g['periodic'])
        assert np.array_equal(shift.centers, centers)
        import torch
        shift_t = PhaseShift.compute(torch.from_numpy(pts).cuda(),
                                     g['periodic'])
        assert np.array_equal(shift_t.centers, centers)
        out = shift.transform(pts)
        assert np.array_equal(out, fwd)
        assert np.array_equal(shift.transform(out, inverse=True), back)
    # wrap-around edge values: exactly on the boundary, tiny negative sums
    shift = PhaseShift.from_params([0, 1], [0.5, 0.25])
    x = np.array([[0.0, 0.75], [1.0 - 2**-53, 0.75 - 1e-20], [0.5, 0.0],
                  [0.25, 1.0 - 2**-53]])
    from oracle import bounds_oracle as bo
    o = bo.OPhaseShift.from_params([0, 1], [0.5, 0.25])
    assert np.array_equal(shift.transform(x), o.transform(x))
    assert np.array_equal(shift.transform(x, inverse=True),
                          o.transform(x, inverse=True))


def test_periodic_nautilus_bound(dev):
    """NautilusBound with a PhaseShift (nautilus.py:91-96, 162-163, 241-243):
    contains() recentres, sample() returns points in the sampler's frame."""
    from nautilus_amd import bounds as nb
    from oracle import philox
    g = load_golden('nautilusbound_periodic_D3')
    ob = nautilus_from_golden(g)
    assert ob.shift is not None
    assert np.array_equal(ob.contains(g['test']), g['contains'])
    b = upload(ob)
    # points hugging the periodic seam as well
    rng = np.random.default_rng(8)
    seam = rng.random((4096, 3))
    seam[:, 0] = (rng.normal(size=4096) * 0.02) % 1
    x = np.vstack([g['test'], seam, g['sample']])
    want = ob.contains(x)
    got = b.contains(x).cpu().numpy()
    assert (got != want).sum() <= 1
    assert got[-len(g['sample']):].all()
    # in a list (shell exclusion) every bound applies its own shift
    ob_plain = nautilus_from_golden(load_golden('nautilusbound_D4'))
    del ob_plain
    lst = dev.DeviceBoundList([b])
    assert np.array_equal(lst.contains_any(x).cpu().numpy(), got)
    assert np.array_equal(lst.first_containing(x).cpu().numpy() == 0, got)

    # sampling: same proposals as the oracle's Philox pipeline, shifted back
    seed, offset, n = 9, 7 * 10**9, 20000
    pts_o, cnt_o = philox.nautilus_sample(ob, seed, offset, n)
    pts_o = ob.shift.transform(pts_o, inverse=True)
    outer = nb.Union.from_members(
        [nb.UnitCubeEllipsoidMixture.from_params(
            m.dim_cube, None if m.ellipsoid is None else
            nb.Ellipsoid.from_params(m.ellipsoid.c, m.ellipsoid.B,
                                     m.ellipsoid.B_inv, m.ellipsoid.A))
         for m in ob.outer_bound.bounds], unit=True)
    outer.log_v_all = ob.outer_bound.log_v_all
    neural = []
    for o in ob.neural_bounds:
        from nautilus_amd.emulator import NeuralNetworkEmulator, Network
        emu = NeuralNetworkEmulator.from_weights(
            o.emulator.mean, o.emulator.scale,
            [Network(n_.coefs, n_.intercepts) for n_ in o.emulator.networks])
        neural.append(nb.NeuralBound.from_parts(
            nb.Ellipsoid.from_params(o.outer_bound.c, o.outer_bound.B,
                                     o.outer_bound.B_inv, o.outer_bound.A),
            emu, o.score_predict_min))
    full = nb.NautilusBound.from_parts(
        outer, neural, rng=np.random.default_rng(1),
        shift=nb.PhaseShift.from_params(g['periodic'], g['centers']))
    full._stream.seed, full._stream.offset = seed, offset
    drawn = full.sample(len(pts_o) // 2)
    assert np.allclose(drawn, pts_o[:len(drawn)], rtol=0, atol=1e-12)
    assert np.all((drawn >= 0) & (drawn < 1))
    assert ob.contains(drawn).mean() > 0.999
    assert abs(full.log_v - float(g['log_v'])) < 0.1


def _product_bound(g, seed):
    """The golden NautilusBound as a nautilus_amd.bounds.NautilusBound with
    its Philox stream at a known place."""
    from nautilus_amd import bounds as nb
    from nautilus_amd.emulator import NeuralNetworkEmulator, Network
    ob = nautilus_from_golden(g)
    outer = nb.Union.from_members(
        [nb.UnitCubeEllipsoidMixture.from_params(
            m.dim_cube, None if m.ellipsoid is None else
            nb.Ellipsoid.from_params(m.ellipsoid.c, m.ellipsoid.B,
                                     m.ellipsoid.B_inv, m.ellipsoid.A))
         for m in ob.outer_bound.bounds], unit=True)
    outer.log_v_all = ob.outer_bound.log_v_all
    neural = []
    for o in ob.neural_bounds:
        emu = NeuralNetworkEmulator.from_weights(
            o.emulator.mean, o.emulator.scale,
            [Network(n_.coefs, n_.intercepts) for n_ in o.emulator.networks])
        neural.append(nb.NeuralBound.from_parts(
            nb.Ellipsoid.from_params(o.outer_bound.c, o.outer_bound.B,
                                     o.outer_bound.B_inv, o.outer_bound.A),
            emu, o.score_predict_min))
    full = nb.NautilusBound.from_parts(outer, neural,
                                       rng=np.random.default_rng(1))
    full._stream.seed, full._stream.offset = seed, 0
    return full


def test_refill_ahead_hands_out_the_same_points(dev):
    """``prefetch`` (the refill of the next batch's bound, launched before
    the host waits for the current batch; sampler.py ``_prefetch_next``)
    draws what the refill-when-asked loop would have drawn -- the bound's
    Philox stream, consumed in order -- so the points a bound hands out do
    not depend on when its refills were launched, a refill in flight lands
    before anything looks at the queue, and the Monte-Carlo volume counters
    count every proposal that was examined."""
    g = load_golden('nautilusbound_D4')
    plain, ahead = _product_bound(g, 123), _product_bound(g, 123)
    asks = (3000, 500, 7000, 1, 2500)
    want = [plain.sample_device(n).clone() for n in asks]
    got = []
    for i, n in enumerate(asks):
        ahead.prefetch(n)                      # in flight ...
        assert ahead.__dict__.get('_pending') is not None or i > 0
        if i == 2:
            assert len(ahead.points) >= 0      # ... lands when looked at
            assert ahead.__dict__.get('_pending') is None
        got.append(ahead.sample_device(n).clone())
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    # a refill in flight when the bound is pickled lands first
    import pickle
    ahead.prefetch(10**5)
    state = pickle.loads(pickle.dumps(ahead))
    assert state.__dict__.get('_pending') is None
    assert ahead.__dict__.get('_pending') is None
    assert ahead.n_sample >= plain.n_sample and ahead.n_sample > 0
    assert abs(ahead.log_v - plain.log_v) < 0.05
    # ... and one that is dropped (reset) leaves no slot behind
    from nautilus_amd.bounds import _PrefetchSlots
    free = len(_PrefetchSlots.free)
    assert ahead.prefetch(10**6)
    assert len(_PrefetchSlots.free) == free - 1
    ahead.reset(np.random.default_rng(2))
    assert len(_PrefetchSlots.free) == free


def test_shell_exclusion_and_association(dev, nautilus_d4, neural_d4):
    """sampler.py:797-798 and 1213-1219 over a list of nested bounds."""
    import torch
    from oracle import bounds_oracle as bo
    g, ob = nautilus_d4
    # three nested bounds: the golden one and two shrunken ellipsoids
    c = 0.5 * np.ones(4)
    e1 = bo.OEllipsoid.from_params(c, 0.2 * np.eye(4))
    e2 = bo.OEllipsoid.from_params(c + 0.05, 0.1 * np.eye(4))
    obs = [ob, e1, e2]
    devs = [upload(o) for o in obs]
    x = np.random.default_rng(4).random((5000, 4))
    contains = np.array([o.contains(x) for o in obs])
    lst = dev.DeviceBoundList(devs)
    assert np.array_equal(lst.contains_any(x).cpu().numpy(),
                          contains.any(axis=0))
    rev = dev.DeviceBoundList(devs[::-1])
    idx = rev.first_containing(x).cpu().numpy()
    want = np.full(len(x), -1)
    for i in range(3):                   # highest index wins
        want[contains[i]] = 2 - i
    want_first = np.full(len(x), -1)
    for pos, i in enumerate([2, 1, 0]):
        sel = contains[i] & (want_first < 0)
        want_first[sel] = pos
    assert np.array_equal(idx, want_first)
    assert np.array_equal(dev.DeviceBoundList([]).contains_any(
        torch.from_numpy(x).cuda()).cpu().numpy(), np.zeros(len(x), bool))


def test_compaction_edge_cases(dev):
    import torch
    gen = torch.Generator(device='cuda').manual_seed(0)
    for n, d in [(1, 3), (2047, 5), (2048, 50), (2049, 7), (100000, 20)]:
        x = torch.rand((n, d), dtype=torch.float64, device='cuda',
                       generator=gen)
        for p in (0.0, 0.01, 0.5, 1.0):
            flags = (torch.rand(n, device='cuda', generator=gen) < p).to(
                torch.uint8) * 3
            out, counts, src = dev.compact_rows(x, flags, 2, want_index=True)
            k = int(counts[1])
            keep = flags.bool()
            assert k == int(keep.sum()) == int(counts[0])
            assert torch.equal(out[:k], x[keep])
            assert torch.equal(src[:k], torch.nonzero(keep).flatten())


def test_shell_stats_match_reference(dev):
    import torch
    from scipy.special import logsumexp
    g = load_golden('shellstats')
    for i in range(len(g['shell_n'])):
        ll = g['log_l_%d' % i]
        out = dev.shell_stats(torch.from_numpy(ll).cuda(),
                              float(np.median(ll))).cpu().numpy()
        assert np.isclose(out[0], logsumexp(ll), rtol=1e-13, atol=1e-13)
        assert np.isclose(out[1], logsumexp(2 * ll), rtol=1e-13, atol=1e-13)
        assert out[2] == ll.max()
        assert out[3] == np.sum(ll >= np.median(ll))
        n_eff = np.exp(2 * out[0] - out[1])
        assert np.isclose(n_eff, g['shell_n_eff'][i], rtol=1e-11)
    ninf = torch.full((100,), -np.inf, dtype=torch.float64, device='cuda')
    out = dev.shell_stats(ninf).cpu().numpy()
    assert out[0] == -np.inf and out[1] == -np.inf
    big = np.random.default_rng(0).normal(size=3_000_000) * 30
    out = dev.shell_stats(torch.from_numpy(big).cuda()).cpu().numpy()
    assert np.isclose(out[0], logsumexp(big), rtol=1e-12)


def test_emulator_training_matches_oracle(dev):
    """nb_mlp_train.hip against the restated MLPRegressor.fit: same Glorot
    draw, same minibatch order, fp64 -> the loss curve and the weights track
    the oracle (differences only from the association of fp64 sums)."""
    import torch
    from nautilus_amd import emulator
    from oracle import mlp_oracle as mo
    g = load_golden('emulator_D5_E1')
    x = (g['x'] - g['mean']) / g['scale']
    y = g['y']
    n_ep = 6
    nets, _ = emulator.train_networks(
        torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), [0, 3],
        max_epochs=n_ep)
    for seed, net in zip([0, 3], nets):
        ref = mo.fit_network(x, y, seed, max_iter=n_ep)
        assert net.n_iter_ == ref.n_iter == n_ep
        assert np.allclose(net.loss_curve_, ref.loss_curve, rtol=1e-9, atol=0)
        for k in range(4):
            assert np.allclose(net.coefs_[k], ref.coefs[k], rtol=0, atol=1e-8)
            assert np.allclose(net.intercepts_[k], ref.intercepts[k], rtol=0,
                               atol=1e-8)


@pytest.mark.parametrize('hidden', [(64, 32, 16), (100, 17, 3), (5, 50, 20)])
def test_emulator_narrow_architectures(dev, hidden):
    """``neural_network_kwargs=dict(hidden_layer_sizes=...)`` (neural.py:79-83
    passes it to MLPRegressor): three hidden layers narrower than the default
    train and predict in the default's tiles with zero weights for the
    missing units -- against the oracle's fit of that architecture (same
    Glorot draw for ITS shapes, same shuffles), a full fit to its stopping
    epoch, and the prediction of the trained emulator."""
    import torch
    from nautilus_amd import emulator
    from oracle import mlp_oracle as mo
    g = load_golden('emulator_D5_E1')
    x, y = g['x'], g['y']
    kw = dict(hidden_layer_sizes=hidden, max_iter=40)
    emu = emulator.NeuralNetworkEmulator.train(x, y, n_networks=2,
                                               neural_network_kwargs=kw)
    ref = mo.Emulator.train(x, y, n_networks=2, neural_network_kwargs=kw)
    units = [5, *hidden, 1]
    for net, rnet in zip(emu.neural_networks, ref.networks):
        assert [c.shape for c in net.coefs_] == list(zip(units[:-1],
                                                         units[1:]))
        assert net.n_iter_ == rnet.n_iter
        assert np.allclose(net.loss_curve_, rnet.loss_curve, rtol=1e-8, atol=0)
        for k in range(4):
            assert np.allclose(net.coefs_[k], rnet.coefs[k], rtol=0, atol=1e-7)
            assert np.allclose(net.intercepts_[k], rnet.intercepts[k], rtol=0,
                               atol=1e-7)
    probe = np.random.default_rng(1).random((3000, 5))
    want = ref.predict(probe)
    assert np.allclose(emu.predict(probe), want, rtol=0, atol=1e-7)
    # the trained weights through the oracle's forward pass: the device
    # prediction itself to rounding
    exact = mo.Emulator.from_weights(
        emu.mean, emu.scale, [(n.coefs_, n.intercepts_)
                              for n in emu.neural_networks]).predict(probe)
    assert np.allclose(emu.predict(probe), exact, rtol=0, atol=1e-11)


def test_emulator_training_ragged_batches(dev):
    """n not a multiple of 200 and n < 200 (last / only minibatch short)."""
    import torch
    from nautilus_amd import emulator
    from oracle import mlp_oracle as mo
    rng = np.random.default_rng(9)
    for n, d in [(437, 20), (90, 3), (1, 2)]:
        x = rng.normal(size=(n, d))
        y = rng.random(n)
        nets, _ = emulator.train_networks(
            torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), [1],
            max_epochs=3)
        ref = mo.fit_network(x, y, 1, max_iter=3)
        assert np.allclose(nets[0].loss_curve_, ref.loss_curve, rtol=1e-9)
        assert np.allclose(nets[0].coefs_[0], ref.coefs[0], rtol=0, atol=1e-8)


@pytest.mark.parametrize('d', [63, 64, 65, 80, 100, 112, 127, 128])
def test_emulator_training_wide_inputs(dev, d):
    """The trainer beyond 64 input dimensions (five to nine k-tiles in layer
    1: other register schedules, a different job list of the gradient phase)
    against the restated MLPRegressor.fit -- configuration 5 trains at 100."""
    import torch
    from nautilus_amd import emulator
    from oracle import mlp_oracle as mo
    rng = np.random.default_rng(100 + d)
    n, n_ep = 1237, 4
    x = rng.normal(size=(n, d))
    y = rng.random(n)
    seeds = [0, 1, 2, 3, 4, 5, 6, 7] if d == 100 else [0, 3]
    nets, _ = emulator.train_networks(
        torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), seeds,
        max_epochs=n_ep)
    for seed, net in zip(seeds, nets):
        ref = mo.fit_network(x, y, seed, max_iter=n_ep)
        assert net.n_iter_ == ref.n_iter == n_ep
        assert np.allclose(net.loss_curve_, ref.loss_curve, rtol=1e-9, atol=0)
        for k in range(4):
            assert np.allclose(net.coefs_[k], ref.coefs[k], rtol=0, atol=1e-8)
            assert np.allclose(net.intercepts_[k], ref.intercepts[k], rtol=0,
                               atol=1e-8)


def test_emulator_training_large_n(dev):
    """The trainer on a training set of the size the samplers reach late in
    a run (120 000 rows: 600 Adam steps per epoch, shuffles from the native
    MT19937 streams) against the restated MLPRegressor.fit: the first epochs
    agree to rounding.  (Later epochs drift apart the way ANY two summation
    orders do -- tests/tools/train_divergence.py prints the growth next to
    that of two CPU runs which differ in summation order only.)"""
    import torch
    from nautilus_amd import emulator
    from oracle import mlp_oracle as mo
    rng = np.random.default_rng(21)
    n, d, n_ep = 120000, 50, 3
    x = rng.normal(size=(n, d))
    r = np.linalg.norm(x[:, :8], axis=1) + 0.3 * rng.normal(size=n)
    y = np.argsort(np.argsort(-r)) / n
    nets, stats = emulator.train_networks(
        torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), [0, 5],
        max_epochs=n_ep)
    assert stats['n_rows'] == n
    for seed, net in zip([0, 5], nets):
        ref = mo.fit_network(x, y, seed, max_iter=n_ep)
        assert net.n_iter_ == ref.n_iter == n_ep
        assert np.allclose(net.loss_curve_, ref.loss_curve, rtol=1e-9, atol=0)
        for k in range(4):
            assert np.allclose(net.coefs_[k], ref.coefs[k], rtol=0, atol=1e-8)
            assert np.allclose(net.intercepts_[k], ref.intercepts[k], rtol=0,
                               atol=1e-8)


@pytest.mark.parametrize('name,e', [('emulator_D5_E1', 1),
                                    ('emulator_D20_E2', 2),
                                    ('emulator_D50_E4', 4)])
def test_emulator_full_fit_equals_sklearn(dev, name, e):
    """A WHOLE fit -- same Glorot draw, same minibatch order for every epoch
    (numpy RandomState shuffles as in scikit-learn), the stopping rule of
    _fit_stochastic -- against scikit-learn's own MLPRegressor.fit (fixtures
    written by make_golden.py through the reference's
    NeuralNetworkEmulator.train): the stop epoch is EQUAL, the loss curve
    agrees over its whole length, weights and predictions to rounding."""
    from nautilus_amd.emulator import NeuralNetworkEmulator
    g = load_golden(name)
    emu = NeuralNetworkEmulator.train(g['x'], g['y'], n_networks=e)
    assert np.allclose(emu.mean, g['mean'], rtol=1e-14, atol=1e-15)
    assert np.allclose(emu.scale, g['scale'], rtol=1e-13)
    for i, net in enumerate(emu.neural_networks):
        ref = g['loss_curve_%d' % i]
        assert net.n_iter_ == int(g['n_iter_%d' % i]) == len(ref)
        assert np.allclose(net.loss_curve_, ref, rtol=1e-9, atol=0)
        for k in range(4):
            assert np.allclose(net.coefs_[k], g['coef_%d_%d' % (i, k)],
                               rtol=0, atol=1e-10)
            assert np.allclose(net.intercepts_[k],
                               g['intercept_%d_%d' % (i, k)], rtol=0,
                               atol=1e-10)
    assert np.allclose(emu.predict(g['test']), g['predict'], rtol=0,
                       atol=1e-11)


def test_fleet_falls_back_without_the_resident_kernel(dev, monkeypatch):
    """Two ensembles with a training set each are packed into one fleet
    trainer, which needs the resident kernel.  Where the library cannot
    provide it (XCD placement probe fails, XCDs held by another trainer --
    forced here with the library-side switch NB_TRAIN_NO_RESIDENT, which the
    Python side does not know) every ensemble falls back to a trainer of its
    own with two launches per step, and the networks are the same."""
    from nautilus_amd import emulator
    rng = np.random.default_rng(4)
    jobs = []
    for n, d in [(700, 6), (450, 6)]:
        x = rng.normal(size=(n, d))
        y = rng.random(n)
        jobs.append(dict(xs=torch.from_numpy(x).cuda(),
                         y=torch.from_numpy(y).cuda(), seeds=[0, 1],
                         max_epochs=4))
    fleet = emulator.train_ensembles([dict(j) for j in jobs])
    monkeypatch.setenv('NB_TRAIN_NO_RESIDENT', '1')
    alone = emulator.train_ensembles([dict(j) for j in jobs])
    for (nets_a, st_a), (nets_b, st_b) in zip(fleet, alone):
        assert st_a['n_iter'] == st_b['n_iter'] == [4, 4]
        for a, b in zip(nets_a, nets_b):
            assert np.allclose(a.loss_curve_, b.loss_curve_, rtol=1e-9)
            for k in range(4):
                assert np.allclose(a.coefs_[k], b.coefs_[k], rtol=0,
                                   atol=1e-9)


def test_emulator_full_training_quality(dev):
    """Reference tests/test_neural.py:6-15: RMSE < 0.3 std on the 5-D radial
    rank target; stopping epoch in the reference's range."""
    from nautilus_amd.emulator import NeuralNetworkEmulator
    g = load_golden('emulator_D5_E1')
    emu = NeuralNetworkEmulator.train(g['x'], g['y'], n_networks=2)
    assert np.allclose(emu.mean, g['mean']) and np.allclose(emu.scale,
                                                            g['scale'])
    pred = emu.predict(g['x'])
    assert np.sqrt(np.mean((pred - g['y'])**2)) < 0.3 * np.std(g['y'])
    n_ref = int(g['n_iter_0'])
    assert 11 <= emu.neural_networks[0].n_iter_ <= 10000
    assert 0.2 * n_ref <= emu.neural_networks[0].n_iter_ <= 5 * n_ref
    # early epochs of network 0 follow sklearn's own loss curve
    assert np.allclose(emu.neural_networks[0].loss_curve_[:5],
                       g['loss_curve_0'][:5], rtol=1e-6)


def test_benchmark_likelihoods_match_numpy(dev):
    """nb_loglike_rosenbrock / nb_loglike_funnel (BASELINE configs C3 / C5;
    the funnel is the D-dimensional form of the reference's
    tests/test_sampler.py:311-314) against numpy / scipy on 10^5 points, and
    the Gaussian / mixture likelihoods of C1, C2, C4 against scipy."""
    from scipy.stats import multivariate_normal, norm
    from nautilus_amd import (FunnelLikelihood, GaussianLikelihood,
                              GaussianMixtureLikelihood, RosenbrockLikelihood)
    rng = np.random.default_rng(0)
    for d in (2, 3, 30, 100, 128):
        u = rng.random((100000 if d <= 30 else 20000, d))
        ros = RosenbrockLikelihood(d)
        x = 10 * u - 5
        want = -np.sum(100 * (x[:, 1:] - x[:, :-1]**2)**2 +
                       (1 - x[:, :-1])**2, axis=1)
        got = ros(torch.from_numpy(u).cuda()).cpu().numpy()
        assert np.allclose(got, want, rtol=1e-12, atol=0)   # fma contraction
        assert np.allclose(ros.numpy(u), want, rtol=1e-13)
        assert np.array_equal(ros(u), got)          # numpy in -> numpy out
        fun = FunnelLikelihood(d)
        s = np.exp(20 * (u[:, 0] - 0.5)) / 100
        want = norm.logpdf(u[:, 0], 0.5, 0.1) + np.sum(
            norm.logpdf(u[:, 1:], 0.5, s[:, None]), axis=1)
        got = fun(torch.from_numpy(u).cuda()).cpu().numpy()
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
        assert np.allclose(fun.numpy(u), want, rtol=1e-12, atol=1e-9)
    # empty batch
    assert RosenbrockLikelihood(4)(torch.empty((0, 4), dtype=torch.float64,
                                               device='cuda')).shape == (0,)
    d = 20
    cov = 0.05**2 * (0.5 * np.ones((d, d)) + 0.5 * np.eye(d))
    u = rng.random((50000, d)) * 0.2 + 0.4
    g = GaussianLikelihood(np.full(d, 0.5), cov)
    want = multivariate_normal(np.full(d, 0.5), cov).logpdf(u)
    assert np.allclose(g(torch.from_numpy(u).cuda()).cpu().numpy(), want,
                       rtol=1e-11, atol=1e-9)
    means = 0.25 + 0.5 * np.random.default_rng(3).random((4, 50))
    mix = GaussianMixtureLikelihood(means, 0.02)
    u = means[rng.integers(0, 4, 20000)] + 0.02 * rng.normal(size=(20000, 50))
    from scipy.special import logsumexp
    want = logsumexp([multivariate_normal(m, 0.02**2 * np.eye(50)).logpdf(u)
                      for m in means], axis=0) - np.log(4)
    assert np.allclose(mix(torch.from_numpy(u).cuda()).cpu().numpy(), want,
                       rtol=1e-11, atol=1e-8)


def test_comm_c_abi_single_rank(dev):
    """nb_comm_* (RCCL through the C ABI): a one-rank communicator -- the
    all-gather is a copy, the all-reduce the identity; rank keys as in
    parallel.rank_key.  (More ranks need one GPU per rank: the driver's
    multi-GPU run; the collective pattern itself is covered by the gloo
    tests.)"""
    import ctypes as C
    from nautilus_amd import _lib, parallel
    lib = _lib.load()
    uid = (C.c_uint8 * 128)()
    if lib.nb_comm_unique_id(uid) != 0:
        pytest.skip('RCCL not available: ' +
                    lib.nb_last_error().decode())
    comm = C.c_void_p()
    _lib.check(lib.nb_comm_init(0, 1, uid, C.byref(comm)))
    try:
        x = torch.arange(1000, dtype=torch.float64, device='cuda') * 0.5
        y = torch.zeros_like(x)
        _lib.check(lib.nb_comm_allgather_f64(
            comm, C.c_void_p(x.data_ptr()), 1000, C.c_void_p(y.data_ptr()),
            None))
        c = torch.tensor([3, -7, 2**40], dtype=torch.int64, device='cuda')
        _lib.check(lib.nb_comm_allreduce_i64(comm, C.c_void_p(c.data_ptr()),
                                             3, None))
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        assert c.cpu().tolist() == [3, -7, 2**40]
    finally:
        lib.nb_comm_destroy(comm)
    for seed in (0, 123456789, 2**62 + 5):
        for r in range(8):
            assert lib.nb_comm_rank_key(seed, r) == parallel.rank_key(seed, r)
    assert lib.nb_comm_init(2, 2, uid, C.byref(comm)) != 0     # bad rank
