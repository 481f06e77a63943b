"""The live set of the exploration phase on the device: ``nb_live_append`` /
``nb_live_select`` / ``nb_live_stats`` through ``device.LivePool`` and
``Sampler._live_sums`` (reference nautilus/sampler.py:1147-1190, which sorts
every stored log L on every iteration).

Kernel level: threshold and counts exact against a numpy sort, the per-shell
sums to 1e-12, with ties at the threshold, -inf batches, the overflow /
rebuild path and plateaus larger than the pool's initial capacity.  Sampler
level: ``f_live`` / ``log_v_live`` against the REFERENCE's own values at nine
points of a reference run (tests/golden/liveset.npz, make_golden.py)."""

import numpy as np
import pytest
from scipy.special import logsumexp

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def gpu_only():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=float)).cuda()


def _check_selection(pool, allv, k):
    thr, n_gt, n_eq = pool.select()
    if len(allv) >= k:
        want = np.sort(allv)[-k]
        assert thr == want
        assert n_gt == np.sum(allv > want)
        assert n_eq == np.sum(allv == want)
    else:
        assert thr == -np.inf
    return thr


def test_selection_and_shell_sums_against_numpy():
    """30 batches of ragged size: the k-th largest value, the counts above /
    at it and the per-shell (count, logsumexp, ties) rows."""
    from nautilus_amd import device
    rng = np.random.default_rng(0)
    k = 500
    vals = []
    pool = device.LivePool(k)
    for it in range(30):
        b = rng.normal(size=rng.integers(1, 3000)) * 3
        if it % 7 == 3:
            b[:50] = b[0]                    # ties inside a batch
        if it == 5:
            b[:] = -np.inf                   # a batch outside the support
        if it == 12:
            b[:20] = np.sort(np.concatenate(vals))[-k]   # ties AT the cut
        vals.append(b)
        pool.add(_dev(b))
        allv = np.concatenate(vals)
        thr = _check_selection(pool, allv, k)
        rows = pool.shell_stats([_dev(v) for v in vals[-3:]])
        for row, v in zip(rows, vals[-3:]):
            assert row[0] == np.sum(v > thr)
            assert row[2] == np.sum(v == thr)
            if row[0] > 0:
                assert abs(row[1] - logsumexp(v[v > thr])) < 1e-12
            else:
                assert row[1] == -np.inf


def test_fewer_values_than_places_and_all_minus_inf():
    from nautilus_amd import device
    pool = device.LivePool(100)
    pool.add(_dev(np.full(40, -np.inf)))
    assert pool.select() == (-np.inf, 0, 40)
    pool.add(_dev(np.arange(59.0)))
    assert pool.select()[0] == -np.inf            # 99 values < 100 places
    pool.add(_dev(np.array([7.5])))
    thr, n_gt, n_eq = pool.select()
    assert thr == -np.inf and n_eq == 40 and n_gt == 60
    pool.add(_dev(np.array([3.25])))
    thr, n_gt, n_eq = pool.select()               # 101 values: one -inf drops
    assert thr == -np.inf and n_eq == 40          # the ties stay in the pool


def test_rebuild_from_many_tensors():
    """Construction from everything a sampler holds (after a new bound, a
    resume or an unpickle): 10^6 values in five tensors."""
    from nautilus_amd import device
    rng = np.random.default_rng(1)
    big = [rng.normal(size=200000) for _ in range(5)]
    pool = device.LivePool(2000, [_dev(b) for b in big])
    _check_selection(pool, np.concatenate(big), 2000)
    # and it keeps working incrementally afterwards
    more = rng.normal(size=5000) + 2.0
    pool.add(_dev(more))
    _check_selection(pool, np.concatenate(big + [more]), 2000)


def test_overflow_is_reported_not_silent():
    """Appending more values than the pool holds between two selections sets
    the overflow flag; ``select`` raises instead of returning a threshold
    computed from a truncated pool."""
    from nautilus_amd import device
    pool = device.LivePool(10)
    room = pool.cap
    pool.add(_dev(np.arange(float(room + 1000))))
    with pytest.raises(OverflowError):
        pool.select()


def test_plateau_larger_than_the_initial_capacity():
    """A likelihood plateau keeps every tied value in the pool (n_gt + n_eq,
    not n_live): 200 000 equal values at n_live = 2000 exceed the 4 k + 2^17
    places the pool starts with."""
    from nautilus_amd import device
    ties = np.zeros(200000)
    pool = device.LivePool(2000, [_dev(ties)])
    assert pool.select() == (0.0, 0, 200000)
    assert pool.cap >= 200000
    pool.add(_dev(np.zeros(4096)))
    pool.add(_dev(np.full(10, 1.0)))
    assert pool.select() == (0.0, 10, 204096)


def _sampler_from_snapshot(g, k):
    """A Sampler holding exactly the shells of snapshot ``k`` (no bounds are
    evaluated by the live-set code; it needs their number only)."""
    from nautilus_amd import Sampler, sampler as sm
    s = Sampler(lambda x: x, lambda x: 0.0, n_dim=3, n_live=int(g['n_live']))
    n_shell = len(g['s%d_shell_n' % k])
    s.bounds = [None] * int(g['s%d_n_bounds' % k])
    s.shell_n = g['s%d_shell_n' % k].copy()
    s.shell_log_v = g['s%d_shell_log_v' % k].copy()
    s.log_l = [g['s%d_log_l_%d' % (k, i)] for i in range(n_shell)]
    s.shell_log_l = np.array([
        logsumexp(ll) - np.log(len(ll)) if len(ll) else np.nan
        for ll in s.log_l])
    s._ll_dev = []
    for ll in s.log_l:
        grow = sm._Grow()
        if len(ll):
            grow.append(_dev(ll))
        s._ll_dev.append(grow)
    return s


def test_f_live_and_log_v_live_match_the_reference():
    """sampler.py:1147-1190 at nine points of a reference run (no ties in
    this run, so the reference's ``argsort`` subset is unique)."""
    g = load_golden('liveset')
    assert int(g['n_snap']) >= 5
    for k in range(int(g['n_snap'])):
        s = _sampler_from_snapshot(g, k)
        assert abs(s.log_z - float(g['s%d_log_z' % k])) < 1e-12
        assert abs(s.f_live - float(g['s%d_f_live' % k])) < 1e-12 * max(
            1.0, float(g['s%d_f_live' % k]))
        assert abs(s.log_v_live - float(g['s%d_log_v_live' % k])) < 1e-12


def test_live_sums_survive_a_pool_overflow():
    """``Sampler._live_sums`` rebuilds the pool from the shells when many
    batches were appended without a selection in between."""
    g = load_golden('liveset')
    k = int(g['n_snap']) - 1
    s = _sampler_from_snapshot(g, k)
    want = (s.f_live, s.log_v_live)
    # flood the pool behind the sampler's back: values below every stored
    # log L would normally be filtered by the threshold, these are not
    s._live.add(_dev(np.full(s._live.cap + 10, np.inf)))
    s._live.counts[2] = 1                   # what nb_live_append reports
    assert (s.f_live, s.log_v_live) == want


def test_constant_likelihood_at_default_sizes():
    """tests/test_sampler.py:334-348 at the DEFAULT n_live = 2000 and f_live
    = 0.01: ~200 000 tied points pass through the live pool, more than it
    starts with; the reference finishes this case with log Z = 0 and the unit
    cube as the only bound."""
    from nautilus_amd import Sampler
    s = Sampler(lambda x: x, lambda x: np.zeros(len(x)), 2, seed=0,
                vectorized=True, n_batch=8192)
    assert s.run(n_eff=0)
    assert np.isclose(s.log_z, 0)
    assert len(s.bounds) == 1
    assert s.n_like >= 190000
