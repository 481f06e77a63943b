#!/usr/bin/env python
"""Benchmark of the nautilus shell-filling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "effective posterior samples/sec + |dlogZ| vs
analytic, 50-dim Gaussian"): 50-D Gaussian, mu = 0.5,
sigma = 0.05, identity prior (analytic log Z = 0), n_live = 2000,
n_networks = 4.  The bound hierarchy is built first (exploration phase,
untimed setup, reported as ``setup_s``); a STEP is then one pass of the hot
path over one batch: pick the shell (sampler.py:489-491), draw proposals from
its bound, drop points inside later bounds, evaluate the likelihood, update
the importance-weight statistics -- ``Sampler.add_samples``.  ``value`` is the
growth of the effective sample size over the K timed steps divided by the
time (max over ranks).  With N GPUs every step fills N x n_batch points
(weak scaling): each rank draws its share, one RCCL all-gather per step.

``value`` is the rate of the hot loop (sampling-phase steps);
``value_full_run`` is SURVEY.md section 8d's end-to-end figure, n_eff divided
by the whole wall time including the exploration phase that builds the bounds
(emulator training, MVEE, mixture fits) -- with N GPUs the exploration is
sharded too (every batch and the emulator networks are dealt out over the
ranks).

Extra objects on the JSON line: ``roofline`` (dominant kernel of the timed
region, HIP-event timed; ``traffic`` is measured only with ``--pmc-traffic``,
which re-runs this command under ``rocprofv3 --pmc`` in child processes),
``roofline_contains`` (the north star's streaming Ellipsoid.contains kernel),
``roofline_draw`` (the proposal draw: proposals/s, bytes written against HBM,
fp64 vector operations against the peak),
``cpu_baseline`` (the CPU oracle continuing the same sampler state through
the reference's multiprocess-pool path on all host cores, and on one core,
for a bounded time).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TF = 78.6      # MI355X datasheet FP64 matrix
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--dim', type=int, default=50)
    p.add_argument('--n-live', type=int, default=2000)
    p.add_argument('--n-batch', type=int, default=65536,
                   help='shell points per timed step and GPU')
    p.add_argument('--n-batch-setup', type=int, default=16384,
                   help='batch size while the bounds are built and every '
                        'shell receives its first batch (untimed setup).  '
                        'Chosen by end-to-end time (DESIGN.md section 11): '
                        'larger batches mean fewer, thicker shells -- 94 / 80 '
                        '/ 72 / 51 / 44 / 41 bounds at 2048 ... 32768 -- and '
                        'the whole run is shortest at 16384; log Z stays '
                        'within 0.003 of the analytic value throughout')
    p.add_argument('--n-networks', type=int, default=4)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--cpu-seconds', type=float, default=14.0,
                   help='wall time of the pooled CPU baseline leg (the '
                        'single-core leg gets half of it)')
    p.add_argument('--cpu-cores', type=int, default=0,
                   help='workers of the pooled CPU baseline (0 = all host '
                        'cores)')
    p.add_argument('--pmc-traffic', action='store_true',
                   help='measure roofline.traffic: re-run this command under '
                        'rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in '
                        'separate child passes; adds two full runs)')
    p.add_argument('--explore-timeout', type=float, default=1500.0)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--timed-region-only', action='store_true',
                   help='skip the kernel micro-benchmarks behind the timed '
                        'steps (the --pmc-traffic child passes: their '
                        'dispatches would be taken for the last ones of the '
                        'timed region)')
    p.add_argument('--host-likelihood', action='store_true',
                   help='evaluate the likelihood with numpy on the host '
                        '(points cross PCIe both ways every step): the '
                        'PCIe-inclusive rate quoted in DESIGN.md, never the '
                        'headline value')
    p.add_argument('--backend', default='nccl',
                   help="torch.distributed backend ('nccl' = RCCL; 'gloo' "
                        "only for functional tests)")
    p.add_argument('--same-device', action='store_true',
                   help='functional test: all ranks share cuda:0')
    p.add_argument('--force-comm', action='store_true',
                   help='functional test: run the collective code path even '
                        'with a single rank')
    return p.parse_args()


class NumpyGaussian:
    """Isotropic Gaussian log-density in plain numpy (picklable: the pooled
    CPU baseline ships it to its workers)."""

    def __init__(self, mean, sigma):
        self.mean, self.sigma = np.asarray(mean, float), float(sigma)

    def __call__(self, x):
        x = np.atleast_2d(x)
        d = x.shape[1]
        return (-0.5 * np.sum(((x - self.mean) / self.sigma)**2, axis=1) -
                d * np.log(self.sigma * np.sqrt(2 * np.pi)))


_THREAD_VARS = ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')


class _OraclePool:
    """``map`` / ``size`` over a multiprocessing pool (what the reference's
    NautilusPool offers, pool.py:65-107).  Fork server: this process has
    initialised the HIP runtime."""

    def __init__(self, n_workers):
        import multiprocessing
        self.size = n_workers
        # one BLAS / OpenMP thread per worker, as the reference's CI and docs
        # prescribe for pools (the fork server inherits the environment)
        saved = {k: os.environ.get(k) for k in _THREAD_VARS}
        os.environ.update({k: '1' for k in _THREAD_VARS})
        try:
            self._pool = multiprocessing.get_context('forkserver').Pool(
                n_workers)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self._pool.map(_warm, range(4 * n_workers))      # start the workers

    def map(self, func, iterable):
        return self._pool.map(func, iterable)

    def close(self):
        self._pool.terminate()


def _warm(i):
    import oracle.bounds_oracle, oracle.mlp_oracle     # noqa: F401
    return i


def oracle_sampler_from(sampler, like_numpy):
    """CPU-oracle copy of the explored sampler state (cpu_baseline leg only:
    the oracle is the checker / baseline, never the measured product)."""
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    from oracle.sampler_oracle import OSampler

    rng = np.random.default_rng(12345)

    def ell(e):
        return bo.OEllipsoid.from_params(e.c, e.B, e.B_inv, e.A, rng=rng)

    def convert(b):
        if not hasattr(b, 'outer_bound'):
            return bo.OCube(b.n_dim, rng=rng)
        members = [bo.OMixture.from_params(
            m.dim_cube, None if m.ellipsoid is None else ell(m.ellipsoid),
            rng=rng) for m in b.outer_bound.bounds]
        outer = bo.OUnion.from_members(members, unit=True, rng=rng)
        outer.log_v_all = np.array(b.outer_bound.log_v_all)
        outer.n_sample = int(b.outer_bound.n_sample)
        outer.n_reject = int(b.outer_bound.n_reject)
        neural = []
        for nb in b.neural_bounds:
            o = bo.ONeural()
            o.n_dim = nb.n_dim
            o.outer_bound = ell(nb.outer_bound)
            o.score_predict_min = nb.score_predict_min
            o.emulator = None
            if nb.emulator is not None:
                o.emulator = mo.Emulator.from_weights(
                    nb.emulator.mean, nb.emulator.scale,
                    [(n.coefs_, n.intercepts_)
                     for n in nb.emulator.neural_networks])
            neural.append(o)
        out = bo.ONautilus.from_parts(outer, neural, rng=rng)
        out.n_sample = int(b.n_sample)
        out.n_reject = int(b.n_reject)
        return out

    o = OSampler(lambda x: x, like_numpy, n_dim=sampler.n_dim,
                 n_live=sampler.n_live, n_networks=sampler.n_networks,
                 vectorized=True, n_batch=100, seed=1)
    o.rng = rng
    o.bounds = [convert(b) for b in sampler.bounds]
    o.points = sampler.points
    o.log_l = [np.array(ll) for ll in sampler.log_l]
    for key in ('shell_n', 'shell_n_sample', 'shell_n_eff', 'shell_log_l_min',
                'shell_log_l', 'shell_log_v', 'shell_n_sample_exp',
                'shell_end_exp'):
        setattr(o, key, np.array(getattr(sampler, key)))
    o.explored = True
    o._discard = sampler.discard_exploration
    o.n_like = sampler.n_like
    return o


def _oracle_leg(sampler, like_numpy, seconds, pool, n_batch):
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        o = oracle_sampler_from(sampler, like_numpy)
        o.pool = pool
        o.n_batch = n_batch
        for s in range(len(o.log_l)):
            o.update_shell_info(s)
        n0, like0 = o.n_eff, o.n_like
        prop0 = sum(b.outer_bound.n_sample for b in o.bounds[1:])
        t0 = time.time()
        steps = 0
        while time.time() - t0 < seconds:
            shell = int(np.argmax(o.shell_log_l + o.shell_log_v -
                                  0.5 * np.log(o.shell_n) -
                                  0.5 * np.log(o.shell_n_eff)))
            o.add_samples(shell)
            steps += 1
        dt = time.time() - t0
        prop1 = sum(b.outer_bound.n_sample for b in o.bounds[1:])
    return dict(value=(o.n_eff - n0) / dt, steps=steps, seconds=dt,
                points_per_s=(o.n_like - like0) / dt,
                proposals_per_s=(prop1 - prop0) / dt, n_bounds=len(o.bounds))


def cpu_baseline(sampler, like_numpy, seconds, cores):
    """ESS/s of the CPU oracle continuing the same explored state through
    the reference's multiprocess-pool path (north star: "the reference's own
    multiprocess-pool CPU path timed on the node's host cores"): bounds
    replicate themselves over the pool in ``sample`` (bounds/nautilus.py:
    223-237), the vectorized likelihood gets one chunk per worker
    (sampler.py:860-873), n_batch = the reference's default for that pool
    (the smallest multiple of its size >= 100, sampler.py:300-303).  The
    one-core figure (pool=None, n_batch=100) rides along."""
    cores = cores or os.cpu_count() or 1
    single = _oracle_leg(sampler, like_numpy, 0.5 * seconds, None, 100)
    out = dict(unit='effective samples/s', kind='port',
               host_cores_available=os.cpu_count(), single_core=dict(
                   value=single['value'], cores=1,
                   points_per_s=single['points_per_s'],
                   proposals_per_s=single['proposals_per_s'],
                   sample='%d add_samples steps of n_batch=100, pool=None, '
                          '%.1f s' % (single['steps'], single['seconds'])))
    if cores == 1:
        out.update(value=single['value'], cores=1,
                   sample=out['single_core']['sample'])
        return out
    t0 = time.time()
    pool = _OraclePool(cores)
    start_s = time.time() - t0
    try:
        n_batch = (100 // cores + (100 % cores != 0)) * cores
        leg = _oracle_leg(sampler, like_numpy, seconds, pool, n_batch)
        # the reference's default batch hands every worker about one point
        # per map(); what the same pool path does when each worker gets ~100
        # points per call rides along (n_batch = 100 x cores)
        big = _oracle_leg(sampler, like_numpy, 0.5 * seconds, pool,
                          100 * cores)
    finally:
        pool.close()
    out['large_batch'] = dict(
        value=big['value'], cores=cores, n_batch=100 * cores,
        points_per_s=big['points_per_s'],
        proposals_per_s=big['proposals_per_s'],
        sample='%d add_samples steps of n_batch=%d, pool=%d, %.1f s' % (
            big['steps'], 100 * cores, cores, big['seconds']))
    out.update(
        value=leg['value'], cores=cores, points_per_s=leg['points_per_s'],
        proposals_per_s=leg['proposals_per_s'], pool_start_s=start_s,
        sample='%d add_samples steps of n_batch=%d on the same explored '
               '%d-bound state with pool=%d (multiprocessing, fork server), '
               '%.1f s; oracle/ numpy restatement of the reference' %
               (leg['steps'], n_batch, leg['n_bounds'], cores,
                leg['seconds']))
    return out


def pmc_traffic(argv):
    """HBM bytes of the bound-evaluation kernels (nb_eval_fast_kernel,
    nb_cand_kernel) over the timed region of this very command: two child
    runs under ``rocprofv3 --pmc`` (FETCH_SIZE and WRITE_SIZE need separate
    passes on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"; FETCH_SIZE
    counts 64 B per 128-B request of wide loads -> x 2).  The timed region's
    dispatches are the last ``roofline.kernel_dispatches`` of each kernel in
    the child run.  Returns a dict or None if the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    totals = {}
    calls = None
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        tmp = tempfile.mkdtemp(prefix='nb_pmc_', dir='/tmp')
        cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace',
               '--output-format', 'csv', '-d', tmp, '-o', 'pmc', '--',
               sys.executable, os.path.abspath(__file__)] + [
                   a for a in argv if a != '--pmc-traffic'] + [
                   '--no-cpu-baseline', '--timed-region-only']
        env = dict(os.environ, TMPDIR='/tmp')
        try:
            child = subprocess.run(cmd, cwd='/tmp', env=env, check=True,
                                   timeout=1800, capture_output=True,
                                   text=True)
            line = [ln for ln in child.stdout.splitlines()
                    if ln.startswith('{"metric"')][-1]
            roof = json.loads(line)['roofline']
            n_timed, calls = roof['kernel_dispatches'], roof['launches']
        except Exception:
            shutil.rmtree(tmp, ignore_errors=True)
            return None
        rows = []
        for path in glob.glob(os.path.join(tmp, '**', '*counter_collection'
                                                      '.csv'),
                              recursive=True):
            with open(path) as fh:
                rows += [r for r in csv.DictReader(fh)
                         if r.get('Counter_Name') == counter]
        shutil.rmtree(tmp, ignore_errors=True)
        total = 0.0
        for name, count in n_timed.items():
            mine = sorted((r for r in rows if name in r.get('Kernel_Name', '')),
                          key=lambda r: int(r['Dispatch_Id']))
            if count > len(mine):
                return None
            total += sum(float(r['Counter_Value']) for r in mine[-count:]) \
                if count else 0.0
        # FETCH_SIZE / WRITE_SIZE are reported in kilobytes; FETCH_SIZE
        # tallies the 128-B requests of wide loads at 64 B on gfx950 (x 2)
        totals[counter] = total * 1024.0 * (2.0 if counter == 'FETCH_SIZE'
                                            else 1.0)
    return dict(bytes_per_launch=(totals['FETCH_SIZE'] +
                                  totals['WRITE_SIZE']) / max(1, calls),
                read_bytes_per_launch=totals['FETCH_SIZE'] / max(1, calls),
                write_bytes_per_launch=totals['WRITE_SIZE'] / max(1, calls),
                launches=calls)


def roctx_region(resume):
    """Under ``rocprofv3 --selected-regions`` only the timed region is
    traced; a no-op otherwise."""
    import ctypes
    for path in ('librocprofiler-sdk-roctx.so',
                 '/opt/rocm/lib/librocprofiler-sdk-roctx.so'):
        try:
            lib = ctypes.CDLL(path)
        except OSError:
            continue
        (lib.roctxProfilerResume if resume else lib.roctxProfilerPause)(0)
        return


def two_stage_roofline(d, k=4, e=4, n=1 << 20, reps=10):
    """``DeviceBound.accept`` of a K = M = 4, E = 4 bound through the staged
    route: flops from the evaluation counters over the median call."""
    import torch
    from nautilus_amd import bounds as nbd, device
    from nautilus_amd.emulator import NeuralNetworkEmulator, Network
    rs = np.random.RandomState(0)
    units = [d, 100, 50, 20, 1]

    def net():
        coefs, icpts = [], []
        for a, b in zip(units[:-1], units[1:]):
            lim = np.sqrt(6.0 / (a + b))
            coefs.append(rs.uniform(-lim, lim, (a, b)))
            icpts.append(rs.uniform(-lim, lim, b))
        return Network(coefs, icpts)
    members, neural = [], []
    for c in 0.25 + 0.5 * rs.rand(k, d):
        a = rs.normal(size=(d, d)) * 0.02 / np.sqrt(d) + 0.04 * np.eye(d)
        cov = a @ a.T
        b_mat = np.linalg.cholesky(cov)
        args = (c, b_mat, np.linalg.inv(b_mat), np.linalg.inv(cov))
        members.append(nbd.Ellipsoid.from_params(*args))
        emu = NeuralNetworkEmulator.from_weights(
            np.zeros(d), np.ones(d), [net() for _ in range(e)])
        neural.append(nbd.NeuralBound.from_parts(
            nbd.Ellipsoid.from_params(*args), emu, 0.0))
    outer = nbd.Union.from_members(members, unit=True)
    outer.log_v_all = np.array([m.log_v for m in members])
    dev = nbd.NautilusBound.from_parts(
        outer, neural, rng=np.random.default_rng(1)).device_bound()
    x = dev.propose(7, 0, n)
    dev.accept(7, 0, x)
    each = []
    with device.EvalCounters() as counters:
        torch.cuda.synchronize()
        for _ in range(reps):
            t0 = time.perf_counter()
            dev.accept(7, 0, x)
            torch.cuda.synchronize()
            each.append(time.perf_counter() - t0)
        work = counters.read()
    flops = ((work['outer_point_evals'] + work['ellipsoid_point_evals']) *
             d * (d + 1) + work['emulator_point_evals'] * 2.0 *
             (100 * d + 6020)) / reps
    ms = float(np.median(each)) * 1e3
    tf = flops / (ms * 1e-3) / 1e12
    return dict(kernel='nb_cand_kernel + nb_eval_fast_kernel<batch>',
                bound='mfma', achieved=tf, peak=FP64_MFMA_PEAK_TF,
                unit='TFLOP/s', frac=tf / FP64_MFMA_PEAK_TF, traffic=None,
                points=n, members=k, neural_bounds=k, networks=e, n_dim=d,
                avg_call_ms=ms, slowest_call_ms=float(max(each)) * 1e3,
                flop_per_call=flops)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.same_device:
        # several processes on one GPU: the resident training kernel owns
        # whole XCDs per process and must not be shared between processes
        os.environ['NB_TRAIN_TWO_LAUNCH'] = '1'
        # ... and the mixture fit's workgroups wait for each other: one per
        # restart where several processes fit at the same time on one GPU
        os.environ['NB_GMM_MAX_WGS'] = '1'
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1 or args.force_comm:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        wait = datetime.timedelta(seconds=args.explore_timeout + 1800)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', timeout=wait,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend, timeout=wait)
        from nautilus_amd.parallel import ShardedComm
        comm = ShardedComm()
        comm_info = comm.describe()
        if comm_info['ranks_seen'] != world:
            raise SystemExit('the communicator counts %d ranks, WORLD_SIZE=%d'
                             % (comm_info['ranks_seen'], world))

    from nautilus_amd import GaussianLikelihood, Sampler, device, unit_prior
    from nautilus_amd.bounds import Ellipsoid

    d = args.dim
    like = GaussianLikelihood(np.full(d, 0.5), np.eye(d) * 0.05**2)
    like_used = like
    if args.host_likelihood:
        def like_used(x_host):             # plain callable -> host path
            return like.numpy(x_host)
    sampler = Sampler(unit_prior, like_used, n_dim=d, n_live=args.n_live,
                      n_networks=args.n_networks,
                      n_batch=args.n_batch_setup, vectorized=True,
                      seed=args.seed, comm=comm)

    # ---- setup: build the bound hierarchy (exploration, untimed) ---------
    t_setup = time.time()

    def explore():
        sampler.run(n_eff=0, n_shell=0, discard_exploration=True,
                    timeout=args.explore_timeout)
        torch.cuda.synchronize()
        if not sampler.explored:
            raise SystemExit('exploration did not finish within %.0f s' %
                             args.explore_timeout)

    # with N ranks the exploration is sharded like everything else: every
    # batch and the emulator networks are dealt out over the ranks, the
    # (deterministic) geometric construction is replicated
    explore()
    if comm is not None:
        comm.assert_identical([sampler.log_z or 0.0, sampler.n_like,
                               len(sampler.bounds)], 'cuda',
                              'exploration state')
    setup_s = time.time() - t_setup

    # per-rank wall time by phase (host clocks): what each rank spent in
    # emulator training, the rest of the bound construction, drawing /
    # evaluating shell points, likelihoods and inside the collectives
    PHASES = ('bound_neural', 'bound_other', 'sample_shell', 'likelihood',
              'collectives')

    def phase_clock():
        t = sampler.timing
        return [t.get('bound_neural', 0.0),
                t.get('bound_decompose', 0.0) + t.get('bound_envelope', 0.0),
                t.get('sample_shell', 0.0), t.get('likelihood', 0.0),
                comm.seconds if comm is not None else 0.0]

    def per_rank(now, before=None):
        mine = [a - b for a, b in zip(now, before or [0.0] * len(now))]
        rows = comm.gather_floats(mine, 'cuda') if comm is not None else [mine]
        return [dict(zip(PHASES, (round(v, 4) for v in row))) for row in rows]
    clock_explored = phase_clock()
    rank_phases = dict(exploration=per_rank(clock_explored))
    if comm is not None:
        sampler.n_batch = args.n_batch_setup * world

    def step():
        if np.any(sampler.shell_n < 1):
            shell = int(np.flatnonzero(sampler.shell_n < 1)[0])
        else:
            shell = sampler._next_shell()
        sampler.add_samples(shell)

    def proposals():
        return sum(b.outer_bound.n_sample for b in sampler.bounds[1:])

    # every shell needs one batch before the steady-state shell selection
    # (sampler.py:482-486); part of the untimed setup
    t_fill = time.time()
    while np.any(sampler.shell_n < 1):
        step()
    torch.cuda.synchronize()
    fill_s = time.time() - t_fill
    sampler.n_batch = args.n_batch * world       # batch of the timed steps
    for _ in range(args.warmup):
        step()

    # ---- timed region -----------------------------------------------------
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize()
    n_eff0, n_like0, prop0 = sampler.n_eff, sampler.n_like, proposals()
    clock_timed = phase_clock()
    roctx_region(True)
    mallocs0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
    disp0 = dict(device.DISPATCHES)
    with device.EvalCounters() as counters, device.KernelTimer() as ktimer:
        t0 = time.time()
        for _ in range(args.steps):
            step()
        # (sharded runs: the points of the last batches are still on their
        # way to the other ranks -- part of the steps' work)
        sampler.land_points()
        torch.cuda.synchronize()
        dt = time.time() - t0
    roctx_region(False)
    mallocs = torch.cuda.memory_stats().get('num_device_alloc', 0) - mallocs0
    rank_phases['timed_steps'] = per_rank(phase_clock(), clock_timed)
    rank_dt = (comm.gather_floats([dt], 'cuda') if comm is not None
               else [[dt]])
    for row, t in zip(rank_phases['timed_steps'], rank_dt):
        row['wall'] = round(t[0], 4)
    if comm is not None:
        comm.barrier()
        dt = comm.max_float(dt, 'cuda')
    n_eff1, n_like1, prop1 = sampler.n_eff, sampler.n_like, proposals()
    kernels = ktimer.totals()
    work = counters.read()
    dispatches = {k: device.DISPATCHES[k] - disp0[k] for k in disp0}

    # ---- roofline of the dominant kernel ----------------------------------
    dominant = max(kernels, key=lambda k: kernels[k]['ms'])
    if dominant == 'bound_eval':
        # the timer brackets the bound-evaluation calls as a family; name the
        # kernel that carries them (43 of its dispatches against 3-5 of the
        # geometric kernel in the headline run)
        dominant = 'nb_eval_fast_kernel (bound evaluation family)'
    e = args.n_networks
    flops = ((work['outer_point_evals'] + work['ellipsoid_point_evals']) *
             d * (d + 1) +
             work['emulator_point_evals'] * 2.0 * (100 * d + 6020))
    ev = kernels.get('bound_eval', dict(ms=0.0, launches=0))
    achieved_tf = flops / (ev['ms'] * 1e-3) / 1e12 if ev['ms'] > 0 else 0.0
    # HBM bytes per launch: PMC counters cannot be collected from inside this
    # process.  --pmc-traffic re-runs this command under rocprofv3 --pmc
    # (separate FETCH_SIZE / WRITE_SIZE passes, only the timed region is
    # profiled) and reports the measured value; otherwise null, with the
    # latest committed measurement under profiles/ named for reference.
    import glob
    committed = sorted(glob.glob(os.path.join(
        os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*',
        'bench_pmc_traffic.json')))
    traffic, traffic_src = None, (
        'not measured in this run (use --pmc-traffic); newest committed '
        'measurement: %s' % (os.path.relpath(
            committed[-1], os.path.dirname(os.path.abspath(__file__)))
            if committed else 'none'))
    if args.pmc_traffic and rank == 0 and world == 1:
        got = pmc_traffic(sys.argv[1:])
        if got is not None:
            traffic = got['bytes_per_launch']
            traffic_src = ('rocprofv3 --pmc child passes of this command: '
                           '%.0f B read (FETCH_SIZE x 2) + %.0f B written per '
                           'launch, %d launches' % (
                               got['read_bytes_per_launch'],
                               got['write_bytes_per_launch'],
                               got['launches']))
    roofline = dict(
        kernel='nb_eval_fast_kernel (+ nb_cand_kernel)', kernel_note=(
            'bound evaluation of the timed steps: proposal acceptance in '
            'nb_eval_fast_kernel (fused cube test, ellipsoid, emulators); '
            'shell exclusion against the later bounds (any number) = '
            'nb_cand_kernel (geometric tests of every point against the whole '
            'list, candidate lists per (bound, neural bound)) + ONE batched '
            'nb_eval_fast_kernel launch over all candidates; calls and '
            'HIP-event time are those of whole queries, kernel_dispatches '
            'counts the launches of each kernel'),
        bound='mfma', achieved=achieved_tf,
        peak=FP64_MFMA_PEAK_TF, unit='TFLOP/s',
        frac=achieved_tf / FP64_MFMA_PEAK_TF, traffic=traffic,
        traffic_unit='bytes/launch', traffic_source=traffic_src,
        launches=ev['launches'], kernel_dispatches=dispatches,
        avg_launch_ms=ev['ms'] / max(1, ev['launches']),
        algorithmic_flops_per_launch=flops / max(1, ev['launches']),
        # every proposal is read once (8 D bytes) and flagged (1 byte);
        # the shell-exclusion launches add the accepted points again
        algorithmic_bytes_per_launch=(
            ((prop1 - prop0) / world + 2.0 * (n_like1 - n_like0) / world)
            * (8 * d + 1) / max(1, ev['launches'])),
        point_evals=work, dominant_by_time=dominant,
        time_share={k: v['ms'] for k, v in kernels.items()},
        # committed counter passes of the same steps (one counter per
        # rocprofv3 pass): SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / kernel cycles
        matrix_pipe_busy='0.80 of the kernel cycles (profiles/r06/'
                         'fast_pmc.txt; not measured in this run)')

    # emulator networks per bound (M x E) against the ranks they are dealt out
    # over (network g trains on rank g mod world, reference neural.py:93-96)
    nets_per_bound = max(
        [sum(len(nbd.emulator.neural_networks) for nbd in b.neural_bounds
             if nbd.emulator is not None) for b in sampler.bounds[1:]] or [0])
    out = dict(
        metric='effective posterior samples/sec + |dlogZ| vs analytic, '
               '50-dim Gaussian',
        ranks_training=min(world, nets_per_bound),
        ranks_idle_while_training=max(0, world - nets_per_bound),
        networks_per_bound=nets_per_bound,
        value=(n_eff1 - n_eff0) / dt, unit='effective samples/s',
        value_definition='growth of the effective sample size over the K '
                         'timed sampling-phase steps / their wall time (the '
                         'hot loop); the end-to-end figure incl. the '
                         'exploration phase is value_full_run',
        value_full_run=n_eff1 / (setup_s + fill_s + dt),
        value_full_run_definition='n_eff / (exploration + first batch of '
                                  'every shell + warm-up and timed steps) '
                                  '(SURVEY.md section 8d)',
        n_gpus=world, steps=args.steps, warmup=args.warmup,
        ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
        scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
        config=dict(workload='%d-D Gaussian mu=0.5 sigma=0.05, identity '
                             'prior, sampling-phase add_samples steps' % d,
                    n_dim=d, n_live=args.n_live, n_networks=e,
                    n_batch_per_gpu=args.n_batch,
                    n_batch_setup=args.n_batch_setup,
                    n_batch_global=args.n_batch * world,
                    discard_exploration=True, seed=args.seed,
                    likelihood='host numpy (PCIe inclusive)'
                    if args.host_likelihood else 'device'),
        log_z=float(sampler.log_z), abs_dlogz=abs(float(sampler.log_z)),
        n_eff=float(n_eff1), n_like=int(n_like1),
        n_bounds=len(sampler.bounds), setup_s=setup_s, shell_fill_s=fill_s,
        points_per_s=(n_like1 - n_like0) / dt,
        device_mallocs_in_timed_region=int(mallocs),
        # refills launched ahead of the batch that needs them (sampler.py
        # _prefetch_next): launches issued, guesses that were right / wrong
        prefetch=dict(getattr(sampler, 'prefetch_stats', {})),
        proposals_per_s=(prop1 - prop0) / dt,
        full_run=dict(wall_s=setup_s + fill_s + dt,
                      ess_per_s=n_eff1 / (setup_s + fill_s + dt)),
        setup_breakdown={k: round(v, 2) for k, v in sampler.timing.items()},
        # multi-GPU runs describe themselves: what the communicator spans
        # (backend, ranks counted by an all-reduce, the device of every rank)
        # and where every rank's wall time went, by phase
        communicator=(comm_info if comm is not None else
                      dict(backend=None, ranks_seen=1, world=1)),
        per_rank_seconds=rank_phases,
        roofline=roofline)

    if rank == 0:
        # the north star's named kernel: streaming Ellipsoid.contains
        nb = sampler.bounds[-1].neural_bounds[0].outer_bound
        ell = Ellipsoid.from_params(nb.c, nb.B, nb.B_inv, nb.A)
        n_pts = 1 << 24
        # a mixed mask: half of the probe points are drawn inside the
        # ellipsoid, half uniformly in the cube (all outside at D = 50)
        x = torch.rand((n_pts, d), dtype=torch.float64, device='cuda')
        x[::2] = ell.device_bound().propose(7, 0, n_pts // 2)
        bound_dev = ell.device_bound()
        bound_dev.contains_stream(x)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        reps = 10
        ev0.record()
        for _ in range(reps):
            mask = bound_dev.contains_stream(x)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        gbs = n_pts * (8 * d + 1) / (ms * 1e-3) / 1e9
        out['roofline_contains'] = dict(
            kernel='nb_ell_stream_pipe_kernel', bound='hbm', achieved=gbs,
            peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS,
            traffic=None,
            traffic_note='committed counter passes of this kernel: '
                         'profiles/r05/fifth_session/stream_pmc.txt '
                         '(FETCH_SIZE: 0.98 x the algorithmic bytes)',
            points=n_pts, bytes_per_point=8 * d + 1,
            avg_launch_ms=ms, inside_fraction=float(mask.double().mean()))
        del x, mask
        # ... and at BASELINE configuration 5's dimension, where D (D + 1)
        # flop per 8 D + 1 bytes put it behind the fp64 MFMA roof (ridge at
        # n_dim 78), not the HBM one
        d2 = 100
        rng2 = np.random.default_rng(d2)
        a2 = rng2.normal(size=(d2, d2))
        cov2 = (a2 @ a2.T / d2 + np.eye(d2)) * 0.02
        b2 = np.linalg.cholesky(cov2)
        ell2 = Ellipsoid.from_params(0.5 * np.ones(d2), b2, np.linalg.inv(b2),
                                     np.linalg.inv(cov2)).device_bound()
        n2 = 1 << 23
        x = torch.rand((n2, d2), dtype=torch.float64, device='cuda')
        x[::2] = 0.5 + 0.6 * (x[::2] - 0.5)
        ell2.contains_stream(x)
        ev0.record()
        for _ in range(reps):
            mask = ell2.contains_stream(x)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        tf = n2 * d2 * (d2 + 1.0) / (ms * 1e-3) / 1e12
        gbs = n2 * (8 * d2 + 1) / (ms * 1e-3) / 1e9
        out['roofline_contains_d100'] = dict(
            kernel='nb_ell_stream_kernel', bound='mfma', achieved=tf,
            peak=FP64_MFMA_PEAK_TF, unit='TFLOP/s',
            frac=tf / FP64_MFMA_PEAK_TF, hbm_gbs=gbs,
            hbm_frac=gbs / HBM_PEAK_GBS, traffic=None, points=n2,
            flop_per_point=d2 * (d2 + 1), bytes_per_point=8 * d2 + 1,
            avg_launch_ms=ms, inside_fraction=float(mask.double().mean()))
        del x, mask
        # ... and the proposal draw (Ellipsoid.sample, basic.py:362-381), the
        # second kernel of the timed step: VALU work (Philox, Box-Muller
        # polynomials, the triangular product x = c + B z), 8 D bytes written
        # per proposal.  Priced against both roofs it could meet; the issue
        # counters behind the statement of what binds it are committed
        # (profiles/r06/draw_pmc.txt: SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES)
        n3 = 1 << 22
        draw_dev = ell.device_bound()
        draw_dev.propose(7, 0, n3)
        ev0.record()
        for _ in range(reps):
            xs = draw_dev.propose(7, 0, n3)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        # fp64 operations per proposal as the kernel issues them: the
        # triangular product D (D + 1), |z|^2 and the rescaling 3 D, and per
        # Box-Muller pair the polynomial log (47), sincos (49), the square
        # root (~12) and four multiplications -- nb_draw.h, counted by hand
        flop_draw = d * (d + 1.0) + 3.0 * d + 0.5 * d * 112.0
        wr = n3 * 8.0 * d / (ms * 1e-3) / 1e9
        tf = n3 * flop_draw / (ms * 1e-3) / 1e12
        out['roofline_draw'] = dict(
            kernel='nb_draw_kernel', bound='valu', proposals=n3,
            avg_launch_ms=ms, proposals_per_s=n3 / (ms * 1e-3),
            bytes_written_per_proposal=8 * d, hbm_write_gbs=wr,
            hbm_frac=wr / HBM_PEAK_GBS,
            fp64_flop_per_proposal=flop_draw, achieved=tf,
            peak=FP64_MFMA_PEAK_TF, unit='TFLOP/s (fp64 vector; the vector '
            'and the matrix peak of this part are the same 78.6)',
            frac=tf / FP64_MFMA_PEAK_TF,
            integer_work='Philox4x32-10: D / 4 calls of 10 rounds (four '
                         '32-bit multiplies + six logic operations each) per '
                         'proposal -- about as many vector instructions as '
                         'the fp64 work, not in the flop count',
            valu_busy='0.77 of the kernel cycles issue a VALU instruction '
                      'on every SIMD (SQ_ACTIVE_INST_VALU, committed pass)',
            counters='profiles/r06/draw_pmc.txt')
        del xs
        # ... and the two-stage bound evaluation (geometric stage + candidate
        # lists + one batched emulator launch) of a bound with four outer
        # members and four neural bounds at the same dimension: median of
        # synchronised calls (single calls of a process that ran kernels of
        # another n_dim before stall for tens of milliseconds while the device
        # re-ramps its clocks, profiles/r05/slow_mode_probe.txt)
        if not args.timed_region_only:
            out['roofline_two_stage_d100'] = two_stage_roofline(100)
        out['mfma_f64_probe_tflops'] = device.mfma_f64_peak(20000)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(
                sampler, NumpyGaussian(np.full(d, 0.5), 0.05),
                args.cpu_seconds, args.cpu_cores)
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
