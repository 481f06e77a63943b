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

Extra objects on the JSON line: ``roofline`` (dominant kernel of the timed
region, HIP-event timed), ``roofline_contains`` (the north star's streaming
Ellipsoid.contains kernel), ``cpu_baseline`` (the CPU oracle continuing the
same sampler state on one host core for a bounded time).
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
    p.add_argument('--n-batch-setup', type=int, default=8192,
                   help='batch size while the bounds are built and every '
                        'shell receives its first batch (untimed setup)')
    p.add_argument('--n-networks', type=int, default=4)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--cpu-seconds', type=float, default=20.0)
    p.add_argument('--explore-timeout', type=float, default=1500.0)
    p.add_argument('--no-cpu-baseline', action='store_true')
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
    p.add_argument('--broadcast-state', action='store_true',
                   help='build the bounds on rank 0 only and broadcast the '
                        'sampler (default: every rank repeats the '
                        'deterministic exploration; the broadcast is also the '
                        'automatic fallback if the replicas disagree)')
    return p.parse_args()


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


def cpu_baseline(sampler, like_numpy, seconds):
    """ESS/s of the CPU oracle continuing the same state, one core,
    n_batch = 100 (the reference's default batch)."""
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        o = oracle_sampler_from(sampler, like_numpy)
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
    return dict(value=(o.n_eff - n0) / dt, unit='effective samples/s',
                cores=1, kind='port',
                sample='%d add_samples steps of n_batch=100 on the same '
                       'explored %d-bound state, %.1f s, oracle/ numpy '
                       'restatement of the reference' %
                       (steps, len(o.bounds), dt),
                points_per_s=(o.n_like - like0) / dt,
                proposals_per_s=(prop1 - prop0) / dt)


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

    def broadcast_from_rank0(s):
        box = [s if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if rank != 0:
            s = box[0]
            s.comm = comm
        return s

    if comm is None:
        explore()
    elif args.broadcast_state:
        if rank == 0:
            explore()
        sampler = broadcast_from_rank0(sampler)
    else:
        # bound construction is "replicas only": identical seeds and
        # deterministic kernels give identical bounds on every rank
        explore()
        try:
            comm.assert_identical([sampler.log_z or 0.0, sampler.n_like,
                                   len(sampler.bounds)], 'cuda',
                                  'exploration state')
        except RuntimeError as err:
            if rank == 0:
                print('replicas differ (%s); broadcasting rank 0' % err,
                      file=sys.stderr)
            sampler = broadcast_from_rank0(sampler)
    setup_s = time.time() - t_setup
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
    roctx_region(True)
    mallocs0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
    with device.EvalCounters() as counters, device.KernelTimer() as ktimer:
        t0 = time.time()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.time() - t0
    roctx_region(False)
    mallocs = torch.cuda.memory_stats().get('num_device_alloc', 0) - mallocs0
    if comm is not None:
        comm.barrier()
        dt = comm.max_float(dt, 'cuda')
    n_eff1, n_like1, prop1 = sampler.n_eff, sampler.n_like, proposals()
    kernels = ktimer.totals()
    work = counters.read()

    # ---- roofline of the dominant kernel ----------------------------------
    dominant = max(kernels, key=lambda k: kernels[k]['ms'])
    e = args.n_networks
    flops = ((work['outer_point_evals'] + work['ellipsoid_point_evals']) *
             d * (d + 1) +
             work['emulator_point_evals'] * 2.0 * (100 * d + 6020))
    ev = kernels.get('nb_eval_kernel', dict(ms=0.0, launches=0))
    achieved_tf = flops / (ev['ms'] * 1e-3) / 1e12 if ev['ms'] > 0 else 0.0
    # HBM bytes per launch: PMC counters cannot be collected from inside this
    # process; profiles/tools/bench_traffic.sh measures them in separate
    # rocprofv3 --pmc passes of this very command (FETCH_SIZE x 2 on gfx950,
    # WRITE_SIZE) and commits the result.  Only quoted for that workload.
    traffic, traffic_src = None, None
    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                            'profiles', 'r01', 'bench_eval_traffic.json')
    default_workload = (d == 50 and args.n_live == 2000 and e == 4 and
                        args.n_batch == 65536 and world == 1 and
                        not args.host_likelihood)
    if default_workload and os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            traffic = float(json.load(fh)['hbm_bytes_per_launch'])
        traffic_src = ('profiles/r01/bench_eval_traffic.json (separate '
                       'rocprofv3 --pmc passes of this command)')
    roofline = dict(
        kernel='nb_eval_kernel', bound='mfma', achieved=achieved_tf,
        peak=FP64_MFMA_PEAK_TF, unit='TFLOP/s',
        frac=achieved_tf / FP64_MFMA_PEAK_TF, traffic=traffic,
        traffic_unit='bytes/launch', traffic_source=traffic_src,
        launches=ev['launches'],
        avg_launch_ms=ev['ms'] / max(1, ev['launches']),
        algorithmic_flops_per_launch=flops / max(1, ev['launches']),
        # every proposal is read once (8 D bytes) and flagged (1 byte);
        # the shell-exclusion launches add the accepted points again
        algorithmic_bytes_per_launch=(
            ((prop1 - prop0) / world + 2.0 * (n_like1 - n_like0) / world)
            * (8 * d + 1) / max(1, ev['launches'])),
        point_evals=work, dominant_by_time=dominant,
        time_share={k: v['ms'] for k, v in kernels.items()})

    out = dict(
        metric='effective posterior samples/sec + |dlogZ| vs analytic, '
               '50-dim Gaussian',
        value=(n_eff1 - n_eff0) / dt, unit='effective samples/s',
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
        proposals_per_s=(prop1 - prop0) / dt,
        full_run=dict(wall_s=setup_s + fill_s + dt,
                      ess_per_s=n_eff1 / (setup_s + fill_s + dt)),
        setup_breakdown={k: round(v, 2) for k, v in sampler.timing.items()},
        roofline=roofline)

    if rank == 0:
        # the north star's named kernel: streaming Ellipsoid.contains
        nb = sampler.bounds[-1].neural_bounds[0].outer_bound
        ell = Ellipsoid.from_params(nb.c, nb.B, nb.B_inv, nb.A)
        n_pts = 1 << 24
        x = torch.rand((n_pts, d), dtype=torch.float64, device='cuda')
        x[::2] = torch.from_numpy(nb.c).cuda() + 3.0 * (
            x[::2] - 0.5) * float(np.sqrt(np.mean(np.diag(nb.B)**2)))
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
            kernel='nb_ell_stream_kernel', bound='hbm', achieved=gbs,
            peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS,
            traffic=None, points=n_pts, bytes_per_point=8 * d + 1,
            avg_launch_ms=ms, inside_fraction=float(mask.double().mean()))
        del x, mask
        out['mfma_f64_probe_tflops'] = device.mfma_f64_peak(20000)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sampler, like.numpy,
                                               args.cpu_seconds)
            out['cpu_baseline']['host_cores_available'] = os.cpu_count()
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
