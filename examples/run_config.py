"""End-to-end runs of the BASELINE.json configurations (SURVEY.md section 8d,
C1-C5) on one GPU:  python examples/run_config.py C4 [--n-eff 10000]

Prints one JSON line per run: wall time, log Z (and the analytic value where
one exists), effective sample size, likelihood calls, bounds.  The likelihoods
are the product's device likelihoods (nautilus_amd/likelihoods.py).
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nautilus_amd import Sampler, unit_prior  # noqa: E402
from nautilus_amd.configs import baseline_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('name')
    ap.add_argument('--n-eff', type=float, default=10000)
    ap.add_argument('--n-batch', type=int, default=None,
                    help='default: the configuration\'s own batch size')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--n-live', type=int, default=None,
                    help='default: the configuration\'s own')
    ap.add_argument('--n-networks', type=int, default=None,
                    help='default: the configuration\'s own')
    ap.add_argument('--timeout', type=float, default=np.inf)
    ap.add_argument('--counters', action='store_true',
                    help='point-evaluation counters of the bound kernels and '
                         'HIP-event time per kernel family')
    ap.add_argument('--keep-exploration', action='store_true',
                    help='the reference\'s default: the points of the '
                         'exploration phase count (discard_exploration=False)')
    ap.add_argument('--progress', default=None,
                    help='append one JSON line per new bound (and per 50 '
                         'sampling batches) to this file')
    ap.add_argument('--watchdog', type=float, default=0,
                    help='dump the Python stack and exit after this many '
                         'seconds (debugging aid)')
    args = ap.parse_args()
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    c = baseline_config(args.name)
    t0 = time.time()
    s = Sampler(unit_prior, c['likelihood'], n_dim=c['n_dim'],
                n_live=args.n_live or c['n_live'],
                n_networks=c['n_networks'] if args.n_networks is None
                else args.n_networks,
                n_batch=args.n_batch or c['n_batch'],
                vectorized=True, seed=args.seed)
    extra = {}
    discard = not args.keep_exploration
    if args.progress:
        calls = dict(add_samples=0)

        def train_stats(bound):
            nbs = getattr(bound, 'neural_bounds', [])
            emu = nbs[0].emulator if nbs else None
            st = getattr(emu, 'trainer_stats', None) or {}
            return dict(n_train=st.get('n_rows'), n_iter=st.get('n_iter'),
                        n_neural=len(nbs))

        def note(kind):
            with open(args.progress, 'a') as f:
                f.write(json.dumps(dict(
                    t=round(time.time() - t0, 1), kind=kind,
                    n_bounds=len(s.bounds), explored=bool(s.explored),
                    log_z=None if s.log_z is None else float(s.log_z),
                    n_eff=float(s.n_eff or 0.0),
                    f_live=float(s.f_live) if not s.explored and
                    s.log_z is not None else None,
                    log_v=float(s.bounds[-1].log_v),
                    **train_stats(s.bounds[-1]),
                    n_like=int(s.n_like), n_dead=int(s.n_dead_bounds),
                    train_s=round(s.timing.get('bound_neural', 0.0), 1),
                    shell_s=round(s.timing.get('sample_shell', 0.0), 1))) +
                    '\n')
        add_bound, add_samples = s.add_bound, s.add_samples

        def add_bound_logged(*a, **k):
            out = add_bound(*a, **k)
            note('bound')
            return out

        def add_samples_logged(*a, **k):
            out = add_samples(*a, **k)
            calls['add_samples'] += 1
            if s.explored and calls['add_samples'] % 50 == 0:
                note('samples')
            return out
        s.add_bound, s.add_samples = add_bound_logged, add_samples_logged
    if args.counters:
        from nautilus_amd import device
        with device.EvalCounters() as counters, \
                device.KernelTimer() as timer:
            ok = s.run(n_eff=args.n_eff, discard_exploration=discard,
                       timeout=args.timeout)
            torch.cuda.synchronize()
            wall = time.time() - t0
            extra = dict(counters=counters.read(), kernels={
                k: dict(launches=v['launches'], s=round(v['ms'] / 1e3, 2))
                for k, v in timer.totals().items()})
    else:
        ok = s.run(n_eff=args.n_eff, discard_exploration=discard,
                   timeout=args.timeout)
        torch.cuda.synchronize()
        wall = time.time() - t0
    # posterior mean of the first three parameters, on the device (config 5
    # holds ~10^7 points of 100 dimensions; Sampler.posterior would bring them
    # all to the host): weights as in Sampler.posterior (sampler.py:602-608)
    s.land_points()
    start = (s.shell_end_exp if s._discard_exploration and s.explored
             else np.zeros(len(s.log_l), dtype=int))
    offset = s.shell_log_v - np.log(np.maximum(s.shell_n, 1))
    num = torch.zeros(3, dtype=torch.float64, device='cuda')
    sq = torch.zeros(3, dtype=torch.float64, device='cuda')
    for p, ll, st, o in zip(s._pts, s._ll_dev, start, offset):
        w = torch.exp(ll.view()[st:] + float(o) - float(s.log_z))
        num += (p.view()[st:, :3] * w[:, None]).sum(0)
        sq += (p.view()[st:, :3]**2 * w[:, None]).sum(0)
    mean = num.cpu().numpy()
    var = sq.cpu().numpy() - mean**2
    print(json.dumps(dict(
        config=args.name, finished=bool(ok), wall_s=round(wall, 2),
        n_batch=args.n_batch or c['n_batch'], seed=args.seed,
        n_live=args.n_live or c['n_live'],
        n_networks=c['n_networks'] if args.n_networks is None
        else args.n_networks,
        mean_x0=float(mean[0]), var_x0=float(var[0]),
        discard_exploration=discard, explored=bool(s.explored),
        log_z=float(s.log_z), analytic_log_z=c['analytic_log_z'],
        n_eff=float(s.n_eff), n_like=int(s.n_like), n_bounds=len(s.bounds),
        n_neural_last=len(s.bounds[-1].neural_bounds),
        n_dead_bounds=s.n_dead_bounds
        if len(s.bounds) > 1 else 0,
        n_proposals=int(s.n_proposals), mean_first3=mean[:3].round(4).tolist(),
        n_in_bound=int(np.sum(s.shell_n_sample)),
        timing={k: round(v, 2) for k, v in s.timing.items()}, **extra)))


if __name__ == '__main__':
    main()
