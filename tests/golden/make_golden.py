"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (``/root/reference`` is mounted there and does
not exist on the GPU box):

    OMP_NUM_THREADS=1 python tests/golden/make_golden.py

It imports johannesulf/nautilus v1.0.6 from /root/reference, drives the
functions on the hot path with fixed seeds and writes
inputs + the reference's outputs as small ``.npz`` files.  Only data is
stored; no reference source travels.  The oracle (``oracle/``) is pinned
against these files by ``tests/test_oracle_golden.py``.
"""

import json
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))

import nautilus  # noqa: E402
from nautilus import bounds  # noqa: E402
from nautilus.bounds.basic import minimum_volume_enclosing_ellipsoid  # noqa
from nautilus.neural import NeuralNetworkEmulator  # noqa: E402

assert nautilus.__version__ == '1.0.6'


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def correlated_cloud(n, d, seed, scale=0.08):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(d, d)) / np.sqrt(d)
    return 0.5 + scale * rng.normal(size=(n, d)) @ (np.eye(d) + 0.5 * a)


def ellipsoid_case(d, n):
    pts = correlated_cloud(n, d, 100 + d)
    ell = bounds.Ellipsoid.compute(pts, enlarge_per_dim=1.1,
                                   rng=np.random.default_rng(0))
    drawn = ell.sample(512)
    test = np.random.default_rng(1).random((512, d))
    # half of the probe points are pulled towards the centre so that the
    # mask has both values
    test[::2] = ell.c + 0.35 * (test[::2] - 0.5)
    y = ell.transform(test)
    save('ellipsoid_D%d' % d, points=pts, enlarge=1.1, c=ell.c, A=ell.A,
         B=ell.B, B_inv=ell.B_inv, log_v=ell.log_v, sample=drawn, test=test,
         transform=y, r2=np.sum(y**2, axis=-1), contains=ell.contains(test))


def mvee_sphere():
    d = 10
    pts = np.zeros((2 * d, d)) + 0.5
    for i in range(2 * d):
        pts[i, i // 2] += 1 if i % 2 else -1
    pts = np.vstack([pts, np.zeros(d) + 0.5])
    c, a, a_inv = minimum_volume_enclosing_ellipsoid(pts)
    save('mvee_sphere_D10', points=pts, c=c, A=a, A_inv=a_inv)


def mixture_case():
    rng = np.random.default_rng(5)
    d = 6
    pts = rng.random((400, d))
    pts[:, :3] = 0.5 + 0.03 * rng.normal(size=(400, 3)) @ np.array(
        [[1, 0.5, 0], [0, 1, 0.3], [0, 0, 1]])
    mix = bounds.UnitCubeEllipsoidMixture.compute(
        pts, enlarge_per_dim=1.1, rng=np.random.default_rng(0))
    drawn = mix.sample(512)
    test = np.random.default_rng(6).random((512, d))
    test[::2, :3] = 0.5 + 0.1 * (test[::2, :3] - 0.5)
    test[5] = -0.01
    save('mixture_D6', points=pts, dim_cube=mix.dim_cube, c=mix.ellipsoid.c,
         B=mix.ellipsoid.B, B_inv=mix.ellipsoid.B_inv, A=mix.ellipsoid.A,
         log_v=mix.log_v, sample=drawn, test=test,
         contains=mix.contains(test), transform=mix.transform(test))


def member_arrays(union):
    out = {}
    for i, b in enumerate(union.bounds):
        if hasattr(b, 'dim_cube'):
            out['dim_cube_%d' % i] = b.dim_cube
            b = b.ellipsoid
            if b is None:
                continue
        out['c_%d' % i] = b.c
        out['B_%d' % i] = b.B
        out['B_inv_%d' % i] = b.B_inv
        out['A_%d' % i] = b.A
    return out


def union_case(name, pts, member_cls, n_split, unit, d):
    union = bounds.Union.compute(pts, enlarge_per_dim=1.1, unit=unit,
                                 bound_class=member_cls,
                                 rng=np.random.default_rng(0))
    for _ in range(n_split):
        union.split()
    k = len(union.bounds)
    union.reset(np.random.default_rng(7))
    drawn = union.sample(1500)
    lo, hi = pts.min(0) - 0.05, pts.max(0) + 0.05
    test = lo + (hi - lo) * np.random.default_rng(8).random((1024, d))
    counts = np.sum([b.contains(test) for b in union.bounds], axis=0)
    save(name, points=pts, K=k, unit=unit, log_v_all=union.log_v_all,
         sample=drawn, n_sample=union.n_sample, n_reject=union.n_reject,
         log_v=union.log_v, fifo=union.points, test=test, counts=counts,
         contains=union.contains(test), **member_arrays(union))


def union_cases():
    rng = np.random.default_rng(11)
    # two overlapping blobs in 3-D, ellipsoid members, no cube clip
    pts = np.vstack([0.4 + 0.05 * rng.normal(size=(300, 3)),
                     0.55 + 0.05 * rng.normal(size=(300, 3))])
    union_case('union_K2_D3', pts, bounds.Ellipsoid, 1, False, 3)
    # four blobs in 8-D, mixture members, with the unit-cube clip; two blobs
    # sit at the cube boundary so that the clip rejects proposals
    cen = np.array([0.03, 0.35, 0.65, 0.97])
    pts = np.vstack([np.clip(c + 0.02 * rng.normal(size=(250, 8)), 0,
                             1 - 1e-9) for c in cen])
    pts[:, 6:] = rng.random((1000, 2))
    union_case('union_K4_D8', pts, bounds.UnitCubeEllipsoidMixture, 3, True,
               8)


def union_d50_case():
    """union_K4_D50: four blobs in 50 dimensions, the
    last five uniform, cube-ellipsoid mixture members, unit-cube clip."""
    rng = np.random.default_rng(12)
    cen = 0.3 + 0.4 * rng.random((4, 50))
    pts = np.vstack([c + 0.01 * rng.normal(size=(300, 50)) for c in cen])
    pts[:, 45:] = rng.random((1200, 5))
    union_case('union_K4_D50', pts, bounds.UnitCubeEllipsoidMixture, 3, True,
               50)


def emulator_case(d, n, e, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, d))
    r = np.linalg.norm(x, axis=1)
    y = (np.argsort(np.argsort(-r)) + 0.5) / n
    emu = NeuralNetworkEmulator.train(x, y, n_networks=e)
    test = rng.normal(size=(256, d))
    arrays = dict(x=x, y=y, mean=emu.mean, scale=emu.scale, test=test,
                  predict=emu.predict(test), n_networks=e)
    for i, net in enumerate(emu.neural_networks):
        arrays['n_iter_%d' % i] = net.n_iter_
        arrays['loss_curve_%d' % i] = np.array(net.loss_curve_)
        for k in range(4):
            arrays['coef_%d_%d' % (i, k)] = net.coefs_[k]
            arrays['intercept_%d_%d' % (i, k)] = net.intercepts_[k]
    save('emulator_D%d_E%d' % (d, e), **arrays)


def neural_arrays(nb, prefix):
    """Every parameter of a NeuralBound (bounds/neural.py:10-26)."""
    out = {prefix + 'c': nb.outer_bound.c, prefix + 'B': nb.outer_bound.B,
           prefix + 'B_inv': nb.outer_bound.B_inv,
           prefix + 'A': nb.outer_bound.A,
           prefix + 'score_predict_min': nb.score_predict_min}
    if nb.emulator is not None:
        out[prefix + 'mean'] = nb.emulator.mean
        out[prefix + 'scale'] = nb.emulator.scale
        out[prefix + 'n_networks'] = len(nb.emulator.neural_networks)
        for i, net in enumerate(nb.emulator.neural_networks):
            for k in range(4):
                out[prefix + 'coef_%d_%d' % (i, k)] = net.coefs_[k]
                out[prefix + 'intercept_%d_%d' % (i, k)] = net.intercepts_[k]
    return out


def neural_and_nautilus_case():
    np.random.seed(0)
    pts = np.random.random(size=(500, 4))
    log_l = -np.linalg.norm(pts - 0.5, axis=1)
    log_l_min = np.median(log_l)
    nb = bounds.NeuralBound.compute(pts, log_l, log_l_min, n_networks=1,
                                    rng=np.random.default_rng(0))
    test = np.random.default_rng(2).random((512, 4))
    save('neuralbound_D4', points=pts, log_l=log_l, log_l_min=log_l_min,
         c=nb.outer_bound.c, B=nb.outer_bound.B, B_inv=nb.outer_bound.B_inv,
         score_predict_min=nb.score_predict_min, test=test,
         contains=nb.contains(test),
         score=nb.emulator.predict(nb.outer_bound.transform(test)),
         **neural_arrays(nb, 'nb_'))

    full = bounds.NautilusBound.compute(
        pts, log_l, log_l_min, np.log(0.5), n_networks=1,
        rng=np.random.default_rng(0))
    full.reset(np.random.default_rng(3))
    drawn = full.sample(2000)
    save('nautilusbound_D4', points=pts, log_l=log_l, log_l_min=log_l_min,
         log_v_target=np.log(0.5), sample=drawn, n_sample=full.n_sample,
         n_reject=full.n_reject, outer_n_sample=full.outer_bound.n_sample,
         outer_n_reject=full.outer_bound.n_reject, log_v=full.log_v,
         test=test, contains=full.contains(test),
         n_neural=len(full.neural_bounds),
         n_outer=len(full.outer_bound.bounds),
         K=len(full.outer_bound.bounds), unit=True,
         log_v_all=full.outer_bound.log_v_all,
         **member_arrays(full.outer_bound),
         **{k: v for i, nb_i in enumerate(full.neural_bounds)
            for k, v in neural_arrays(nb_i, 'nb%d_' % i).items()})


def periodic_cases():
    """PhaseShift (bounds/periodic.py) alone, inside a NautilusBound, and in a
    full run (the reference's tests/test_bounds.py:314-327 and
    tests/test_sampler.py:395-416 cover the same ground)."""
    np.random.seed(0)
    pts_all, centers, fwd, back = [], [], [], []
    for i in range(20):
        pts = (np.random.random(size=(10, 4)) * 0.1 +
               np.random.random(size=4)) % 1
        shift = bounds.PhaseShift.compute(pts, np.array([0, 2]))
        pts_all.append(pts)
        centers.append(shift.centers)
        fwd.append(shift.transform(pts))
        back.append(shift.transform(fwd[-1], inverse=True))
    save('phaseshift', points=np.array(pts_all), periodic=np.array([0, 2]),
         centers=np.array(centers), forward=np.array(fwd),
         inverse=np.array(back))

    rng = np.random.default_rng(11)
    pts = rng.random((600, 3))
    delta = np.abs(pts - np.array([0.03, 0.5, 0.97]))
    delta = np.minimum(delta, 1 - delta)            # periodic distance
    log_l = -np.linalg.norm(delta, axis=1)
    log_l_min = np.median(log_l)
    periodic = np.array([0, 2])
    full = bounds.NautilusBound.compute(
        pts, log_l, log_l_min, np.log(0.5), n_networks=1, periodic=periodic,
        rng=np.random.default_rng(0))
    full.reset(np.random.default_rng(3))
    drawn = full.sample(2000)
    test = np.random.default_rng(2).random((2048, 3))
    save('nautilusbound_periodic_D3', points=pts, log_l=log_l,
         log_l_min=log_l_min, log_v_target=np.log(0.5), sample=drawn,
         n_sample=full.n_sample, n_reject=full.n_reject,
         outer_n_sample=full.outer_bound.n_sample,
         outer_n_reject=full.outer_bound.n_reject, log_v=full.log_v,
         test=test, contains=full.contains(test),
         periodic=periodic, centers=full.shift.centers,
         n_neural=len(full.neural_bounds),
         n_outer=len(full.outer_bound.bounds),
         K=len(full.outer_bound.bounds), unit=True,
         log_v_all=full.outer_bound.log_v_all,
         **member_arrays(full.outer_bound),
         **{k: v for i, nb_i in enumerate(full.neural_bounds)
            for k, v in neural_arrays(nb_i, 'nb%d_' % i).items()})

    def wrapped(x):
        return -0.5 * np.sum((np.abs(x - 0.5) - 0.5)**2, axis=-1) / 0.1

    rows = []
    for per, n_networks in [(True, 0), (False, 0), (True, 1)]:
        s = nautilus.Sampler(lambda x: x, wrapped, n_dim=2, n_live=400,
                             periodic=np.arange(2) if per else None,
                             n_networks=n_networks, vectorized=True, seed=0)
        s.run(n_eff=2000, discard_exploration=True)
        p, log_w, _ = s.posterior()
        rows.append(dict(
            periodic=per, n_networks=n_networks, seed=0, n_live=400,
            n_eff_target=2000, log_z=float(s.log_z), n_eff=float(s.n_eff),
            n_like=int(s.n_like), n_bounds=len(s.bounds),
            n_neural_per_bound=[len(b.neural_bounds) for b in s.bounds[1:]],
            shell_n=s.shell_n.tolist(),
            shell_log_v=s.shell_log_v.tolist(),
            shell_log_l=s.shell_log_l.tolist()))
        print('periodic e2e', per, n_networks, rows[-1]['log_z'],
              rows[-1]['n_neural_per_bound'][-3:])
    with open(os.path.join(HERE, 'e2e_periodic.json'), 'w') as f:
        json.dump(dict(problem='2-D Gaussian wrapped around the corners of '
                               'the unit square: -0.5 * |abs(x - 0.5) - 0.5|^2'
                               ' / 0.1', runs=rows), f, indent=1)


def gauss3(x):
    return -0.5 * np.sum(((x - np.array([0.4, 0.5, 0.6])) / 0.1)**2, axis=-1)


def e2e_cases():
    """Full reference runs: the oracle driver must reproduce them exactly,
    and they give the statistical band for the device path."""
    rows = []
    for n_networks, seed in [(0, 0), (0, 1), (1, 0), (1, 1), (2, 2)]:
        s = nautilus.Sampler(lambda x: x, gauss3, n_dim=3, n_live=400,
                             n_networks=n_networks, vectorized=True,
                             seed=seed)
        s.run(n_eff=3000, discard_exploration=True)
        pts, log_w, log_l = s.posterior()
        w = np.exp(log_w)
        rows.append(dict(
            n_networks=n_networks, seed=seed, n_live=400, n_eff_target=3000,
            log_z=float(s.log_z), n_eff=float(s.n_eff), n_like=int(s.n_like),
            n_bounds=len(s.bounds), eta=float(s.eta),
            shell_n=s.shell_n.tolist(),
            shell_n_sample=s.shell_n_sample.tolist(),
            shell_log_v=s.shell_log_v.tolist(),
            shell_log_l=s.shell_log_l.tolist(),
            shell_n_eff=s.shell_n_eff.tolist(),
            mean=np.average(pts, weights=w, axis=0).tolist(),
            var=np.average((pts - [0.4, 0.5, 0.6])**2, weights=w,
                           axis=0).tolist()))
        print('e2e', rows[-1]['n_networks'], seed, rows[-1]['log_z'],
              rows[-1]['n_like'])
    with open(os.path.join(HERE, 'e2e_gauss3.json'), 'w') as f:
        json.dump(dict(problem='3-D Gaussian mu=(0.4,0.5,0.6) sigma=0.1, '
                               'identity prior; analytic log_z=-6.4e-5 + '
                               'log((0.1*sqrt(2pi))^3)',
                       analytic_log_z=float(3 * np.log(0.1 * np.sqrt(
                           2 * np.pi)) - 6.4e-5),
                       runs=rows), f, indent=1)

    # per-shell statistics of one finished run (shellstats fixture)
    s = nautilus.Sampler(lambda x: x, gauss3, n_dim=3, n_live=400,
                         n_networks=0, vectorized=True, seed=5)
    s.run(n_eff=2000)
    pts, log_w, log_l = s.posterior()
    arrays = dict(shell_n_sample=s.shell_n_sample, shell_n=s.shell_n,
                  bound_log_v=np.array([b.log_v for b in s.bounds]),
                  shell_log_v=s.shell_log_v, shell_log_l=s.shell_log_l,
                  shell_n_eff=s.shell_n_eff, log_z=s.log_z, n_eff=s.n_eff,
                  eta=s.eta, log_w=log_w)
    for i, ll in enumerate(s.log_l):
        arrays['log_l_%d' % i] = ll
    save('shellstats', **arrays)


def liveset_case():
    """The live set of the exploration phase (sampler.py:1147-1190): snapshots
    of the per-shell log L, volumes and counts of a reference run in progress
    with the reference's own ``f_live`` / ``log_v_live`` at that moment."""
    s = nautilus.Sampler(lambda x: x, gauss3, n_dim=3, n_live=300,
                         n_networks=0, vectorized=True, seed=11, n_batch=64)
    snaps = []
    orig = s.add_samples
    calls = [0]

    def recording(*args, **kwargs):
        out = orig(*args, **kwargs)
        calls[0] += 1
        if not s.explored and calls[0] % 9 == 0 and len(s.bounds) > 0:
            snaps.append(dict(
                log_l=[np.copy(ll) for ll in s.log_l],
                shell_log_v=np.copy(s.shell_log_v),
                shell_n=np.copy(s.shell_n), f_live=s.f_live,
                log_v_live=s.log_v_live, log_z=s.log_z,
                n_bounds=len(s.bounds)))
        return out
    s.add_samples = recording
    s.run(n_eff=500)
    arrays = dict(n_snap=len(snaps), n_live=300)
    for k, sn in enumerate(snaps):
        arrays['s%d_shell_log_v' % k] = sn['shell_log_v']
        arrays['s%d_shell_n' % k] = sn['shell_n']
        arrays['s%d_f_live' % k] = sn['f_live']
        arrays['s%d_log_v_live' % k] = sn['log_v_live']
        arrays['s%d_log_z' % k] = sn['log_z']
        arrays['s%d_n_bounds' % k] = sn['n_bounds']
        for i, ll in enumerate(sn['log_l']):
            arrays['s%d_log_l_%d' % (k, i)] = ll
    print('liveset: %d snapshots, last with %d bounds' % (
        len(snaps), snaps[-1]['n_bounds']))
    save('liveset', **arrays)


def gauss20(x):
    """BASELINE config 2: mu = 0.5, Sigma = sigma^2
    (0.5 11^T + 0.5 I), sigma = 0.05; analytic log Z = 0."""
    d, s = 20, 0.05
    cov = s**2 * (0.5 * np.ones((d, d)) + 0.5 * np.eye(d))
    if not hasattr(gauss20, 'prec'):
        gauss20.prec = np.linalg.inv(cov)
        gauss20.norm = -0.5 * (d * np.log(2 * np.pi) +
                               np.linalg.slogdet(cov)[1])
    y = np.atleast_2d(x) - 0.5
    return gauss20.norm - 0.5 * np.einsum('ij,jk,ik->i', y, gauss20.prec, y)


def gauss3_c1(x):
    """BASELINE config 1 (README example): mu = (0.4, 0.5, 0.6), sigma = 0.1,
    normalised; analytic log Z = -6.4e-5."""
    return (gauss3(x) - 3 * np.log(0.1 * np.sqrt(2 * np.pi)))


def _summary(s, seed, discard, mu):
    pts, log_w, log_l = s.posterior()
    w = np.exp(log_w)
    mean = np.average(pts, weights=w, axis=0)
    return dict(seed=seed, discard_exploration=discard,
                log_z=float(s.log_z), n_eff=float(s.n_eff),
                n_like=int(s.n_like), n_bounds=len(s.bounds),
                eta=float(s.eta), mean=mean.tolist(),
                var=np.average((pts - mu)**2, weights=w, axis=0).tolist())


def e2e_config_run(config, seed, discard):
    """One full reference run of BASELINE config C1 / C2."""
    if config == 'C1':
        s = nautilus.Sampler(lambda x: x, gauss3_c1, n_dim=3, n_live=1000,
                             vectorized=True, seed=seed)
        mu = np.array([0.4, 0.5, 0.6])
    else:
        s = nautilus.Sampler(lambda x: x, gauss20, n_dim=20, n_live=2000,
                             vectorized=True, seed=seed)
        mu = np.full(20, 0.5)
    s.run(discard_exploration=discard)
    return _summary(s, seed, discard, mu)


def e2e_config_sweep(config, seeds, discards, n_proc):
    """Seed sweep of a BASELINE config with the reference (the tolerance band
    of log Z, n_like and the number of bounds for the device path)."""
    import multiprocessing
    jobs = [(config, seed, discard) for discard in discards for seed in seeds]
    with multiprocessing.Pool(n_proc) as pool:
        rows = pool.starmap(e2e_config_run, jobs)
    analytic = -6.4e-5 if config == 'C1' else 0.0
    problem = ('3-D Gaussian mu=(0.4,0.5,0.6) sigma=0.1 (normalised), '
               'n_live=1000' if config == 'C1' else
               '20-D correlated Gaussian mu=0.5 Sigma=0.05^2 (0.5 11^T + '
               '0.5 I), n_live=2000')
    with open(os.path.join(HERE, 'e2e_%s.json' % config), 'w') as f:
        json.dump(dict(problem=problem + ', identity prior, n_networks=4, '
                       'n_eff=10000 (reference defaults)',
                       analytic_log_z=analytic, runs=rows), f, indent=1)
    print(config, [round(r['log_z'], 4) for r in rows])


if __name__ == '__main__':
    if '--round2-small' in sys.argv:
        union_d50_case()
        emulator_case(50, 1500, 4, 23)
        sys.exit(0)
    if '--e2e-C1' in sys.argv:
        e2e_config_sweep('C1', range(10), [False, True], 7)
        sys.exit(0)
    if '--e2e-C2' in sys.argv:
        e2e_config_sweep('C2', range(6), [True], 6)
        sys.exit(0)
    if '--liveset-only' in sys.argv:
        liveset_case()
        sys.exit(0)
    if '--periodic-only' in sys.argv:
        periodic_cases()
        sys.exit(0)
    if '--neural-only' in sys.argv:
        neural_and_nautilus_case()
        sys.exit(0)
    for d, n in [(3, 200), (20, 400), (50, 600)]:
        ellipsoid_case(d, n)
    mvee_sphere()
    mixture_case()
    union_cases()
    emulator_case(5, 1000, 1, 21)
    emulator_case(20, 600, 2, 22)
    neural_and_nautilus_case()
    periodic_cases()
    e2e_cases()
    liveset_case()
