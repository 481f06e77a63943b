"""Test-side glue: upload an oracle bound's parameters through the product's
C-ABI wrapper so that HIP output can be compared with the oracle."""

import numpy as np


def member_from_oracle(b):
    from nautilus_amd import device
    if hasattr(b, 'dim_cube'):                       # OMixture
        if b.ellipsoid is None:
            return device.member()
        idx = np.flatnonzero(~b.dim_cube).astype(np.int32)
        e = b.ellipsoid
        return device.member(e.c, e.B, e.B_inv, idx_ell=idx)
    if hasattr(b, 'B'):                              # OEllipsoid
        return device.member(b.c, b.B, b.B_inv)
    return device.member()                           # OCube


def neural_from_oracle(nb):
    from nautilus_amd import device
    e = nb.outer_bound
    out = dict(ellipsoid=device.member(e.c, e.B, e.B_inv),
               score_predict_min=float(nb.score_predict_min), mlp=None)
    if nb.emulator is not None:
        out['mlp'] = dict(mean=nb.emulator.mean, scale=nb.emulator.scale,
                          nets=[(n.coefs, n.intercepts)
                                for n in nb.emulator.networks])
    return out


def upload(bound):
    """Oracle bound (any class) -> nautilus_amd.device.DeviceBound."""
    from nautilus_amd import device
    from oracle import bounds_oracle as bo
    if isinstance(bound, bo.ONautilus):
        u = bound.outer_bound
        return device.DeviceBound(
            bound.n_dim, [member_from_oracle(m) for m in u.bounds],
            u.log_v_all, u.cube is not None,
            [neural_from_oracle(nb) for nb in bound.neural_bounds],
            shift=None if bound.shift is None else
            (bound.shift.periodic, bound.shift.centers))
    if isinstance(bound, bo.OUnion):
        return device.DeviceBound(
            bound.n_dim, [member_from_oracle(m) for m in bound.bounds],
            bound.log_v_all, bound.cube is not None)
    if isinstance(bound, bo.ONeural):
        return device.DeviceBound(bound.n_dim, [], None, False,
                                  [neural_from_oracle(bound)])
    if isinstance(bound, bo.OCube):
        return device.DeviceBound(bound.n_dim, [member_from_oracle(bound)],
                                  [0.0], True)
    return device.DeviceBound(bound.n_dim, [member_from_oracle(bound)],
                              [bound.log_v], False)


def near_boundary(values, edge, tol):
    return np.abs(np.asarray(values) - edge) < tol


def neural_from_golden(g, prefix):
    """Rebuild an oracle NeuralBound from the parameters stored in a golden
    file (no retraining: the GPU box may have a different CPU / BLAS)."""
    from oracle import bounds_oracle as bo
    from oracle import mlp_oracle as mo
    nb = bo.ONeural()
    nb.outer_bound = bo.OEllipsoid.from_params(
        g[prefix + 'c'], g[prefix + 'B'], g[prefix + 'B_inv'],
        g[prefix + 'A'])
    nb.n_dim = nb.outer_bound.n_dim
    nb.score_predict_min = float(g[prefix + 'score_predict_min'])
    nb.emulator = None
    if prefix + 'mean' in g:
        e = int(g[prefix + 'n_networks'])
        nets = [([g[prefix + 'coef_%d_%d' % (i, k)] for k in range(4)],
                 [g[prefix + 'intercept_%d_%d' % (i, k)] for k in range(4)])
                for i in range(e)]
        nb.emulator = mo.Emulator.from_weights(g[prefix + 'mean'],
                                               g[prefix + 'scale'], nets)
    return nb


def union_from_golden(g, mixture):
    from oracle import bounds_oracle as bo
    members = []
    for i in range(int(g['K'])):
        ell = None
        if 'B_%d' % i in g:
            ell = bo.OEllipsoid.from_params(g['c_%d' % i], g['B_%d' % i],
                                            g['B_inv_%d' % i], g['A_%d' % i])
        members.append(bo.OMixture.from_params(g['dim_cube_%d' % i], ell)
                       if mixture else ell)
    u = bo.OUnion.from_members(members, unit=bool(g['unit']))
    u.log_v_all = g['log_v_all']
    return u


def nautilus_from_golden(g):
    from oracle import bounds_oracle as bo
    outer = union_from_golden(g, True)
    neural = [neural_from_golden(g, 'nb%d_' % i)
              for i in range(int(g['n_neural']))]
    shift = None
    if 'centers' in g:
        shift = bo.OPhaseShift.from_params(g['periodic'], g['centers'])
    return bo.ONautilus.from_parts(outer, neural, shift=shift)


def khachiyan_weights_numpy(points, n_max=100, n_batch=20):
    """Weights u of the batched Khachiyan iteration (reference
    bounds/basic.py:175-232) in numpy, with the inverse tracked through the
    rank-one updates of a sweep (the reference re-inverts after every
    update).  Test-side comparison for the device kernels."""
    from scipy.linalg.lapack import dpotrf, dpotri
    n, d = points.shape
    q = np.empty((n, d + 1))
    q[:, :d] = points
    q[:, d] = 1.0
    u = np.full(n, 1.0 / n)
    v = (q * u[:, None]).T @ q
    for _ in range(n_max):
        tri = dpotri(dpotrf(v)[0])[0]
        v_inv = tri + tri.T - np.diag(np.diag(tri))
        g_all = np.einsum('ij,ij->i', q @ v_inv, q)
        first = True
        for j in np.argsort(g_all)[-n_batch:][::-1]:
            qj = q[j]
            w = v_inv @ qj
            g = g_all[j] if first else qj @ w
            first = False
            if g < d + 1:
                continue
            step = (g - (d + 1)) / ((d + 1) * (g - 1))
            v = v * (1 - step) + step * np.outer(qj, qj)
            ratio = step / (1 - step)
            v_inv = (v_inv - np.outer(w, w) * (ratio / (1 + ratio * g))) / \
                (1 - step)
            u *= (1 - step)
            u[j] += step
    return u


def mvee_numpy_batch(point_sets, n_max=100, n_batch=20):
    """Stand-in for ``geometry.mvee_batch`` in the CPU-only tests of the host
    logic: the finishing steps of basic.py:233-241 on top of
    ``khachiyan_weights_numpy``."""
    out = []
    for points in point_sets:
        points = np.ascontiguousarray(points, dtype=float)
        u = khachiyan_weights_numpy(points, n_max, n_batch)
        c = np.atleast_1d(np.average(points, weights=u, axis=0))
        a_inv = np.atleast_2d(np.cov(points, aweights=u, rowvar=False,
                                     bias=True))
        a = np.linalg.inv(a_inv)
        diff = points - c
        scale = np.amax(np.einsum('ij,ij->i', diff @ a, diff))
        out.append((c, a / scale, a_inv * scale))
    return out


def rosenbrock_log_z_exact(n_dim, m=2000):
    """Evidence of the Rosenbrock likelihood of BASELINE config 3 (x = 10 u - 5,
    identity prior) by transfer quadrature: the integrand is a chain, so
    f_k(x_k) = exp(-(1 - x_k)^2) * int exp(-100 (x_{k+1} - x_k^2)^2) f_{k+1}
    is one matrix-vector product per dimension on an m-point grid (converged
    to 1e-8 at m = 2000; -137.4875 for n_dim = 30)."""
    from scipy.special import logsumexp
    h = 10.0 / m
    x = -5 + (np.arange(m) + 0.5) * h
    kern = -100.0 * (x[None, :] - x[:, None]**2)**2
    logf = np.zeros(m)
    for _ in range(n_dim - 1):
        logf = logsumexp(kern + logf[None, :], axis=1) + np.log(h) - (1 - x)**2
    return logsumexp(logf) + np.log(h) - n_dim * np.log(10.0)
