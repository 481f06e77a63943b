"""Host-side construction of the bound geometry (small, sequential numerics).

Bound *construction* runs a handful of times per bound on ~n_live points.
Everything that touches the points -- the Khachiyan iteration of the
minimum-volume enclosing ellipsoid, its weighted moments and scaling, the
two-component mixture fit -- runs on the GPU (``nb_mvee_khachiyan``,
``nb_weighted_moments``, ``nb_quadform_max``, ``nb_gmm_fit``), batched over the
point sets that are independent of each other; O(D^3) operations on single
(D x D) matrices, the greedy cube/ellipsoid choice and the ellipsoid overlap
test stay on the host.  What is built here is uploaded once through ``nb_bound_create``;
every per-point operation afterwards (draw, contains, emulator, compaction)
runs on the GPU.

Semantics follow the reference (paths relative to /root/reference/nautilus):
bounds/basic.py:154-241 (MVEE), 265-316 (Ellipsoid.compute), 471-563
(UnitCubeEllipsoidMixture.compute); bounds/union.py:14-40 (overlap test),
153-229 (split), 231-267 (trim).
"""

import hashlib
import itertools
from collections import OrderedDict
from contextlib import contextmanager

import numpy as np
from scipy.linalg.lapack import dpotrf, dpotri
from scipy.optimize import minimize
from scipy.special import gammaln, logsumexp
from threadpoolctl import threadpool_limits

try:                      # ~10x faster than hashlib; optional
    import xxhash
except ImportError:       # pragma: no cover
    xxhash = None

_BLAS_LIMITED = 0


@contextmanager
def single_threaded_blas():
    """Host BLAS pinned to one thread, as the reference does around the bound
    construction (sampler.py:1022): it works on tiny matrices.  Re-entrant:
    only the outermost region pays for threadpoolctl's library scan (~1 ms,
    which added up to 2 % of an exploration when every MVEE batch did it)."""
    global _BLAS_LIMITED
    if _BLAS_LIMITED:
        yield
        return
    _BLAS_LIMITED += 1
    try:
        with threadpool_limits(limits=1):
            yield
    finally:
        _BLAS_LIMITED -= 1


def inv_spd(m):
    """Inverse of a symmetric positive definite matrix via Cholesky."""
    tri = dpotri(dpotrf(m)[0])[0]
    return tri + tri.T - np.diag(np.diag(tri))


def mvee_batch(point_sets, n_max=100, n_batch=20):
    """Minimum-volume enclosing ellipsoids (reference bounds/basic.py:175-241)
    of several point sets at once; everything that touches the points runs on
    the GPU (``device.mvee_fit_batch``).  Returns a list of (c, A, A^-1)."""
    from . import device
    out = [None] * len(point_sets)
    by_dim = {}
    for i, pts in enumerate(point_sets):
        by_dim.setdefault(pts.shape[1], []).append(i)
    for idx in by_dim.values():
        res = device.mvee_fit_batch([point_sets[i] for i in idx], n_max,
                                    n_batch)
        for i, r in zip(idx, res):
            out[i] = r
    return out


def mvee(points, n_max=100, n_batch=20):
    """One minimum-volume enclosing ellipsoid: centre c, shape matrix A
    ((x-c)^T A (x-c) <= 1) and A^-1."""
    return mvee_batch([points], n_max, n_batch)[0]


# Ellipsoids are functions of (point set, enlargement) only -- no random
# numbers are consumed -- and the construction of one NautilusBound asks for
# the ellipsoid of the same rows several times (the decomposition, the neural
# bounds and the sampling envelope all start from the live points,
# nautilus.py:100-133).  A few recent results are kept.
_ELL_CACHE = OrderedDict()
_ELL_CACHE_SIZE = 16


def _host_rows(points):
    if hasattr(points, 'is_cuda'):
        return None
    return np.ascontiguousarray(points, dtype=float)


def _cache_key(points, enlarge_per_dim):
    rows = _host_rows(points)
    if rows is None:
        return None
    flat = rows.view(np.uint8).reshape(-1)
    if xxhash is not None:
        digest = xxhash.xxh64(flat).digest()
    else:
        digest = hashlib.blake2b(flat, digest_size=16).digest()
    return (rows.shape, digest, float(enlarge_per_dim))


def _ellipsoid_task(points, enlarge_per_dim):
    """Ellipsoid.compute (bounds/basic.py:265-316) as a coroutine: yields the
    point set whose MVEE it needs, receives (c, A, A^-1) and returns
    dict(c, A, B, B_inv) with B = chol(A^-1) lower triangular, B_inv = B^-1."""
    n, d = points.shape
    if enlarge_per_dim < 1.0:
        raise ValueError("The 'enlarge_per_dim' factor cannot be smaller "
                         "than unity.")
    if not n > d:
        raise ValueError('Number of points must be larger than number '
                         'dimensions.')
    key = _cache_key(points, enlarge_per_dim)
    if key is not None and key in _ELL_CACHE:
        _ELL_CACHE.move_to_end(key)
        return dict(_ELL_CACHE[key])
    c, a, a_inv = yield points
    a = a / enlarge_per_dim**2.0
    a_inv = a_inv * enlarge_per_dim**2.0
    b = np.linalg.cholesky(a_inv)
    b_inv = np.tril(np.linalg.inv(b))
    out = dict(c=c, A=a, B=b, B_inv=b_inv)
    if key is not None:
        _ELL_CACHE[key] = dict(out)
        while len(_ELL_CACHE) > _ELL_CACHE_SIZE:
            _ELL_CACHE.popitem(last=False)
    return out


def run_tasks(tasks):
    """Drive several construction coroutines in lockstep: the MVEE requests
    they have pending at the same time (the two children of a split, the
    neural-bound ellipsoids of one NautilusBound) go to the GPU as ONE batch.
    Returns the coroutines' return values."""
    out = [None] * len(tasks)
    pending = {}
    for i, task in enumerate(tasks):
        try:
            pending[i] = next(task)
        except StopIteration as stop:
            out[i] = stop.value
    while pending:
        idx = list(pending)
        with single_threaded_blas():
            res = mvee_batch([pending[i] for i in idx])
        pending = {}
        for i, r in zip(idx, res):
            try:
                pending[i] = tasks[i].send(r)
            except StopIteration as stop:
                out[i] = stop.value
    return out


def ellipsoid_params_batch(point_sets, enlarge_per_dim=1.1):
    return run_tasks([_ellipsoid_task(p, enlarge_per_dim)
                      for p in point_sets])


def ellipsoid_params(points, enlarge_per_dim=1.1):
    """Ellipsoid.compute (bounds/basic.py:265-316): returns dict(c, A, B,
    B_inv) with B = chol(A^-1) lower triangular and B_inv = B^-1."""
    return ellipsoid_params_batch([points], enlarge_per_dim)[0]


def ellipsoid_log_volume(b):
    """bounds/basic.py:393-394."""
    d = b.shape[0]
    return (np.linalg.slogdet(b)[1] + d * np.log(2.) + d * gammaln(1.5) -
            gammaln(d / 2.0 + 1))


def _mixture_task(points, enlarge_per_dim):
    """Greedy choice of the dimensions bounded by the unit cube
    (bounds/basic.py:471-563) as a coroutine (see ``_ellipsoid_task``).
    Returns (dim_cube, ellipsoid dict or None)."""
    d = points.shape[1]
    ell = yield from _ellipsoid_task(points, enlarge_per_dim)
    log_v = ellipsoid_log_volume(ell['B'])
    dim_cube = np.zeros(d, dtype=bool)

    while np.sum(~dim_cube) > 1:
        free = np.flatnonzero(~dim_cube)
        # Volume of the ellipsoid that is left when one dimension is dropped
        # (basic.py:522-531), for every candidate at once.  The reference
        # inverts the marginal (k-1)x(k-1) matrix and re-evaluates n quadratic
        # forms per candidate, O(n k^3) per round; with the shape matrix A of
        # the full ellipsoid, x = point - centre and y = A x, the marginal
        # form is x^T A x - y_i^2 / A_ii and its determinant det(A) / A_ii
        # (Schur complement), O(n k^2) per round and equal to 1e-13.
        k = len(free)
        x = points[:, free] - ell['c']
        y = x @ ell['A']
        diag = np.diag(ell['A'])
        scale = np.amax(np.einsum('ij,ij->i', y, x)[:, None] -
                        y**2 / diag[None, :], axis=0)
        trial_v = (np.log(diag) - np.linalg.slogdet(ell['A'])[1] +
                   (k - 1) * np.log(scale))
        dim = free[np.argmin(trial_v)]
        dim_cube[dim] = True
        cand = yield from _ellipsoid_task(points[:, ~dim_cube],
                                          enlarge_per_dim)
        cand_v = ellipsoid_log_volume(cand['B'])
        if cand_v < log_v:
            ell, log_v = cand, cand_v
        else:
            dim_cube[dim] = False
            break

    if log_v > 0:
        # the ellipsoid is larger than the cube: start from the cube and move
        # dimensions into an ellipsoid while that shrinks the volume
        ell, log_v = None, 0.0
        dim_cube = np.ones(d, dtype=bool)
        tested = np.zeros(d, dtype=bool)
        while not np.all(tested):
            for dim in np.flatnonzero(~tested):
                dim_cube[dim] = False
                tested[dim] = True
                cand = yield from _ellipsoid_task(points[:, ~dim_cube],
                                                  enlarge_per_dim)
                cand_v = ellipsoid_log_volume(cand['B'])
                if log_v > cand_v:
                    ell, log_v = cand, cand_v
                    tested[dim_cube] = False
                else:
                    dim_cube[dim] = True
    if np.all(dim_cube):
        ell = None
    return dim_cube, ell


def mixture_params_batch(point_sets, enlarge_per_dim=1.1):
    return run_tasks([_mixture_task(np.asarray(p), enlarge_per_dim)
                      for p in point_sets])


def mixture_params(points, enlarge_per_dim=1.1):
    """UnitCubeEllipsoidMixture.compute (bounds/basic.py:471-563): returns
    (dim_cube, ellipsoid dict or None)."""
    return mixture_params_batch([points], enlarge_per_dim)[0]


def ellipsoids_overlap(params):
    """Exact pairwise intersection test (bounds/union.py:14-40): minimise
    1 - d^T (A1^-1/(1-s) + A2^-1/s)^-1 d over s in (0, 1)."""
    cs = [p['c'] for p in params]
    covs = [np.linalg.inv(p['A']) for p in params]
    for i, j in itertools.combinations(range(len(cs)), 2):
        delta = cs[i] - cs[j]

        def k(s):
            return 1 - delta @ np.linalg.inv(
                covs[i] / (1 - s) + covs[j] / s) @ delta
        if minimize(k, 0.5, bounds=[(1e-9, 1 - 1e-9)]).fun > 0:
            return True
    return False


N_INIT = 10          # bounds/union.py:186


class _Mixture:
    """The three attributes of a fitted sklearn GaussianMixture that
    Union.split reads (union.py:188-190)."""

    def __init__(self, weights, means, covariances, lower_bound, logp=None):
        self.weights_ = weights
        self.means_ = means
        self.covariances_ = covariances
        self.lower_bound_ = lower_bound
        self.logp = logp          # (2, n) cuda tensor from the device fit


class DegenerateMixture(ValueError):
    """Every restart of the two-component mixture fit ended with an empty
    cluster or a covariance that is not positive definite."""


def _best_of_inits(points_t, random_state):
    """Best of N_INIT restarts of the two-component mixture
    (mixture/_base.py:fit_predict keeps the largest lower bound).  All
    restarts run concurrently on the GPU (``nb_gmm_fit``, n_dim <= 128)."""
    from . import device
    fits = [f for f in device.gmm_fit(points_t, n_init=N_INIT,
                                      seed=random_state) if not f['failed']]
    if not fits:
        # scikit-learn's fit raises here as well (mixture/
        # _gaussian_mixture.py, _compute_precision_cholesky: "Fitting the
        # mixture model failed because some components have ill-defined
        # empirical covariance"); nothing on this path falls back to the host
        raise DegenerateMixture(
            'Fitting the mixture model failed because some components have '
            'ill-defined empirical covariance (for instance caused by '
            'singleton or collapsed samples) in every one of the %d device '
            'restarts (%d points, %d dimensions).' %
            (N_INIT, points_t.shape[0], points_t.shape[1]))
    best = max(fits, key=lambda f: f['lower_bound'])
    return _Mixture(best['weights'], best['means'], best['covariances'],
                    best['lower_bound'], best['logp'])


def two_component_labels(points_t, n_points_min, random_state):
    """Hard assignment of points to the two components of a full-covariance
    Gaussian mixture, re-balanced so that both clusters keep at least
    ``n_points_min`` members (bounds/union.py:185-197).  The EM fit
    (scikit-learn's ``GaussianMixture`` restated) leaves the
    weighted log probabilities of its final parameters on the device; only
    2 n doubles come back."""
    gmm = _best_of_inits(points_t, random_state)
    logp = gmm.logp.t().cpu().numpy()
    labels = np.argmax(logp, axis=1)
    if not np.all(np.bincount(labels, minlength=2) >= n_points_min):
        small = np.argmin(np.bincount(labels, minlength=2))
        labels[np.argsort(-logp[:, small])[:n_points_min]] = small
    return labels


def log_volume_union(log_v_all, n_reject, n_sample):
    """bounds/union.py:342-343."""
    return logsumexp(log_v_all) + np.log(1.0 - n_reject / n_sample)
