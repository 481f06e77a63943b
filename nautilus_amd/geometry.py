"""Host-side construction of the bound geometry (small, sequential numerics).

Bound *construction* runs a handful of times per bound on ~n_live points.  Its
numerically heavy piece, the Khachiyan iteration of the minimum-volume
enclosing ellipsoid, runs on the GPU (``nb_mvee_weights``); the O(D^3)
finishing steps, the greedy cube/ellipsoid choice, the two-component mixture
split and the ellipsoid overlap test stay on the host (SURVEY.md section 8,
rows f1/f2).  What is built here is uploaded once through ``nb_bound_create``;
every per-point operation afterwards (draw, contains, emulator, compaction)
runs on the GPU.

Semantics follow the reference (paths relative to /root/reference/nautilus):
bounds/basic.py:154-241 (MVEE), 265-316 (Ellipsoid.compute), 471-563
(UnitCubeEllipsoidMixture.compute); bounds/union.py:14-40 (overlap test),
153-229 (split), 231-267 (trim).
"""

import itertools

import numpy as np
from scipy.linalg.lapack import dpotrf, dpotri
from scipy.optimize import minimize
from scipy.special import gammaln, logsumexp
from threadpoolctl import threadpool_limits


def inv_spd(m):
    """Inverse of a symmetric positive definite matrix via Cholesky."""
    tri = dpotri(dpotrf(m)[0])[0]
    return tri + tri.T - np.diag(np.diag(tri))


def khachiyan_weights_host(points, n_max=100, n_batch=20):
    """Weights u of the batched Khachiyan iteration (reference
    bounds/basic.py:175-232) on the host -- used for n_dim > 63, where the
    device kernel's matrices no longer fit the LDS.  Unlike the reference the
    (n, D+1, D+1) tensor of outer products is never materialised: the
    quadratic forms are computed as row sums of (Q V^-1) * Q."""
    n, d = points.shape
    q = np.empty((n, d + 1))
    q[:, :d] = points
    q[:, d] = 1.0
    u = np.full(n, 1.0 / n)
    v = (q * u[:, None]).T @ q
    for _ in range(n_max):
        # V^-1 is re-factorised once per sweep and then tracked through the
        # <= n_batch rank-one updates with the Sherman-Morrison identity
        # (the reference re-factorises after every update, basic.py:230)
        v_inv = inv_spd(v)
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


def khachiyan_weights(points, n_max=100, n_batch=20):
    """The iteration runs on the GPU (``nb_mvee_weights``: one persistent
    workgroup, quadratic forms on the matrix cores) whenever the dimension
    allows it."""
    from . import device
    if points.shape[1] <= device.MVEE_MAX_DIM:
        return device.mvee_weights(points, n_max, n_batch).cpu().numpy()
    return khachiyan_weights_host(points, n_max, n_batch)


def mvee(points, n_max=100, n_batch=20):
    """Minimum-volume enclosing ellipsoid (reference bounds/basic.py:175-241):
    Khachiyan weights, then centre / covariance / scaling (:233-241).

    Returns centre c, shape matrix A ((x-c)^T A (x-c) <= 1) and A^-1.
    """
    points = np.ascontiguousarray(points, dtype=float)
    u = khachiyan_weights(points, n_max, n_batch)
    c = np.atleast_1d(np.average(points, weights=u, axis=0))
    a_inv = np.atleast_2d(np.cov(points, aweights=u, rowvar=False, bias=True))
    a = np.linalg.inv(a_inv)
    diff = points - c
    scale = np.amax(np.einsum('ij,ij->i', diff @ a, diff))
    return c, a / scale, a_inv * scale


def ellipsoid_params(points, enlarge_per_dim=1.1):
    """Ellipsoid.compute (bounds/basic.py:265-316): returns dict(c, A, B,
    B_inv) with B = chol(A^-1) lower triangular and B_inv = B^-1."""
    n, d = points.shape
    if enlarge_per_dim < 1.0:
        raise ValueError("The 'enlarge_per_dim' factor cannot be smaller "
                         "than unity.")
    if not n > d:
        raise ValueError('Number of points must be larger than number '
                         'dimensions.')
    with threadpool_limits(limits=1):
        c, a, a_inv = mvee(points)
    a = a / enlarge_per_dim**2.0
    a_inv = a_inv * enlarge_per_dim**2.0
    b = np.linalg.cholesky(a_inv)
    b_inv = np.tril(np.linalg.inv(b))
    return dict(c=c, A=a, B=b, B_inv=b_inv)


def ellipsoid_log_volume(b):
    """bounds/basic.py:393-394."""
    d = b.shape[0]
    return (np.linalg.slogdet(b)[1] + d * np.log(2.) + d * gammaln(1.5) -
            gammaln(d / 2.0 + 1))


def mixture_params(points, enlarge_per_dim=1.1):
    """Greedy choice of the dimensions bounded by the unit cube
    (bounds/basic.py:471-563).  Returns (dim_cube, ellipsoid dict or None)."""
    d = points.shape[1]
    ell = ellipsoid_params(points, enlarge_per_dim)
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
        cand = ellipsoid_params(points[:, ~dim_cube], enlarge_per_dim)
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
                cand = ellipsoid_params(points[:, ~dim_cube], enlarge_per_dim)
                cand_v = ellipsoid_log_volume(cand['B'])
                if log_v > cand_v:
                    ell, log_v = cand, cand_v
                    tested[dim_cube] = False
                else:
                    dim_cube[dim] = True
    if np.all(dim_cube):
        ell = None
    return dim_cube, ell


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
_GMM_POOL = None


class _Mixture:
    """The three attributes of a fitted sklearn GaussianMixture that
    Union.split reads (union.py:188-190)."""

    def __init__(self, weights, means, covariances, lower_bound):
        self.weights_ = weights
        self.means_ = means
        self.covariances_ = covariances
        self.lower_bound_ = lower_bound


def _fit_one(args):
    from sklearn.mixture import GaussianMixture
    points_t, seed = args
    with threadpool_limits(limits=1):
        return GaussianMixture(n_components=2, n_init=1,
                               random_state=seed).fit(points_t)


def _best_of_inits_host(points_t, random_state):
    """scikit-learn on host threads -- used for n_dim > 63 only.  The
    reference runs the restarts sequentially inside one
    ``GaussianMixture(n_init=10)`` call; here they get seeds derived from
    ``random_state`` and run concurrently (numpy releases the GIL inside
    BLAS).  Same estimator, same selection rule (largest lower bound)."""
    global _GMM_POOL
    from concurrent.futures import ThreadPoolExecutor
    seeds = np.random.RandomState(random_state).randint(2**31 - 1,
                                                        size=N_INIT)
    if _GMM_POOL is None:
        _GMM_POOL = ThreadPoolExecutor(max_workers=N_INIT)
    fits = list(_GMM_POOL.map(_fit_one, [(points_t, int(sd)) for sd in seeds]))
    return max(fits, key=lambda g: g.lower_bound_)


def _best_of_inits(points_t, random_state):
    """Best of N_INIT restarts of the two-component mixture
    (mixture/_base.py:fit_predict keeps the largest lower bound).  All
    restarts run concurrently on the GPU (``nb_gmm_fit``)."""
    from . import device
    if points_t.shape[1] > device.GMM_MAX_DIM:
        return _best_of_inits_host(points_t, random_state)
    fits = [f for f in device.gmm_fit(points_t, n_init=N_INIT,
                                      seed=random_state) if not f['failed']]
    if not fits:
        # every restart hit an empty cluster or a covariance that is not
        # positive definite (degenerate point sets): scikit-learn relocates
        # empty clusters and may still succeed -- let it decide
        return _best_of_inits_host(points_t, random_state)
    best = max(fits, key=lambda f: f['lower_bound'])
    return _Mixture(best['weights'], best['means'], best['covariances'],
                    best['lower_bound'])


def two_component_labels(points_t, n_points_min, random_state):
    """Hard assignment of points to the two components of a full-covariance
    Gaussian mixture, re-balanced so that both clusters keep at least
    ``n_points_min`` members (bounds/union.py:185-197).  The EM fit itself is
    scikit-learn's ``GaussianMixture`` -- the reference's own dependency for
    this step."""
    from scipy.stats import multivariate_normal
    gmm = _best_of_inits(points_t, random_state)
    logp = np.vstack([multivariate_normal.logpdf(
        points_t, mean=gmm.means_[i], cov=gmm.covariances_[i]) +
        np.log(gmm.weights_[i]) for i in range(2)]).T
    labels = np.argmax(logp, axis=1)
    if not np.all(np.bincount(labels, minlength=2) >= n_points_min):
        small = np.argmin(np.bincount(labels, minlength=2))
        labels[np.argsort(-logp[:, small])[:n_points_min]] = small
    return labels


def log_volume_union(log_v_all, n_reject, n_sample):
    """bounds/union.py:342-343."""
    return logsumexp(log_v_all) + np.log(1.0 - n_reject / n_sample)
