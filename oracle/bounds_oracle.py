"""Oracle restatement of the reference's bound geometry (numpy, fp64).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Each routine cites the
reference lines (relative to /root/reference) whose behaviour it restates.
RNG consumption order is kept identical to the reference so that golden
vectors generated from the reference with the same ``numpy.random.Generator``
seed match exactly.
"""

import itertools

import numpy as np
from scipy.linalg.lapack import dpotrf, dpotri
from scipy.optimize import minimize
from scipy.special import gammaln, logsumexp
from scipy.stats import multivariate_normal, rankdata
from threadpoolctl import threadpool_limits

from . import mlp_oracle

CHUNK = 1000  # rejection-loop chunk, nautilus/bounds/union.py:306, nautilus.py:215


def _rng(rng):
    return np.random.default_rng() if rng is None else rng


# --------------------------------------------------------------------------
# linear algebra helpers
# --------------------------------------------------------------------------

def spd_inverse(m):
    """nautilus/bounds/basic.py:154-172 (dpotrf + dpotri, mirrored triangle)."""
    tri = dpotri(dpotrf(m)[0])[0]
    return tri + tri.T - np.diag(np.diag(tri))


def mvee(points, n_max=100, n_batch=20):
    """Batched Khachiyan MVEE, nautilus/bounds/basic.py:175-241.

    Returns (c, A, A_inv) with (x-c)^T A (x-c) <= 1 for every input point.
    """
    n, d = points.shape
    q = np.append(points, np.ones((n, 1)), axis=1)
    u = np.repeat(1.0 / n, n)
    outer = np.array([np.outer(row, row) for row in q])   # basic.py:214

    for it in range(n_max):
        if it % 1000 == 0:                                  # basic.py:217-219
            v = np.einsum('ji,j,jk', q, u, q)
            v_inv = spd_inverse(v)
        g = np.einsum('ijk,jk', outer, v_inv)               # basic.py:220
        for j in np.argsort(g)[-n_batch:][::-1]:            # basic.py:221
            # basic.py:222-225: first pass indexes the vector g, afterwards g
            # is a scalar and the quadratic form is recomputed with the
            # current v_inv.
            try:
                g = g[j]
            except IndexError:
                g = np.einsum('jk,jk', outer[j], v_inv)
            if g < d + 1:
                continue
            a = (g - (d + 1)) / ((d + 1) * (g - 1))
            v = v * (1 - a) + a * outer[j]
            v_inv = spd_inverse(v)
            u = u * (1 - a) + a * (np.arange(n) == j)

    c = np.atleast_1d(np.average(points, weights=u, axis=0))      # basic.py:233
    a_inv = np.atleast_2d(np.cov(points, aweights=u, rowvar=False, bias=True))
    a_mat = np.linalg.inv(a_inv)
    scale = np.amax(np.einsum('...i,ij,...j', points - c, a_mat, points - c))
    a_mat /= scale
    a_inv *= scale
    return c, a_mat, a_inv


# --------------------------------------------------------------------------
# primitive bounds
# --------------------------------------------------------------------------

class OCube:
    """Unit hyper-cube, nautilus/bounds/basic.py:9-151."""

    def __init__(self, n_dim, rng=None):
        self.n_dim = n_dim
        self.rng = _rng(rng)

    def contains(self, x):
        return np.all((x >= 0) & (x < 1), axis=-1)          # basic.py:67

    def sample(self, n=100, pool=None):
        return self.rng.random(size=(n, self.n_dim))        # basic.py:85

    log_v = 0                                               # basic.py:98

    def reset(self, rng=None):
        if rng is not None:
            self.rng = rng


class OEllipsoid:
    """nautilus/bounds/basic.py:244-449."""

    @classmethod
    def build(cls, points, enlarge_per_dim=1.1, rng=None):
        """basic.py:265-316."""
        self = cls()
        self.n_dim = points.shape[1]
        if enlarge_per_dim < 1.0:
            raise ValueError("The 'enlarge_per_dim' factor cannot be smaller "
                             "than unity.")
        if not points.shape[0] > self.n_dim:
            raise ValueError('Number of points must be larger than number '
                             'dimensions.')
        with threadpool_limits(limits=1):               # basic.py:302
            self.c, self.A, a_inv = mvee(points)
        self.A /= enlarge_per_dim**2.0
        a_inv *= enlarge_per_dim**2.0
        self.B = np.linalg.cholesky(a_inv)
        self.B_inv = np.linalg.inv(self.B)
        self.rng = _rng(rng)
        return self

    @classmethod
    def from_params(cls, c, B, B_inv=None, A=None, rng=None):
        self = cls()
        self.c = np.asarray(c, float)
        self.n_dim = len(self.c)
        self.B = np.asarray(B, float)
        self.B_inv = np.linalg.inv(self.B) if B_inv is None else B_inv
        self.A = self.B_inv.T @ self.B_inv if A is None else A
        self.rng = _rng(rng)
        return self

    def transform(self, x, inverse=False):
        """basic.py:339-342."""
        if not inverse:
            return np.einsum('ij, ...j', self.B_inv, x - self.c)
        return np.einsum('ij, ...j', self.B, x) + self.c

    def contains(self, x):
        return np.sum(self.transform(x)**2, axis=-1) < 1     # basic.py:360

    def sample(self, n=100):
        """basic.py:376-381: normal block first, then the uniform block."""
        z = self.rng.normal(size=(n, self.n_dim))
        z = z / np.sqrt(np.sum(z**2, axis=1))[:, np.newaxis]
        z *= self.rng.uniform(size=n)[:, np.newaxis]**(1.0 / self.n_dim)
        return self.transform(z, inverse=True)

    @property
    def log_v(self):
        """basic.py:393-394."""
        return (np.linalg.slogdet(self.B)[1] + self.n_dim * np.log(2.) +
                self.n_dim * gammaln(1.5) - gammaln(self.n_dim / 2.0 + 1))

    def reset(self, rng=None):
        if rng is not None:
            self.rng = rng


class OMixture:
    """Cube/ellipsoid mixture, nautilus/bounds/basic.py:452-726."""

    @classmethod
    def build(cls, points, enlarge_per_dim=1.1, rng=None):
        """Greedy choice of cube-bounded dimensions, basic.py:471-563."""
        self = cls()
        d = self.n_dim = points.shape[1]
        kw = dict(enlarge_per_dim=enlarge_per_dim, rng=rng)
        ell = OEllipsoid.build(points, **kw)
        self.dim_cube = np.zeros(d, dtype=bool)

        while np.sum(~self.dim_cube) > 1:                   # basic.py:501
            c = ell.c
            a_inv = np.linalg.inv(ell.A)
            n_free = np.sum(~self.dim_cube)
            lv = np.zeros(n_free)
            for i in range(n_free):                         # basic.py:509-517
                p_proj = np.delete(points[:, ~self.dim_cube], i, axis=1)
                c_proj = np.delete(c, i)
                a_inv_proj = np.delete(np.delete(a_inv, i, axis=0), i, axis=1)
                a_proj = np.linalg.inv(a_inv_proj)
                scale = np.amax(np.einsum('...i,ij,...j', p_proj - c_proj,
                                          a_proj, p_proj - c_proj))
                a_proj /= scale
                lv[i] = np.linalg.slogdet(np.linalg.inv(a_proj))[1]
            dim = np.arange(d)[~self.dim_cube][np.argmin(lv)]
            self.dim_cube[dim] = True
            trial = OEllipsoid.build(points[:, ~self.dim_cube], **kw)
            if trial.log_v < ell.log_v:
                ell = trial
            else:
                self.dim_cube[dim] = False
                break

        if ell.log_v > 0:                                   # basic.py:535-551
            ell = OCube(points)        # basic.py:536 quirk: only log_v=0 is used
            self.dim_cube = np.ones(d, dtype=bool)
            tested = np.zeros(d, dtype=bool)
            while ~np.all(tested):
                for dim in np.arange(d)[~tested]:
                    self.dim_cube[dim] = False
                    tested[dim] = True
                    trial = OEllipsoid.build(points[:, ~self.dim_cube], **kw)
                    if ell.log_v > trial.log_v:
                        ell = trial
                        tested[self.dim_cube] = False
                    else:
                        self.dim_cube[dim] = True

        self.cube = (OCube(int(np.sum(self.dim_cube)), rng=rng)
                     if np.any(self.dim_cube) else None)
        self.ellipsoid = None if np.all(self.dim_cube) else ell
        return self

    @classmethod
    def from_params(cls, dim_cube, ellipsoid, rng=None):
        self = cls()
        self.dim_cube = np.asarray(dim_cube, bool)
        self.n_dim = len(self.dim_cube)
        self.cube = (OCube(int(np.sum(self.dim_cube)), rng=rng)
                     if np.any(self.dim_cube) else None)
        self.ellipsoid = ellipsoid
        return self

    def transform(self, x):
        """basic.py:585-592."""
        y = np.copy(x)
        if self.cube is not None:
            idx = np.arange(self.n_dim)[self.dim_cube]
            y[:, idx] = x[:, idx] * 2 - 1
        if self.ellipsoid is not None:
            idx = np.arange(self.n_dim)[~self.dim_cube]
            y[:, idx] = self.ellipsoid.transform(x[:, idx])
        return y

    def contains(self, x):
        """basic.py:610-617."""
        ok = np.ones(x.shape[:-1], dtype=bool)
        if self.cube is not None:
            idx = np.arange(self.n_dim)[self.dim_cube]
            ok = ok & self.cube.contains(x[..., idx])
        if self.ellipsoid is not None:
            idx = np.arange(self.n_dim)[~self.dim_cube]
            ok = ok & self.ellipsoid.contains(x[..., idx])
        return ok

    def sample(self, n=100):
        """basic.py:633-640: cube columns are drawn before ellipsoid columns."""
        x = np.zeros((n, self.n_dim))
        if self.cube is not None:
            x[:, np.arange(self.n_dim)[self.dim_cube]] = self.cube.sample(n)
        if self.ellipsoid is not None:
            x[:, np.arange(self.n_dim)[~self.dim_cube]] = \
                self.ellipsoid.sample(n)
        return x

    @property
    def log_v(self):
        return 0 if self.ellipsoid is None else self.ellipsoid.log_v

    def reset(self, rng=None):
        if rng is not None:
            if self.ellipsoid is not None:
                self.ellipsoid.reset(rng)
            if self.cube is not None:
                self.cube.reset(rng)


# --------------------------------------------------------------------------
# unions
# --------------------------------------------------------------------------

def ellipsoids_overlap(ells):
    """Pairwise exact intersection test, nautilus/bounds/union.py:14-40."""
    cs = [e.c for e in ells]
    a_invs = [np.linalg.inv(e.A) for e in ells]
    for i, j in itertools.combinations(range(len(cs)), 2):
        dvec = cs[i] - cs[j]

        def k(s):
            return 1 - np.dot(np.dot(dvec, np.linalg.inv(
                a_invs[i] / (1 - s) + a_invs[j] / s)), dvec)
        if minimize(k, 0.5, bounds=[(1e-9, 1 - 1e-9)]).fun > 0:
            return True
    return False


class OUnion:
    """Union of ellipsoids / mixtures, nautilus/bounds/union.py:43-451."""

    @classmethod
    def build(cls, points, enlarge_per_dim=1.1, n_points_min=None, unit=True,
              member_cls=OEllipsoid, rng=None):
        """union.py:78-151."""
        self = cls()
        self.n_dim = points.shape[1]
        self.enlarge_per_dim = enlarge_per_dim
        if n_points_min is None:
            self.n_points_min = self.n_dim + 1
        else:
            if n_points_min < self.n_dim + 1:
                raise ValueError('The number of points per bound must be '
                                 'larger than the number of dimensions.')
            self.n_points_min = n_points_min
        self.cube = OCube(self.n_dim, rng=rng) if unit else None
        self.points_bounds = [points]
        self.bounds = [member_cls.build(points, enlarge_per_dim=enlarge_per_dim,
                                        rng=rng)]
        self.log_v_all = np.array([self.bounds[0].log_v])
        self.block = np.atleast_1d(len(points) < 2 * self.n_points_min)
        self.points = np.zeros((0, self.n_dim))
        self.n_sample = 0
        self.n_reject = 0
        self.rng = _rng(rng)
        return self

    @classmethod
    def from_members(cls, members, unit=True, rng=None):
        self = cls()
        self.bounds = list(members)
        self.n_dim = members[0].n_dim
        self.cube = OCube(self.n_dim, rng=rng) if unit else None
        self.log_v_all = np.array([m.log_v for m in members])
        self.points = np.zeros((0, self.n_dim))
        self.n_sample = 0
        self.n_reject = 0
        self.rng = _rng(rng)
        return self

    def split(self, allow_overlap=True):
        """Two-component GMM split of the largest splittable member,
        union.py:153-229 (sklearn GaussianMixture is the reference's own
        third-party dependency here)."""
        from sklearn.mixture import GaussianMixture
        if not allow_overlap and not isinstance(self.bounds[0], OEllipsoid):
            raise ValueError("'allow_overlap' can only be False if bounds are "
                             "ellipsoids.")
        if not np.any(~self.block):
            return False
        index = np.argmax(np.where(~self.block, self.log_v_all, -np.inf))
        pts_t = self.bounds[index].transform(self.points_bounds[index])
        gmm = GaussianMixture(
            n_components=2, n_init=10,
            random_state=self.rng.integers(2**32 - 1)).fit(pts_t)
        p = np.vstack([multivariate_normal.logpdf(
            pts_t, mean=gmm.means_[i], cov=gmm.covariances_[i]) +
            np.log(gmm.weights_[i]) for i in range(2)]).T
        labels = np.argmax(p, axis=1)
        if not np.all(np.bincount(labels) >= self.n_points_min):  # :195-197
            small = np.argmin(np.bincount(labels))
            labels[np.argsort(-p[:, small])[:self.n_points_min]] = small

        pts = self.points_bounds[index]
        fresh = [type(self.bounds[0]).build(
            pts[labels == lab], enlarge_per_dim=self.enlarge_per_dim,
            rng=self.rng) for lab in (0, 1)]

        if not allow_overlap and ellipsoids_overlap(
                self.bounds[:index] + self.bounds[index + 1:] + fresh):
            return False
        if logsumexp([fresh[0].log_v, fresh[1].log_v]) > \
                self.bounds[index].log_v:                      # :210-213
            self.block[index] = True
            return self.split(allow_overlap=allow_overlap)

        self.points_bounds.pop(index)
        self.points_bounds.append(pts[labels == 0])
        self.points_bounds.append(pts[labels == 1])
        self.bounds.pop(index)
        self.bounds = self.bounds + fresh
        self.log_v_all = np.array([b.log_v for b in self.bounds])
        self.block = np.concatenate((
            np.delete(self.block, index),
            [len(self.points_bounds[-2]) < 2 * self.n_points_min,
             len(self.points_bounds[-1]) < 2 * self.n_points_min]))
        self.reset()
        return True

    def trim(self, threshold=1e3):
        """union.py:231-267."""
        if len(self.bounds) == 1:
            return False
        log_n = np.array([np.log(len(p)) for p in self.points_bounds])
        log_v = np.array([b.log_v for b in self.bounds])
        log_r = log_n - log_v
        index = np.argmin(log_r)
        if log_r[index] - np.median(np.delete(log_r, index)) < \
                -np.log(threshold):
            self.points_bounds.pop(index)
            self.bounds.pop(index)
            self.log_v_all = np.array([b.log_v for b in self.bounds])
            self.reset()
            return True
        return False

    def member_count(self, x):
        """k_i = number of members containing x_i (union.py:316-317)."""
        return np.sum([b.contains(x) for b in self.bounds], axis=0)

    def contains(self, x):
        """union.py:285-289."""
        ok = np.any([b.contains(x) for b in self.bounds], axis=0)
        if self.cube is not None:
            ok = ok & self.cube.contains(x)
        return ok

    def sample(self, n=100):
        """Overlap-corrected rejection loop in chunks of 1000, union.py:305-327."""
        while len(self.points) < n:
            p = np.exp(np.array(self.log_v_all) - logsumexp(self.log_v_all))
            per_member = self.rng.multinomial(CHUNK, p)
            x = np.vstack([b.sample(m) for b, m in
                           zip(self.bounds, per_member)])
            if self.cube is not None:
                x = x[self.cube.contains(x)]
            self.rng.shuffle(x)
            k = self.member_count(x)
            x = x[self.rng.random(size=len(x)) > 1 - 1.0 / k]
            self.points = np.vstack([self.points, x])
            self.n_sample += CHUNK
            self.n_reject += CHUNK - len(x)
        out = self.points[:n]
        self.points = self.points[n:]
        return out

    @property
    def log_v(self):
        """union.py:339-343 (draws one chunk if nothing was sampled yet)."""
        if self.n_sample == 0:
            self.sample()
        return logsumexp(self.log_v_all) + np.log(
            1.0 - self.n_reject / self.n_sample)

    def reset(self, rng=None):
        """union.py:441-450."""
        self.points = np.zeros((0, self.n_dim))
        self.n_sample = 0
        self.n_reject = 0
        if rng is not None:
            self.rng = rng
            if self.cube is not None:
                self.cube.reset(rng)
            for b in self.bounds:
                b.reset(rng)


# --------------------------------------------------------------------------
# neural bound and the composite nautilus bound
# --------------------------------------------------------------------------

def rank_scores(log_l, log_l_min):
    """Training targets, nautilus/bounds/neural.py:82-88."""
    score = np.zeros(len(log_l))
    hi = log_l >= log_l_min
    score[hi] = 0.5 * (1 + (rankdata(log_l[hi]) - 0.5) / np.sum(hi))
    score[~hi] = 0.5 * ((rankdata(log_l[~hi]) - 0.5) / np.sum(~hi))
    return score, hi


class ONeural:
    """Ellipsoid AND (emulator score above threshold),
    nautilus/bounds/neural.py:10-174."""

    @classmethod
    def build(cls, points, log_l, log_l_min, enlarge_per_dim=1.1,
              n_networks=4, neural_network_kwargs={}, pool=None, rng=None):
        """bounds/neural.py:58-97."""
        self = cls()
        self.n_dim = points.shape[1]
        rng = _rng(rng)
        self.outer_bound = OEllipsoid.build(
            points[log_l >= log_l_min], enlarge_per_dim=enlarge_per_dim,
            rng=rng)
        if n_networks == 0:
            self.emulator = None
            self.score_predict_min = 0
            return self
        inside = self.outer_bound.contains(points)
        points = points[inside]
        log_l = log_l[inside]
        x_t = self.outer_bound.transform(points)
        score, hi = rank_scores(log_l, log_l_min)
        self.emulator = mlp_oracle.Emulator.train(
            x_t, score, n_networks=n_networks,
            neural_network_kwargs=neural_network_kwargs, pool=pool)
        self.score_predict_min = np.polyval(np.polyfit(
            score, self.emulator.predict(x_t), 3), np.amin(score[hi]))
        return self

    def contains(self, x):
        """bounds/neural.py:115-126."""
        x = np.atleast_2d(x)
        ok = self.outer_bound.contains(x)
        if np.any(ok) and self.emulator is not None:
            x_t = self.outer_bound.transform(x)
            ok[ok] = (self.emulator.predict(x_t[ok]) >
                      self.score_predict_min - 1e-9)
        return ok


class OPhaseShift:
    """Recentring of periodic dimensions, nautilus/bounds/periodic.py:6-72."""

    @classmethod
    def build(cls, points, periodic):
        """periodic.py:21-46: put the largest gap across the boundary."""
        self = cls()
        self.periodic = periodic
        self.centers = np.zeros(len(periodic))
        for i, dim in enumerate(periodic):
            x = np.sort(points[:, dim])
            gaps = np.append(np.diff(x), x[0] - (x[-1] - 1))
            self.centers[i] = (x[np.argmax(gaps)] + np.amax(gaps) / 2.0 +
                               0.5) % 1
        return self

    @classmethod
    def from_params(cls, periodic, centers):
        self = cls()
        self.periodic = np.asarray(periodic)
        self.centers = np.asarray(centers, float)
        return self

    def transform(self, points, inverse=False):
        """periodic.py:50-72."""
        out = np.copy(points)
        sign = -1 if inverse else +1
        for i, dim in enumerate(self.periodic):
            out[:, dim] = (out[:, dim] + sign * (-self.centers[i] + 0.5)) % 1
        return out


class ONautilus:
    """Composite bound, nautilus/bounds/nautilus.py:13-398."""

    shift = None

    @classmethod
    def build(cls, points, log_l, log_l_min, log_v_target,
              enlarge_per_dim=1.1, n_points_min=None, split_threshold=100,
              periodic=None, n_networks=4, neural_network_kwargs={},
              pool=None, rng=None):
        """nautilus.py:88-144."""
        self = cls()
        self.n_dim = points.shape[1]
        if periodic is not None:                                # :91-96
            self.shift = OPhaseShift.build(points[log_l >= log_l_min],
                                           periodic)
            points = self.shift.transform(points)
        self.neural_bounds = []
        live = points[log_l >= log_l_min]

        multi = OUnion.build(live, enlarge_per_dim=enlarge_per_dim,
                             n_points_min=n_points_min,
                             member_cls=OEllipsoid, rng=rng)
        while multi.split(allow_overlap=False):
            pass
        for ell in multi.bounds:
            sel = ell.contains(points)
            self.neural_bounds.append(ONeural.build(
                points[sel], log_l[sel], log_l_min,
                enlarge_per_dim=enlarge_per_dim, n_networks=n_networks,
                neural_network_kwargs=neural_network_kwargs, pool=pool,
                rng=rng))

        self.outer_bound = OUnion.build(
            live, enlarge_per_dim=enlarge_per_dim, n_points_min=n_points_min,
            member_cls=OMixture, rng=rng)
        limit = np.log(split_threshold * enlarge_per_dim**points.shape[1])
        while self.outer_bound.log_v - log_v_target > limit:   # :123-126
            if not self.outer_bound.split():
                break
        while self.outer_bound.log_v - log_v_target > limit:   # :130-133
            if not self.outer_bound.trim():
                break

        self.rng = _rng(rng)
        self.points = np.zeros((0, self.n_dim))
        self.n_sample = 0
        self.n_reject = 0
        return self

    @classmethod
    def from_parts(cls, outer_bound, neural_bounds, rng=None, shift=None):
        self = cls()
        self.n_dim = outer_bound.n_dim
        self.shift = shift
        self.outer_bound = outer_bound
        self.neural_bounds = list(neural_bounds)
        self.rng = _rng(rng)
        self.points = np.zeros((0, self.n_dim))
        self.n_sample = 0
        self.n_reject = 0
        return self

    def neural_contains(self, x):
        return np.any([b.contains(x) for b in self.neural_bounds], axis=0)

    def contains(self, x):
        """nautilus.py:162-169."""
        if self.shift is not None:
            x = self.shift.transform(x)
        ok = self.outer_bound.contains(x)
        if len(self.neural_bounds) > 0:
            ok = ok & self.neural_contains(x)
        return ok

    def _reset_and_sample(self, n=100, rng=None):
        self.reset(rng=rng)
        self.sample(n, return_points=False)
        return self

    def sample(self, n=100, return_points=True, pool=None):
        """nautilus.py:212-244."""
        if len(self.points) < n:
            if pool is None:
                while len(self.points) < n:
                    x = self.outer_bound.sample(CHUNK)
                    x = x[self.neural_contains(x)]
                    self.points = np.vstack([self.points, x])
                    self.n_sample += CHUNK
                    self.n_reject += CHUNK - len(x)
            else:
                from functools import partial
                n_jobs = pool.size
                per_job = (max(n - len(self.points), 10000) // n_jobs) + 1
                func = partial(self._reset_and_sample, per_job)
                rngs = [np.random.default_rng(s) for s in
                        np.random.SeedSequence(self.rng.integers(
                            2**32 - 1)).spawn(n_jobs)]
                for b in pool.map(func, rngs):
                    self.points = np.vstack([self.points, b.points])
                    self.n_sample += b.n_sample
                    self.n_reject += b.n_reject
                    self.outer_bound.n_sample += b.outer_bound.n_sample
                    self.outer_bound.n_reject += b.outer_bound.n_reject
        if return_points:
            out = self.points[:n]
            self.points = self.points[n:]
            if self.shift is not None:                          # :241-243
                out = self.shift.transform(out, inverse=True)
            return out

    @property
    def log_v(self):
        """nautilus.py:257-261."""
        if self.n_sample == 0:
            self.sample(return_points=False)
        return self.outer_bound.log_v + np.log(
            1.0 - self.n_reject / self.n_sample)

    @property
    def n_ell(self):
        return np.sum([np.any(~b.dim_cube) for b in self.outer_bound.bounds])

    @property
    def n_net(self):
        if self.neural_bounds[0].emulator is not None:
            return len(self.neural_bounds) * len(
                self.neural_bounds[0].emulator.networks)
        return 0

    def reset(self, rng=None):
        """nautilus.py:392-397."""
        self.points = np.zeros((0, self.n_dim))
        self.n_sample = 0
        self.n_reject = 0
        self.outer_bound.reset(rng)
        if rng is not None:
            self.rng = rng
