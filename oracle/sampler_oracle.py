"""Oracle restatement of the reference's driver for the shell-filling path:
``Sampler.sample_shell / evaluate_likelihood / update_shell_info / add_samples
/ add_bound / run`` and the evidence / ESS reductions (nautilus/sampler.py).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Only the features the
BASELINE configs use are restated: identity-style callable prior or none,
``vectorized`` likelihood (or per-point), no blobs, no checkpointing.  An
optional ``pool`` attribute (an object with ``map`` / ``size``) switches on
the reference's multiprocess path: bounds replicate themselves over the pool
in ``sample`` (bounds/nautilus.py:223-237) and a vectorized likelihood gets
one chunk per worker (sampler.py:860-873) -- used by ``bench.py``'s
``cpu_baseline`` leg.
"""

from time import time

import numpy as np
from scipy.special import logsumexp
from threadpoolctl import threadpool_limits

from .bounds_oracle import OCube, ONautilus


def shell_stats(log_l, bound_log_v, shell_n_sample):
    """Per-shell summary, nautilus/sampler.py:927-943.

    Returns (shell_log_v, shell_log_l, shell_n_eff)."""
    n = len(log_l)
    if n == 0:
        return -np.inf, np.nan, 0.0
    log_v = bound_log_v + np.log(n / shell_n_sample)
    mean_l = logsumexp(log_l) - np.log(n)
    if not np.all(log_l == -np.inf):
        n_eff = np.exp(2 * logsumexp(log_l) - logsumexp(2 * log_l))
    else:
        n_eff = float(n)
    return log_v, mean_l, n_eff


def evidence(shell_log_l, shell_log_v):
    """nautilus/sampler.py:691-694."""
    keep = ~np.isnan(shell_log_l)
    return logsumexp(shell_log_l[keep] + shell_log_v[keep])


def total_n_eff(shell_log_l, shell_log_v, shell_n_eff):
    """nautilus/sampler.py:659-665."""
    if np.all(shell_n_eff == 0):
        return 0
    keep = shell_n_eff > 0
    s = shell_log_l + shell_log_v
    w = np.exp(s - np.nanmax(s))[keep]
    return np.sum(w)**2 / np.sum(w**2 / shell_n_eff[keep])


def point_log_weights(shell_log_v, shell_n, log_l_per_shell):
    """Importance weights, nautilus/sampler.py:602-606 and 642."""
    log_v = np.repeat(shell_log_v - np.log(np.maximum(shell_n, 1)), shell_n)
    log_w = log_v + np.concatenate(log_l_per_shell)
    return log_w - logsumexp(log_w)


class OSampler:
    """Minimal single-process restatement of ``nautilus.Sampler``."""

    def __init__(self, prior, likelihood, n_dim, n_live=2000, n_update=None,
                 enlarge_per_dim=1.1, n_points_min=None, split_threshold=100,
                 n_networks=4, neural_network_kwargs={}, n_batch=None,
                 n_like_new_bound=None, vectorized=False, seed=None,
                 periodic=None):
        self.prior = prior
        self.periodic = periodic
        self.likelihood = likelihood
        self.n_dim = n_dim
        if n_dim <= 1:
            raise ValueError('Cannot run Nautilus with less than 2 '
                             'parameters.')
        self.n_live = n_live
        self.n_update = n_live if n_update is None else n_update
        self.n_like_new_bound = (10 * n_live if n_like_new_bound is None
                                 else n_like_new_bound)
        self.enlarge_per_dim = enlarge_per_dim
        self.n_points_min = (n_dim + 50 if n_points_min is None
                             else n_points_min)
        self.split_threshold = split_threshold
        self.n_networks = n_networks
        self.neural_network_kwargs = neural_network_kwargs
        self.vectorized = vectorized
        self.n_batch = 100 if n_batch is None else n_batch
        self.rng = np.random.default_rng(seed)        # sampler.py:305

        self.n_like = 0
        self.explored = False
        self.bounds = []
        self.points = []
        self.log_l = []
        self._discard = False
        self.shell_n = np.zeros(0, dtype=int)
        self.shell_n_sample = np.zeros(0, dtype=int)
        self.shell_n_eff = np.zeros(0)
        self.shell_log_l_min = np.zeros(0)
        self.shell_log_l = np.zeros(0)
        self.shell_log_v = np.zeros(0)
        self.shell_n_sample_exp = np.zeros(0, dtype=int)
        self.shell_end_exp = np.zeros(0, dtype=int)
        self.points_t = np.zeros((0, n_dim))
        self.shell_t = np.zeros(0, dtype=int)
        self.log_l_t = np.zeros(0)
        self.timing = {}

    # -- evidence-related properties -------------------------------------
    @property
    def n_eff(self):
        return total_n_eff(self.shell_log_l, self.shell_log_v,
                           self.shell_n_eff)

    @property
    def log_z(self):
        if np.sum(self.shell_n) == 0:
            return None
        return evidence(self.shell_log_l, self.shell_log_v)

    def _flat_weights(self):
        log_v = np.repeat(
            self.shell_log_v - np.log(np.maximum(self.shell_n, 1)),
            self.shell_n)
        return log_v, np.concatenate(self.log_l)

    @property
    def f_live(self):
        """sampler.py:1158-1169."""
        if self.explored:
            return None
        if np.sum(self.shell_n) == 0:
            return 1.0
        log_v, log_l = self._flat_weights()
        log_w = log_v + log_l
        live = log_w[np.argsort(log_l)][-self.n_live:]
        return np.exp(logsumexp(live) - logsumexp(log_w))

    @property
    def log_v_live(self):
        """sampler.py:1181-1190."""
        if len(self.bounds) == 0:
            return 1.0
        log_v, log_l = self._flat_weights()
        return logsumexp(log_v[np.argsort(log_l)][-self.n_live:])

    # -- main loop ---------------------------------------------------------
    def run(self, f_live=0.01, n_shell=1, n_eff=10000, n_like_max=np.inf,
            discard_exploration=False, timeout=np.inf):
        """sampler.py:415-505."""
        t0 = time()
        if len(self.bounds) == 0:
            self.add_bound()
            self.n_update_iter = -self.n_live
            self.n_like_iter = 0

        def done():
            return (self.explored and np.all(self.shell_n >= n_shell) and
                    self.n_eff >= n_eff)

        while self.n_like < n_like_max and time() - t0 < timeout and \
                not done():
            if not self.explored:
                if ((self.n_update_iter >= self.n_update or
                     self.n_like_iter >= self.n_like_new_bound) and
                        np.sum(self.shell_n) > self.n_live):
                    self.add_bound()
                    self.n_update_iter = 0
                    self.n_like_iter = 0
                self.n_update_iter += self.add_samples(-1)
                self.n_like_iter += self.n_batch
                if self.f_live <= f_live:
                    for s in np.flatnonzero(self.shell_n == 0)[::-1]:
                        self.bounds.pop(s)
                        self.points.pop(s)
                        self.log_l.pop(s)
                        for key in ('shell_n', 'shell_n_sample',
                                    'shell_n_eff', 'shell_log_l_min',
                                    'shell_log_l', 'shell_log_v'):
                            setattr(self, key,
                                    np.delete(getattr(self, key), s))
                    self.shell_n_sample_exp = np.copy(self.shell_n_sample)
                    self.shell_end_exp = np.array(
                        [len(p) for p in self.points])
                    self.explored = True
                    self.set_discard(discard_exploration)
            elif np.any(self.shell_n < n_shell):
                self.add_samples(np.flatnonzero(self.shell_n < n_shell)[0])
            elif self.n_eff < n_eff:
                self.add_samples(np.argmax(
                    self.shell_log_l + self.shell_log_v -
                    0.5 * np.log(self.shell_n) -
                    0.5 * np.log(self.shell_n_eff)))
        return done()

    def set_discard(self, flag):
        """sampler.py:519-539."""
        self._discard = flag
        for s in range(len(self.log_l)):
            self.update_shell_info(s)

    def posterior(self):
        """sampler.py:597-647 (unweighted branch, identity transform)."""
        if self._discard and self.explored:
            start = self.shell_end_exp
        else:
            start = np.zeros(len(self.points), dtype=int)
        pts = np.concatenate([p[s:] for p, s in zip(self.points, start)])
        log_l = np.concatenate([ll[s:] for ll, s in zip(self.log_l, start)])
        log_v = np.repeat(self.shell_log_v -
                          np.log(np.maximum(self.shell_n, 1)), self.shell_n)
        log_w = log_v + log_l
        return pts, log_w - logsumexp(log_w), log_l

    def shell_association(self, x, n_max=None):
        """Highest-index bound containing each point, sampler.py:1210-1221."""
        if n_max is None:
            n_max = len(self.bounds)
        shell = np.repeat(-1, len(x))
        for i in range(n_max - 1, -1, -1):
            todo = shell < 0
            if not np.any(todo):
                break
            hit = self.bounds[i].contains(x[todo])
            idx = np.flatnonzero(todo)[hit]
            shell[idx] = i
        return shell

    @threadpool_limits.wrap(limits=1)                   # sampler.py:789
    def sample_shell(self, index, shell_t=None):
        """sampler.py:784-830."""
        n_bound = 0
        n_have = 0
        idx_t = np.zeros(0, dtype=int)
        chunks = []
        while n_have < self.n_batch:
            # sampler.py:791-792: the bound replicates itself over pool_s
            x = self.bounds[index].sample(self.n_batch - n_have,
                                          pool=getattr(self, 'pool', None))
            n_bound += self.n_batch - n_have
            keep = np.ones(len(x), dtype=bool)
            for later in self.bounds[index:][1:]:
                keep = keep & ~later.contains(x)
            x = x[keep]
            swap = np.zeros(len(x), dtype=bool)
            if shell_t is not None and len(shell_t) > 0:
                shell_p = self.shell_association(
                    x, n_max=len(self.bounds) - 1)
                for s in range(len(self.bounds) - 1):
                    cand = np.flatnonzero(shell_t == s)
                    fresh = np.flatnonzero(shell_p == s)
                    m = min(len(cand), len(fresh))
                    if m > 0:
                        idx_t = np.append(idx_t, self.rng.choice(
                            cand, size=m, replace=False))
                        shell_t[idx_t] = -1
                        swap[self.rng.choice(fresh, size=m,
                                             replace=False)] = True
            x = x[~swap]
            if len(x) > 0:
                chunks.append(x)
                n_have += len(x)
        x = np.concatenate(chunks)
        if shell_t is None:
            return x, n_bound
        return x, n_bound, idx_t

    def evaluate_likelihood(self, x):
        """sampler.py:856-908 without pools / dict priors / blobs."""
        pool = getattr(self, 'pool', None)
        if self.vectorized and pool is not None:
            # sampler.py:860-873: one chunk per worker of pool_l
            args = x if self.prior is None else self.prior(x)
            log_l = np.concatenate([np.atleast_1d(r) for r in pool.map(
                self.likelihood, np.array_split(args, pool.size))]).astype(
                    float)
        elif self.vectorized:
            args = x if self.prior is None else self.prior(x)
            log_l = np.asarray(self.likelihood(args), float)
        else:
            rows = np.copy(x)
            if self.prior is not None:
                rows = [self.prior(r) for r in rows]
            log_l = np.array([self.likelihood(r) for r in rows])
        self.n_like += len(log_l)
        return log_l

    def update_shell_info(self, s):
        """sampler.py:919-943."""
        n_sample = self.shell_n_sample[s]
        if self._discard and self.explored:
            start = self.shell_end_exp[s]
            n_sample = n_sample - self.shell_n_sample_exp[s]
        else:
            start = 0
        log_l = self.log_l[s][start:]
        self.shell_n[s] = len(log_l)
        if len(log_l) > 0:
            v, l, e = shell_stats(log_l, self.bounds[s].log_v, n_sample)
        else:
            v, l, e = -np.inf, np.nan, 0
        self.shell_log_v[s], self.shell_log_l[s], self.shell_n_eff[s] = v, l, e

    def add_bound(self):
        """sampler.py:999-1091."""
        t0 = time()
        if len(self.bounds) == 0:
            log_l_min = -np.inf
            self.bounds.append(OCube(self.n_dim, rng=self.rng))
            ok = True
        else:
            log_l = np.concatenate(self.log_l)
            pts = np.concatenate(self.points)[np.argsort(log_l)]
            log_l = np.sort(log_l)
            log_l_min = log_l[-self.n_live]
            if (np.sum(log_l == log_l_min) > 1 and
                    np.sum(log_l > log_l_min) >= self.n_points_min):
                log_l_min = np.amin(log_l[log_l > log_l_min])
            if np.all(log_l >= log_l_min):
                ok = False
            else:
                with threadpool_limits(limits=1):        # sampler.py:1022
                    b = ONautilus.build(
                        pts, log_l, log_l_min, self.log_v_live,
                        enlarge_per_dim=self.enlarge_per_dim,
                        n_points_min=self.n_points_min,
                        split_threshold=self.split_threshold,
                        periodic=self.periodic,
                        n_networks=self.n_networks,
                        neural_network_kwargs=self.neural_network_kwargs,
                        rng=self.rng)
                    b.sample(1000, return_points=False)
                ok = b.log_v < self.bounds[-1].log_v
                if ok:
                    self.bounds.append(b)
        self.timing['add_bound'] = self.timing.get('add_bound', 0) + \
            time() - t0
        if not ok:
            self.shell_log_l_min[-1] = log_l_min
            return False

        self.shell_n = np.append(self.shell_n, 0)
        self.shell_n_sample = np.append(self.shell_n_sample, 0)
        self.shell_n_eff = np.append(self.shell_n_eff, 0)
        self.shell_log_l = np.append(self.shell_log_l, np.nan)
        self.shell_log_v = np.append(self.shell_log_v, np.nan)
        self.shell_log_l_min = np.append(self.shell_log_l_min, log_l_min)
        self.points.append(np.zeros((0, self.n_dim)))
        self.log_l.append(np.zeros(0))

        if len(self.bounds) > 1:                          # :1059-1089
            st, pt, lt = [], [], []
            for s in range(len(self.bounds) - 1):
                inside = self.bounds[-1].contains(self.points[s])
                st.append(np.repeat(s, np.sum(inside)))
                pt.append(self.points[s][inside])
                self.points[s] = self.points[s][~inside]
                lt.append(self.log_l[s][inside])
                self.log_l[s] = self.log_l[s][~inside]
                self.shell_n[s] -= np.sum(inside)
                self.update_shell_info(s)
            self.shell_t = np.concatenate(st)
            self.points_t = np.concatenate(pt)
            self.log_l_t = np.concatenate(lt)
        return True

    def add_samples(self, shell):
        """sampler.py:1115-1144."""
        t0 = time()
        if shell == -1 and len(self.shell_t) > 0:
            x, n_bound, idx_t = self.sample_shell(-1, self.shell_t)
            assert len(x) + len(idx_t) == n_bound
            if len(idx_t) > 0:
                self.points[-1] = np.concatenate(
                    (self.points[-1], self.points_t[idx_t]))
                self.log_l[-1] = np.concatenate(
                    (self.log_l[-1], self.log_l_t[idx_t]))
        else:
            x, n_bound = self.sample_shell(shell)
        self.shell_n_sample[shell] += n_bound
        log_l = self.evaluate_likelihood(x)
        self.points[shell] = np.append(self.points[shell], x, axis=0)
        self.log_l[shell] = np.append(self.log_l[shell], log_l, axis=0)
        self.update_shell_info(shell)
        self.timing['add_samples'] = self.timing.get('add_samples', 0) + \
            time() - t0
        return np.sum(log_l >= self.shell_log_l_min[shell])
