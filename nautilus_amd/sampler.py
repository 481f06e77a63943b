"""Importance nested sampling driver with the ``nautilus.Sampler`` interface,
driving the MI355X hot path.

Control flow follows the reference (nautilus/sampler.py: ``run`` 373-505,
``add_bound`` 982-1091, ``add_samples`` 1093-1144, ``sample_shell`` 751-830,
``update_shell_info`` 910-943, evidence / ESS properties 650-730, 1147-1190)
so that evidence and posterior agree with the reference sampler; what differs
is where the work happens:

* all points stay in HBM (one growable tensor per shell); only ``log_l``
  (8 bytes per point) is mirrored on the host for the driver's bookkeeping;
* proposals, bound membership, shell exclusion and shell association are HIP
  kernels over whole batches (``bounds.py`` / ``device.py``);
* a shell batch is drawn as "trials until the n_batch-th success" over one
  large block of in-bound points -- the same negative-binomial law as the
  reference's shrinking-deficit loop (sampler.py:790-823), with one or two
  kernel launches instead of dozens;
* per-shell evidence statistics are wavefront reductions (``nb_shell_stats``).

A likelihood marked with ``device = True`` (see ``likelihoods.py``) receives
and returns cuda tensors, so nothing crosses PCIe inside the loop; any other
callable is evaluated on the host exactly like in the reference, including
``vectorized`` / ``pass_dict`` / ``pool`` handling.
"""

import os
import warnings
from functools import partial
from time import time

import numpy as np
import torch
from scipy.special import logsumexp

from . import device, geometry
from .bounds import BarrenBound, NautilusBound, UnitCube
from .pool import NautilusPool, likelihood_worker


class _Grow:
    """Append-only device array with amortised doubling (replaces the
    reference's ``np.append`` growth, sampler.py:1135-1136)."""

    def __init__(self, width=None):
        shape = (0,) if width is None else (0, width)
        self.data = torch.empty(shape, dtype=torch.float64, device='cuda')
        self.n = 0

    def view(self):
        return self.data[:self.n]

    def append(self, rows):
        k = rows.shape[0]
        if self.n + k > self.data.shape[0]:
            cap = max(2 * self.data.shape[0], self.n + k, 1024)
            new = torch.empty((cap,) + tuple(self.data.shape[1:]),
                              dtype=torch.float64, device='cuda')
            new[:self.n] = self.data[:self.n]
            self.data = new
        self.data[self.n:self.n + k] = rows
        self.n += k

    def reserve(self, k):
        """Room for ``k`` rows that land later (``Sampler.land_points``);
        returns the index of the first.  NaN until then."""
        start = self.n
        self.append(torch.full((k,) + tuple(self.data.shape[1:]),
                               float('nan'), dtype=torch.float64,
                               device='cuda'))
        return start

    def keep(self, mask):
        kept = self.data[:self.n][mask]
        self.data = kept.clone()
        self.n = kept.shape[0]

    def __getstate__(self):
        return dict(data=self.data[:self.n].cpu().numpy(), n=self.n)

    def __setstate__(self, state):
        self.data = torch.from_numpy(state['data']).cuda()
        self.n = state['n']


class _RowsInFlight:
    """Points of a sharded batch whose all-gather has not landed yet."""

    def __init__(self, handle, total):
        self.handle, self.total = handle, total


MAX_IN_FLIGHT = 4        # gathers of points a rank keeps pending


class Sampler:
    """Drop-in for ``nautilus.Sampler`` (constructor signature and public
    attributes of the reference, sampler.py:121-129 and 307-327)."""

    def __init__(self, prior, likelihood, n_dim=None, n_live=2000,
                 n_update=None, enlarge_per_dim=1.1, n_points_min=None,
                 split_threshold=100, periodic=None, n_networks=4,
                 neural_network_kwargs={}, prior_args=[], prior_kwargs={},
                 likelihood_args=[], likelihood_kwargs={}, n_batch=None,
                 n_like_new_bound=None, vectorized=False, pass_dict=None,
                 pool=None, seed=None, blobs_dtype=None, filepath=None,
                 resume=True, comm=None):
        if filepath is not None:
            from . import io
            io.h5py()          # ImportError now rather than at the first write

        self._device_likelihood = bool(getattr(likelihood, 'device', False))
        from .likelihoods import unit_prior
        self._prior_is_identity = (prior is unit_prior or
                                   getattr(prior, 'identity', False) is True)
        if self._device_likelihood and not getattr(prior, 'device', False):
            raise ValueError(
                'a device likelihood needs a prior transform that works on '
                'cuda tensors (mark it with `.device = True`, e.g. '
                'nautilus_amd.unit_prior)')
        if callable(prior):
            self.prior = partial(prior, *prior_args, **prior_kwargs)
            if n_dim is None:
                raise ValueError("When passing a function as the 'prior' "
                                 "argument, 'n_dim' cannot be None.")
            self.n_dim = n_dim
            pass_dict = False if pass_dict is None else pass_dict
        else:
            self.prior = prior
            self.n_dim = prior.dimensionality()
            pass_dict = True if pass_dict is None else pass_dict
        if likelihood_args or likelihood_kwargs:
            self.likelihood = partial(likelihood, *likelihood_args,
                                      **likelihood_kwargs)
        else:
            self.likelihood = likelihood
        if self.n_dim <= 1:
            raise ValueError('Cannot run Nautilus with less than 2 '
                             'parameters.')

        self.n_live = n_live
        self.n_update = n_live if n_update is None else n_update
        self.n_like_new_bound = (10 * n_live if n_like_new_bound is None
                                 else n_like_new_bound)
        self.enlarge_per_dim = enlarge_per_dim
        self.n_points_min = (self.n_dim + 50 if n_points_min is None
                             else n_points_min)
        self.split_threshold = split_threshold
        self.periodic = periodic
        self.n_networks = n_networks
        self.neural_network_kwargs = neural_network_kwargs
        if n_networks > 0:
            # ValueError here, not at the first add_bound, for MLPRegressor
            # options the device trainer does not hold (emulator.check_hidden)
            from .emulator import check_network_kwargs
            check_network_kwargs(neural_network_kwargs)
        self.vectorized = vectorized
        self.pass_dict = pass_dict

        # pool normalisation, sampler.py:283-298
        try:
            pools = list(pool)
        except TypeError:
            pools = [pool]
        for i in range(len(pools)):
            if pools[i] in [None, 1]:
                pools[i] = None
            elif i == 0 and isinstance(pools[i], int):
                pools[i] = NautilusPool(pools[i], likelihood=self.likelihood)
                self.likelihood = likelihood_worker
            else:
                pools[i] = NautilusPool(pools[i])
        self.pool_l = pools[0]
        self.pool_s = pools[-1]

        if n_batch is None:
            s = 1 if self.pool_l is None else self.pool_l.size
            n_batch = (100 // s + (100 % s != 0)) * s
        self.n_batch = n_batch
        self.rng = np.random.default_rng(seed)

        self.n_like = 0
        self.n_dead_bounds = 0      # bounds dropped for a dead emulator
        self.explored = False
        self.bounds = []
        self._pts = []           # per shell: _Grow (n, n_dim) on the device
        self._in_flight = []     # (shell, first row, gather handle)
        self._ll_dev = []        # per shell: _Grow (n,) on the device
        self.log_l = []          # per shell: numpy mirror of log_l
        self.blobs = None        # per shell: numpy (structured) arrays
        self.blobs_dtype = blobs_dtype
        self.blobs_t = None
        self._discard_exploration = False
        self.shell_n = np.zeros(0, dtype=int)
        self.shell_n_sample = np.zeros(0, dtype=int)
        self.shell_n_eff = np.zeros(0, dtype=float)
        self.shell_log_l_min = np.zeros(0, dtype=float)
        self.shell_log_l = np.zeros(0, dtype=float)
        self.shell_log_v = np.zeros(0, dtype=float)
        self.shell_n_sample_exp = np.zeros(0, dtype=int)
        self.shell_end_exp = np.zeros(0, dtype=int)
        self._shell_max = np.zeros(0)   # per shell: largest stored log L
        self._live = None               # device.LivePool (exploration)
        self._pts_t = torch.empty((0, self.n_dim), dtype=torch.float64,
                                  device='cuda')
        self.shell_t = np.zeros(0, dtype=int)
        self.log_l_t = np.zeros(0)
        self.comm = comm         # parallel.ShardedComm or None
        self._later = {}         # cache: shell index -> DeviceBoundList
        self.timing = dict(add_bound=0.0, sample_shell=0.0, likelihood=0.0,
                           bookkeeping=0.0)
        self.filepath = filepath
        if resume and filepath is not None and os.path.exists(filepath):
            from . import io
            io.read_sampler(self, filepath)     # sampler.py:330-371

    # ------------------------------------------------------------------
    # pickling (the reference's sampler is picklable; device handles and
    # caches are rebuilt lazily, tensors travel as numpy arrays)
    # ------------------------------------------------------------------
    def __getstate__(self):
        self.land_points()
        state = dict(self.__dict__)
        state['_later'] = {}
        state['_live'] = None
        state['comm'] = None
        state['_in_flight'] = []
        state.pop('_pinned', None)
        state.pop('_predicted', None)
        state['_pts_t'] = self._pts_t.cpu().numpy()
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._pts_t = torch.from_numpy(state['_pts_t']).cuda()

    # ------------------------------------------------------------------
    # public views
    # ------------------------------------------------------------------
    @property
    def points(self):
        """Per-shell points as numpy arrays (reference attribute)."""
        self.land_points()
        return [p.view().cpu().numpy() for p in self._pts]

    @property
    def n_proposals(self):
        """Raw proposal evaluations so far: points drawn from the outer
        multi-ellipsoids of all bounds (``outer_bound.n_sample``,
        union.py:322)."""
        return int(sum(b.outer_bound.n_sample for b in self.bounds
                       if hasattr(b, 'outer_bound')))

    @property
    def points_t(self):
        return self._pts_t.cpu().numpy()

    @property
    def n_eff(self):
        """sampler.py:650-665."""
        if np.all(self.shell_n_eff == 0):
            return 0
        use = self.shell_n_eff > 0
        s = self.shell_log_l + self.shell_log_v
        w = np.exp(s - np.nanmax(s))[use]
        return np.sum(w)**2 / np.sum(w**2 / self.shell_n_eff[use])

    @property
    def log_z(self):
        """sampler.py:682-694."""
        if np.sum(self.shell_n) == 0:
            return None
        use = ~np.isnan(self.shell_log_l)
        return logsumexp(self.shell_log_l[use] + self.shell_log_v[use])

    @property
    def eta(self):
        """sampler.py:710-730."""
        use = ~np.isnan(self.shell_log_l)
        lz = (self.shell_log_l + self.shell_log_v)[use]
        eff = (self.shell_n_eff / self.shell_n)[use]
        return np.exp(2 * logsumexp(lz) - 2 * logsumexp(lz - 0.5 * np.log(eff)))

    def effective_sample_size(self):
        """Deprecated alias of ``n_eff`` (sampler.py:667-679)."""
        warnings.warn("The function 'effective_sample_size' is deprecated. "
                      "Please use the 'n_eff' property, instead.",
                      DeprecationWarning, stacklevel=2)
        return self.n_eff

    def evidence(self):
        """Deprecated alias of ``log_z`` (sampler.py:696-707)."""
        warnings.warn("The function 'evidence' is deprecated. Please use the "
                      "'log_z' property, instead.", DeprecationWarning,
                      stacklevel=2)
        return self.log_z

    def asymptotic_sampling_efficiency(self):
        """Deprecated alias of ``eta`` (sampler.py:732-744)."""
        warnings.warn("The function 'asymptotic_sampling_efficiency' is "
                      "deprecated. Please use the 'eta' property, instead.",
                      DeprecationWarning, stacklevel=2)
        return self.eta

    # -- the live set (exploration phase), on the device -------------------
    def _live_pool(self):
        """All log L at or above the n_live-th largest, maintained on the
        device (device.LivePool).  Built from the points stored in the shells
        -- transfer candidates waiting between shells do not count, as in the
        reference, which concatenates ``self.log_l`` -- whenever a bound was
        added (points leave their shells) or the sampler was unpickled /
        resumed; in between every batch appends to it."""
        pool = self.__dict__.get('_live')
        if pool is None or pool.k != self.n_live:
            pool = device.LivePool(self.n_live,
                                   [ll.view() for ll in self._ll_dev])
            self._live = pool
            self._shell_max = np.array([
                float(device.shell_stats(ll.view())[2]) if ll.n > 0
                else -np.inf for ll in self._ll_dev])
        return pool

    def _live_threshold(self):
        """(n_live-th largest stored log L, #above, #equal) from the pool."""
        pool = self._live_pool()
        try:
            return pool.select()
        except OverflowError:               # many batches without a selection
            self._live = None
            return self._live_pool().select()

    def _live_sums(self):
        """(log sum of the weights of the n_live points of largest log L,
        log sum of their volumes): sampler.py:1162-1169 and 1186-1190 with the
        selection and the per-shell sums on the device.  Ties at the
        threshold (likelihood plateaus) share the remaining places equally --
        the reference takes an arbitrary subset of them."""
        pool = self._live_pool()
        rows = None
        if pool.dirty:
            # one wait for the selection AND the per-shell sums: the shells
            # that reached the PREVIOUS threshold are a superset of those that
            # reach the new one (it only rises while the pool lives)
            thr_prev = float(pool._host[0])
            shells = [s for s in range(len(self._ll_dev))
                      if self._ll_dev[s].n > 0 and
                      self._shell_max[s] >= thr_prev]
            try:
                (thr, n_gt, n_eq), rows = pool.select_with_stats(
                    [self._ll_dev[s].view() for s in shells])
                keep = [i for i, s in enumerate(shells)
                        if self._shell_max[s] >= thr]
                shells = [shells[i] for i in keep]
                rows = rows[keep]
            except OverflowError:           # many batches without a selection
                self._live = None
                rows = None
        if rows is None:
            thr, n_gt, n_eq = self._live_threshold()
            pool = self._live
            shells = [s for s in range(len(self._ll_dev))
                      if self._ll_dev[s].n > 0 and self._shell_max[s] >= thr]
            rows = pool.shell_stats([self._ll_dev[s].view() for s in shells])
        frac = min(1.0, max(0.0, (self.n_live - n_gt) / n_eq)) if n_eq > 0 \
            else 0.0
        log_w, log_v = [-np.inf], [-np.inf]
        for s, (c_gt, lse_gt, c_eq) in zip(shells, rows):
            a = self.shell_log_v[s] - np.log(self.shell_n[s])
            if c_gt > 0:
                log_w.append(a + lse_gt)
            if c_eq > 0 and frac > 0:
                log_w.append(a + thr + np.log(frac * c_eq))
            if c_gt + frac * c_eq > 0:
                log_v.append(a + np.log(c_gt + frac * c_eq))
        return logsumexp(log_w), logsumexp(log_v)

    @property
    def f_live(self):
        """sampler.py:1147-1169."""
        if self.explored:
            return None
        if np.sum(self.shell_n) == 0:
            return 1.0
        # the weights of ALL points sum to the evidence (sampler.py:691-694)
        return float(np.exp(self._live_sums()[0] - self.log_z))

    @property
    def log_v_live(self):
        """sampler.py:1171-1190."""
        if len(self.bounds) == 0:
            return 1.0
        if np.sum(self.shell_n) == 0:
            return -np.inf
        return float(self._live_sums()[1])

    @property
    def discard_exploration(self):
        return self._discard_exploration

    @discard_exploration.setter
    def discard_exploration(self, flag):
        if not isinstance(flag, bool):
            raise ValueError("'discard_exploration' must be a bool.")
        self._discard_exploration = flag
        for s in range(len(self.log_l)):
            self.update_shell_info(s)

    # ------------------------------------------------------------------
    # main loop
    # ------------------------------------------------------------------
    def run(self, f_live=0.01, n_shell=1, n_eff=10000, n_like_max=np.inf,
            discard_exploration=False, timeout=np.inf, verbose=False):
        """sampler.py:373-505."""
        t0 = time()
        self._n_shell_min = n_shell
        if verbose:
            print('Starting the nautilus_amd sampler (MI355X hot path)...')
            self.print_status(header=True)
        if len(self.bounds) == 0:
            self.add_bound()
            self.n_update_iter = -self.n_live
            self.n_like_iter = 0

        def finished():
            return bool(self.explored and np.all(self.shell_n >= n_shell) and
                        self.n_eff >= n_eff)

        def out_of_time():
            late = time() - t0 >= timeout
            # wall clocks differ between the ranks of a sharded run; the
            # decision to enter another (collective) batch must not
            if self.comm is not None and timeout < np.inf:
                late = self.comm.any_flag(late)
            return late

        done = finished()
        while self.n_like < n_like_max and not done and not out_of_time():
            if not self.explored:
                if ((self.n_update_iter >= self.n_update or
                     self.n_like_iter >= self.n_like_new_bound) and
                        np.sum(self.shell_n) > self.n_live):
                    self.add_bound(verbose=verbose)
                    self.n_update_iter = 0
                    self.n_like_iter = 0
                    if self._writes_checkpoints():
                        self.write(self.filepath, overwrite=True)
                self.n_update_iter += self.add_samples(-1, verbose=verbose)
                self.n_like_iter += self.n_batch
                if self._writes_checkpoints():
                    # the complete file after the first batch (:449-453)
                    if self.n_like == self.n_batch:
                        self.write(self.filepath, overwrite=True)
                    self.write_shell_update(self.filepath, -1)
                if self.f_live <= f_live:
                    self._finish_exploration(discard_exploration)
                    if self._writes_checkpoints():
                        self.write(self.filepath, overwrite=True)
            elif np.any(self.shell_n < n_shell):
                shell = int(np.flatnonzero(self.shell_n < n_shell)[0])
                self.add_samples(shell, verbose=verbose)
                if self._writes_checkpoints():
                    self.write_shell_update(self.filepath, shell)
            elif self.n_eff < n_eff:
                shell = self._next_shell()
                self.add_samples(shell, verbose=verbose)
                if self._writes_checkpoints():
                    self.write_shell_update(self.filepath, shell)
            done = finished()
        self.land_points()
        if verbose:
            self.print_status('Finished' if done else 'Stopped')
        return done

    def _writes_checkpoints(self):
        """Every rank of a sharded run holds the same state; rank 0 writes."""
        return self.filepath is not None and (self.comm is None or
                                              self.comm.rank == 0)

    def _next_shell(self):
        """Shell with the largest expected gain (sampler.py:489-491)."""
        return int(np.argmax(self.shell_log_l + self.shell_log_v -
                             0.5 * np.log(self.shell_n) -
                             0.5 * np.log(self.shell_n_eff)))

    def _finish_exploration(self, discard_exploration):
        """sampler.py:457-480."""
        for s in np.flatnonzero(self.shell_n == 0)[::-1]:
            self.bounds.pop(s)
            self._pts.pop(s)
            self._ll_dev.pop(s)
            self.log_l.pop(s)
            if self.blobs is not None:
                self.blobs.pop(s)
            for key in ('shell_n', 'shell_n_sample', 'shell_n_eff',
                        'shell_log_l_min', 'shell_log_l', 'shell_log_v',
                        '_shell_max'):
                setattr(self, key, np.delete(getattr(self, key), s))
        self._later = {}
        self._live = None
        self.shell_n_sample_exp = np.copy(self.shell_n_sample)
        self.shell_end_exp = np.array([len(ll) for ll in self.log_l])
        self.explored = True
        self.discard_exploration = discard_exploration

    # ------------------------------------------------------------------
    # shells
    # ------------------------------------------------------------------
    def _later_bounds(self, index):
        index = index % len(self.bounds)
        key = (index, len(self.bounds))
        if key not in self._later:
            devs = [b.device_bound() for b in self.bounds[index + 1:]]
            self._later[key] = device.DeviceBoundList(devs) if devs else None
        return self._later[key]

    def shell_association(self, points, n_max=None):
        """Highest-index bound (below ``n_max``) containing each point
        (sampler.py:1192-1221)."""
        if n_max is None:
            n_max = len(self.bounds)
        x = device.as_device_points(points, self.n_dim)
        # (the list of a given length is built once: the pairing of
        # sample_shell asks for the same one several times per batch)
        key = ('association', n_max, len(self.bounds))
        lst = self._later.get(key)
        if lst is None:
            lst = self._later[key] = device.DeviceBoundList(
                [b.device_bound() for b in self.bounds[:n_max][::-1]])
        first = lst.first_containing(x).cpu().numpy().astype(int)
        shell = np.where(first >= 0, n_max - 1 - first, -1)
        return shell

    def sample_shell(self, index, shell_t=None, n_target=None):
        """Fill one batch of the shell ``index`` (sampler.py:751-830).

        Returns (points on the device, n_bound[, idx_t]).  ``n_target``
        (default ``n_batch``) is this rank's share of the batch.  In an
        un-sharded run the points are a view of a scratch buffer: valid until
        the next call (``add_samples`` copies them into the shell's storage
        right away)."""
        n_target = self.n_batch if n_target is None else n_target
        if shell_t is not None and index not in [-1, len(self.bounds) - 1]:
            raise ValueError("'shell_t' must be empty list if not sampling "
                             "from the last bound/shell.")
        bound = self.bounds[index]
        later = self._later_bounds(index)
        s_idx = index % len(self.bounds)
        n_bound = 0
        have = 0
        idx_t = np.zeros(0, dtype=int)
        transfer = shell_t is not None and len(shell_t) > 0
        # The batch is assembled in a buffer of its own: the rows a bound
        # hands out are views of its queue, and the flags / compacted rows of
        # a round live in the process-wide scratch buffers (no allocation of
        # a new size per round).  Un-sharded runs reuse the batch buffer too:
        # add_samples copies the batch into the shell's storage before the
        # next one is drawn; a sharded run's batch may still be travelling.
        out = device._buffer('shell_batch', (n_target, self.n_dim),
                             torch.float64, self.comm is None)

        while have < n_target:
            need = n_target - have
            if transfer or later is None:
                n_req = need             # every drawn point is in the shell
            else:
                n_req = self._shell_request(s_idx, need)
            x = bound.sample_device(n_req)
            if later is not None:
                # sampler.py:796-799 on the device: flag the points inside a
                # later bound, compact the others in order; the source row of
                # the need-th survivor tells how many draws were examined
                inside = later.inside_flags(x, reuse=True)
                rows, counts, src = device.compact_rows(
                    x, inside, mask=device.GS_INSIDE, want_index=True,
                    flip=device.GS_INSIDE, reuse=True)
                probe = torch.cat([counts, src[max(0, min(
                    need, n_req) - 1):max(1, min(need, n_req))]]).cpu()
                total = int(probe[1])
                if total >= need:
                    # stop at the need-th success; the rest goes back
                    used = int(probe[2]) + 1
                    if hasattr(bound, '_queue'):
                        bound._queue().unpop(n_req - used)
                    # (points of a bound without FIFO are i.i.d. draws; the
                    # unexamined tail is independent of the stopping rule
                    # and is simply dropped)
                    x = rows[:need]
                else:
                    used = n_req
                    x = rows[:total]
                n_bound += used
            else:
                n_bound += n_req

            if transfer and x.shape[0] > 0:
                x, idx_t = self._pair_with_candidates(x, shell_t, idx_t)
            if x.shape[0] > 0:
                out[have:have + x.shape[0]].copy_(x)
                have += x.shape[0]

        pts = out[:have]
        if shell_t is None:
            return pts, n_bound
        return pts, n_bound, idx_t

    def _shell_request(self, s_idx, need):
        """Points to ask the bound of shell ``s_idx`` for when ``need`` of
        them must lie outside all later bounds: the in-shell fraction f seen
        so far (exploration points count even when they are discarded from the
        estimate) with four standard deviations on top -- of the estimate of f
        from the shell's n_s draws and of the binomial count of this request.
        (A flat 15 % + 256 until round 6: 13 % of every exclusion launch of
        the headline step examined rows nobody needed.)  A request that falls
        short costs one more round of the caller's loop, nothing else."""
        if s_idx == len(self.bounds) - 1:
            return need
        n_s = self.shell_n_sample[s_idx]
        frac = (len(self.log_l[s_idx]) + 1.0) / (n_s + 2.0) \
            if n_s > 0 else 0.5
        rel = 4.0 * np.sqrt((1.0 - frac) / frac *
                            (1.0 / max(n_s, 1.0) + frac / max(need, 1.0)))
        return int(min(4 * device_block(),
                       need / frac * (1.0 + min(rel, 1.0)) + 64))

    def _predict_next_shell(self, shell, n_new):
        """The shell the loop of ``run`` will most likely sample after the
        batch of ``n_new`` points that was just drawn for ``shell`` -- BEFORE
        that batch's likelihoods are known: rule of sampler.py:482-491 with
        the shell's count grown by the batch and its mean likelihood and
        N_eff / N taken as unchanged.  Only a guess (``prefetch`` acts on it:
        a wrong guess costs nothing but the order of the work)."""
        s = shell % len(self.bounds)
        if not self.explored:
            return len(self.bounds) - 1
        n = np.array(self.shell_n, dtype=float)
        n_eff = np.array(self.shell_n_eff, dtype=float)
        known = n[s] > 0
        n[s] += n_new
        n_min = getattr(self, '_n_shell_min', 1)
        if np.any(n < n_min):
            return int(np.flatnonzero(n < n_min)[0])
        if not known or not np.all(n_eff > 0):
            return None
        n_eff[s] *= n[s] / (n[s] - n_new)
        with np.errstate(all='ignore'):
            gain = (self.shell_log_l + self.shell_log_v - 0.5 * np.log(n) -
                    0.5 * np.log(n_eff))
        if not np.any(np.isfinite(gain)):
            return None
        return int(np.nanargmax(gain))

    def _prefetch_plan(self, shell, n_new):
        """(bound, points to ask it for) of the refill ``_prefetch_next`` will
        queue, or None -- the host arithmetic of the guess, done while the GPU
        still works on the batch."""
        if not PREFETCH or self.comm is not None:
            return None
        nxt = self._predict_next_shell(shell, n_new)
        if nxt is None or not hasattr(self.bounds[nxt], 'prefetch'):
            return None
        return nxt, self._shell_request(nxt, self.n_batch)

    def _prefetch_next(self, shell, n_new, plan=False):
        """Queue the refill of the bound the next batch will most likely ask
        (``_RejectionSampler.prefetch``) behind the launches of the current
        batch, so that the GPU draws and filters proposals while the host
        waits for this batch's numbers and does its bookkeeping.  ``plan``:
        what ``_prefetch_plan`` returned earlier for the same batch."""
        if plan is False:
            plan = self._prefetch_plan(shell, n_new)
        if plan is None:
            return
        nxt, n_ask = plan
        stats = self.__dict__.setdefault(
            'prefetch_stats', dict(issued=0, enough=0, right=0, wrong=0))
        if self.bounds[nxt].prefetch(n_ask):
            stats['issued'] += 1
        else:
            stats['enough'] += 1
        self._predicted = nxt

    def _pair_with_candidates(self, x, shell_t, idx_t):
        """Pair fresh points of the newest shell with stored candidates of
        the same earlier shell (sampler.py:803-819): the candidates move into
        the new shell, the paired fresh points are dropped.  Host RNG; in a
        sharded run every rank does this on the same gathered points."""
        waiting = shell_t[shell_t >= 0]
        if waiting.size == 0:
            return x, idx_t              # every candidate has moved already
        shell_p = self.shell_association(x, n_max=len(self.bounds) - 1)
        swap = np.zeros(x.shape[0], dtype=bool)
        # the reference walks ALL shells in ascending order and draws from its
        # generator for those that have both candidates and fresh points; the
        # same shells in the same order, found without n_bounds passes over
        # the two arrays (a funnel run has hundreds of bounds and tens of
        # thousands of these rounds)
        for s in np.intersect1d(waiting, shell_p):
            cand = np.flatnonzero(shell_t == s)
            fresh = np.flatnonzero(shell_p == s)
            m = min(len(cand), len(fresh))
            if m > 0:
                idx_t = np.append(idx_t, self.rng.choice(
                    cand, size=m, replace=False))
                shell_t[idx_t] = -1
                swap[self.rng.choice(fresh, size=m, replace=False)] = True
        if not swap.any():
            return x, idx_t
        return x[torch.from_numpy(~swap).cuda()], idx_t

    def _fetch_async(self, t):
        """Start the transfer of the 1-D float64 cuda tensor ``t`` into
        pinned host memory; returns (host view, event).  What is launched
        between this call and ``event.synchronize()`` runs BEHIND the transfer
        in the stream -- ``t.cpu()`` issued after such launches would wait
        for them."""
        n = int(t.shape[0])
        buf = self.__dict__.get('_pinned')
        if buf is None or buf.shape[0] < n:
            buf = self._pinned = torch.empty(
                max(2 * n, 4096), dtype=torch.float64).pin_memory()
        view = buf[:n]
        view.copy_(t, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        return view, event

    def evaluate_likelihood(self, points, fetch=True, after_fetch=None):
        """sampler.py:832-908.  ``points`` is a cuda tensor (n, n_dim);
        returns (log_l numpy, log_l cuda tensor, blobs or None).  ``fetch =
        False``: a device likelihood's values stay on the device (None in
        place of the numpy array) -- ``add_samples`` brings them to the host
        together with the shell statistics, one wait instead of two."""
        if self._device_likelihood:
            args = points
            if callable(self.prior):
                if not self._prior_is_identity:
                    args = self.prior(points)
            elif self.pass_dict:
                args = self.prior.unit_to_dictionary(points)
            else:
                args = self.prior.unit_to_physical(points)
            ll = self.likelihood(args)
            blobs = None
            if isinstance(ll, tuple):
                # a device likelihood's blobs are cuda tensors (or arrays) of
                # one row per point; they join the host bookkeeping like a
                # vectorized host likelihood's (sampler.py:875-904)
                blobs = self._pack_blobs([
                    b.cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
                    for b in ll[1:]])
                ll = ll[0]
            # (whatever the likelihood returns -- float32, a column, a strided
            # view -- joins the shell's storage as float64 values in a row)
            ll = ll.reshape(-1).to(torch.float64).contiguous()
            self.n_like += ll.shape[0]
            return (ll.cpu().numpy() if fetch else None), ll, blobs

        if callable(self.prior):
            transform = self.prior
        elif self.pass_dict:
            transform = self.prior.unit_to_dictionary
        else:
            transform = self.prior.unit_to_physical
        if after_fetch is None:
            host = points.cpu().numpy()
        else:
            # the points start their way to the host, THEN the next batch's
            # refill is queued (``after_fetch``): the GPU draws and filters
            # proposals while the CPU evaluates this batch
            view, event = self._fetch_async(points.reshape(-1))
            after_fetch()
            event.synchronize()
            host = view.numpy().reshape(points.shape).copy()
        if not self.vectorized:
            args = list(map(transform, np.copy(host)))
        else:
            args = list(map(transform, np.array_split(
                host, 1 if self.pool_l is None else self.pool_l.size)))
        if self.pool_l is not None:
            result = list(self.pool_l.map(self.likelihood, args))
        else:
            result = list(map(self.likelihood, args))
        # blobs: everything the likelihood returns after the first element
        # (sampler.py:875-905)
        blobs = None
        if isinstance(result[0], tuple):
            blobs = [r[1:] for r in result]
            result = [r[0] for r in result]
        log_l = (np.concatenate(result) if self.vectorized
                 else np.array(result)).astype(float)
        if blobs is not None:
            n_col = len(blobs[0])
            if self.vectorized:
                blobs = [np.concatenate([row[col] for row in blobs])
                         for col in range(n_col)]
            else:
                blobs = [np.array([row[col] for row in blobs])
                         for col in range(n_col)]
            blobs = self._pack_blobs(blobs)
        self.n_like += len(log_l)
        return log_l, torch.from_numpy(log_l).cuda(), blobs

    def _pack_blobs(self, cols):
        """The blob columns of one batch (one array per returned value, one
        row per point) as the reference stores them: a plain array for one
        column, a structured one for several (sampler.py:892-904)."""
        if self.blobs_dtype is None:
            if len(cols) > 1:
                self.blobs_dtype = [('blob_{}'.format(i), b.dtype)
                                    for i, b in enumerate(cols)]
            else:
                self.blobs_dtype = cols[0].dtype
        return np.squeeze(np.array(list(zip(*cols)), dtype=self.blobs_dtype))

    def _gather_blobs(self, blobs, n_local, interleaved_to=None):
        """The blobs of a sharded batch follow their points: ``n_local`` rows
        per rank travel as bytes through the same all-gather (rank-major, the
        order ``gather_rows`` gives the points).  ``interleaved_to = n``: the
        ranks evaluated rows r, r + world, ... of one batch of n points
        (``_sharded_likelihood``), padded to n_local rows each; the rows come
        back in the order of the batch."""
        comm = self.comm
        flat = np.ascontiguousarray(blobs).reshape(-1)
        have = n_local if interleaved_to is None else len(
            range(comm.rank, interleaved_to, comm.world))
        dtype = flat.dtype
        # (an empty share knows its dtype but not its row width: the widths
        # of all ranks are reduced along)
        width = flat.view(np.uint8).size // have if have > 0 else 0
        width = int(comm.max_float(float(width), 'cuda'))
        raw = np.zeros((n_local, width), dtype=np.uint8)
        if have > 0:
            raw[:have] = flat.view(np.uint8).reshape(have, width)
        table = comm.gather_rows(
            torch.from_numpy(raw).to(comm._dev('cuda'))).cpu().numpy()
        if interleaved_to is not None:
            table = np.ascontiguousarray(
                table.reshape(comm.world, n_local, width).transpose(1, 0, 2)
            ).reshape(-1, width)[:interleaved_to]
        n = table.shape[0]
        out = np.ascontiguousarray(table).reshape(-1).view(dtype)
        # The shape of one row is agreed on ONCE per run: a rank whose share
        # has a single row (np.squeeze in _pack_blobs drops the row axis) or
        # none cannot tell (2, 3) from (6,) -- the widest description any rank
        # offers wins (ranks with several rows know it) and is kept.
        if self.__dict__.get('_blob_tail') is None:
            mine = tuple(np.shape(blobs)[1:]) if np.ndim(blobs) > 1 \
                and have > 1 else ()
            code = [float(len(mine))] + [float(v) for v in mine] + \
                [0.0] * (4 - len(mine))
            code = [comm.max_float(v, 'cuda') for v in code[:5]]
            tail = tuple(int(v) for v in code[1:1 + int(code[0])])
            if not tail and n > 0 and out.size != n:
                tail = (out.size // n,)
            if n > 0 and (code[0] > 0 or out.size == n):
                self._blob_tail = tail
        else:
            tail = self._blob_tail
        return out.reshape((n,) + tuple(tail))

    def _shell_slice(self, index):
        """The log L of shell ``index`` that count (device view), the number
        of draws they come from and the offset of the view in the shell."""
        n_sample = self.shell_n_sample[index]
        if self._discard_exploration and self.explored:
            start = self.shell_end_exp[index]
            n_sample = n_sample - self.shell_n_sample_exp[index]
        else:
            start = 0
        return self._ll_dev[index].view()[start:], n_sample, start

    def update_shell_info(self, index, stats=None):
        """sampler.py:910-943 with the reductions on the device (``stats``:
        the result of ``device.shell_stats`` for the shell's current slice,
        already on the host)."""
        ll, n_sample, start = self._shell_slice(index)
        n = ll.shape[0]
        if len(self._shell_max) != len(self._ll_dev):     # resumed / unpickled
            self._shell_max = np.full(len(self._ll_dev), np.inf)
        self.shell_n[index] = n
        if n > 0:
            st = (device.shell_stats(ll).cpu().numpy() if stats is None
                  else stats)
            if start == 0:
                self._shell_max[index] = st[2]
            self.shell_log_v[index] = (self.bounds[index].log_v +
                                       np.log(n / n_sample))
            self.shell_log_l[index] = st[0] - np.log(n)
            if st[2] > -np.inf:
                self.shell_n_eff[index] = np.exp(2 * st[0] - st[1])
            else:
                self.shell_n_eff[index] = n
        else:
            if start == 0:
                self._shell_max[index] = -np.inf
            self.shell_log_v[index] = -np.inf
            self.shell_log_l[index] = np.nan
            self.shell_n_eff[index] = 0

    def add_samples(self, shell, verbose=False):
        """sampler.py:1093-1144."""
        if verbose:
            self.print_status('Sampling', end='\r')
        t0 = time()
        log_l = None
        blobs = None
        if self.__dict__.get('_predicted') is not None:
            hit = self._predicted == shell % len(self.bounds)
            self.prefetch_stats['right' if hit else 'wrong'] += 1
            self._predicted = None
        if shell == -1 and len(self.shell_t) > 0:
            if self.comm is not None:
                pts, n_bound, idx_t = self._sharded_transfer_batch()
            else:
                pts, n_bound, idx_t = self.sample_shell(-1, self.shell_t)
            assert pts.shape[0] + len(idx_t) == n_bound
            if len(idx_t) > 0:
                sel = torch.from_numpy(idx_t).cuda()
                self._pts[-1].append(self._pts_t[sel])
                moved = torch.from_numpy(self.log_l_t[idx_t]).cuda()
                self._ll_dev[-1].append(moved)
                if self.__dict__.get('_live') is not None:
                    self._live.add(moved)
                self.log_l[-1] = np.concatenate(
                    (self.log_l[-1], self.log_l_t[idx_t]))
                if self.blobs is not None:
                    self.blobs[-1] = np.concatenate(
                        (self.blobs[-1], self.blobs_t[idx_t]))
        elif self.comm is not None:
            pts, log_l, log_l_dev, n_bound, blobs = self._sharded_batch(shell)
        else:
            pts, n_bound = self.sample_shell(shell)
        t1 = time()
        self.shell_n_sample[shell] += n_bound
        if log_l is None:
            if self.comm is not None and not self._device_likelihood:
                log_l, log_l_dev, blobs = self._sharded_likelihood(pts)
            else:
                # (a device likelihood's values come to the host further
                # down, in one transfer with the shell statistics)
                ahead = None
                if not self._device_likelihood and self.comm is None and (
                        self.explored or len(self.shell_t) == 0):
                    # a host likelihood: the GPU refills the next batch's
                    # queue while the CPU evaluates this one
                    ahead = partial(self._prefetch_next, shell, pts.shape[0])
                log_l, log_l_dev, blobs = self.evaluate_likelihood(
                    pts, fetch=self.comm is not None or not DEFER_FETCH,
                    after_fetch=ahead)
        t2 = time()
        # (the guess of the next batch's shell, while the likelihood kernel
        # runs: the refill itself is queued further down)
        plan = self._prefetch_plan(shell, pts.shape[0]) \
            if self.explored and log_l is None and \
            not isinstance(pts, _RowsInFlight) else None
        if isinstance(pts, _RowsInFlight):
            # sharded sampling phase: the rows are on their way to this rank
            first = self._pts[shell].reserve(pts.total)
            self._in_flight.append((shell, first, pts.handle))
            while len(self._in_flight) > MAX_IN_FLIGHT:
                self._land(self._in_flight.pop(0))
        else:
            self._pts[shell].append(pts)
        self._ll_dev[shell].append(log_l_dev)
        if self.explored:
            self._live = None      # log_v_live rebuilds it from the shells
        elif self.__dict__.get('_live') is not None:
            self._live.add(log_l_dev)
        stats = None
        if log_l is None:
            view = self._shell_slice(shell)[0]
            if view.shape[0] > 0:
                both = torch.cat([device.shell_stats(view), log_l_dev])
                if self.explored and PREFETCH and self.comm is None:
                    # the numbers start their way to the host, the next
                    # batch's refill is queued behind them, then the wait
                    host, event = self._fetch_async(both)
                    self._prefetch_next(shell, log_l_dev.shape[0], plan)
                    event.synchronize()
                    both = host.numpy().copy()
                else:
                    both = both.cpu().numpy()
                stats, log_l = both[:4], both[4:]
            else:
                log_l = log_l_dev.cpu().numpy()
        self.log_l[shell] = _grow(self.log_l[shell], log_l)
        if blobs is not None:                      # sampler.py:1137-1141
            if self.blobs is None:
                self.blobs = [blobs]
            else:
                self.blobs[shell] = np.append(self.blobs[shell], blobs,
                                              axis=0)
        self.update_shell_info(shell, stats)
        t3 = time()
        self.timing['sample_shell'] += t1 - t0
        self.timing['likelihood'] += t2 - t1
        self.timing['bookkeeping'] += t3 - t2
        return int(np.sum(log_l >= self.shell_log_l_min[shell]))

    # -- multi-GPU (parallel.py): every batch of every phase is sharded ----
    @staticmethod
    def _counter_owners(bound):
        owners = [bound] + ([bound.outer_bound]
                            if hasattr(bound, 'outer_bound') else [])
        return [o for o in owners if hasattr(o, 'n_sample')]

    def _counters(self, bound):
        return [getattr(o, c) for o in self._counter_owners(bound)
                for c in ('n_sample', 'n_reject')]

    def _rank_keyed(self, bound):
        """From its first sharded draw on, rank r takes the proposals of
        ``bound`` from its own Philox stream."""
        if not getattr(bound, '_is_rank_keyed', False):
            from . import parallel
            bound._stream.seed = parallel.rank_key(bound._stream.seed,
                                                   self.comm.rank)
            if hasattr(bound, '_queue'):
                bound._queue().clear()
            bound._is_rank_keyed = True
        return bound

    def _sum_counters(self, bound, before, extra=()):
        """The MC-volume counters of ``bound`` advance by the sum of what all
        ranks drew since ``before`` (nautilus.py:232-237); ``extra`` ints are
        summed along.  Returns the summed extras."""
        after = self._counters(bound)
        delta = [a - b for a, b in zip(after, before)]
        n_extra = len(extra)
        totals = self.comm.sum_ints(list(extra) + delta, 'cuda')
        k = 0
        for o in self._counter_owners(bound):
            for c in ('n_sample', 'n_reject'):
                setattr(o, c, before[k] + totals[n_extra + k])
                k += 1
        return totals[:n_extra]

    def _land(self, entry):
        shell, first, handle = entry
        rows = handle.wait()
        self._pts[shell].data[first:first + rows.shape[0]] = rows

    def land_points(self):
        """Wait for the points of sharded batches that are still travelling
        (see ``_sharded_batch``); collective in effect -- every rank issued
        the same gathers.  ``run()`` ends with it."""
        pending = self.__dict__.get('_in_flight')
        while pending:
            self._land(pending.pop(0))

    def _sharded_batch(self, shell):
        """One batch spread over all ranks: local draw + likelihood, ONE
        all-gather of the accepted points with their log L, one all-reduce
        of the integer counters.  In the sampling phase of a run without
        checkpoints nothing reads the points before the run ends (no bound is
        built any more): the log L are gathered at once -- the shell
        statistics need them -- and the points, 8 D of the 8 (D + 1) bytes per
        row, by an asynchronous all-gather that lands behind the next batches
        (``land_points``)."""
        from . import parallel
        comm = self.comm
        bound = self._rank_keyed(self.bounds[shell])
        before = self._counters(bound)
        n_local = parallel.split_batch(self.n_batch, comm.world)
        pts, n_bound = self.sample_shell(shell, n_target=n_local)
        n_like0 = self.n_like
        _, ll_dev, blobs = self.evaluate_likelihood(pts)
        if blobs is not None:
            blobs = self._gather_blobs(blobs, pts.shape[0])
        self.n_like = n_like0
        if self.explored and self.filepath is None:
            # (collectives of one communicator run in the order of issue: the
            # small ones this batch waits for go first, the points last --
            # they have the next batch's draw and evaluation to get through)
            ll_dev = comm.gather_rows(ll_dev[:, None])[:, 0].contiguous()
            n_bound, = self._sum_counters(bound, before, [n_bound])
            handle = comm.gather_rows_async(pts)
            self.n_like += ll_dev.shape[0]
            return (_RowsInFlight(handle, ll_dev.shape[0]),
                    ll_dev.cpu().numpy(), ll_dev, n_bound, blobs)
        packed = torch.cat([pts, ll_dev[:, None]], dim=1)
        gathered = comm.gather_rows(packed)
        n_bound, = self._sum_counters(bound, before, [n_bound])
        pts = gathered[:, :-1].contiguous()
        ll_dev = gathered[:, -1].contiguous()
        self.n_like += pts.shape[0]
        return pts, ll_dev.cpu().numpy(), ll_dev, n_bound, blobs

    def _sharded_transfer_batch(self):
        """A batch of the newest shell while transfer candidates are waiting
        (sampler.py:790-823): the ranks draw their shares, the points are
        gathered, and the pairing with the candidates -- host RNG, identical
        on every rank -- runs on the gathered batch."""
        comm = self.comm
        bound = self._rank_keyed(self.bounds[-1])
        before = self._counters(bound)
        have, n_bound = 0, 0
        chunks = []
        idx_t = np.zeros(0, dtype=int)
        while have < self.n_batch:
            need = self.n_batch - have
            x = bound.sample_device(-(-need // comm.world))
            x = comm.gather_rows(x)[:need]   # newest bound: all in its shell
            n_bound += need
            if len(self.shell_t) > 0:
                x, idx_t = self._pair_with_candidates(x, self.shell_t, idx_t)
            if x.shape[0] > 0:
                chunks.append(x)
                have += x.shape[0]
        self._sum_counters(bound, before)
        pts = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
        return pts, n_bound, idx_t

    def _sharded_likelihood(self, pts):
        """Host likelihood of a batch every rank holds: rank r evaluates rows
        r, r + world, ... and one all-gather returns all values."""
        comm = self.comm
        n = pts.shape[0]
        per = -(-n // comm.world)
        mine = pts[comm.rank::comm.world]
        n_like0 = self.n_like
        ll, _, blobs = self.evaluate_likelihood(mine)
        if blobs is not None:
            blobs = self._gather_blobs(blobs, per, interleaved_to=n)
        self.n_like = n_like0 + n
        padded = torch.zeros(per, dtype=torch.float64, device='cuda')
        padded[:len(ll)] = torch.from_numpy(ll).cuda()
        table = comm.gather_rows(padded[:, None]).reshape(comm.world, per)
        ll_dev = table.t().reshape(-1)[:n].contiguous()
        return ll_dev.cpu().numpy(), ll_dev, blobs

    # ------------------------------------------------------------------
    # bounds
    # ------------------------------------------------------------------
    def add_bound(self, verbose=False):
        """sampler.py:982-1091."""
        t0 = time()
        if len(self.bounds) == 0:
            log_l_min = -np.inf
            self.bounds.append(UnitCube.compute(self.n_dim, rng=self.rng))
            ok = True
        else:
            if verbose:
                self.print_status('Bounding', end='\r')
            # sampler.py:1004-1020 without sorting anything: the n_live-th
            # largest log L and the counts above / at it come from the live
            # pool's radix selection on the device
            thr, n_gt, n_eq = self._live_threshold()
            n_total = int(np.sum(self.shell_n))
            log_l_min = thr
            # likelihood plateaus, sampler.py:1012-1020
            if n_eq > 1 and n_gt >= self.n_points_min:
                log_l_min = self._live.smallest_above()
                n_at_or_above = n_gt
            else:
                n_at_or_above = n_gt + n_eq
            if n_at_or_above == n_total:
                ok = False
            else:
                # (the reference hands the points over sorted by log L; no
                # step of the construction depends on their order)
                log_l = np.concatenate(self.log_l)
                pts = torch.cat([p.view() for p in self._pts])
                # host BLAS pinned to one thread as in the reference
                # (sampler.py:1022): the construction works on tiny matrices
                with geometry.single_threaded_blas():
                    bound = NautilusBound.compute(
                        pts, log_l, log_l_min, self.log_v_live,
                        enlarge_per_dim=self.enlarge_per_dim,
                        n_points_min=self.n_points_min,
                        split_threshold=self.split_threshold,
                        periodic=self.periodic,
                        n_networks=self.n_networks,
                        neural_network_kwargs=self.neural_network_kwargs,
                        pool=self.pool_s, rng=self.rng, comm=self.comm)
                    barren = bound.emulators_dead
                    if barren:
                        pass
                    elif self.comm is None:
                        try:
                            bound.sample(1000, return_points=False,
                                         guard=True)
                        except BarrenBound:
                            # measured: the pre-fill accepted next to nothing
                            barren = True
                    else:
                        # the pre-fill behind the first volume estimate
                        # (sampler.py:1032), a share per rank; every rank
                        # takes part in both collectives whatever it measured
                        before = self._counters(self._rank_keyed(bound))
                        try:
                            bound.sample(-(-1000 // self.comm.world),
                                         return_points=False, guard=True)
                        except BarrenBound:
                            barren = True
                        self._sum_counters(bound, before)
                        barren = bool(self.comm.any_flag(barren))
                for key, val in bound.timing.items():
                    self.timing[key] = self.timing.get(key, 0.0) + val
                if barren:
                    # Deviation: every network of an ensemble ended no
                    # better than a constant (or the pre-fill measured an
                    # acceptance below 1e-7 over 2.7 x 10^8 proposals): the
                    # bound accepts next to nothing, and the reference's
                    # pre-fill (nautilus.py:217-240) would take from hours
                    # to for ever.  (With an exactly constant emulator the
                    # reference's threshold, bounds/neural.py:121-126, falls
                    # 1e-9 below the prediction and the bound degenerates to
                    # its ellipsoid instead -- a behaviour change, not only a
                    # hang fix.)  Treated like a bound that fails to shrink
                    # (sampler.py:1034): the last bound is filled for another
                    # n_update points and the next attempt trains on the
                    # larger set.
                    self.n_dead_bounds += 1
                    ok = False
                else:
                    ok = bool(bound.log_v < self.bounds[-1].log_v)
                if ok:
                    self.bounds.append(bound)
        if not ok:
            self.shell_log_l_min[-1] = log_l_min
            self.timing['add_bound'] += time() - t0
            return False

        self.shell_n = np.append(self.shell_n, 0)
        self.shell_n_sample = np.append(self.shell_n_sample, 0)
        self.shell_n_eff = np.append(self.shell_n_eff, 0)
        self._shell_max = np.append(self._shell_max, -np.inf)
        self.shell_log_l = np.append(self.shell_log_l, np.nan)
        self.shell_log_v = np.append(self.shell_log_v, np.nan)
        self.shell_log_l_min = np.append(self.shell_log_l_min, log_l_min)
        self._pts.append(_Grow(self.n_dim))
        self._ll_dev.append(_Grow())
        self.log_l.append(np.zeros(0))
        if self.blobs is not None:                 # sampler.py:1050-1052
            self.blobs.append(np.zeros(self.blobs[-1][:0].shape,
                                       dtype=self.blobs_dtype))
        self._later = {}

        if len(self.bounds) > 1:
            # candidates for transfer into the new shell, sampler.py:1057-1089
            st, pt, lt, bt = [], [], [], []
            new = self.bounds[-1]
            for s in range(len(self.bounds) - 1):
                if self._pts[s].n == 0:
                    continue
                inside = new.contains_device(self._pts[s].view())
                k = int(inside.sum())
                if k == 0:
                    continue
                inside_h = inside.cpu().numpy()
                st.append(np.repeat(s, k))
                pt.append(self._pts[s].view()[inside])
                lt.append(self.log_l[s][inside_h])
                if self.blobs is not None:
                    bt.append(self.blobs[s][inside_h])
                    self.blobs[s] = self.blobs[s][~inside_h]
                self._pts[s].keep(~inside)
                self._ll_dev[s].keep(~inside)
                self.log_l[s] = self.log_l[s][~inside_h]
                self.shell_n[s] -= k
                self.update_shell_info(s)
            if st:
                self.shell_t = np.concatenate(st)
                self._pts_t = torch.cat(pt)
                self.log_l_t = np.concatenate(lt)
                if self.blobs is not None:
                    self.blobs_t = np.concatenate(bt)
            else:
                self.shell_t = np.zeros(0, dtype=int)
                self._pts_t = self._pts_t[:0]
                self.log_l_t = np.zeros(0)
                if self.blobs is not None:
                    self.blobs_t = self.blobs[0][:0]
        # points left their shells for the transfer set: the live pool is
        # rebuilt from the shells on its next use
        self._live = None
        self.timing['add_bound'] += time() - t0
        return True

    # ------------------------------------------------------------------
    # results
    # ------------------------------------------------------------------
    def posterior(self, return_as_dict=None, equal_weight=False,
                  equal_weight_boost=1.0, return_blobs=False):
        """sampler.py:541-647."""
        if return_blobs and self.blobs is None:
            raise ValueError('No blobs have been calculated.')
        self.land_points()
        if return_as_dict is None:
            return_as_dict = bool(callable(self.prior) and self.pass_dict)
        if self._discard_exploration and self.explored:
            start = self.shell_end_exp
        else:
            start = np.zeros(len(self.log_l), dtype=int)
        pts = torch.cat([p.view()[s:] for p, s in zip(self._pts, start)]
                        ).cpu().numpy()
        log_l = np.concatenate([ll[s:] for ll, s in zip(self.log_l, start)])
        # log w_i = shell_log_v - log shell_n + log L_i (sampler.py:602-608),
        # per shell on the device; their normalisation is the evidence, which
        # the per-shell device reductions already hold (sampler.py:691-694)
        offset = self.shell_log_v - np.log(np.maximum(self.shell_n, 1))
        log_w = torch.cat([ll.view()[s:] + float(o) for ll, s, o in
                           zip(self._ll_dev, start, offset)]).cpu().numpy()
        log_norm = self.log_z if np.sum(self.shell_n) > 0 else 0.0
        blobs = None
        if return_blobs:
            blobs = np.concatenate([b[s:] for b, s in zip(self.blobs, start)])
        if equal_weight:
            rep = np.exp(log_w - np.amax(log_w)) * equal_weight_boost
            rep = np.floor(rep).astype(int) + (
                self.rng.random(len(rep)) < rep - np.floor(rep)).astype(int)
            pts = np.repeat(pts, rep, axis=0)
            log_w = np.zeros(np.sum(rep))
            log_l = np.repeat(log_l, rep, axis=0)
            if return_blobs:
                blobs = np.repeat(blobs, rep, axis=0)
        if callable(self.prior):
            transform = self.prior
        elif return_as_dict:
            transform = self.prior.unit_to_dictionary
        else:
            transform = self.prior.unit_to_physical
        if not self.vectorized and callable(self.prior):
            pts = np.array(list(map(transform, pts)))
        else:
            pts = transform(pts)
        if not return_as_dict and callable(self.prior) and self.pass_dict:
            raise ValueError('Cannot return points as numpy array. The prior '
                             'function only returns dictionaries.')
        if equal_weight:
            log_norm = logsumexp(log_w)
        if return_blobs:
            return pts, log_w - log_norm, log_l, blobs
        return pts, log_w - log_norm, log_l

    def write(self, filepath, overwrite=False):
        """Write the sampler to an HDF5 file in the reference's layout
        (sampler.py:1253-1332)."""
        from . import io
        io.write_sampler(self, filepath, overwrite=overwrite)

    def write_shell_update(self, filepath, shell):
        """sampler.py:1334-1377."""
        from . import io
        io.write_shell_update(self, filepath, shell)

    def shell_bound_occupation(self, fractional=True):
        """sampler.py:1223-1251."""
        self.land_points()
        m = np.zeros((len(self.bounds), len(self.bounds)), dtype=int)
        for i, p in enumerate(self._pts):
            for k, b in enumerate(self.bounds):
                m[i, k] = int(b.contains_device(p.view()).sum()) \
                    if p.n > 0 else 0
        if fractional:
            m = m / np.diag(m)[:, np.newaxis]
        return m

    def print_status(self, status='', header=False, end='\n'):
        """One line of the status table (sampler.py:945-980)."""
        if header:
            cells = ['Status', 'Bounds', 'Ellipses', 'Networks', 'Calls',
                     'f_live', 'N_eff', 'log Z']
        else:
            last = self.bounds[-1] if len(self.bounds) > 1 else None
            vals = [status, len(self.bounds),
                    last.n_ell if last is not None else 0,
                    last.n_net if last is not None else 0,
                    self.n_like, self.f_live, self.n_eff, self.log_z]
            fmts = ['{}', '{:d}', '{:d}', '{:d}', '{:d}', '{:.4f}', '{:.0f}',
                    '{:+.2f}']
            cells = ['N/A' if v is None else f.format(v)
                     for v, f in zip(vals, fmts)]
        widths = [9, 6, 8, 8, 8, 6, 5, 7]
        print(' | '.join('{:<{}}'.format(c, w)
                         for c, w in zip(cells, widths)), end=end, flush=True)


# add_samples fetches a device likelihood's values together with the shell
# statistics (one wait per batch instead of two; profiles/tools/step_ab.py
# measures both)
DEFER_FETCH = True
# refills launched ahead of the batch that will ask for them (``prefetch``);
# NB_PREFETCH=0 restores the launch-when-asked order (profiles/r06 A/B)
PREFETCH = os.environ.get('NB_PREFETCH', '1') not in ('0', '')


def _grow(cur, new):
    """``np.append(cur, new)`` for a 1-D float array that is appended to
    batch after batch (sampler.py:1135-1136 re-allocates and copies the
    whole shell every time: 10 MB per 65 536-point batch once a shell holds
    10^6 values, 0.3 ms of the bench's step).  The result is a view of a
    buffer with spare room; the next call writes behind it in place where
    ``cur`` still is the leading view of such a buffer, and copies once into
    a buffer of twice the size where it is not."""
    new = np.asarray(new, dtype=float)
    n, k = len(cur), len(new)
    base = cur.base
    if (isinstance(base, np.ndarray) and base.ndim == 1 and
            base.dtype == cur.dtype == np.float64 and base.flags.owndata and
            base.flags.writeable and base.size >= n + k and
            cur.ctypes.data == base.ctypes.data and cur.strides == (8,)):
        base[n:n + k] = new
        return base[:n + k]
    buf = np.empty(max(2 * (n + k), 1024), dtype=np.float64)
    buf[:n] = cur
    buf[n:n + k] = new
    return buf[:n + k]


def device_block():
    """Upper limit of in-bound points examined per launch."""
    return 1 << 20
