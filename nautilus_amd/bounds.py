"""Device-backed bounds with the reference's duck-typed protocol.

Every class offers what ``nautilus.Sampler`` touches on a bound (reference
nautilus/sampler.py:791-798, 932, 1002, 1023-1035, 1069, 1218):
``Class.compute(...)``, ``.contains(points)``, ``.sample(n_points)``,
``.log_v``, ``.reset(rng)`` (and ``n_ell`` / ``n_net`` where the reference has
them).  Construction runs on the host (``geometry.py``); the parameters are
uploaded once (``device.DeviceBound``) and every per-point operation runs in
the HIP kernels.  ``contains`` / ``sample`` accept and return numpy arrays
like the reference; the ``*_device`` variants keep cuda tensors on the GPU
(used by the sampler's hot loop).

Random numbers: a bound owns a Philox stream whose 64-bit key is drawn from
the shared ``numpy.random.Generator`` when the bound is created or ``reset``
(DESIGN.md "RNG contract"); the same generator state therefore reproduces the
same points bit for bit, as in the reference (tests/test_bounds.py:59-74,
412-441 of the reference), while different pool sizes / GPU counts agree
statistically only.
"""

from time import time

import numpy as np
import torch
from scipy.special import logsumexp
from scipy.stats import rankdata

from . import device, geometry
from .emulator import NeuralNetworkEmulator

import os as _os
# debugging aids: NB_FILL_TRACE=1 prints one line per refill launch and per
# trained emulator to stderr; NB_DUMP_COLLAPSE=<file.npz> saves the training
# set of the first emulator whose whole ensemble died (tests/tools/collapse_check.py)
_FILL_TRACE = bool(_os.environ.get('NB_FILL_TRACE'))
MIN_DRAW = 1 << 14          # proposals per launch, lower limit
MAX_DRAW = 1 << int(_os.environ.get('NB_MAX_DRAW_LOG2', '22'))   # upper limit
#                             (bounds the scratch memory)
DEAD_MARGIN = 5e-3          # loss within this of the constant predictor's
MAX_BARREN = 256            # launches of MAX_DRAW without one accepted point
# measured guard of the pre-fill: after GUARD_LAUNCHES launches of MAX_DRAW
# proposals in one refill with an acceptance below GUARD_ACCEPTANCE the bound
# is reported barren (BarrenBound) -- an ensemble with one marginally alive
# network passes the loss test of NeuralBound.compute_many and resets the
# MAX_BARREN counter with every stray accepted point



def _MARGIN(need, accepted=0):
    """Proposals drawn per accepted point needed, over the inverse of the
    acceptance measured so far: four standard deviations on top -- of the
    count of this launch and of the acceptance estimate itself, which rests on
    the ``accepted`` points the bound has handed out so far -- between 2 % and
    the flat 20 % of the rounds before 6 (0.2 ms of surplus proposals per
    65 536-point step of the headline run; a young bound of configuration 4,
    whose acceptance is known from a few hundred points, keeps the 20 %: with
    2 % its refills fell short every other time and the run took 187 s instead
    of 170).  NB_DRAW_MARGIN=loose restores the flat 20 %.  A launch that falls
    short is followed by another, as before."""
    if _os.environ.get('NB_DRAW_MARGIN') == 'loose':
        return 1.2
    rel = 4.0 * max(1.0 / np.sqrt(max(need, 1.0)),
                    1.0 / np.sqrt(max(accepted, 1.0)))
    return 1.0 + min(0.2, max(0.02, rel))


PREFETCH_LAUNCHES = 1       # launches of one refill issued ahead (prefetch)
GUARD_LAUNCHES = 64
GUARD_ACCEPTANCE = 1e-7


class BarrenBound(RuntimeError):
    """A bound whose rejection sampler accepts (next to) nothing."""


def _default_rng(rng):
    return np.random.default_rng() if rng is None else rng


def _to_numpy_mask(mask, like):
    return mask if isinstance(like, torch.Tensor) else mask.cpu().numpy()


class _PhiloxStream:
    """64-bit key + running proposal index of one bound."""

    def __init__(self, rng):
        self.rekey(rng)

    def rekey(self, rng):
        self.seed = int(rng.integers(0, 2**63 - 1))
        self.offset = 0

    def take(self, n):
        start = self.offset
        self.offset += int(n)
        return self.seed, start


class _Fifo:
    """Accepted points waiting to be handed out (``self.points`` of the
    reference's Union / NautilusBound), kept on the device in one grow-only
    buffer: rows [head, tail) are queued.  (Rebuilding the queue with
    ``torch.cat`` on every refill asked the allocator for a block of a new
    size every time -- device mallocs in the middle of the sampling phase.)"""

    def __init__(self, n_dim):
        self.n_dim = n_dim
        self.data = torch.empty((0, n_dim), dtype=torch.float64,
                                device='cuda')
        self.head = 0
        self.tail = 0

    def __len__(self):
        return self.tail - self.head

    @property
    def buf(self):
        """Rows 0 .. tail (the queue is ``buf[head:]``)."""
        return self.data[:self.tail]

    def push(self, rows):
        k, m = int(rows.shape[0]), len(self)
        cap = self.data.shape[0]
        if self.tail + k > cap:
            if m + k <= cap and self.head >= m:
                # slide the queue to the front (source and target disjoint)
                self.data[:m].copy_(self.data[self.head:self.tail])
            else:
                new = torch.empty((max(2 * cap, 2 * (m + k), 4096),
                                   self.n_dim), dtype=torch.float64,
                                  device='cuda')
                new[:m].copy_(self.data[self.head:self.tail])
                self.data = new
            self.head, self.tail = 0, m
        self.data[self.tail:self.tail + k].copy_(rows)
        self.tail += k

    def pop(self, n):
        """The next ``n`` rows as a VIEW of the queue's buffer: valid until
        the next ``push`` (a refill may slide the queue to the front or
        replace the buffer) -- copy what has to live longer."""
        out = self.data[self.head:self.head + n]
        self.head += n
        return out

    def unpop(self, n):
        """Return the last ``n`` popped rows to the front of the queue."""
        self.head -= n

    def clear(self):
        self.head = self.tail = 0

    def __getstate__(self):
        return dict(n_dim=self.n_dim,
                    rows=self.data[self.head:self.tail].cpu().numpy())

    def __setstate__(self, state):
        self.n_dim = state['n_dim']
        self.data = torch.from_numpy(state['rows']).cuda()
        self.head = 0
        self.tail = int(self.data.shape[0])


class _Persistent:
    """``write`` / ``read`` / ``update`` of the reference's bounds (HDF5 groups
    in the reference's layout; implemented in io.py)."""

    def write(self, group):
        from . import io
        io.write_bound(self, group)

    def update(self, group):
        from . import io
        io.update_bound(self, group)

    @classmethod
    def read(cls, group, rng=None):
        from . import io
        return io.read_bound(cls, group, rng) if cls is not PhaseShift \
            else io.read_bound(cls, group)


class _DeviceBoundBase(_Persistent):
    """Shared contains / upload plumbing."""

    _dev = None

    def device_bound(self):
        if self._dev is None:
            self._dev = self._upload()
        return self._dev

    def contains_device(self, x):
        return self.device_bound().contains(x)

    def contains(self, points):
        single = (not isinstance(points, torch.Tensor) and
                  np.ndim(points) == 1)
        mask = self.contains_device(points)
        out = _to_numpy_mask(mask, points)
        return bool(out[0]) if single else out

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_dev'] = None
        return state


# ---------------------------------------------------------------------------
# primitive bounds
# ---------------------------------------------------------------------------

def minimum_volume_enclosing_ellipsoid(points, n_max=100, n_batch=20):
    """Reference bounds/basic.py:175-241 under its own name: centre ``c``,
    shape matrix ``A`` ((x - c)^T A (x - c) <= 1) and ``A^-1`` of the batched
    Khachiyan iteration, computed on the device (``geometry.mvee``)."""
    return geometry.mvee(points, n_max, n_batch)


def invert_symmetric_positive_semidefinite_matrix(m):
    """Reference bounds/basic.py:154-172: inverse through the Cholesky factor
    (a single small matrix: host LAPACK)."""
    from scipy.linalg import cho_factor, cho_solve
    m = np.asarray(m, dtype=float)
    return cho_solve(cho_factor(m, lower=True), np.eye(len(m)))


class UnitCube(_DeviceBoundBase):
    """Unit hypercube (reference bounds/basic.py:9-151)."""

    @classmethod
    def compute(cls, n_dim, rng=None):
        self = cls()
        self.n_dim = n_dim
        self.rng = _default_rng(rng)
        self._stream = _PhiloxStream(self.rng)
        return self

    def _member(self):
        return device.member()

    def _upload(self):
        return device.DeviceBound(self.n_dim, [self._member()], [0.0], True)

    def sample_device(self, n_points=100):
        seed, off = self._stream.take(n_points)
        return self.device_bound().propose(seed, off, n_points)

    def sample(self, n_points=100, pool=None):
        return self.sample_device(n_points).cpu().numpy()

    @property
    def log_v(self):
        return 0

    def reset(self, rng=None):
        if rng is not None:
            self.rng = rng
            self._stream.rekey(rng)


class Ellipsoid(_DeviceBoundBase):
    """Ellipsoid (reference bounds/basic.py:244-449)."""

    @classmethod
    def compute(cls, points, enlarge_per_dim=1.1, rng=None):
        return cls.compute_many([points], enlarge_per_dim, rng)[0]

    @classmethod
    def compute_many(cls, point_sets, enlarge_per_dim=1.1, rng=None):
        """``compute`` (basic.py:265-316) for several independent point sets;
        their minimum-volume ellipsoids are fitted in the same GPU launches.
        No random numbers are consumed by the fits; the Philox keys are drawn
        in list order."""
        params = geometry.ellipsoid_params_batch(
            [np.asarray(p) for p in point_sets], enlarge_per_dim)
        out = []
        for p in params:
            self = cls()
            self.n_dim = len(p['c'])
            self.c, self.A, self.B, self.B_inv = (p['c'], p['A'], p['B'],
                                                  p['B_inv'])
            self.rng = _default_rng(rng)
            self._stream = _PhiloxStream(self.rng)
            out.append(self)
        return out

    @classmethod
    def from_params(cls, c, B, B_inv=None, A=None, rng=None):
        self = cls()
        self.c = np.asarray(c, float)
        self.n_dim = len(self.c)
        self.B = np.tril(np.asarray(B, float))
        self.B_inv = np.tril(np.linalg.inv(self.B) if B_inv is None
                             else np.asarray(B_inv, float))
        self.A = self.B_inv.T @ self.B_inv if A is None else A
        self.rng = _default_rng(rng)
        self._stream = _PhiloxStream(self.rng)
        return self

    def params(self):
        return dict(c=self.c, A=self.A, B=self.B, B_inv=self.B_inv)

    def _member(self, idx_ell=None, free_dims=False):
        return device.member(self.c, self.B, self.B_inv, idx_ell=idx_ell,
                             free_dims=free_dims)

    def _upload(self):
        return device.DeviceBound(self.n_dim, [self._member()], [self.log_v],
                                  False)

    def transform(self, points, inverse=False):
        """basic.py:318-342 (host helper; the device path fuses it)."""
        if not inverse:
            return np.einsum('ij, ...j', self.B_inv, points - self.c)
        return np.einsum('ij, ...j', self.B, points) + self.c

    def contains_device(self, x):
        x = device.as_device_points(x, self.n_dim)
        if x.shape[0] >= 4096:
            return self.device_bound().contains_stream(x)
        return self.device_bound().contains(x)

    def sample_device(self, n_points=100):
        seed, off = self._stream.take(n_points)
        return self.device_bound().propose(seed, off, n_points)

    def sample(self, n_points=100):
        return self.sample_device(n_points).cpu().numpy()

    @property
    def log_v(self):
        return geometry.ellipsoid_log_volume(self.B)

    def reset(self, rng=None):
        if rng is not None:
            self.rng = rng
            self._stream.rekey(rng)


class UnitCubeEllipsoidMixture(_DeviceBoundBase):
    """Cube along some dimensions, ellipsoid along the others (reference
    bounds/basic.py:452-726)."""

    @classmethod
    def compute(cls, points, enlarge_per_dim=1.1, rng=None):
        return cls.compute_many([points], enlarge_per_dim, rng)[0]

    @classmethod
    def compute_many(cls, point_sets, enlarge_per_dim=1.1, rng=None):
        """``compute`` (basic.py:471-563) for several independent point sets:
        their greedy cube / ellipsoid searches advance in lockstep and share
        the GPU launches of their ellipsoid fits."""
        params = geometry.mixture_params_batch(
            [np.asarray(p) for p in point_sets], enlarge_per_dim)
        out = []
        for dim_cube, ell in params:
            self = cls()
            self.dim_cube = dim_cube
            self.n_dim = len(dim_cube)
            self.rng = _default_rng(rng)
            self.ellipsoid = None if ell is None else Ellipsoid.from_params(
                ell['c'], ell['B'], ell['B_inv'], ell['A'], rng=self.rng)
            self.cube = (UnitCube.compute(int(np.sum(self.dim_cube)),
                                          rng=self.rng)
                         if np.any(self.dim_cube) else None)
            self._stream = _PhiloxStream(self.rng)
            out.append(self)
        return out

    @classmethod
    def from_params(cls, dim_cube, ellipsoid, rng=None):
        self = cls()
        self.dim_cube = np.asarray(dim_cube, bool)
        self.n_dim = len(self.dim_cube)
        self.rng = _default_rng(rng)
        self.ellipsoid = ellipsoid
        self.cube = (UnitCube.compute(int(np.sum(self.dim_cube)),
                                      rng=self.rng)
                     if np.any(self.dim_cube) else None)
        self._stream = _PhiloxStream(self.rng)
        return self

    def _member(self):
        if self.ellipsoid is None:
            return device.member()
        idx = np.flatnonzero(~self.dim_cube).astype(np.int32)
        return self.ellipsoid._member(idx_ell=idx)

    def _upload(self):
        return device.DeviceBound(self.n_dim, [self._member()], [self.log_v],
                                  False)

    def transform(self, points):
        """basic.py:565-592 (host helper used by the mixture split)."""
        out = np.copy(points)
        if self.cube is not None:
            idx = np.flatnonzero(self.dim_cube)
            out[:, idx] = points[:, idx] * 2 - 1
        if self.ellipsoid is not None:
            idx = np.flatnonzero(~self.dim_cube)
            out[:, idx] = self.ellipsoid.transform(points[:, idx])
        return out

    def sample_device(self, n_points=100):
        seed, off = self._stream.take(n_points)
        return self.device_bound().propose(seed, off, n_points)

    def sample(self, n_points=100):
        return self.sample_device(n_points).cpu().numpy()

    @property
    def log_v(self):
        return 0 if self.ellipsoid is None else self.ellipsoid.log_v

    def reset(self, rng=None):
        if rng is not None:
            self.rng = rng
            self._stream.rekey(rng)
            if self.ellipsoid is not None:
                self.ellipsoid.reset(rng)
            if self.cube is not None:
                self.cube.reset(rng)


# ---------------------------------------------------------------------------
# rejection sampling shared by Union and NautilusBound
# ---------------------------------------------------------------------------

class _PrefetchSlots:
    """Output buffers of refills that are in flight (``prefetch``): a refill
    in flight keeps its compacted rows in a scratch buffer of its own until
    they land in the bound's queue.  At most N_SLOTS are outstanding in a
    process; asking for another lands the oldest (long finished) first."""

    N_SLOTS = 4
    free = list(range(N_SLOTS))
    owners = []                # (bound, slot), oldest first

    @classmethod
    def acquire(cls, bound):
        """A free slot for ``bound`` (which may hold slots of the refill it
        is issuing); with none free the oldest other refill in flight lands
        first."""
        if not cls.free:
            oldest = next(b for b, _ in cls.owners if b is not bound)
            oldest._land_pending()
        slot = cls.free.pop()
        cls.owners.append((bound, slot))
        return slot

    @classmethod
    def release(cls, slot):
        cls.owners = [(b, s) for b, s in cls.owners if s != slot]
        cls.free.append(slot)


class _RejectionSampler(_DeviceBoundBase):
    """FIFO + Monte-Carlo volume counters around ``DeviceBound.sample_launch``
    (the chunked loops of bounds/union.py:305-327 and bounds/nautilus.py:
    212-244, with one launch of >= 16384 proposals instead of chunks of
    1000)."""

    def _init_sampling(self, rng):
        self.rng = _default_rng(rng)
        self._stream = _PhiloxStream(self.rng)
        self._fifo = None
        self.n_sample = 0
        self.n_reject = 0

    def _queue(self, land=True):
        """The FIFO; a refill launched ahead of time (``prefetch``) lands in
        it first -- whoever looks at the queue sees every accepted point."""
        if land and self.__dict__.get('_pending') is not None:
            self._land_pending()
        if self._fifo is None:
            self._fifo = _Fifo(self.n_dim)
        return self._fifo

    def _land_pending(self):
        """Counters and rows of the launch ``prefetch`` issued: its counts
        come to the host now (no wait if the launch has finished meanwhile),
        the rows move into the queue, the slot is free again."""
        pending, self._pending = self._pending, None
        for rows, counts, n_draw, slot in pending:
            _PrefetchSlots.release(slot)
            self._collect(rows, counts.cpu().numpy(), n_draw)

    def prefetch(self, n_points):
        """Launch the refill that ``sample_device(n_points)`` would need --
        WITHOUT waiting for it: draw, accept and compact are queued behind
        whatever the stream holds, the counts stay on the device until the
        queue is looked at next (``_queue``).  The sampler calls this for the
        shell it expects to sample next, behind the last launch of the
        current batch: the GPU works on the refill while the host does the
        batch's bookkeeping (reference: the refill of nautilus.py:212-244
        starts when ``sample`` is called, with the host idle meanwhile).  The
        proposals are those the synchronous refill would have drawn -- the
        same Philox stream, consumed in order -- so the points a bound hands
        out do not depend on when its refills were launched."""
        if self.__dict__.get('_pending') is not None:
            return False
        need = n_points - len(self._queue(land=False))
        if need <= 0:
            return False
        # A launch is limited to MAX_DRAW proposals.  Where that is expected
        # to fall short of ``need`` a second launch could follow at once
        # (PREFETCH_LAUNCHES = 2; the refill loop issues it after a wait for
        # the first one's count, sized by the real shortfall) -- measured on
        # the headline step: 10.20-10.27 ms against 10.11-10.12 with one
        # (the second launch is sized from the estimate and draws more).
        pending = []
        acc = max(self._acceptance(), 1e-7)
        while need > 0 and len(pending) < PREFETCH_LAUNCHES:
            n_draw = self._launch(need)
            seed, off = self._stream.take(n_draw)
            slot = _PrefetchSlots.acquire(self)
            rows, counts = self.device_bound().sample_launch(
                seed, off, n_draw, reuse=True, out_role='prefetch%d' % slot)
            pending.append((rows, counts, n_draw, slot))
            need -= int(n_draw * acc / 1.05)
            self._pending = pending
        return True

    def drop_pending(self):
        """Forget a prefetched launch (its proposals were drawn and are simply
        not looked at: i.i.d. draws, independent of everything kept)."""
        if self.__dict__.get('_pending') is not None:
            for entry in self._pending:
                _PrefetchSlots.release(entry[3])
            self._pending = None

    def __getstate__(self):
        if self.__dict__.get('_pending') is not None:
            self._land_pending()
        state = super().__getstate__()
        state.pop('_pending', None)
        return state

    @property
    def points(self):
        """The reference's ``self.points`` (numpy view of the FIFO)."""
        q = self._queue()
        return q.buf[q.head:].cpu().numpy()

    def _acceptance(self):
        raise NotImplementedError

    def _account(self, n_draw, n_outer, n_final):
        raise NotImplementedError

    def _launch(self, need):
        """Number of proposals of one refill launch for ``need`` more
        points."""
        acc = max(self._acceptance(), 1e-7)
        margin = _MARGIN(need, self.n_sample - self.n_reject)
        n_draw = int(min(MAX_DRAW, max(MIN_DRAW, margin * need / acc)))
        n_draw = (n_draw + 63) // 64 * 64
        return n_draw

    def _collect(self, rows, c, n_draw):
        """Counters and queue after a launch (``c`` = its counts, host)."""
        self._account(n_draw, int(c[0]), int(c[1]))
        rows = rows[:int(c[1])]
        shift = getattr(self, 'shift', None)
        if shift is not None:      # back to the sampler's frame (:241-243)
            device.phase_shift_(rows, shift.periodic, shift.centers,
                                inverse=True)
        self._queue().push(rows)

    def _fill(self, n_points, guard=False):
        """Refill the queue to ``n_points``.  ``guard`` (the pre-fill of a
        new bound, Sampler.add_bound): give up with BarrenBound once
        GUARD_LAUNCHES full launches have accepted next to nothing -- there
        the caller drops the bound; inside a run the reference's loop is kept
        (only a bound that accepts NOTHING over MAX_BARREN launches ends it)."""
        q = self._queue()
        barren = 0
        full_launches = full_accepted = 0
        while len(q) < n_points:
            need = n_points - len(q)
            n_draw = self._launch(need)
            seed, off = self._stream.take(n_draw)
            rows, counts = self.device_bound().sample_launch(
                seed, off, n_draw, reuse=True)     # q.push copies the rows
            c = counts.cpu().numpy()
            # the reference's loop (nautilus.py:217-240) never ends for a
            # bound that accepts nothing; say so instead
            barren = barren + 1 if int(c[1]) == 0 and n_draw == MAX_DRAW \
                else 0
            if barren >= MAX_BARREN:
                raise BarrenBound(
                    'the bound accepted none of %d proposals (%d inside its '
                    'ellipsoids): its emulator rejects everything' %
                    (barren * MAX_DRAW, int(c[0])))
            if n_draw == MAX_DRAW:
                full_launches += 1
                full_accepted += int(c[1])
                if guard and full_launches >= GUARD_LAUNCHES and \
                        full_accepted < \
                        GUARD_ACCEPTANCE * full_launches * MAX_DRAW:
                    raise BarrenBound(
                        'the bound accepted %d of %d proposals: its emulators '
                        'reject (next to) everything' %
                        (full_accepted, full_launches * MAX_DRAW))
            if _FILL_TRACE:
                import sys
                dev = self.device_bound()
                print('[fill] need=%d n_draw=%d outer=%d final=%d K=%d M=%d '
                      'dense_need=%s' % (need, n_draw, c[0], c[1],
                                         dev.n_members, dev.n_neural,
                                         dev.dense_need), file=sys.stderr,
                      flush=True)
            self._collect(rows, c, n_draw)

    def sample_device(self, n_points=100):
        self._fill(n_points)
        return self._queue().pop(n_points)

    def _reset_sampling(self, rng=None):
        self.drop_pending()
        self._queue().clear()
        self.n_sample = 0
        self.n_reject = 0
        if rng is not None:
            self.rng = rng
            self._stream.rekey(rng)


class Union(_RejectionSampler):
    """Union of ellipsoids or cube-ellipsoid mixtures (reference
    bounds/union.py:43-451)."""

    _final_bit = 1

    @classmethod
    def compute(cls, points, enlarge_per_dim=1.1, n_points_min=None,
                unit=True, bound_class=Ellipsoid, rng=None):
        self = cls()
        points = np.asarray(points)
        self.n_dim = points.shape[1]
        self.enlarge_per_dim = enlarge_per_dim
        if n_points_min is None:
            self.n_points_min = self.n_dim + 1
        else:
            if n_points_min < self.n_dim + 1:
                raise ValueError('The number of points per bound must be '
                                 'larger than the number of dimensions.')
            self.n_points_min = n_points_min
        self._init_sampling(rng)
        self.cube = UnitCube.compute(self.n_dim, rng=self.rng) if unit \
            else None
        self.points_bounds = [points]
        self.bounds = [bound_class.compute(
            points, enlarge_per_dim=enlarge_per_dim, rng=self.rng)]
        self.log_v_all = np.array([self.bounds[0].log_v])
        self.block = np.atleast_1d(len(points) < 2 * self.n_points_min)
        return self

    @classmethod
    def from_members(cls, members, unit=True, rng=None):
        self = cls()
        self.bounds = list(members)
        self.n_dim = self.bounds[0].n_dim
        self._init_sampling(rng)
        self.cube = UnitCube.compute(self.n_dim, rng=self.rng) if unit \
            else None
        self.log_v_all = np.array([b.log_v for b in self.bounds])
        return self

    def _upload(self):
        return device.DeviceBound(
            self.n_dim, [b._member() for b in self.bounds], self.log_v_all,
            self.cube is not None)

    def _invalidate(self):
        self._dev = None
        self._reset_sampling()

    # -- decomposition (host) ---------------------------------------------
    def split(self, allow_overlap=True):
        """union.py:153-229."""
        if not allow_overlap and not isinstance(self.bounds[0], Ellipsoid):
            raise ValueError("'allow_overlap' can only be False if bounds are "
                             "ellipsoids.")
        if not np.any(~self.block):
            return False
        index = int(np.argmax(np.where(~self.block, self.log_v_all, -np.inf)))
        pts = self.points_bounds[index]
        try:
            labels = geometry.two_component_labels(
                self.bounds[index].transform(pts), self.n_points_min,
                int(self.rng.integers(2**32 - 1)))
        except geometry.DegenerateMixture:
            # every device restart ended with an empty cluster or a singular
            # covariance (duplicated / collapsed points): these points cannot
            # be divided -- the outcome of a split that does not shrink the
            # volume (union.py:204-207), not the end of the run
            self.block[index] = True
            return self.split(allow_overlap=allow_overlap)
        cls = type(self.bounds[0])
        halves = cls.compute_many([pts[labels == lab] for lab in (0, 1)],
                                  enlarge_per_dim=self.enlarge_per_dim,
                                  rng=self.rng)
        if not allow_overlap and geometry.ellipsoids_overlap(
                [b.params() for b in
                 self.bounds[:index] + self.bounds[index + 1:] + halves]):
            return False
        if logsumexp([halves[0].log_v, halves[1].log_v]) > \
                self.bounds[index].log_v:
            self.block[index] = True
            return self.split(allow_overlap=allow_overlap)
        self.points_bounds.pop(index)
        self.points_bounds += [pts[labels == 0], pts[labels == 1]]
        self.bounds.pop(index)
        self.bounds = self.bounds + halves
        self.log_v_all = np.array([b.log_v for b in self.bounds])
        self.block = np.concatenate((
            np.delete(self.block, index),
            [len(self.points_bounds[-2]) < 2 * self.n_points_min,
             len(self.points_bounds[-1]) < 2 * self.n_points_min]))
        self._invalidate()
        return True

    def trim(self, threshold=1e3):
        """union.py:231-267."""
        if len(self.bounds) == 1:
            return False
        log_r = (np.log([len(p) for p in self.points_bounds]) -
                 np.array([b.log_v for b in self.bounds]))
        index = int(np.argmin(log_r))
        if log_r[index] - np.median(np.delete(log_r, index)) < \
                -np.log(threshold):
            self.points_bounds.pop(index)
            self.bounds.pop(index)
            self.block = np.delete(self.block, index)
            self.log_v_all = np.array([b.log_v for b in self.bounds])
            self._invalidate()
            return True
        return False

    # -- sampling ---------------------------------------------------------
    def _acceptance(self):
        if self.n_sample == 0:
            return 1.0
        return 1.0 - self.n_reject / self.n_sample

    def _account(self, n_draw, n_outer, n_final):
        self.n_sample += n_draw                    # union.py:322
        self.n_reject += n_draw - n_outer          # union.py:323

    def _fill(self, n_points):
        q = self._queue()
        while len(q) < n_points:
            need = n_points - len(q)
            acc = max(self._acceptance(), 1e-7)
            n_draw = int(min(MAX_DRAW, max(MIN_DRAW, 1.2 * need / acc)))
            n_draw = (n_draw + 63) // 64 * 64
            seed, off = self._stream.take(n_draw)
            dev = self.device_bound()
            rows, counts = dev.sample_launch(seed, off, n_draw, mask=1,
                                             reuse=True)   # push copies
            k = int(counts[1])
            self._account(n_draw, k, k)
            q.push(rows[:k])

    def sample(self, n_points=100):
        return self.sample_device(n_points).cpu().numpy()

    @property
    def log_v(self):
        """union.py:329-343."""
        if self.n_sample == 0:
            self._fill(1)
        return geometry.log_volume_union(self.log_v_all, self.n_reject,
                                         self.n_sample)

    def reset(self, rng=None):
        """union.py:431-450."""
        self._reset_sampling(rng)
        if rng is not None:
            if self.cube is not None:
                self.cube.reset(rng)
            for b in self.bounds:
                b.reset(rng)


class NeuralBound(_DeviceBoundBase):
    """Ellipsoid AND emulator score above a threshold (reference
    bounds/neural.py:10-174)."""

    @classmethod
    def compute(cls, points, log_l, log_l_min, enlarge_per_dim=1.1,
                n_networks=4, neural_network_kwargs={}, pool=None, rng=None,
                comm=None):
        """``points`` may be a numpy array or a cuda tensor (the sampler keeps
        all points on the device); ``log_l`` is a numpy array."""
        return cls.compute_many([(points, log_l)], log_l_min,
                                enlarge_per_dim=enlarge_per_dim,
                                n_networks=n_networks,
                                neural_network_kwargs=neural_network_kwargs,
                                rng=rng, comm=comm)[0]

    @classmethod
    def compute_many(cls, data, log_l_min, enlarge_per_dim=1.1, n_networks=4,
                     neural_network_kwargs={}, rng=None, comm=None):
        """bounds/neural.py:58-97 for several (points, log_l) sets -- the
        neural bounds of one NautilusBound (nautilus.py:107-114).  The
        reference trains their emulators one after the other; here all
        ensembles train side by side on the GPU.  No random numbers are
        consumed (the networks are seeded 0..n_networks-1, neural.py:88), so
        the order of the work does not matter."""
        rng = _default_rng(rng)
        bounds, train = [], []
        xs, lives = [], []
        for points, log_l in data:
            x = device.as_device_points(points)
            xs.append(x)
            lives.append(x[torch.from_numpy(
                np.asarray(log_l) >= log_l_min).cuda()].cpu().numpy())
        outer = Ellipsoid.compute_many(lives, enlarge_per_dim=enlarge_per_dim,
                                       rng=rng)
        for (points, log_l), x, ell in zip(data, xs, outer):
            self = cls()
            log_l = np.asarray(log_l)
            self.n_dim = x.shape[1]
            self.outer_bound = ell
            bounds.append(self)
            self.emulator_dead = False
            if n_networks == 0:
                self.emulator = None
                self.score_predict_min = 0
                continue
            inside = self.outer_bound.contains_device(x)
            log_l = log_l[inside.cpu().numpy()]
            # ellipsoid-frame coordinates on the device (basic.py:340)
            x_t = transform_device(self.outer_bound, x[inside])
            score = np.zeros(len(log_l))
            hi = log_l >= log_l_min
            score[hi] = 0.5 * (1 + (rankdata(log_l[hi]) - 0.5) / np.sum(hi))
            score[~hi] = 0.5 * ((rankdata(log_l[~hi]) - 0.5) / np.sum(~hi))
            train.append((self, x_t, score, hi))
        if train:
            emus = NeuralNetworkEmulator.train_many(
                [(x_t, score) for _, x_t, score, _ in train],
                n_networks=n_networks,
                neural_network_kwargs=neural_network_kwargs, comm=comm)
            for (self, x_t, score, hi), emu in zip(train, emus):
                self.emulator = emu
                # A network whose last epoch is no better than the constant
                # predictor has died (all units of a layer inactive, a hazard
                # of Adam at the reference's learning rate 1e-2).  A bound
                # whose whole ensemble died accepts nothing, and the
                # reference's sampling loop would spin on it for ever.
                floor = 0.5 * np.var(score) * (1.0 - DEAD_MARGIN)
                self.emulator_dead = all(
                    n.loss_curve_[-1] >= floor for n in emu.neural_networks)
                pred = emu.predict_device(x_t).cpu().numpy()
                self.score_predict_min = np.polyval(
                    np.polyfit(score, pred, 3), np.amin(score[hi]))
                if _FILL_TRACE:
                    import sys
                    print('[neural] n=%d hi=%d spm=%.6f pred[min %.4f q50 %.4f'
                          ' q99 %.4f max %.4f] pred(hi)[min %.4f q50 %.4f] '
                          'n_iter=%s loss=%s' % (
                              len(score), int(np.sum(hi)),
                              self.score_predict_min, pred.min(),
                              np.median(pred), np.quantile(pred, 0.99),
                              pred.max(), pred[hi].min(),
                              np.median(pred[hi]),
                              [n.n_iter_ for n in emu.neural_networks],
                              ['%.2e' % n.loss_curve_[-1]
                               for n in emu.neural_networks]),
                          file=sys.stderr, flush=True)
                    # the training set of a dead ensemble, for
                    # tests/tools/collapse_check.py
                    dump = _os.environ.get('NB_DUMP_COLLAPSE')
                    if dump and self.emulator_dead and \
                            not _os.path.exists(dump):
                        np.savez(dump, x_t=x_t.cpu().numpy(), score=score,
                                 hi=hi, mean=emu.mean, scale=emu.scale)
        return bounds

    @classmethod
    def from_parts(cls, ellipsoid, emulator, score_predict_min):
        self = cls()
        self.n_dim = ellipsoid.n_dim
        self.outer_bound = ellipsoid
        self.emulator = emulator
        self.score_predict_min = score_predict_min
        return self

    def _neural(self):
        return dict(ellipsoid=self.outer_bound._member(),
                    score_predict_min=float(self.score_predict_min),
                    mlp=None if self.emulator is None
                    else self.emulator.mlp_desc())

    def _upload(self):
        return device.DeviceBound(self.n_dim, [], None, False,
                                  [self._neural()])

    def contains(self, points):
        pts = points if isinstance(points, torch.Tensor) else \
            np.atleast_2d(points)
        return _to_numpy_mask(self.contains_device(pts), points)


class PhaseShift(_Persistent):
    """Recentring of periodic dimensions (reference bounds/periodic.py:6-72):
    the largest gap between the points of a periodic dimension is moved onto
    the boundary of the unit interval."""

    @classmethod
    def compute(cls, points, periodic):
        """periodic.py:21-46.  ``points``: host array or cuda tensor of the
        live points; only the ``len(periodic)`` columns are sorted."""
        self = cls()
        self.periodic = np.asarray(periodic)
        self.centers = np.zeros(len(self.periodic))
        if isinstance(points, torch.Tensor):
            cols = torch.sort(points[:, torch.as_tensor(
                self.periodic, device=points.device, dtype=torch.long)],
                dim=0).values.cpu().numpy()
        else:
            cols = np.sort(np.asarray(points)[:, self.periodic], axis=0)
        for i in range(len(self.periodic)):
            x = cols[:, i]
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

    def transform_device(self, x, inverse=False):
        """New cuda tensor with the shift applied (``nb_phase_shift``)."""
        out = device.as_device_points(x, None).clone()
        return device.phase_shift_(out, self.periodic, self.centers, inverse)

    def transform(self, points, inverse=False):
        """periodic.py:50-72; numpy in -> numpy out, tensor in -> tensor."""
        out = self.transform_device(points, inverse)
        return out if isinstance(points, torch.Tensor) else out.cpu().numpy()

    def params(self):
        return self.periodic, self.centers


def transform_device(ellipsoid, x):
    """B_inv (x - c) for a cuda tensor of points (reference basic.py:340), on
    the matrix cores like every later evaluation of the same transform."""
    return ellipsoid.device_bound().transform(x)


class NautilusBound(_RejectionSampler):
    """Outer multi-ellipsoid union AND any neural bound (reference
    bounds/nautilus.py:13-398)."""

    _final_bit = 2

    @classmethod
    def compute(cls, points, log_l, log_l_min, log_v_target,
                enlarge_per_dim=1.1, n_points_min=None, split_threshold=100,
                periodic=None, n_networks=4, neural_network_kwargs={},
                pool=None, rng=None, comm=None):
        """nautilus.py:39-144.  ``comm`` (a ``parallel.ShardedComm``) deals
        the emulator networks out over the GPUs of a sharded run, the way the
        reference maps them over ``pool`` (neural.py:93-96)."""
        self = cls()
        t0 = time()
        log_l = np.asarray(log_l)
        x = device.as_device_points(points)
        self.n_dim = x.shape[1]
        self._init_sampling(rng)
        is_live = torch.from_numpy(log_l >= log_l_min).cuda()
        self.shift = None
        if periodic is not None:                          # nautilus.py:91-96
            self.shift = PhaseShift.compute(x[is_live], periodic)
            x = self.shift.transform_device(x)
        live = x[is_live].cpu().numpy()

        # non-overlapping ellipsoids -> one neural bound each (:100-114)
        multi = Union.compute(live, enlarge_per_dim=enlarge_per_dim,
                              n_points_min=n_points_min,
                              bound_class=Ellipsoid, rng=self.rng)
        while multi.split(allow_overlap=False):
            pass
        t1 = time()
        data = []
        for ell in multi.bounds:
            sel = ell.contains_device(x)
            data.append((x[sel], log_l[sel.cpu().numpy()]))
        self.neural_bounds = NeuralBound.compute_many(
            data, log_l_min, enlarge_per_dim=enlarge_per_dim,
            n_networks=n_networks,
            neural_network_kwargs=neural_network_kwargs, rng=self.rng,
            comm=comm)

        t2 = time()
        # sampling envelope (:116-133)
        self.outer_bound = Union.compute(
            live, enlarge_per_dim=enlarge_per_dim, n_points_min=n_points_min,
            bound_class=UnitCubeEllipsoidMixture, rng=self.rng)
        limit = np.log(split_threshold * enlarge_per_dim**self.n_dim)
        while self.outer_bound.log_v - log_v_target > limit:
            if not self.outer_bound.split():
                break
        while self.outer_bound.log_v - log_v_target > limit:
            if not self.outer_bound.trim():
                break
        # the composite draws through its own fused pipeline; the envelope's
        # counters keep accumulating (they are valid MC samples of the same
        # volume), only its private FIFO is dropped
        self.outer_bound._queue().clear()
        self.timing = dict(bound_decompose=t1 - t0, bound_neural=t2 - t1,
                           bound_envelope=time() - t2)
        return self

    @classmethod
    def from_parts(cls, outer_bound, neural_bounds, rng=None, shift=None):
        self = cls()
        self.n_dim = outer_bound.n_dim
        self.shift = shift
        self.outer_bound = outer_bound
        self.neural_bounds = list(neural_bounds)
        self._init_sampling(rng)
        return self

    @property
    def emulators_dead(self):
        """True if the whole ensemble of one of the neural bounds died in
        training (``NeuralBound.emulator_dead``): the bound accepts nothing."""
        return any(getattr(nb, 'emulator_dead', False)
                   for nb in self.neural_bounds)

    def _upload(self):
        u = self.outer_bound
        return device.DeviceBound(
            self.n_dim, [b._member() for b in u.bounds], u.log_v_all,
            u.cube is not None, [nb._neural() for nb in self.neural_bounds],
            shift=None if self.shift is None else self.shift.params())

    def _acceptance(self):
        a = 1.0
        if self.outer_bound.n_sample > 0:
            a *= 1.0 - self.outer_bound.n_reject / self.outer_bound.n_sample
        if self.n_sample > 0:
            a *= 1.0 - self.n_reject / self.n_sample
        return a

    def _account(self, n_draw, n_outer, n_final):
        self.outer_bound.n_sample += n_draw        # union.py:322-323
        self.outer_bound.n_reject += n_draw - n_outer
        self.n_sample += n_outer                   # nautilus.py:221-222
        self.n_reject += n_outer - n_final

    def sample(self, n_points=100, return_points=True, pool=None,
               guard=False):
        """nautilus.py:193-244.  ``pool`` is accepted for compatibility: the
        proposals of one launch are already spread over the whole GPU (and
        over all GPUs of a ``DevicePool``).  ``guard``: see ``_fill``."""
        if not return_points:
            self._fill(n_points, guard=guard)
            return None
        return self.sample_device(n_points).cpu().numpy()

    @property
    def log_v(self):
        """nautilus.py:246-261."""
        if self.n_sample == 0:
            self._fill(1)
        u = self.outer_bound
        return (geometry.log_volume_union(u.log_v_all, u.n_reject,
                                          u.n_sample) +
                np.log(1.0 - self.n_reject / self.n_sample))

    @property
    def n_ell(self):
        return int(np.sum([np.any(~b.dim_cube)
                           for b in self.outer_bound.bounds]))

    @property
    def n_net(self):
        if self.neural_bounds[0].emulator is not None:
            return len(self.neural_bounds) * len(
                self.neural_bounds[0].emulator.neural_networks)
        return 0

    def reset(self, rng=None):
        """nautilus.py:382-397."""
        self._reset_sampling(rng)
        self.outer_bound.reset(rng)
