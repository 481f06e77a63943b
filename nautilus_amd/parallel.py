"""Multi-GPU sharding of the shell-filling loop: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm).

The reference's only parallel pattern on this path is "replicate the bound,
draw independent streams, concatenate the accepted points and add up the
counters" (nautilus/bounds/nautilus.py:223-237).
The MI355X version of it:

* every rank holds the (tiny) bound parameters and the full sampler state;
* every batch of ``n_batch`` shell points -- in the exploration phase, in the
  pre-fill of a new bound (sampler.py:1032) and in the sampling phase -- is
  split into ``n_batch / world`` per rank; each rank draws its share from its
  own Philox stream (key mixed with the rank) and evaluates the likelihood
  for it;
* ONE ``all_gather`` per batch moves the accepted points (+ their log L) to
  every rank -- equal counts, so no padding; in the sampling phase, where
  nothing reads the points before the run ends, the log L travel at once and
  the points asynchronously behind the next batches -- and one ``all_reduce`` adds the
  integer counters (``n_bound`` and the four MC-volume counters) that the
  evidence depends on (sampler.py:1133, nautilus.py:232-237).  A batch that
  pairs fresh points with transfer candidates (sampler.py:803-819) gathers
  the points first, runs the (replicated, host-RNG) pairing on the gathered
  batch and gathers the likelihoods afterwards;
* the networks of an emulator ensemble are dealt out over the ranks
  (reference neural.py:93-96 maps them over its pool); one ``all_reduce`` of
  the zero-padded weight blobs brings every network to every rank, bit for
  bit what a single rank would have trained.

There is no collective inside the kernels; the geometric part of the bound
construction (MVEE, mixture fit) is replicated: identical seeds give
identical bounds on every rank.
"""

import torch
import torch.distributed as dist

_MIX = 0x9E3779B97F4A7C15


def rank_key(seed, rank):
    """Philox key of ``rank`` derived from the shared key (rank 0 keeps the
    single-GPU stream)."""
    return (int(seed) ^ ((rank * _MIX) & (2**63 - 1))) & (2**63 - 1)


def _timed(fn):
    """Accumulate the host wall time of a collective on the communicator."""
    import functools
    import time

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        depth = getattr(self, '_depth', 0)
        self._depth = depth + 1
        t0 = time.perf_counter()
        try:
            return fn(self, *args, **kwargs)
        finally:
            self._depth = depth
            if depth == 0:            # (gather_rows_async calls gather_rows)
                self.seconds += time.perf_counter() - t0
                self.calls += 1
    return wrapper


class _Gathered:
    """A gather in flight (input kept alive until it has landed)."""

    def __init__(self, work, out, rows, comm=None):
        self.work, self.out, self.rows, self.comm = work, out, rows, comm

    def wait(self):
        if self.work is not None:
            import time
            t0 = time.perf_counter()
            self.work.wait()
            self.work = None
            if self.comm is not None:
                self.comm.seconds += time.perf_counter() - t0
        self.rows = None
        return self.out


class ShardedComm:
    """Thin wrapper of the collectives the sampler needs."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # host wall time spent inside the collectives (incl. waiting for the
        # slowest rank) and their number, for the bench line's per-rank
        # breakdown
        self.seconds = 0.0
        self.calls = 0
        self._guard_shared_devices()

    def _guard_shared_devices(self):
        """Ranks that SHARE a GPU (tests, a CPX partition handed to several
        processes) cannot count on all workgroups of a mixture fit being
        resident while another rank's fit runs on the same CUs: one workgroup
        per restart then (``nb_gmm_set_max_wgs``)."""
        if not torch.cuda.is_available():
            return
        import socket
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = (socket.gethostname(),
                 str(getattr(props, 'uuid', '')) or 'dev',
                 getattr(props, 'pci_bus_id', -1),
                 getattr(props, 'pci_device_id', -1),
                 getattr(props, 'pci_domain_id', -1))
        seen = [None] * self.world
        dist.all_gather_object(seen, ident, group=self.group)
        self.shared_device = len(set(seen)) < len(seen)
        if self.shared_device:
            from . import _lib
            _lib.check(_lib.load().nb_gmm_set_max_wgs(1))

    def describe(self, device='cuda'):
        """What the communicator really spans: backend, ranks counted by an
        all-reduce of ones, and the device every rank runs on (gathered)."""
        backend = dist.get_backend(self.group)
        seen = self.sum_ints([1], device)[0]
        name = 'cpu'
        if torch.cuda.is_available():
            i = torch.cuda.current_device()
            props = torch.cuda.get_device_properties(i)
            name = '%s #%d (%s)' % (props.name, i, getattr(
                props, 'gcnArchName', ''))
        names = [None] * self.world
        dist.all_gather_object(names, name, group=self.group)
        out = dict(backend=backend, ranks_seen=int(seen), world=self.world,
                   devices=names)
        if backend == 'nccl':
            try:
                out['rccl_version'] = '.'.join(
                    str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        return out

    def gather_floats(self, values, device):
        """(world, len(values)) nested list: every rank's short list of
        floats (per-rank timings of the bench line)."""
        t = torch.tensor([list(values)], dtype=torch.float64,
                         device=self._dev(device))
        out = torch.empty((self.world, t.shape[1]), dtype=torch.float64,
                          device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().tolist()

    @_timed
    def gather_rows(self, rows):
        """Concatenate equally sized (n_local, k) blocks of all ranks in rank
        order."""
        rows = rows.contiguous()
        out = torch.empty((self.world * rows.shape[0],) + tuple(rows.shape[1:]),
                          dtype=rows.dtype, device=rows.device)
        if rows.is_cuda and dist.get_backend(self.group) == 'gloo':
            # functional-test path (gloo has no device all-gather)
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, rows.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, rows, group=self.group)
        return out

    @_timed
    def gather_rows_async(self, rows):
        """``gather_rows`` without waiting: returns a handle whose ``wait()``
        gives the gathered tensor (and makes the current stream wait for the
        collective, not the host).  The sampling phase moves its points this
        way -- 8 D bytes per point to every rank, the bulk of a batch's
        traffic, and nothing needs them before the run ends -- while the
        next batches are drawn (sampler.py, ``_sharded_batch``)."""
        rows = rows.contiguous()
        if rows.is_cuda and dist.get_backend(self.group) == 'gloo':
            return _Gathered(None, self.gather_rows(rows), rows, self)
        out = torch.empty((self.world * rows.shape[0],) + tuple(rows.shape[1:]),
                          dtype=rows.dtype, device=rows.device)
        work = dist.all_gather_into_tensor(out, rows, group=self.group,
                                           async_op=True)
        return _Gathered(work, out, rows, self)

    @_timed
    def sum_ints(self, values, device):
        """Element-wise sum of a short list of python ints over all ranks."""
        t = torch.tensor(list(values), dtype=torch.int64,
                         device=self._dev(device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.cpu()]

    def _dev(self, device):
        return 'cpu' if dist.get_backend(self.group) == 'gloo' else device

    @_timed
    def max_float(self, value, device):
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self._dev(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.cpu()[0])

    def assert_identical(self, values, device, what='state'):
        """All ranks must hold bit-identical float values (replicated
        exploration)."""
        t = torch.tensor(list(values), dtype=torch.float64,
                         device=self._dev(device))
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise RuntimeError('replicated %s diverged between ranks: %s vs %s'
                               % (what, lo.tolist(), hi.tolist()))

    @_timed
    def sum_rows(self, rows):
        """Element-wise sum of equally shaped float tensors over all ranks
        (used with disjoint non-zero rows: x + 0 = x exactly, so the result is
        a bit-exact gather of what the owners wrote)."""
        if rows.is_cuda and dist.get_backend(self.group) == 'gloo':
            host = rows.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            return host.to(rows.device)
        out = rows.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    @_timed
    def any_flag(self, flag):
        """True on every rank if ``flag`` is true on any (collective stop /
        continue decisions: wall-clock limits differ between ranks)."""
        t = torch.tensor([1 if flag else 0], dtype=torch.int64,
                         device=self._dev('cuda'))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.cpu()[0]))

    def barrier(self):
        dist.barrier(group=self.group)


def split_batch(n_batch, world):
    """Per-rank share of a batch; the global batch must divide evenly so that
    the all-gather needs no padding."""
    if n_batch % world != 0:
        raise ValueError('n_batch=%d must be a multiple of the number of '
                         'ranks (%d)' % (n_batch, world))
    return n_batch // world


def shard_shell_batch(comm, local_rows, local_log_l, local_counts):
    """Exchange one batch: returns (all rows, all log_l, summed counters).

    local_rows (n_local, D) and local_log_l (n_local,) live on the same
    device; local_counts is a list of python ints."""
    packed = torch.cat([local_rows, local_log_l[:, None]], dim=1)
    gathered = comm.gather_rows(packed)
    totals = comm.sum_ints(local_counts, local_rows.device)
    return gathered[:, :-1].contiguous(), gathered[:, -1].contiguous(), totals
