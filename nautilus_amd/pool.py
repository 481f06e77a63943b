"""Worker pools for host-side likelihoods.

``NautilusPool`` keeps the interface of ``nautilus.pool.NautilusPool``
(reference nautilus/pool.py:36-107): ``map(func, iterable)`` and ``size``
over a ``multiprocessing.Pool``, a dask ``Client``, an MPI executor or
anything with ``.map``.  On the MI355X path the proposal / bound work never
goes through a pool (it is one kernel launch spread over the whole GPU); a
pool is only used to evaluate *host* likelihood functions, exactly at the
reference's call site C1 (sampler.py:863-873).
"""

from multiprocessing import Pool

_LIKELIHOOD = None


def initialize_worker(likelihood):
    """Cache the likelihood in the worker process (pool.py:6-16)."""
    global _LIKELIHOOD
    _LIKELIHOOD = likelihood


def likelihood_worker(*args):
    """Evaluate the cached likelihood (pool.py:19-33)."""
    return _LIKELIHOOD(*args)


class NautilusPool:
    """Uniform ``map`` / ``size`` over different pool flavours."""

    def __init__(self, pool, likelihood=None):
        if isinstance(pool, int):
            self.pool = Pool(pool, initializer=initialize_worker,
                             initargs=(likelihood, ))
        else:
            self.pool = pool

    def _is_dask(self):
        return 'distributed.client.Client' in str(type(self.pool))

    def map(self, func, iterable):
        if self._is_dask():
            return list(self.pool.gather(self.pool.map(func, iterable)))
        return list(self.pool.map(func, iterable))

    @property
    def size(self):
        if self._is_dask():
            return len(self.pool.nthreads())
        for attr in ('_processes', '_max_workers', 'size', 'nt'):
            if hasattr(self.pool, attr):
                return getattr(self.pool, attr)
        raise ValueError('Cannot determine size of pool.')
