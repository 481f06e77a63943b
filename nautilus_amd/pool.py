"""Worker pools for host-side likelihoods.

On the MI355X path proposals and bound tests never go through a pool (one
launch covers the whole GPU); a pool only serves *host* likelihood callables,
at the reference's call site C1 (nautilus/sampler.py:863-873).
``NautilusPool`` offers what the sampler needs from any pool flavour --
``map(func, iterable)`` returning a list and ``size`` -- like
``nautilus.pool.NautilusPool`` (reference nautilus/pool.py:36-107).
"""

import multiprocessing
import numbers

# attribute that holds the worker count, by pool flavour: multiprocessing.Pool,
# concurrent.futures executors, mpi4py / schwimmbad pools, ...
_SIZE_ATTRIBUTES = ('_processes', '_max_workers', 'size', 'nt')

_worker_likelihood = None


def _install_likelihood(likelihood):
    global _worker_likelihood
    _worker_likelihood = likelihood


def likelihood_worker(*args):
    """Runs in a worker of an integer-sized pool: the likelihood was shipped
    once when the worker started, only the points travel per call."""
    return _worker_likelihood(*args)


def _start_method():
    """Workers are forked (closures and interactively defined likelihoods
    work, as with the reference's default pool) unless this process has
    already initialised the HIP runtime: a forked child would inherit an
    unusable GPU context, so a fork server is used then and the likelihood
    has to be picklable."""
    try:
        import torch
        if torch.cuda.is_initialized():
            return 'forkserver'
    except ImportError:
        pass
    return 'fork'


def _require_picklable(likelihood):
    """A fork server starts its workers from a clean process: the likelihood
    reaches them by pickle.  Say so instead of a bare PicklingError from the
    pool's initializer."""
    import pickle
    try:
        pickle.dumps(likelihood)
    except Exception as err:
        raise ValueError(
            'The HIP runtime of this process is already initialised, so the '
            'likelihood workers are started through a fork server and the '
            'likelihood must be picklable (a module-level function, not a '
            'lambda or closure): {}.  Create the Sampler with pool=<int> '
            'before any other GPU use, or pass a pool object.'.format(err)
        ) from err


class NautilusPool:
    """``map`` / ``size`` over an integer (a new ``multiprocessing`` pool), a
    ``multiprocessing.Pool``, an executor, a dask client or anything else
    with ``map``."""

    def __init__(self, pool, likelihood=None):
        if isinstance(pool, numbers.Integral) and not isinstance(pool, bool):
            method = _start_method()
            if method != 'fork' and likelihood is not None:
                _require_picklable(likelihood)
            context = multiprocessing.get_context(method)
            self.pool = context.Pool(int(pool), initializer=_install_likelihood,
                                     initargs=(likelihood, ))
        else:
            self.pool = pool

    def _is_dask(self):
        # a dask.distributed client returns futures from map() and knows its
        # workers through nthreads()
        return hasattr(self.pool, 'gather') and hasattr(self.pool, 'nthreads')

    def map(self, func, iterable):
        result = self.pool.map(func, iterable)
        if self._is_dask():
            result = self.pool.gather(result)
        return list(result)

    @property
    def size(self):
        if self._is_dask():
            return len(self.pool.nthreads())
        for name in _SIZE_ATTRIBUTES:
            value = getattr(self.pool, name, None)
            if value is not None:
                return value
        raise ValueError('Cannot determine size of pool.')
