import os
import sys

import numpy as np
import pytest
from threadpoolctl import threadpool_limits

# The reference pins BLAS to one thread in its hot regions (basic.py:302,
# neural.py:10, sampler.py:789, 1022) and its CI sets OMP_NUM_THREADS=1; the
# golden vectors were generated that way.  Threaded BLAS re-associates sums.
threadpool_limits(limits=1)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'),
                        allow_pickle=False))


@pytest.fixture
def golden():
    return load_golden
