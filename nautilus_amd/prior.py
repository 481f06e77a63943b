"""Prior helper with the interface of ``nautilus.Prior`` (reference
nautilus/prior.py:9-181).  Host-side and elementwise; out of the GPU hot path
(SURVEY.md section 2.1)."""

import numbers

import numpy as np
from scipy.stats import uniform


def _is_free(dist):
    return hasattr(dist, 'isf')


class Prior:
    """Ordered collection of named model parameters."""

    def __init__(self):
        self.keys = []
        self.dists = []

    def add_parameter(self, key=None, dist=(0, 1)):
        """Add a parameter: ``dist`` is a (low, high) tuple (uniform), a
        number (fixed), the name of an earlier parameter (tied) or an object
        with an ``isf`` method (prior.py:25-73)."""
        if key is None:
            name = 'x_{}'.format(len(self.keys))
        elif not isinstance(key, str):
            raise TypeError("Keyword argument 'key' must be a string.")
        elif key in self.keys:
            raise ValueError("Key '{}' already in key list.".format(key))
        else:
            name = key

        if isinstance(dist, tuple):
            entry = uniform(loc=dist[0], scale=dist[1] - dist[0])
        elif isinstance(dist, numbers.Number) or _is_free(dist):
            entry = dist
        elif isinstance(dist, str):
            if dist not in self.keys or dist == str(key):
                raise ValueError('Key {} not defined previously.'.format(dist))
            entry = dist
            while isinstance(self.dists[self.keys.index(entry)], str):
                entry = self.dists[self.keys.index(entry)]
        else:
            raise TypeError("Keyword argument 'dist' does not have the "
                            "correct type")
        self.keys.append(name)
        self.dists.append(entry)

    def dimensionality(self):
        return sum(_is_free(d) for d in self.dists)

    def _check(self, arr):
        if self.dimensionality() != arr.shape[-1]:
            raise ValueError('Dimensionality of points does not match prior.')

    def unit_to_physical(self, points):
        """Inverse-survival transform per free parameter (prior.py:85-120)."""
        points = np.asarray(points)
        self._check(points)
        out = np.zeros_like(points)
        col = 0
        for dist in self.dists:
            if _is_free(dist):
                out[..., col] = dist.isf(1 - points[..., col])
                col += 1
        return out

    def physical_to_dictionary(self, phys_points):
        """prior.py:122-162."""
        phys_points = np.asarray(phys_points)
        self._check(phys_points)
        out = {}
        col = 0
        for key, dist in zip(self.keys, self.dists):
            if _is_free(dist):
                out[key] = phys_points[..., col]
                col += 1
            elif isinstance(dist, numbers.Number):
                out[key] = np.ones(phys_points[..., 0].shape) * dist
        for key, dist in zip(self.keys, self.dists):
            if isinstance(dist, str):
                out[key] = out[dist]
        return out

    def unit_to_dictionary(self, points):
        return self.physical_to_dictionary(self.unit_to_physical(points))
