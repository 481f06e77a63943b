"""Prior helper with the interface of ``nautilus.Prior`` (reference
nautilus/prior.py:9-181).  Numpy points are transformed on the host exactly as
in the reference; cuda tensors (the batches of a device likelihood) are
transformed on the GPU when every free parameter is uniform or normal
(``nb_prior_transform``)."""

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

    def device_spec(self):
        """(kind, loc, scale) arrays if every free parameter is a frozen scipy
        ``uniform`` (kind 0) or ``norm`` (kind 1), else None."""
        kind, loc, scale = [], [], []
        for dist in self.dists:
            if not _is_free(dist):
                continue
            name = getattr(getattr(dist, 'dist', None), 'name', None)
            if name not in ('uniform', 'norm'):
                return None
            try:
                _, lo, sc = dist.dist._parse_args(*dist.args, **dist.kwds)
            except Exception:
                return None
            kind.append(0 if name == 'uniform' else 1)
            loc.append(float(lo))
            scale.append(float(sc))
        return np.array(kind, np.uint8), np.array(loc), np.array(scale)

    @property
    def device(self):
        """True if batches can be transformed on the GPU."""
        return self.device_spec() is not None

    def unit_to_physical(self, points):
        """Inverse-survival transform per free parameter (prior.py:85-120)."""
        import torch
        if isinstance(points, torch.Tensor):
            from . import device
            spec = self.device_spec()
            if spec is None:
                raise ValueError(
                    'only uniform and normal parameters can be transformed '
                    'on the device')
            if self.dimensionality() != points.shape[-1]:
                raise ValueError('Dimensionality of points does not match '
                                 'prior.')
            return device.prior_transform(points, *spec)
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
        """prior.py:122-162 (numpy arrays or cuda tensors)."""
        import torch
        if isinstance(phys_points, torch.Tensor):
            out = {}
            col = 0
            for key, dist in zip(self.keys, self.dists):
                if _is_free(dist):
                    out[key] = phys_points[..., col]
                    col += 1
                elif isinstance(dist, numbers.Number):
                    out[key] = torch.full_like(phys_points[..., 0], dist)
            for key, dist in zip(self.keys, self.dists):
                if isinstance(dist, str):
                    out[key] = out[dist]
            return out
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
