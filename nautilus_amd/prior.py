"""Prior helper with the interface of ``nautilus.Prior`` (reference
nautilus/prior.py:9-181): named parameters that are free (a distribution
with an ``isf`` method), fixed (a number) or tied to an earlier parameter (its
name).  Numpy points are transformed on the host; cuda tensors (the batches of
a device likelihood) are transformed on the GPU when every free parameter is
uniform or normal (``nb_prior_transform``)."""

import numbers

import numpy as np
from scipy.stats import uniform

FREE, FIXED, TIED = 'free', 'fixed', 'tied'


def _kind(dist):
    if hasattr(dist, 'isf'):
        return FREE
    if isinstance(dist, numbers.Number):
        return FIXED
    if isinstance(dist, str):
        return TIED
    return None


class Prior:
    """Ordered table of model parameters.  ``keys`` and ``dists`` are the
    reference's public attributes (prior.py:22-23); everything else is derived
    from them through ``_table``."""

    def __init__(self):
        self.keys = []
        self.dists = []

    def add_parameter(self, key=None, dist=(0, 1)):
        """prior.py:25-73: ``dist`` is a (low, high) tuple (uniform), an
        object with ``isf``, a number (fixed) or the name of an earlier
        parameter (tied; chains of names resolve to their root)."""
        if key is not None and not isinstance(key, str):
            raise TypeError("Keyword argument 'key' must be a string.")
        if key is not None and key in self.keys:
            raise ValueError("Key '{}' already in key list.".format(key))
        if isinstance(dist, tuple):
            dist = uniform(loc=dist[0], scale=dist[1] - dist[0])
        kind = _kind(dist)
        if kind is None:
            raise TypeError("Keyword argument 'dist' does not have the "
                            "correct type")
        if kind == TIED:
            if dist not in self.keys or dist == str(key):
                raise ValueError('Key {} not defined previously.'.format(dist))
            while _kind(self.dists[self.keys.index(dist)]) == TIED:
                dist = self.dists[self.keys.index(dist)]
        self.keys.append('x_{}'.format(len(self.keys)) if key is None else key)
        self.dists.append(dist)

    def _table(self):
        """[(key, kind, dist, column of the free parameter or None)]."""
        rows, column = [], 0
        for key, dist in zip(self.keys, self.dists):
            kind = _kind(dist)
            rows.append((key, kind, dist, column if kind == FREE else None))
            column += kind == FREE
        return rows

    def dimensionality(self):
        return sum(kind == FREE for _, kind, _, _ in self._table())

    def _require_width(self, width):
        if self.dimensionality() != width:
            raise ValueError('Dimensionality of points does not match prior.')

    def device_spec(self):
        """(kind, loc, scale) arrays if every free parameter is a frozen scipy
        ``uniform`` (kind 0) or ``norm`` (kind 1), else None."""
        kind, loc, scale = [], [], []
        for _, row_kind, dist, _ in self._table():
            if row_kind != FREE:
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
        """x = dist.isf(1 - u) for every free parameter (prior.py:85-120)."""
        import torch
        if isinstance(points, torch.Tensor):
            from . import device
            spec = self.device_spec()
            if spec is None:
                raise ValueError(
                    'only uniform and normal parameters can be transformed '
                    'on the device')
            self._require_width(points.shape[-1])
            return device.prior_transform(points, *spec)
        points = np.asarray(points)
        self._require_width(points.shape[-1])
        physical = np.zeros_like(points)
        for _, kind, dist, column in self._table():
            if kind == FREE:
                physical[..., column] = dist.isf(1 - points[..., column])
        return physical

    def physical_to_dictionary(self, phys_points):
        """One entry per key: the column of a free parameter, a constant array
        for a fixed one, the entry of its root for a tied one
        (prior.py:122-162; numpy arrays or cuda tensors)."""
        import torch
        if isinstance(phys_points, torch.Tensor):
            constant = torch.full_like
        else:
            phys_points = np.asarray(phys_points)
            self._require_width(phys_points.shape[-1])
            constant = np.full_like
        table = self._table()
        values = {}
        for key, kind, dist, column in table:
            if kind == FREE:
                values[key] = phys_points[..., column]
            elif kind == FIXED:
                values[key] = constant(phys_points[..., 0], dist)
        for key, kind, dist, _ in table:
            if kind == TIED:
                values[key] = values[dist]
        return {key: values[key] for key in self.keys}

    def unit_to_dictionary(self, points):
        return self.physical_to_dictionary(self.unit_to_physical(points))
