"""Test-side glue: upload an oracle bound's parameters through the product's
C-ABI wrapper so that HIP output can be compared with the oracle."""

import numpy as np


def member_from_oracle(b):
    from nautilus_amd import device
    if hasattr(b, 'dim_cube'):                       # OMixture
        if b.ellipsoid is None:
            return device.member()
        idx = np.flatnonzero(~b.dim_cube).astype(np.int32)
        e = b.ellipsoid
        return device.member(e.c, e.B, e.B_inv, idx_ell=idx)
    if hasattr(b, 'B'):                              # OEllipsoid
        return device.member(b.c, b.B, b.B_inv)
    return device.member()                           # OCube


def neural_from_oracle(nb):
    from nautilus_amd import device
    e = nb.outer_bound
    out = dict(ellipsoid=device.member(e.c, e.B, e.B_inv),
               score_predict_min=float(nb.score_predict_min), mlp=None)
    if nb.emulator is not None:
        out['mlp'] = dict(mean=nb.emulator.mean, scale=nb.emulator.scale,
                          nets=[(n.coefs, n.intercepts)
                                for n in nb.emulator.networks])
    return out


def upload(bound):
    """Oracle bound (any class) -> nautilus_amd.device.DeviceBound."""
    from nautilus_amd import device
    from oracle import bounds_oracle as bo
    if isinstance(bound, bo.ONautilus):
        u = bound.outer_bound
        return device.DeviceBound(
            bound.n_dim, [member_from_oracle(m) for m in u.bounds],
            u.log_v_all, u.cube is not None,
            [neural_from_oracle(nb) for nb in bound.neural_bounds])
    if isinstance(bound, bo.OUnion):
        return device.DeviceBound(
            bound.n_dim, [member_from_oracle(m) for m in bound.bounds],
            bound.log_v_all, bound.cube is not None)
    if isinstance(bound, bo.ONeural):
        return device.DeviceBound(bound.n_dim, [], None, False,
                                  [neural_from_oracle(bound)])
    if isinstance(bound, bo.OCube):
        return device.DeviceBound(bound.n_dim, [member_from_oracle(bound)],
                                  [0.0], True)
    return device.DeviceBound(bound.n_dim, [member_from_oracle(bound)],
                              [bound.log_v], False)


def near_boundary(values, edge, tol):
    return np.abs(np.asarray(values) - edge) < tol
