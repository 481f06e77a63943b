"""Brute-force check of the shell bookkeeping on a finished funnel run
(non-nested bounds): for a sample of shells, fresh points of bound i are
tested against the LATER bounds (sampler.py:796-799) on the device
(nb_list_eval) and in numpy from the bounds' own parameters (ellipsoid
transforms with B_inv, cube limits, emulator forward pass), and the volume
estimate of bound i (Monte-Carlo counters) is compared with a fresh
estimate.

    python profiles/tools/funnel_shell_check.py [n_dim] [n_batch] [seed]
"""
import sys
import numpy as np
import torch

sys.path.insert(0, '.')
from nautilus_amd import Sampler, unit_prior  # noqa: E402
from nautilus_amd.configs import baseline_config  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
c = baseline_config('C5-D%d' % d)
s = Sampler(unit_prior, c['likelihood'], n_dim=d, n_live=2000, n_networks=4,
            n_batch=n_batch, vectorized=True, seed=seed)
s.run(discard_exploration=True)
print('run: log_z %.4f (analytic %.4f) bounds %d n_like %d' % (
    s.log_z, c['analytic_log_z'], len(s.bounds), s.n_like), flush=True)


def ell_inside(e, x):
    y = (x - e.c) @ e.B_inv.T
    return np.sum(y * y, axis=1) < 1.0


def member_inside(m, x):
    if hasattr(m, 'dim_cube'):
        ok = np.ones(len(x), dtype=bool)
        if np.any(m.dim_cube):
            xc = x[:, m.dim_cube]
            ok &= np.all((xc >= 0) & (xc < 1), axis=1)
        if m.ellipsoid is not None:
            ok &= ell_inside(m.ellipsoid, x[:, ~m.dim_cube])
        return ok
    return ell_inside(m, x)


def emulator_predict(emu, y):
    z = (y - emu.mean) / emu.scale
    out = 0.0
    for net in emu.neural_networks:
        h = z
        for k in range(4):
            h = h @ net.coefs_[k] + net.intercepts_[k]
            if k < 3:
                h = np.maximum(h, 0.0)
        out = out + h[:, 0]
    return out / len(emu.neural_networks)


def bound_inside(b, x):
    if not hasattr(b, 'neural_bounds'):          # the unit cube
        return np.all((x >= 0) & (x < 1), axis=1), np.zeros(len(x), bool)
    u = b.outer_bound
    ok = np.zeros(len(x), dtype=bool)
    for m in u.bounds:
        ok |= member_inside(m, x)
    if u.cube is not None:
        ok &= np.all((x >= 0) & (x < 1), axis=1)
    any_nb = np.zeros(len(x), dtype=bool)
    edge = np.zeros(len(x), dtype=bool)
    for nb in b.neural_bounds:
        e = nb.outer_bound
        y = (x - e.c) @ e.B_inv.T
        r2 = np.sum(y * y, axis=1)
        edge |= np.abs(r2 - 1.0) < 1e-9
        inside = r2 < 1.0
        if nb.emulator is not None:
            score = emulator_predict(nb.emulator, y)
            edge |= np.abs(score - nb.score_predict_min) < 1e-7
            inside &= score > nb.score_predict_min - 1e-9
        any_nb |= inside
    return ok & any_nb, edge


n_b = len(s.bounds)
rows = []
for i in sorted(set(np.linspace(1, n_b - 2, 12).astype(int))):
    b = s.bounds[i]
    n0, r0 = b.n_sample, b.n_reject
    u0, ur0 = b.outer_bound.n_sample, b.outer_bound.n_reject
    log_v_run = b.log_v
    x = b.sample_device(200000).clone()
    # fresh volume estimate from the proposals this drew
    dn, dr = b.n_sample - n0, b.n_reject - r0
    du, dur = b.outer_bound.n_sample - u0, b.outer_bound.n_reject - ur0
    from scipy.special import logsumexp
    log_v_new = (logsumexp(b.outer_bound.log_v_all) + np.log(1 - dur / du) +
                 np.log(1 - dr / dn))
    later = s._later_bounds(i)
    got = later.contains_any(x).cpu().numpy()
    xh = x.cpu().numpy()
    want = np.zeros(len(xh), dtype=bool)
    edge = np.zeros(len(xh), dtype=bool)
    for j in range(i + 1, n_b):
        w, e = bound_inside(s.bounds[j], xh)
        want |= w
        edge |= e
    # the bound's own contains on its own samples
    own, own_edge = bound_inside(b, xh)
    frac_run = s.shell_n[i] / max(1, s.shell_n_sample[i] -
                                  s.shell_n_sample_exp[i])
    rows.append((i, log_v_run, log_v_new, 1 - got.mean(), 1 - want.mean(),
                 frac_run, int(np.sum((got != want) & ~edge)),
                 int(edge.sum()), float(own[~own_edge].mean())))
    print('shell %3d: log_v run %.4f fresh %.4f (diff %+.4f) | in-shell '
          'fraction device %.5f numpy %.5f run %.5f | mismatches %d (edge %d) '
          '| own samples inside own bound (numpy) %.6f' % (
              i, log_v_run, log_v_new, log_v_new - log_v_run, 1 - got.mean(),
              1 - want.mean(), frac_run, rows[-1][6], rows[-1][7],
              rows[-1][8]), flush=True)
