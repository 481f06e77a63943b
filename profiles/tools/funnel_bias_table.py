"""profiles/r05/funnel_bias.json: the funnel family (configuration 5's
problem) -- log Z - analytic, E[x_0] - analytic, bounds, likelihood calls,
wall -- for this build (examples/run_config.py lines in profiles/r05/
funnel_*.jsonl, MI355X) and for the REFERENCE (tests/golden/e2e_funnel.json,
CPU), grouped by (n_dim, n_live, n_networks, n_batch, exploration kept or
discarded).

    python profiles/tools/funnel_bias_table.py
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(
    __file__))))
sys.path.insert(0, ROOT)
from nautilus_amd.configs import funnel_log_z, funnel_moments  # noqa: E402

rows = {}


def add(who, d, n_live, e, n_batch, keep, run):
    key = (d, n_live, e, n_batch, 'kept' if keep else 'discarded', who)
    if all(r['seed'] != run['seed'] for r in rows.get(key, [])):
        rows.setdefault(key, []).append(run)


for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r05',
                                          'funnel_*.jsonl'))):
    for line in open(path):
        r = json.loads(line)
        d = int(r['config'][4:])
        add('this build (MI355X)', d, r['n_live'], r['n_networks'],
            r['n_batch'], not r['discard_exploration'],
            dict(seed=r['seed'], d_log_z=r['log_z'] - funnel_log_z(d),
                 d_mean_x0=r['mean_x0'] - funnel_moments(d)[0],
                 var_x0=r['var_x0'], n_bounds=r['n_bounds'],
                 n_like=r['n_like'], n_eff=r['n_eff'], wall_s=r['wall_s']))
with open(os.path.join(ROOT, 'tests', 'golden', 'e2e_funnel.json')) as f:
    for r in json.load(f)['runs']:
        d = r['n_dim']
        add('reference (CPU, 1 core)', d, r['n_live'], r['n_networks'], 100,
            not r['discard_exploration'],
            dict(seed=r['seed'], d_log_z=r['log_z'] - funnel_log_z(d),
                 d_mean_x0=r['mean_x0'] - funnel_moments(d)[0],
                 var_x0=r['var_x0'], n_bounds=r['n_bounds'],
                 n_like=r['n_like'], n_eff=r['n_eff'], wall_s=r['wall_s']))

out = []
for key in sorted(rows):
    runs = rows[key]
    z = np.array([r['d_log_z'] for r in runs])
    m = np.array([r['d_mean_x0'] for r in runs])
    out.append(dict(
        n_dim=key[0], n_live=key[1], n_networks=key[2], n_batch=key[3],
        exploration=key[4], sampler=key[5], runs=len(runs),
        d_log_z_mean=float(z.mean()),
        d_log_z_sd=float(z.std(ddof=1)) if len(z) > 1 else None,
        d_mean_x0_mean=float(m.mean()),
        var_x0_mean=float(np.mean([r['var_x0'] for r in runs])),
        var_x0_analytic=funnel_moments(key[0])[1],
        n_bounds_mean=float(np.mean([r['n_bounds'] for r in runs])),
        n_like_mean=float(np.mean([r['n_like'] for r in runs])),
        wall_s_mean=float(np.mean([r['wall_s'] for r in runs])),
        per_run=runs))
with open(os.path.join(ROOT, 'profiles', 'r05', 'funnel_bias.json'), 'w') as f:
    json.dump(dict(
        problem='Neal funnel on the unit cube (configs.FunnelLikelihood); '
                'analytic values by quadrature (configs.funnel_log_z, '
                'funnel_moments)', table=out), f, indent=1)
for o in out:
    print('D=%-3d n_live=%-5d E=%d n_batch=%-4d %-9s %-24s runs %d  dlogZ '
          '%+.4f%s  dE[x0] %+.5f  bounds %.0f  n_like %.0f  wall %.0f s' % (
              o['n_dim'], o['n_live'], o['n_networks'], o['n_batch'],
              o['exploration'], o['sampler'], o['runs'], o['d_log_z_mean'],
              '' if o['d_log_z_sd'] is None else ' (sd %.4f)' %
              o['d_log_z_sd'], o['d_mean_x0_mean'], o['n_bounds_mean'],
              o['n_like_mean'], o['wall_s_mean']))
