#!/bin/bash
# configuration 5's own settings (n_live 10000, 8 networks, n_batch 8192) with
# the exploration discarded, at the dimensions that finish
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { timeout 2400 python examples/run_config.py "$@" 2>>$O/funnel_c.err | tail -1 >> $O/funnel_c.jsonl; }
run C5-D50 --seed 0
run C5-D20 --seed 0
run C5-D10 --seed 0
python - <<'P'
import json
for l in open('gpurun_out/r05/funnel_c.jsonl'):
    d=json.loads(l); print(d['config'], d['seed'], 'keep' if not d['discard_exploration'] else 'disc', round(d['log_z']-d['analytic_log_z'],4), round(d['mean_x0'],5), d['n_like'], d['n_bounds'], round(d['n_eff']), d['wall_s'])
P
