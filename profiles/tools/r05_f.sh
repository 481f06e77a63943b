#!/bin/bash
# host wake-up after long device waits: blocking synchronize vs ROCr polling
# (HSA_ENABLE_INTERRUPT=0) vs a spin on Event.query(); then the funnel family
# at the reference's reduced settings over 8 seeds
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
rm -f $O/sync_modes.txt
for mode in default nointerrupt spin; do
  for i in 1 2 3 4 5 6; do
    case $mode in
      default) E="";;
      nointerrupt) E="HSA_ENABLE_INTERRUPT=0";;
      spin) E="NB_BENCH_SPIN=1";;
    esac
    echo "== $mode process $i" >> $O/sync_modes.txt
    env $E timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/sync_modes.txt
  done
done
grep -E "^==|D=100" $O/sync_modes.txt | paste - - | cut -c1-150
run() { timeout 1500 python examples/run_config.py "$@" 2>>$O/funnel_b.err | tail -1 >> $O/funnel_b.jsonl; }
for S in 0 1 2 3 4 5 6 7; do
  run C5-D10 --n-live 2000 --n-networks 4 --n-batch 100 --seed $S
  run C5-D20 --n-live 2000 --n-networks 4 --n-batch 100 --seed $S
done
for S in 3 4 5; do
  run C5-D10 --n-live 2000 --n-networks 4 --n-batch 100 --seed $S --keep-exploration
done
run C5-D20 --n-live 2000 --n-networks 4 --n-batch 100 --seed 3 --keep-exploration
python - <<'P'
import json
for l in open('gpurun_out/r05/funnel_b.jsonl'):
    d=json.loads(l); print(d['config'], d['seed'], 'keep' if not d['discard_exploration'] else 'disc', round(d['log_z']-d['analytic_log_z'],4), round(d['mean_x0'],5), d['n_like'], d['n_bounds'], round(d['n_eff']))
P
