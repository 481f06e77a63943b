#!/bin/bash
# Round 5, first funnel experiment: which knob moves the evidence bias?
#  (1) the reference's own settings of tests/golden/make_golden_funnel.py
#      ("reduced": n_live 2000, n_networks 4) at n_batch 100 / 1024, D 10-50
#  (2) config 5's settings at D = 30: n_batch 8192 / 1024, exploration kept /
#      discarded
# one JSON line per run -> gpurun_out/r05/funnel_a.jsonl
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { timeout 1500 python examples/run_config.py "$@" 2>>$O/funnel_a.err | tail -1 >> $O/funnel_a.jsonl; }
for D in 10 20 30; do
  for S in 0 1; do
    run C5-D$D --n-live 2000 --n-networks 4 --n-batch 100 --seed $S
    run C5-D$D --n-live 2000 --n-networks 4 --n-batch 1024 --seed $S
  done
done
run C5-D50 --n-live 2000 --n-networks 4 --n-batch 100 --seed 0
run C5-D30 --n-batch 8192 --seed 0
run C5-D30 --n-batch 1024 --seed 0
run C5-D30 --n-batch 8192 --seed 0 --keep-exploration
cat $O/funnel_a.jsonl
