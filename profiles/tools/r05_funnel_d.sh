#!/bin/bash
# more seeds at the reference's reduced settings, 30 and 50 dimensions
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { timeout 1500 python examples/run_config.py "$@" 2>>$O/funnel_d.err | tail -1 >> $O/funnel_d.jsonl; }
for S in 2 3 4 5; do run C5-D30 --n-live 2000 --n-networks 4 --n-batch 100 --seed $S; done
run C5-D50 --n-live 2000 --n-networks 4 --n-batch 100 --seed 1
run C5-D30 --n-live 2000 --n-networks 4 --n-batch 100 --seed 2 --keep-exploration
wc -l $O/funnel_d.jsonl
