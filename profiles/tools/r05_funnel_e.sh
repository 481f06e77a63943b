#!/bin/bash
# three more seeds of the funnel at 50 dimensions with the reference's reduced
# settings (the reference's own runs at that size: tests/golden/
# make_golden_funnel.py reduced_D50_seed0 / 1, ~4 h each on a CPU core)
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { timeout 900 python examples/run_config.py "$@" 2>>$O/funnel_e.err | tail -1 >> $O/funnel_e.jsonl; }
for S in 2 3 4; do run C5-D50 --n-live 2000 --n-networks 4 --n-batch 100 --seed $S; done
wc -l $O/funnel_e.jsonl; cut -c1-330 $O/funnel_e.jsonl
