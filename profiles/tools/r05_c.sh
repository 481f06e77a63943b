#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "list_eval_against" 2>&1 | tail -60 > $O/parity_c.log
cat $O/parity_c.log | tail -40
rm -f $O/stage_timing.txt
for i in 1 2 3 4 5 6; do
  echo "== process $i" >> $O/stage_timing.txt
  NB_STAGE_TIMING=1 timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|\[stage\]" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/stage_timing.txt
done
cat $O/stage_timing.txt
