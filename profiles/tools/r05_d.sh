#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -30 > $O/parity_d.log
tail -5 $O/parity_d.log
rm -f $O/stage_timing.txt
for i in 1 2 3 4 5; do
  echo "== process $i" >> $O/stage_timing.txt
  NB_STAGE_TIMING=1 timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|\[stage\]" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/stage_timing.txt
done
grep -E "^==|D=100|call (7|8|9):" $O/stage_timing.txt
