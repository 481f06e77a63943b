#!/bin/bash
# accept_bench.py 50 100 in one process, shipped (one-tile) instantiations:
# how often is the D = 100 call slow, and under which ROCr scratch switch?
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b/scratch_experiment_2.txt
rm -f $O
for env in "" "HSA_SCRATCH_SINGLE_LIMIT=2147483648" "HSA_ENABLE_SCRATCH_ALT=1" "HSA_SCRATCH_SINGLE_LIMIT=2147483648 HSA_NO_SCRATCH_RECLAIM=1" "HSA_SCRATCH_SINGLE_LIMIT_ASYNC=8589934592 HSA_SCRATCH_SINGLE_LIMIT=8589934592"; do
  echo "=== env: $env" >> $O
  for i in 1 2 3 4 5; do
    env $env timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep "D=100.*proposals:" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O
  done
done
cat $O
