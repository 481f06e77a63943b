#!/bin/bash
# Scratch experiment of nb_cand_kernel<7, 2, 2> (440 bytes of scratch per lane)
# behind n_dim = 50 kernels in one process, under the ROCr scratch switches.
# gpurun -- bash profiles/tools/r04b_scratch.sh
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
rm -f $O/scratch_experiment.txt
for env in "" "NB_CAND_TWO_TILES=1" "NB_CAND_TWO_TILES=1 HSA_SCRATCH_SINGLE_LIMIT=2147483648" "NB_CAND_TWO_TILES=1 HSA_NO_SCRATCH_RECLAIM=1" "NB_CAND_TWO_TILES=1 HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "NB_CAND_TWO_TILES=1 HSA_NO_SCRATCH_THREAD_LIMITER=1" "NB_CAND_TWO_TILES=1 HSA_ENABLE_SCRATCH_ALT=1"; do
  echo "=== env: $env" >> $O/scratch_experiment.txt
  env $env timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -v "^$" | tail -12 >> $O/scratch_experiment.txt
done
cat $O/scratch_experiment.txt
