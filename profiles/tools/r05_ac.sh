#!/bin/bash
# 65 <= n_dim <= 112: the pipelined kernel with units of ONE tile (the next
# tile's loads in flight while the current one is multiplied, operands read
# ahead) against the plain kernel (NB_STREAM_PLAIN=1), same box, three runs.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  for i in 1 2 3; do
  echo "pipelined, one tile per unit (shipped), run $i"
  NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 65 72 80 84 96 99 100 112 | grep stream
  echo "plain (NB_STREAM_PLAIN=1), run $i"
  NB_STREAM_PLAIN=1 NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 65 72 80 84 96 99 100 112 | grep stream
  done
} > $O/stream_pipe1_ab.txt 2>&1
cut -c1-110 $O/stream_pipe1_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -3 | tee $O/stream_tests.log
