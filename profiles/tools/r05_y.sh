#!/bin/bash
# nb_ell_stream: software-pipelined units of two tiles (shipped) against four
# tiles loaded and then multiplied (NB_STREAM_NO_PIPE=1), same box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  for i in 1 2; do
  echo "pipelined (shipped), run $i"
  timeout 300 python profiles/tools/stream_bench.py 8 19 20 33 49 50 63 64 | grep stream
  echo "four tiles, no pipelining (NB_STREAM_NO_PIPE=1), run $i"
  NB_STREAM_NO_PIPE=1 timeout 300 python profiles/tools/stream_bench.py 8 19 20 33 49 50 63 64 | grep stream
  done
} > $O/stream_pipe_ab.txt 2>&1
cat $O/stream_pipe_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -3 | tee $O/stream_tests.log
