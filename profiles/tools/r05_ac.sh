#!/bin/bash
# nb_ell_stream_wide_kernel (65 <= n_dim <= 112: two tiles per wavefront, the
# centre folded into the accumulators) against the one- / two-tile kernel with
# the centred inputs (NB_STREAM_NARROW=1), same box, three runs each.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  for i in 1 2 3; do
  echo "wide (shipped), run $i"
  NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 65 72 80 84 96 99 100 112 | grep stream
  echo "narrow (NB_STREAM_NARROW=1), run $i"
  NB_STREAM_NARROW=1 NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 65 72 80 84 96 99 100 112 | grep stream
  done
} > $O/stream_wide_ab.txt 2>&1
cut -c1-110 $O/stream_wide_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -3 | tee $O/stream_tests.log
