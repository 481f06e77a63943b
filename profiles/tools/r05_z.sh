#!/bin/bash
# nb_ell_stream_kernel beyond 64 dimensions: A operands read one chunk of four
# k-steps ahead (shipped) against the build before (libnautilus_hip_varb.so:
# every LDS read directly in front of its MFMAs), same box, two runs each.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  for i in 1 2 3; do
  echo "A operands read ahead (shipped), run $i"
  NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 68 84 100 116 | grep stream
  echo "before (variant library), run $i"
  NAUTILUS_HIP_LIB=$R/nautilus_amd/lib/libnautilus_hip_varb.so NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 68 84 100 116 | grep stream
  done
} > $O/stream_ahead_small_ab.txt 2>&1
cat $O/stream_ahead_small_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -3 | tee $O/stream_tests.log
