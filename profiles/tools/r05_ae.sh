#!/bin/bash
# Pipelined stream kernel without the masking selects (shipped) against the
# build before (libnautilus_hip_varb.so), same box, three runs each.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  for i in 1 2 3; do
  echo "no selects (shipped), run $i"
  timeout 300 python profiles/tools/stream_bench.py 19 20 33 49 50 64 | grep stream
  NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 84 100 | grep stream
  echo "before (variant library), run $i"
  NAUTILUS_HIP_LIB=$R/nautilus_amd/lib/libnautilus_hip_varb.so timeout 300 python profiles/tools/stream_bench.py 19 20 33 49 50 64 | grep stream
  NAUTILUS_HIP_LIB=$R/nautilus_amd/lib/libnautilus_hip_varb.so NB_STREAM_N=4194304 timeout 300 python profiles/tools/stream_bench.py 84 100 | grep stream
  done
} > $O/stream_nomask_ab.txt 2>&1
cut -c1-100 $O/stream_nomask_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -3 | tee $O/stream_tests.log
