#!/bin/bash
# nb_ell_stream_pipe_kernel: the A operands of a row tile read one MFMA chain
# ahead (shipped) against the build before (nautilus_amd/lib/
# libnautilus_hip_varb.so: every LDS read directly in front of its MFMAs), same
# box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  for i in 1 2; do
  echo "A operands read ahead (shipped), run $i"
  timeout 300 python profiles/tools/stream_bench.py 8 19 20 33 49 50 63 64 | grep stream
  echo "before (variant library), run $i"
  NAUTILUS_HIP_LIB=$R/nautilus_amd/lib/libnautilus_hip_varb.so timeout 300 python profiles/tools/stream_bench.py 8 19 20 33 49 50 63 64 | grep stream
  done
} > $O/stream_prefetch_ab.txt 2>&1
cat $O/stream_prefetch_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -3 | tee $O/stream_tests.log
