#!/bin/bash
# Odd n_dim in nb_ell_stream_kernel: 16-byte loads from 8-byte-aligned rows
# (shipped) against the LDS-DMA kernel (NB_STREAM_ODD_DMA=1), same box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
{
  echo "unaligned 16-byte loads (shipped)"
  timeout 300 python profiles/tools/stream_bench.py 19 33 49 50 63 | grep stream
  echo "LDS DMA kernel (NB_STREAM_ODD_DMA=1)"
  NB_STREAM_ODD_DMA=1 timeout 300 python profiles/tools/stream_bench.py 19 33 49 50 63 | grep stream
} > $O/stream_odd_ab.txt 2>&1
cat $O/stream_odd_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "stream or ellipsoid or contains" 2>&1 | tail -5 | tee $O/stream_tests.log
