#!/bin/bash
# Odd n_dim in the bound-evaluation kernels: 16-byte pair loads from
# 8-byte-aligned rows (shipped) against the 8-byte loads of the tree before
# (nautilus_amd/lib/libnautilus_hip_varb.so), same box; then the parity and
# fuzz suites of the kernels.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
V=$R/nautilus_amd/lib/libnautilus_hip_varb.so
{
  for d in 49 33 99; do
    echo "pair loads (shipped), n_dim $d"
    timeout 300 python profiles/tools/fast_time.py $d 2>&1 | grep "D="
    timeout 300 python profiles/tools/accept_bench.py $d 2>&1 | grep "D=" | cut -c1-200
    echo "8-byte loads (before), n_dim $d"
    NAUTILUS_HIP_LIB=$V timeout 300 python profiles/tools/fast_time.py $d 2>&1 | grep "D="
    NAUTILUS_HIP_LIB=$V timeout 300 python profiles/tools/accept_bench.py $d 2>&1 | grep "D=" | cut -c1-200
  done
} > $O/odd_loads_ab.txt 2>&1
cat $O/odd_loads_ab.txt
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py -q -m gpu 2>&1 | tail -5 | tee $O/parity_fuzz.log
