#!/bin/bash
# nb_eval_fast_kernel: the ellipsoid stage's A operands read one k-tile ahead
# (variant library, -DNB_FAST_ELL_AHEAD) against the shipped build, same box.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
V=$R/nautilus_amd/lib/libnautilus_hip_varb.so
{
  for i in 1 2 3; do
    echo "shipped, run $i"
    timeout 300 python profiles/tools/fast_time.py 50 20 2>&1 | grep "D="
    echo "ellipsoid operands read ahead (variant), run $i"
    NAUTILUS_HIP_LIB=$V timeout 300 python profiles/tools/fast_time.py 50 20 2>&1 | grep "D="
  done
} > $O/fast_ell_ahead_ab.txt 2>&1
cat $O/fast_ell_ahead_ab.txt
