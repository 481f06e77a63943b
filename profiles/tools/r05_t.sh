#!/bin/bash
# A/B of two trainer builds in one process sequence (same box): shipped library
# against nautilus_amd/lib/libnautilus_hip_varb.so, each twice.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
T=${1:-ab}
for i in 1 2; do
  echo "shipped, run $i"; timeout 300 python profiles/tools/train_speed.py 2>&1 | grep "us/step"
  echo "variant, run $i"; NAUTILUS_HIP_LIB=$R/nautilus_amd/lib/libnautilus_hip_varb.so timeout 300 python profiles/tools/train_speed.py 2>&1 | grep "us/step"
done > $O/train_ab_$T.txt
cat $O/train_ab_$T.txt
