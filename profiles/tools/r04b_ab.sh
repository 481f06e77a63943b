#!/bin/bash
# Same-box comparison of trainer builds: libraries nautilus_amd/lib/libnautilus_hip_<tag>.so
# (built from other commits, not shipped) against the current one.
#   gpurun -- bash profiles/tools/r04b_ab.sh R A
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b/train_speed_ab.txt
rm -f $O
for rep in 1 2; do
  for tag in "$@" current; do
    lib=nautilus_amd/lib/libnautilus_hip_$tag.so
    [ $tag = current ] && lib=nautilus_amd/lib/libnautilus_hip.so
    echo "== $tag (pass $rep)" >> $O
    NAUTILUS_HIP_LIB=$lib timeout 300 python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
