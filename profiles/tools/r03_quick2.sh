#!/bin/bash
for v in f2 f4; do
  echo "== few-poll sleep $v"
  NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_$v.so python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | head -4
done
