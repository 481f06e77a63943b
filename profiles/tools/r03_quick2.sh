#!/bin/bash
for v in s6 s10; do
  echo "== poll sleep $v"
  NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_$v.so python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | head -1
  NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_$v.so python profiles/tools/train_many.py 2>&1 | grep -v amdgpu.ids
done
echo "== product (16)"; python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | head -1
