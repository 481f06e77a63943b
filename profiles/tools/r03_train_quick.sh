#!/bin/bash
mkdir -p gpurun_out/r3c
echo "== sc1 (product)"; python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | head -2
echo "== sc0 variant"; NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_sc0.so python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | head -2
NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_sc0.so python -m pytest tests/test_hip_parity.py -x -q -k "emulator_full_fit or emulator_training" 2>&1 | tail -3
