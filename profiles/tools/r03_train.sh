#!/bin/bash
# trainer: parity tests, per-step latency, in-kernel phase stamps (debug build)
mkdir -p gpurun_out/r3b
python -m pytest tests/test_hip_parity.py -x -q -k "emulator or train" 2>&1 | tail -4 | tee gpurun_out/r3b/train_tests.log
NB_TRAIN_TWO_LAUNCH=1 python -m pytest tests/test_hip_parity.py -x -q -k "emulator_full_fit or emulator_training" 2>&1 | tail -4 | tee gpurun_out/r3b/train_tests_two_launch.log
python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b/train_speed.txt
NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_dbg.so python profiles/tools/train_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b/train_phases.txt
