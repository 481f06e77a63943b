#!/bin/bash
# trainer: parity tests, then step times
mkdir -p gpurun_out/r03
python -m pytest tests/test_hip_parity.py -x -q -k "emulator" 2>&1 | tail -4
python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/train_speed.txt
python profiles/tools/train_many.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/train_many.txt
