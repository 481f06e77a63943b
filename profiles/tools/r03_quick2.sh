#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 600 python profiles/tools/train_many.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/train_many.txt
python -m pytest tests/test_hip_parity.py -x -q -k "emulator or train" 2>&1 | tail -3
