#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 1700 python -m pytest tests/test_configs_gpu.py -x -q --durations=8 > gpurun_out/r3f/configs.log 2>&1
grep -v "^  File\|Extension modules" gpurun_out/r3f/configs.log | tail -40
