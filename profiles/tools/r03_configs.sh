#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests/test_configs_gpu.py -x -q --durations=8 2>&1 | tail -25 | tee gpurun_out/r3f/configs.log
