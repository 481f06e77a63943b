#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_sampler_gpu.py tests/test_sampler_behaviour_gpu.py tests/test_io_gpu.py tests/test_live_gpu.py -x -q --durations=8 2>&1 | tail -25 | tee gpurun_out/r3e/parity.log
