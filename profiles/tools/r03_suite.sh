#!/bin/bash
# whole GPU suite with per-test durations
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r03/suite.log 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r03/suite.log | tail -40
