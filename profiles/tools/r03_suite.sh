#!/bin/bash
# the GPU suite behind the three configuration tests that passed in the
# previous call (C1, C2, C3: 201 s)
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 \
  --deselect "tests/test_configs_gpu.py::test_gaussian_configs_against_reference_runs" \
  --deselect tests/test_configs_gpu.py::test_C3_rosenbrock_full_run_against_the_reference \
  > gpurun_out/r03/suite2.log 2>&1
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r03/suite2.log | tail -40
python profiles/tools/stream_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/stream_bench2.txt
