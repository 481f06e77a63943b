#!/bin/bash
# whole GPU suite with per-test durations (the first three configuration tests
# passed in the previous call: 217 s)
mkdir -p gpurun_out/r03
./build/l2_read_bench > gpurun_out/r03/l2_read_bench.txt 2>&1
cat gpurun_out/r03/l2_read_bench.txt
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 \
  --deselect "tests/test_configs_gpu.py::test_gaussian_configs_against_reference_runs" \
  --deselect tests/test_configs_gpu.py::test_C3_rosenbrock_full_run_against_the_reference \
  > gpurun_out/r03/suite2.log 2>&1
tail -45 gpurun_out/r03/suite2.log
