#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 200 python -X faulthandler -m pytest tests/test_configs_gpu.py -x -q -s -k "C2 or C5" > gpurun_out/r3e/c2c4.log 2>&1
grep -v "Extension modules\|pluggy\|_pytest" gpurun_out/r3e/c2c4.log | head -30 | cut -c1-200
