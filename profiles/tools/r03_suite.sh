#!/bin/bash
# whole GPU suite with per-test durations, then the bench
mkdir -p gpurun_out/r3d
python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r3d/suite.log 2>&1
tail -40 gpurun_out/r3d/suite.log
python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
tail -c 2500 gpurun_out/r3d/bench.json; tail -3 gpurun_out/r3d/bench.err
