#!/bin/bash
# configurations 1-3 end to end through examples/run_config.py (C4 / C5: own runs)
mkdir -p gpurun_out/r03
for c in C1 C2 C3; do
  timeout 600 python examples/run_config.py $c > gpurun_out/r03/cfg_$c.json 2> gpurun_out/r03/cfg_$c.err
  cut -c1-400 gpurun_out/r03/cfg_$c.json
done
