#!/bin/bash
mkdir -p gpurun_out/r3g
NB_FILL_TRACE=1 timeout 700 python examples/run_config.py C4 --timeout 600 --n-eff 2000 --watchdog 680 > gpurun_out/r3g/c4_wd.json 2> gpurun_out/r3g/c4_wd.err
grep -v "^\[fill\]" gpurun_out/r3g/c4_wd.err | tail -3 | cut -c1-400
awk '/^\[fill\]/{c++; if (c>3000) next} {print}' gpurun_out/r3g/c4_wd.err > gpurun_out/r3g/c4_wd.trim; mv gpurun_out/r3g/c4_wd.trim gpurun_out/r3g/c4_wd.err
cut -c1-700 gpurun_out/r3g/c4_wd.json
