#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 300 python profiles/tools/explore_profile.py 2>&1 | grep -E "^wall|^bounds" | cut -c1-500 | tee $O/explore_summary_overlap.txt
timeout 900 python -m pytest tests/test_sampler_gpu.py -q -m gpu -x -k "barren or envelope" 2>&1 | tail -8
timeout 2400 python -m pytest tests/test_configs_gpu.py -q -m gpu --durations=12 2>&1 | tail -25 > $O/configs_m.log; tail -22 $O/configs_m.log
