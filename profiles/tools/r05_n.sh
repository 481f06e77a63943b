#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 300 python profiles/tools/explore_profile.py 2>&1 | grep -E "^wall|^bounds" | cut -c1-700 | tee $O/explore_summary_overlap2.txt
timeout 900 python -m pytest tests/test_sampler_gpu.py -q -m gpu -x -k "barren or envelope" 2>&1 | tail -4
