#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 300 python profiles/tools/explore_profile.py 2>&1 | grep -E "^wall|^bounds" | cut -c1-500 | tee $O/explore_summary_overlap.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/suite_a.log
tail -6 $O/suite_a.log
