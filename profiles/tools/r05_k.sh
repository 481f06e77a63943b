#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 400 python profiles/tools/explore_profile.py > $O/explore_gmm_phases.txt 2>&1
grep "\[gmm\]" $O/explore_gmm_phases.txt | awk 'NR%5==1' | cut -c1-260
grep -E "^wall" $O/explore_gmm_phases.txt | cut -c1-400
