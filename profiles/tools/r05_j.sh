#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gmm" 2>&1 | tail -15 > $O/parity_j.log; tail -4 $O/parity_j.log
timeout 300 python profiles/tools/gmm_bench.py > $O/gmm_bench_5.txt 2>&1
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 300 python profiles/tools/gmm_bench.py 50 2000 50 10000 100 10000 20 2000 > $O/gmm_phases_5.txt 2>&1
grep "^d=" $O/gmm_bench_5.txt; grep "\[gmm\]" $O/gmm_phases_5.txt | sort | uniq -c | sort -rn | awk 'NR%5==1' | head -8
timeout 300 python profiles/tools/explore_profile.py 2>&1 | grep -E "^wall|^bounds" | cut -c1-500
