#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gmm" 2>&1 | tail -15 > $O/parity_l.log; tail -4 $O/parity_l.log
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 400 python profiles/tools/explore_profile.py > $O/explore_gmm_phases.txt 2>&1
grep "\[gmm\]" $O/explore_gmm_phases.txt | awk 'NR%7==1' | cut -c1-260
grep -E "^wall" $O/explore_gmm_phases.txt | cut -c1-400
timeout 300 python profiles/tools/gmm_bench.py > $O/gmm_bench_6.txt 2>&1; grep "^d=" $O/gmm_bench_6.txt
timeout 600 python -m pytest tests/test_sampler_gpu.py -x -q -m gpu -k "split or union or mixture or mode or C4 or multimodal" 2>&1 | tail -4
