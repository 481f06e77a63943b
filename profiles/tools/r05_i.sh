#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gmm or mvee or moments" 2>&1 | tail -15 > $O/parity_i.log; tail -4 $O/parity_i.log
timeout 300 python profiles/tools/gmm_bench.py > $O/gmm_bench_4.txt 2>&1
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 300 python profiles/tools/gmm_bench.py 50 2000 50 10000 100 10000 > $O/gmm_phases_4.txt 2>&1
grep "^d=" $O/gmm_bench_4.txt; grep "\[gmm\]" $O/gmm_phases_4.txt | sort | uniq -c | sort -rn | awk 'NR%5==1' | head -6
rm -f $O/accept_bench_10_processes.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|index list|gathered" >> $O/accept_bench_10_processes.txt
done
grep "^D=" $O/accept_bench_10_processes.txt | cut -c1-200
cd /tmp; rm -rf /tmp/r05_explore
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_explore -o ex -- python $R/profiles/tools/explore_profile.py > /tmp/explore.log 2>&1
grep -E "^wall|^bounds|^ rows" /tmp/explore.log | head -20 > $O/explore_summary.txt
find /tmp/r05_explore -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/explore_kernel_stats.csv
head -3 $O/explore_summary.txt | cut -c1-600
head -8 $O/explore_kernel_stats.csv | cut -c1-160
