#!/bin/bash
# Round 5, second call: parity of the kernels that changed (scratch-free
# acceptance / candidate kernels, oracle-direct list test), D = 100 slow-mode
# check over 10 processes, mixture-fit phase times, exploration profile.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/parity_b.log
cat $O/parity_b.log
rm -f $O/accept_bench_10_processes.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|index list|gathered" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/accept_bench_10_processes.txt
done
cat $O/accept_bench_10_processes.txt | grep "^D="
timeout 300 python profiles/tools/accept_bench.py 20 > $O/accept_bench_20.txt 2>&1; grep "^D=" $O/accept_bench_20.txt
timeout 300 python profiles/tools/gmm_bench.py > $O/gmm_bench.txt 2>&1
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 300 python profiles/tools/gmm_bench.py 50 2000 50 10000 100 10000 > $O/gmm_phases.txt 2>&1
cat $O/gmm_bench.txt; grep "\[gmm\]" $O/gmm_phases.txt | sort | uniq -c | sort -rn | head -12
cd /tmp; rm -rf /tmp/r05_explore
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_explore -o ex -- python $R/profiles/tools/explore_profile.py > /tmp/explore.log 2>&1
grep -E "^wall|^bounds|^ rows" /tmp/explore.log | head -20 > $O/explore_summary.txt
find /tmp/r05_explore -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/explore_kernel_stats.csv
head -3 $O/explore_summary.txt | cut -c1-600
head -14 $O/explore_kernel_stats.csv | cut -c1-160
