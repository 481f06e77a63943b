# exploration phase of the headline run under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp
rm -rf /tmp/r03_explore
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_explore -o ex -- python $R/profiles/tools/explore_profile.py > /tmp/explore.log 2>&1
grep -E "^wall|^bounds|^ rows" /tmp/explore.log | head -20 > $OUT/explore_summary.txt
find /tmp/r03_explore -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/explore_kernel_stats.csv
cat $OUT/explore_summary.txt | head -4 | cut -c1-400
head -8 $OUT/explore_kernel_stats.csv | cut -c1-150
