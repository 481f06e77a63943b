# exploration phase of the headline run under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd /tmp
rm -rf /tmp/r02_explore
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02_explore -o ex -- python $R/profiles/tools/explore_profile.py > /tmp/explore.log 2>&1
grep -E "^wall|^bounds|^ rows" /tmp/explore.log | head -20 > $OUT/explore_summary.txt
find /tmp/r02_explore -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/explore_kernel_stats.csv
cat $OUT/explore_summary.txt | head -4
cd $R
timeout 600 python bench.py --pmc-traffic --no-cpu-baseline > /tmp/bench_pmc.log 2>&1
grep '^{"metric"' /tmp/bench_pmc.log | tail -1 > $OUT/bench_pmc_traffic.json
python3 -c "
import json; r=json.load(open('$OUT/bench_pmc_traffic.json')); print('traffic', r['roofline']['traffic'], r['roofline']['traffic_source'], 'alg', r['roofline']['algorithmic_bytes_per_launch'], 'frac', r['roofline']['frac'])"
du -sh $R/gpurun_out
