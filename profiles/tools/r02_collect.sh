# Round-2 evidence (run on the GPU box from the repo root):
#   1. bench.py under rocprofv3 --kernel-trace --stats, timed region only
#   2. bench.py --pmc-traffic (HBM bytes per launch of nb_eval_kernel, measured
#      by child rocprofv3 --pmc passes of the same command)
#   3. the exploration phase of the headline run under rocprofv3 --kernel-trace
#      (emulator training, MVEE, mixture fits, draw kernels)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd /tmp
rm -rf /tmp/r02_bench /tmp/r02_explore
timeout 900 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d /tmp/r02_bench -o bench -- python $R/bench.py > $OUT/bench_prof.log 2>&1
grep '^{"metric"' $OUT/bench_prof.log | tail -1 > $OUT/bench_profiled.json
cp $(find /tmp/r02_bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
cp $(find /tmp/r02_bench -name '*domain_stats.csv' | head -1) $OUT/bench_domain_stats.csv 2>/dev/null
cd $R
timeout 1500 python bench.py --pmc-traffic --no-cpu-baseline > $OUT/bench_pmc.log 2>&1
grep '^{"metric"' $OUT/bench_pmc.log | tail -1 > $OUT/bench_pmc_traffic.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02_explore -o ex -- python $R/profiles/tools/explore_profile.py > $OUT/explore.log 2>&1
cp $(find /tmp/r02_explore -name '*kernel_stats.csv' | head -1) $OUT/explore_kernel_stats.csv
tail -16 $OUT/explore.log | head -15
python3 - <<PY
import json
for f in ('bench_profiled.json', 'bench_pmc_traffic.json'):
    try:
        r = json.load(open('$OUT/' + f))
        print(f, 'value %.4g full %.4g frac %.3f traffic %s' % (r['value'], r['value_full_run'], r['roofline']['frac'], r['roofline']['traffic']))
    except Exception as e:
        print(f, 'missing', e)
PY
