#!/bin/bash
# Round-6 evidence for the bench line: (1) the plain default run, (2) the same
# command under rocprofv3 --kernel-trace --stats restricted to the timed
# region, (3) HBM traffic of the bound-evaluation kernels (bench.py
# --pmc-traffic: FETCH_SIZE / WRITE_SIZE in separate child passes), (4)
# configuration 4 (full run) through examples/run_config.py.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_r06.json 2> $OUT/bench_r06.err
tail -c 400 $OUT/bench_r06.json
cd /tmp
rm -rf /tmp/r06_bench
timeout 600 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d /tmp/r06_bench -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench_prof.log 2>&1
grep '^{"metric"' /tmp/bench_prof.log | tail -1 > $OUT/bench_profiled.json
find /tmp/r06_bench -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
cd $R
timeout 1500 python bench.py --no-cpu-baseline --pmc-traffic > $OUT/bench_pmc_traffic.json 2> $OUT/bench_pmc.err
python - <<'PY'
import json, os
p = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r06/bench_pmc_traffic.json')
try:
    r = json.loads(open(p).read().strip().splitlines()[-1])['roofline']
    print('traffic', r['traffic'], 'algorithmic', r['algorithmic_bytes_per_launch'], r['traffic_source'])
except Exception as e:
    print('pmc failed', e)
PY
for c in C4; do
  timeout 1200 python examples/run_config.py $c > $OUT/cfg_$c.json 2> $OUT/cfg_$c.err
  cut -c1-300 $OUT/cfg_$c.json
done
# kernel statistics of the headline run's exploration
cd /tmp && rm -rf /tmp/r06_explore
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_explore -o ex -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /tmp/explore_prof.log 2>&1
find /tmp/r06_explore -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c "head -25 {} | cut -c1-260 > $OUT/explore_kernel_stats.csv"
head -8 $OUT/explore_kernel_stats.csv | cut -c1-160
