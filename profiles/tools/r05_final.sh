#!/bin/bash
# end of round 5: HBM traffic of the timed region, then the whole GPU suite
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 1500 python bench.py --no-cpu-baseline --pmc-traffic > $O/bench_pmc_traffic.json 2> $O/bench_pmc.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r05/bench_pmc_traffic.json').read().strip().splitlines()[-1])['roofline']
print('traffic', r['traffic'], 'algorithmic', r['algorithmic_bytes_per_launch'], r['traffic_source'])
PY
timeout 2700 python -m pytest tests -q -m gpu --durations=25 2>&1 | tail -45 > $O/suite_final.log
tail -40 $O/suite_final.log
