#!/bin/bash
# End of the fifth session: the default bench command under rocprofv3
# --kernel-trace --stats (timed region only) and the tests that compare with
# the reference's mixture runs.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
cd /tmp && rm -rf /tmp/s5_bench
timeout 600 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d /tmp/s5_bench -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench_prof.log 2>&1
grep '^{"metric"' /tmp/bench_prof.log | tail -1 > $O/bench_profiled.json
find /tmp/s5_bench -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
head -7 $O/bench_kernel_stats.csv | cut -c1-160
cd $R
timeout 900 python -m pytest tests/test_configs_gpu.py -q -m gpu -k "mixture_against" 2>&1 | tail -4 | tee $O/mixture_tests.log
