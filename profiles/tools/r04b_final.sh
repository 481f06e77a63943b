#!/bin/bash
# Round 4, second session: evidence at the final state of the tree.
#   gpurun --timeout 3000 -- bash profiles/tools/r04b_final.sh [suite]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04b_final
mkdir -p $O
cd $R
# 1. the bench line
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; echo
# 2. kernel statistics of the timed region
cd /tmp && rm -rf /tmp/r04b_bench
timeout 600 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d /tmp/r04b_bench -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench_prof.log 2>&1
grep '^{"metric"' /tmp/bench_prof.log | tail -1 > $O/bench_profiled.json
find /tmp/r04b_bench -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
head -6 $O/bench_kernel_stats.csv | cut -c1-160
# 3. trainer: speed, whole exploration, a training-only run under rocprofv3
cd $R
python profiles/tools/train_speed.py 2>/dev/null > $O/train_speed.txt; cat $O/train_speed.txt
python profiles/tools/train_speed.py --big 2>/dev/null > $O/train_speed_big.txt; cat $O/train_speed_big.txt
python profiles/tools/explore_profile.py 2>/dev/null > $O/explore_summary.txt; head -3 $O/explore_summary.txt
cd /tmp && rm -rf /tmp/r04b_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04b_train -o train -- python $R/profiles/tools/train_speed.py > /tmp/train_prof.log 2>&1
find /tmp/r04b_train -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/train_only_kernel_stats.csv
head -5 $O/train_only_kernel_stats.csv | cut -c1-160
cd $R
# 4. two-stage route and the stream kernel (unchanged code; same-tree numbers)
python profiles/tools/accept_bench.py 50 100 2>/dev/null > $O/accept_bench.txt; grep "D=" $O/accept_bench.txt
# 5. the suite
if [ "$1" = suite ]; then
  python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 > $O/suite.log; tail -4 $O/suite.log
fi
