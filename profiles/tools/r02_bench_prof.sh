# bench.py under rocprofv3 --kernel-trace --stats, timed region only
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd /tmp
rm -rf /tmp/r02_bench
timeout 420 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d /tmp/r02_bench -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench_prof.log 2>&1
grep '^{"metric"' /tmp/bench_prof.log | tail -1 > $OUT/bench_profiled.json
tail -5 /tmp/bench_prof.log | cut -c1-300
find /tmp/r02_bench -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
find /tmp/r02_bench -name '*domain_stats.csv' | head -1 | xargs -I{} cp {} $OUT/bench_domain_stats.csv
du -sh $R/gpurun_out; ls -la $OUT
