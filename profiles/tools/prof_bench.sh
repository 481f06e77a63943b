export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_r01f
timeout 600 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d $R/gpurun_out/prof_r01f -o bench -- python $R/bench.py > $R/gpurun_out/bench_prof.log 2>&1
grep '^{"metric"' $R/gpurun_out/bench_prof.log | tail -1 > $R/gpurun_out/bench_r01.json
