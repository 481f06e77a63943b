export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_r01f
timeout 600 rocprofv3 --kernel-trace --stats --marker-trace --selected-regions --output-format csv -d $R/gpurun_out/prof_r01f -o bench -- python $R/bench.py > $R/gpurun_out/bench_prof.log 2>&1
tail -1 $R/gpurun_out/bench_prof.log > $R/gpurun_out/bench_r01.json
cd $R
for s in 1 2 3; do timeout 300 python bench.py --seed $s --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_seed$s.json; done
timeout 300 python bench.py --host-likelihood --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_host.json
bash profiles/tools/eval_pmc.sh > gpurun_out/eval_pmc.txt 2>&1
