# round 6, third call: prefetch with the transfer queued first, pairing
# without n_bounds passes, cand kernel at three wavefronts per SIMD (n_dim 33-64)
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "two_stage or list_eval or accept_routes or nested" > $O/tests_parity.txt 2>&1; tail -2 $O/tests_parity.txt
python -m pytest tests/test_sampler_gpu.py tests/test_sampler_behaviour_gpu.py tests/test_fuzz_gpu.py -x -q > $O/tests_sampler.txt 2>&1; tail -2 $O/tests_sampler.txt
for i in 1 2; do
  timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|index list|gathered" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/accept_bench.txt
done
grep "^D=" $O/accept_bench.txt
NB_STAGE_TIMING=1 timeout 300 python profiles/tools/accept_bench.py 50 2>&1 | grep -E "\[stage\]" | tail -2 | cut -c1-120
for i in 1 2; do
  python bench.py --no-cpu-baseline > $O/bench_prefetch_$i.json 2>/dev/null
done
NB_PREFETCH=0 python bench.py --no-cpu-baseline > $O/bench_noprefetch_1.json 2>/dev/null
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r06c/bench_*.json')):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print(p, 'value %.4g ms_per_step %.3f full %.4g setup %.2f prefetch %s' % (d['value'], d['ms_per_step'], d['value_full_run'], d['setup_s'], d.get('prefetch')))
PY
NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_dbg.so python profiles/tools/mvee_phases.py > $O/mvee_phases.txt 2>&1; cat $O/mvee_phases.txt
timeout 900 python profiles/tools/small_batch_profile.py 20 > $O/small_batch_profile.txt 2>&1; grep -E "exploration|sampling phase|log Z" $O/small_batch_profile.txt
