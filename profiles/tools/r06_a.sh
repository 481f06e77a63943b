# round 6, first measurement call: parity of what changed, prefetch A/B,
# two-stage bench, draw counters, small-batch host profile
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "two_stage or gmm or list_eval or accept_routes or pipelined or union_proposals" > $O/tests_parity.txt 2>&1; tail -3 $O/tests_parity.txt
python -m pytest tests/test_sampler_gpu.py tests/test_sampler_behaviour_gpu.py tests/test_fuzz_gpu.py -x -q > $O/tests_sampler.txt 2>&1; tail -3 $O/tests_sampler.txt
for i in 1 2; do
  NB_PREFETCH=0 python bench.py --no-cpu-baseline > $O/bench_noprefetch_$i.json 2>/dev/null
  NB_PREFETCH=1 python bench.py --no-cpu-baseline > $O/bench_prefetch_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/r06a/bench_*.json')):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(p, 'value %.4g ms_per_step %.3f full %.4g setup %.2f fill %.3f roofline %.3f draw %s' % (
            d['value'], d['ms_per_step'], d['value_full_run'], d['setup_s'], d['shell_fill_s'], d['roofline']['frac'],
            {k: round(v, 4) for k, v in d.get('roofline_draw', {}).items() if k in ('avg_launch_ms', 'hbm_frac', 'frac')}))
    except Exception as e:
        print(p, 'failed', e)
PY
for i in 1 2 3; do
  timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|index list|gathered" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/accept_bench.txt
done
grep "^D=" $O/accept_bench.txt
NB_STAGE_TIMING=1 timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|\[stage\]" | tail -12 > $O/stage_timing.txt
tail -4 $O/stage_timing.txt
bash profiles/tools/draw_pmc.sh 50 > $O/draw_pmc.txt 2>&1; cat $O/draw_pmc.txt
timeout 600 python profiles/tools/small_batch_profile.py 20 > $O/small_batch_profile.txt 2>&1; head -5 $O/small_batch_profile.txt
