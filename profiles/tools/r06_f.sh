# round 6, sixth call: long lists through the sphere mask
O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "long_list or two_stage or sampler or gaussian or list_eval or accept_routes or nested or periodic or exclusion or shell" > $O/tests_parity.txt 2>&1; tail -3 $O/tests_parity.txt
python -m pytest tests/test_sampler_gpu.py tests/test_sampler_behaviour_gpu.py tests/test_fuzz_gpu.py -x -q > $O/tests_sampler.txt 2>&1; tail -2 $O/tests_sampler.txt
rm -f gpurun_out/r06_anchors.jsonl
python profiles/tools/r06_anchor_runs.py funnel 20 0 2>&1 | tail -1 | cut -c1-300
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rm -rf /tmp/r06_f50 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_f50 -o f50 -- python $R/profiles/tools/r06_anchor_runs.py funnel 50 0 > /tmp/f50.log 2>&1)
tail -1 /tmp/f50.log | cut -c1-700
f=$(find /tmp/r06_f50 -name '*kernel_stats.csv' | head -1)
head -16 "$f" | cut -c1-230 > $O/funnel50_kernel_stats.csv; cut -c1-160 $O/funnel50_kernel_stats.csv
cp gpurun_out/r06_anchors.jsonl $O/anchors_sphere_mask.jsonl
python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
print('value %.4g ms_per_step %.3f full %.4g setup %.2f' % (d['value'], d['ms_per_step'], d['value_full_run'], d['setup_s']), d['setup_breakdown'])
PY
