# round 6, second call: cand kernel with the centre strip, prefetch
# prediction statistics + timeline, MVEE phase stamps, small-batch profile
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "two_stage or list_eval or accept_routes or nested or nautilus_bound or union" > $O/tests_parity.txt 2>&1; tail -3 $O/tests_parity.txt
for i in 1 2 3; do
  timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|index list|gathered" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/accept_bench.txt
done
grep "^D=" $O/accept_bench.txt
NB_STAGE_TIMING=1 timeout 300 python profiles/tools/accept_bench.py 50 2>&1 | grep -E "\[stage\]" | tail -3 | cut -c1-120
python bench.py --no-cpu-baseline > $O/bench_prefetch.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06b/bench_prefetch.json').read().strip().splitlines()[-1])
print('value %.4g ms_per_step %.3f full %.4g prefetch %s' % (d['value'], d['ms_per_step'], d['value_full_run'], d.get('prefetch')))
PY
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rm -rf /tmp/r06_trace && timeout 600 rocprofv3 --kernel-trace --marker-trace --selected-regions --output-format csv -d /tmp/r06_trace -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench_trace.log 2>&1)
f=$(find /tmp/r06_trace -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/bench_timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-150:]
prev = None
tot_gap = tot_busy = 0.0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('::')[-1].split('(')[0].split('<')[0]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    if prev is not None: tot_gap += max(gap, 0.0)
    tot_busy += (e - s) / 1e3
    print('%-40s dur %8.1f us   gap before %7.1f us' % (name[:40], (e - s) / 1e3, gap))
    prev = max(e, prev or 0)
print('busy %.1f us, idle %.1f us' % (tot_busy, tot_gap))
PY
tail -60 $O/bench_timeline.txt
NAUTILUS_HIP_LIB=$PWD/nautilus_amd/lib/libnautilus_hip_dbg.so python profiles/tools/mvee_phases.py > $O/mvee_phases.txt 2>&1; cat $O/mvee_phases.txt
timeout 900 python profiles/tools/small_batch_profile.py 20 > $O/small_batch_profile.txt 2>&1; grep -E "exploration|sampling phase|log Z" $O/small_batch_profile.txt
