# timeline of the bench's last timed steps (rocprofv3 --kernel-trace, timed region only)
export TMPDIR=/tmp
R=$PWD; O=gpurun_out/r06t; mkdir -p $O
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms_per_step %.3f full %.4g setup %.2f' % (d['value'], d['ms_per_step'], d['value_full_run'], d['setup_s']))"; done
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
tail -46 $O/bench_timeline.txt
