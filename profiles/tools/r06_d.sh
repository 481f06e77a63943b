# round 6, fourth call: timeline of the timed steps with the refill ahead,
# final cand configuration, run times of the reference-settings runs
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "two_stage or list_eval or accept_routes or nested or gmm" > $O/tests_parity.txt 2>&1; tail -2 $O/tests_parity.txt
python -m pytest tests/test_configs_gpu.py -x -q -k "C5_prefix" --durations=5 > $O/tests_prefix.txt 2>&1; tail -8 $O/tests_prefix.txt
for i in 1 2; do
  timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -E "proposals:|index list|gathered" | sed 's/, 1048576 proposals//; s/(accepted.*//' >> $O/accept_bench.txt
done
grep "^D=" $O/accept_bench.txt
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
tail -48 $O/bench_timeline.txt
rm -f gpurun_out/r06_anchors.jsonl
python profiles/tools/r06_anchor_runs.py funnel 20 0 2>&1 | tail -1 | cut -c1-400
python profiles/tools/r06_anchor_runs.py mixture 30 0 2>&1 | tail -1 | cut -c1-400
python profiles/tools/r06_anchor_runs.py funnel 50 0 2>&1 | tail -1 | cut -c1-600
cp gpurun_out/r06_anchors.jsonl $O/anchors_after_pairing_fix.jsonl
