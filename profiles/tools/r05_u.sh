#!/bin/bash
# Round 5, fifth session: where the idle queue of a timed step is -- the
# dispatch timeline of the bench's last steps (rocprofv3 --kernel-trace) and
# the host profile of 20 sampling-phase steps.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5
mkdir -p $O
cd /tmp && rm -rf /tmp/s5_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/s5_trace -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_traced.log 2>&1
f=$(find /tmp/s5_trace -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/bench_timeline${1}.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# one nb_lse_final_kernel per add_samples step: the window of the last steps
ends = [i for i, r in enumerate(rows) if 'nb_lse_final' in r['Kernel_Name']]
lo, hi = ends[-7], ends[-1]
prev = int(rows[lo]['End_Timestamp'])
tot_gap = tot_busy = 0.0
steps = 0
for r in rows[lo + 1:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('::')[-1].split('(')[0].split('<')[0]
    gap = (s - prev) / 1e3
    tot_gap += max(gap, 0.0)
    tot_busy += (e - s) / 1e3
    print('%-40s dur %8.1f us   gap before %7.1f us' % (name[:40], (e - s) / 1e3, gap))
    if 'nb_lse_final' in r['Kernel_Name']:
        steps += 1
        print('---- step: busy %.1f us, idle %.1f us' % (tot_busy, tot_gap))
        tot_gap = tot_busy = 0.0
    prev = max(e, prev)
PY
cd $R
timeout 600 python profiles/tools/step_cprofile.py > $O/step_cprofile${1}.txt 2>&1
tail -3 $O/bench_traced.log | cut -c1-400
tail -40 $O/step_cprofile${1}.txt
