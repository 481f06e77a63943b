"""Per-dispatch durations and the idle gaps between consecutive dispatches
from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): the last `n`
dispatches whose kernel name contains one of the given substrings.
python profiles/tools/trace_gaps.py trace.csv 40 nb_cand nb_eval_fast"""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2])
keys = sys.argv[3:]
rows = [r for r in csv.DictReader(open(path))
        if any(k in r['Kernel_Name'] for k in keys)]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n:]
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('::')[-1].split('(')[0]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print('%-44s dur %8.1f us   gap before %8.1f us   scratch %s' % (
        name[:44], (e - s) / 1e3, gap, r.get('Scratch_Size', r.get('Private_Segment_Size', '?'))))
    prev_end = e
