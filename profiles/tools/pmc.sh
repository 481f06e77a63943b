export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc1 -o f -- python dev/stream_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc2 -o w -- python dev/stream_bench.py > /dev/null 2>&1
ls /tmp/pmc1 /tmp/pmc2
python - <<'PY'
import csv, glob, collections
for tag, d in (('FETCH_SIZE','/tmp/pmc1'),('WRITE_SIZE','/tmp/pmc2')):
    f = glob.glob(d+'/*counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'stream' in r['Kernel_Name'] and r['Counter_Name']==tag:
            agg[(r['Kernel_Name'][:60], r['Grid_Size'])].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(tag, k, 'n=%d mean=%.1f' % (len(v), sum(v)/len(v)))
PY
