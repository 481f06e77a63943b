# HBM traffic of nb_eval_kernel over the timed region of the default bench.py
# run: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kernel
# trace only, as MI355X_MICROARCH.md prescribes); the timed region's launches
# are the last `roofline.launches` nb_eval_kernel dispatches of the run.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bt_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/bt_$c -o p -- python $R/bench.py --no-cpu-baseline > /tmp/bt_$c.log 2>&1
  grep '^{"metric"' /tmp/bt_$c.log | tail -1 > /tmp/bt_$c.json
done
python3 - <<'PY'
import csv, glob, json, os
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    line = json.load(open('/tmp/bt_%s.json' % c))
    n = line['roofline']['launches']
    f = glob.glob('/tmp/bt_%s/*counter_collection.csv' % c)[0]
    rows = [r for r in csv.DictReader(open(f))
            if 'nb_eval_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    vals = [float(r['Counter_Value']) for r in rows[-n:]]
    out[c] = dict(launches=n, mean_kb=sum(vals) / len(vals),
                  ms_per_step=line['ms_per_step'])
# FETCH_SIZE counts 64-byte units as "KB/2" on gfx950 (x2 correction)
read_b = out['FETCH_SIZE']['mean_kb'] * 1024 * 2
write_b = out['WRITE_SIZE']['mean_kb'] * 1024
res = dict(kernel='nb_eval_kernel', command='python bench.py --no-cpu-baseline',
           launches=out['FETCH_SIZE']['launches'],
           hbm_read_bytes_per_launch=read_b, hbm_write_bytes_per_launch=write_b,
           hbm_bytes_per_launch=read_b + write_b, raw=out)
root = os.environ.get('GRAFT_REPO_ROOT', '.')
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(root, 'gpurun_out', 'bench_eval_traffic.json'), 'w'), indent=1)
print(json.dumps(res))
PY
