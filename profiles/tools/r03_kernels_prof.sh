#!/bin/bash
# Round-3 kernel evidence: streaming contains at several dimensions, dense
# emulator evaluation at D = 50 / 64 / 100 / 128, acceptance of a K = 4, M = 4
# bound -- each with its rocprofv3 kernel statistics; one PMC pass for the
# streaming kernel.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp
for name in stream_bench; do
  rm -rf /tmp/p_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o p -- python $R/profiles/tools/$name.py > $OUT/$name.txt 2>&1
  find /tmp/p_$name -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${name}_kernel_stats.csv
  grep -v amdgpu.ids $OUT/$name.txt | tail -12
done
# acceptance through the two stages, one process per case (D[:K:M])
: > $OUT/accept_bench.txt
for c in 50 100 50:2:4 50:3:3 20:4:4; do
  rm -rf /tmp/p_acc
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_acc -o p -- python $R/profiles/tools/accept_bench.py $c > /tmp/acc.txt 2>&1
  grep "^D=\|gathered" /tmp/acc.txt >> $OUT/accept_bench.txt
  find /tmp/p_acc -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/accept_bench_${c//:/_}_kernel_stats.csv
done
cat $OUT/accept_bench.txt
rm -rf /tmp/p_fast
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fast -o p -- python $R/profiles/tools/fast_time.py 50 64 100 128 > $OUT/fast_time.txt 2>&1
find /tmp/p_fast -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/fast_time_kernel_stats.csv
grep -v amdgpu.ids $OUT/fast_time.txt | tail -10
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_s_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_s_$c -o p -- python $R/profiles/tools/stream_bench.py > /dev/null 2>&1
  python3 - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob('/tmp/p_s_%s/**/*counter_collection.csv' % c, recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'nb_ell_stream' in r['Kernel_Name'] and r['Counter_Name'] == c] if f else []
print(c, 'stream dispatches', len(rows), 'mean KB', sum(float(r['Counter_Value']) for r in rows) / max(1, len(rows)))
PY
done > $OUT/stream_pmc.txt 2>&1
cat $OUT/stream_pmc.txt
