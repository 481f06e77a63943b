#!/bin/bash
# Round-4 evidence for the two-stage bound evaluation (nb_cand.hip + batched
# nb_eval_fast.hip): accept_bench per dimension (own process each), the same
# under rocprofv3 --kernel-trace --stats (CSV per case), and PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, gfx950) for nb_cand_kernel.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd $R
: > $OUT/accept_bench.txt
for c in 50 100 50:2:4 50:3:3 20; do
  python profiles/tools/accept_bench.py $c 2>/dev/null | tee -a $OUT/accept_bench.txt
done
cd /tmp
for c in 50 100; do
  rm -rf /tmp/ab_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$c -o ab -- python $R/profiles/tools/accept_bench.py $c > /dev/null 2>&1
  find /tmp/ab_$c -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/accept_bench_${c}_kernel_stats.csv
  head -6 $OUT/accept_bench_${c}_kernel_stats.csv | cut -c1-160
done
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pm_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm_$ctr -o p -- python $R/profiles/tools/accept_bench.py 50 > /dev/null 2>&1
  python - "$ctr" <<'PY' | tee -a $OUT/cand_pmc.txt
import csv, glob, sys
c = sys.argv[1]
fs = glob.glob('/tmp/pm_%s/**/*counter_collection.csv' % c, recursive=True)
if not fs:
    print(c, 'no data'); sys.exit()
rows = [r for r in csv.DictReader(open(fs[0])) if r['Counter_Name'] == c]
for kern in ('nb_cand_kernel', 'nb_cand_compact_kernel', 'nb_eval_fast_kernel'):
    vals = [float(r['Counter_Value']) for r in rows
            if ('::' + kern + '<') in r['Kernel_Name'] or ('::' + kern + '(') in r['Kernel_Name']]
    if vals:
        print('%s %s dispatches %d mean %.4g (raw counter; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE x 2 on gfx950 for bytes)' % (c, kern, len(vals), sum(vals) / len(vals)))
PY
done
